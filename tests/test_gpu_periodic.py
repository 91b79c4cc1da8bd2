"""Periodic boundaries through the C ABI (pib_set_periodic): the wrapped operators against the oracle's assembled
matrices (src/mesh/cartesianmesh.cpp:595-681 wraps the neighbour indices, :259-266 adds the extra velocity point),
bit-exact, and the Krylov / multigrid solves on them against the oracle's restatement."""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops
from test_gpu_parity import amgx_cfg, gmg_cfg, iters_close, rhs_for

pytestmark = pytest.mark.gpu

CASES = {
    "2d_xy": ((12, 10), (True, True), None),
    "2d_y": ((12, 10), (False, True), (1.05, 1.0)),
    "2d_x_stretched": ((9, 14), (True, False), (1.1, 0.95)),
    "3d_xz": ((8, 6, 10), (True, False, True), None),
    "3d_all": ((8, 9, 7), (True, True, True), (1.0, 1.03, 1.0)),
}


@pytest.fixture(scope="module")
def lin():
    from petibm_amd import linsolver
    return linsolver


def system(case, dt=0.0125, pinned=False):
    n, per, ratios = CASES[case]
    cfg = omesh.periodic_config(n, per, lo=-0.5, hi=1.5, ratios=ratios)
    m = omesh.create_mesh(cfg)
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, DBNG = oops.create_poisson_operator(D, G, L, dt, 0.5 * 0.01)
    if pinned:
        DBNG = oops.pin_row0(DBNG)
    return m, per, DBNG, L


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("pinned", [False, True])
def test_periodic_poisson_assembly_bit_exact(lin, case, pinned):
    from petibm_amd import capi
    dt = 0.0125
    m, per, DBNG, _ = system(case, dt, pinned)
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg())
    s.setPeriodic(per)
    n = [int(v) for v in m.n[3][: m.dim]]
    s.assemblePoisson(n, [m.dL[3][d].true for d in range(m.dim)], dt,
                      capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    rp, cl, vl = s.getCSR()
    assert np.array_equal(rp, DBNG.rowptr)
    assert np.array_equal(cl, DBNG.col)
    assert np.array_equal(vl, DBNG.val)
    s.destroy()


@pytest.mark.parametrize("case,n,per", [("2d", (64, 48), (True, True)), ("2d_y", (48, 40), (False, True)),
                                        ("3d", (32, 32, 32), (True, True, True)), ("3d_odd", (21, 18, 13), (True, False, True))])
def test_periodic_gmg_pcg_matches_oracle(lin, case, n, per):
    from petibm_amd import capi
    dt = 0.01
    cfg = omesh.periodic_config(n, per)
    m = omesh.create_mesh(cfg)
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.005)
    xs, b = rhs_for(A)
    w = [m.dL[3][d].true for d in range(m.dim)]
    s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=1, post=1))
    s.setPeriodic(per)
    s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=1, pre=1, post=1, omega=0.9, coarsest_sweeps=32, periodic=per)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert ref["iters"] <= 22  # 17 with walls: the transfers treat the periodic seam like a wall
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 8)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-8)
    s.destroy()
