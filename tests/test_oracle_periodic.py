"""Periodic boundaries in the oracle: mesh counts (tests/mesh/cartesianmesh2d_yperiodic.cpp), the wrapped operators,
and the multigrid restatement's level operator against the assembled D*BN*G."""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops


@pytest.mark.parametrize("n,per", [((12, 10), (True, True)), ((12, 10), (False, True)), ((8, 6, 10), (True, False, True)),
                                   ((8, 8, 8), (True, True, True))])
def test_gmg_level_operator_is_dbng(n, per):
    dim = len(n)
    cfg = omesh.periodic_config(n, per, ratios=[1.0] * dim)
    m = omesh.create_mesh(cfg)
    for d in range(dim):
        assert m.n[d][d] == (n[d] if per[d] else n[d] - 1)
    dt = 0.01
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, DBNG = oops.create_poisson_operator(D, G, L, dt, 0.5 * 0.01, 1)
    w = [np.array([m.dL[3][d][i] for i in range(n[d])]) for d in range(dim)]
    g = clib.GMG(n, w, dt, nullspace=1, pre=1, post=1, omega=0.8, periodic=per)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(m.pN)
    y_ref = DBNG.to_scipy() @ x
    y = g.apply_operator(x)
    assert np.abs(y - y_ref).max() <= 1e-12 * np.abs(y_ref).max()
    # symmetric, constants in the null space
    A = DBNG.to_dense()
    assert np.abs(A - A.T).max() <= 1e-12 * np.abs(A).max()
    assert np.abs(A.sum(axis=1)).max() <= 1e-10 * np.abs(A).max()
    # PCG with the periodic V-cycle converges like the wall-bounded case
    b = y_ref
    res = g.pcg(DBNG, b, rtol=1e-10, maxit=60)
    assert res["reason"] > 0 and res["iters"] <= 25, res
    r = b - DBNG.to_scipy() @ res["x"]
    assert np.linalg.norm(r) <= 1.01e-10 * np.linalg.norm(b)
