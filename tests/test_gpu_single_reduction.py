"""CG with PETSc's single-reduction recurrences (`-<name>_ksp_cg_single_reduction` / `pib_cg_single_reduction=1`;
csrc/krylov.hip:solve_cg_sr) against its restatement in the oracle (oracle/csrc/oracle.c:orc_cg_single_reduction,
oracle/csrc/gmg.c:orc_pcg_gmg_single_reduction) -- the bars of tests/test_gpu_parity.py: iteration counts within 1 (or
3 %), residual histories to 1e-8 ... 1e-9, solutions to 1e-8, the residual contract recomputed with the CSR operator --
and, on several ranks, the point of the variant: ONE all-reduce per iteration.
"""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops
from test_gpu_parity import (STRETCHED_2D, amgx_cfg, gmg_cfg, iters_close, poisson_system, rhs_for, stretched_3d)

pytestmark = pytest.mark.gpu

SR = "pib_cg_single_reduction=1\n"


@pytest.fixture(scope="module")
def lin():
    from petibm_amd import linsolver
    return linsolver


@pytest.mark.parametrize("pc", ["NOSOLVER", "BLOCK_JACOBI"])
@pytest.mark.parametrize("case", ["2d", "3d"])
def test_single_reduction_cg_matches_oracle_amgx_flavour(lin, pc, case):
    """true-residual L2 norm, constant null space (the lazy mean shift commutes with the product: A 1 = 0)"""
    from petibm_amd import capi
    dt = 0.01
    m, A, _ = poisson_system(STRETCHED_2D if case == "2d" else stretched_3d((14, 12, 10)), dt=dt)
    xs, b = rhs_for(A)
    nn, ww = [int(v) for v in m.n[3][: m.dim]], [m.dL[3][d].true for d in range(m.dim)]
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc=pc, tol=1e-10, extra=SR + "pib_initial_guess_nonzero=0\n"))
    s.assemblePoisson(nn, ww, dt, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    ref = clib.cg(A, b, single_reduction=True, pc="jacobi" if pc != "NOSOLVER" else "none", nullspace=1, norm="unpreconditioned",
                  rtol=1e-10, atol=0.0, dtol=1e300, maxit=5000)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert iters_close(s.getIters(), ref["iters"])
    h = s.getResidualHistory()
    k = min(len(h), len(ref["history"]), 12)
    assert np.allclose(h[:k], ref["history"][:k], rtol=1e-9)
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    e = (x - x.mean()) - (ref["x"] - ref["x"].mean())
    assert np.linalg.norm(e) <= 1e-7 * np.linalg.norm(ref["x"])
    # the standard recurrence on the same system: the same iteration in exact arithmetic
    t = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc=pc, tol=1e-10, extra="pib_initial_guess_nonzero=0\n"))
    t.assemblePoisson(nn, ww, dt, capi.NULLSPACE_CONSTANT)
    y = np.zeros(A.n_rows)
    t.solve(y, b)
    assert iters_close(s.getIters(), t.getIters())
    assert np.allclose(h[:k], t.getResidualHistory()[:k], rtol=1e-9)
    s.destroy()
    t.destroy()


def test_single_reduction_cg_ksp_flavour_and_its_own_option_name(lin):
    """preconditioned norm, zeroed guess, pinned (non-singular) system, the option spelled as PETSc spells it"""
    m, A, _ = poisson_system(stretched_3d((14, 12, 10)), pinned=True)
    xs, b = rhs_for(A, zero_mean=False)
    b[0] = 0.0
    petsc = ("-poisson_ksp_type cg\n-poisson_ksp_rtol 1.0E-10\n-poisson_ksp_atol 1.0E-50\n-poisson_pc_type jacobi\n"
             "-poisson_ksp_cg_single_reduction\n")
    s = lin.LinSolverHIP("poisson", config_text=petsc)
    assert s.getType() == "PETSc KSP"
    s.setMatrix(A)
    x = np.full(A.n_rows, 7.0)  # ignored: KSP zeroes the guess
    s.solve(x, b)
    ref = clib.cg(A, b, single_reduction=True, pc="jacobi", norm="preconditioned", rtol=1e-10, atol=1e-50, maxit=10000)
    assert ref["reason"] > 0 and s.getReason() > 0 and iters_close(s.getIters(), ref["iters"])
    h = s.getResidualHistory()
    k = min(len(h), len(ref["history"]), 12)
    assert np.allclose(h[:k], ref["history"][:k], rtol=1e-9)
    assert np.linalg.norm(x - ref["x"]) <= 1e-7 * np.linalg.norm(ref["x"])
    assert np.isclose(s.getResidual(), h[-1])
    # max_it ends it like the standard recurrence: DIVERGED_ITS, an error for a KSP-flavoured solver (linsolverksp.cpp:96-104)
    from petibm_amd.capi import PibError, ERR_CONV_FAILED
    t = lin.LinSolverHIP("poisson", config_text=petsc + "-poisson_ksp_max_it 3\n")
    t.setMatrix(A)
    with pytest.raises(PibError) as ei:
        t.solve(np.zeros(A.n_rows), b)
    assert ei.value.code == ERR_CONV_FAILED and t.getIters() == 3 and t.getReason() == -3
    s.destroy()
    t.destroy()


@pytest.mark.parametrize("case,pre,post", [("2d_stretched", 1, 1), ("3d_uniform", 2, 2), ("3d_stretched", 1, 1), ("3d_uniform_odd", 2, 2)])
@pytest.mark.parametrize("norm", ["L2", "preconditioned"])
def test_single_reduction_multigrid_pcg_matches_oracle(lin, case, pre, post, norm):
    from petibm_amd import capi
    cfg = {"2d_stretched": STRETCHED_2D, "3d_uniform": omesh.uniform_config((32, 32, 32)),
           "3d_stretched": stretched_3d((24, 20, 16)), "3d_uniform_odd": omesh.uniform_config((21, 18, 13))}[case]
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    if norm == "L2":
        text = gmg_cfg(pre=pre, post=post, extra=SR)
    else:
        text = (f"-poisson_ksp_type cg\n-poisson_ksp_rtol 1.0E-10\n-poisson_ksp_atol 1.0E-50\n-poisson_pc_type gamg\n"
                f"-poisson_ksp_cg_single_reduction true\n-poisson_pib_smoother jacobi\n-poisson_pib_presweeps {pre}\n-poisson_pib_postsweeps {post}\n-poisson_pib_sweep_pairs 0\n")
    s = lin.LinSolverHIP("poisson", config_text=text)
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=1, pre=pre, post=post, omega=0.9, coarsest_sweeps=32)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200, single_reduction=True, norm="unpreconditioned" if norm == "L2" else "preconditioned")
    assert ref["reason"] > 0 and s.getReason() > 0
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= (1.5e-10 if norm == "L2" else 1e-8) * np.linalg.norm(b)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 8)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-8)
    e = (x - x.mean()) - (ref["x"] - ref["x"].mean())
    assert np.linalg.norm(e) <= 1e-8 * np.linalg.norm(ref["x"])
    s.destroy()


@pytest.mark.parametrize("case,pre,post", [("3d_uniform", 2, 2), ("3d_stretched", 1, 1), ("2d_stretched", 1, 1)])
def test_single_reduction_with_a_pinned_row_and_the_multigrid_matches_oracle(lin, case, pre, post):
    """Round 5 (a PIB_ERR_SUP until then): the convention every `type: GPU` run of PetIBM uses (row 0 pinned by
    MatZeroRowsColumns, navierstokes.cpp:414-420) with KSPCGUseSingleReduction's recurrences.  The shift by z[0] does not
    commute with the product of the pinned matrix, which this recurrence applies to z itself: it is applied in a pass of its
    own behind the cycle (krylov.hip OpPinShift) -- the oracle's PCAPPLY for nullspace 2 (oracle/csrc/gmg.c).  Against
    orc_pcg_gmg_single_reduction on the pinned system, and against the standard recurrence on the device (same counts)."""
    from petibm_amd import capi
    cfg = {"2d_stretched": STRETCHED_2D, "3d_uniform": omesh.uniform_config((32, 32, 32)), "3d_stretched": stretched_3d((24, 20, 16))}[case]
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt, pinned=True)
    xs, b = rhs_for(A, zero_mean=False)
    b[0] = 0.0
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    out = []
    for extra in (SR, ""):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post, extra=extra))
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_PINNED)
        x = np.zeros(A.n_rows)
        s.solve(x, b)
        assert s.getReason() > 0
        out.append((x, s.getIters(), s.getResidualHistory().copy()))
        s.destroy()
    g = clib.GMG(n, w, dt, nullspace=2, pre=pre, post=post, omega=0.9, coarsest_sweeps=32)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200, single_reduction=True)
    assert ref["reason"] > 0 and iters_close(out[0][1], ref["iters"]) and abs(out[0][1] - out[1][1]) <= 1
    assert np.linalg.norm(b - clib.spmv(A, out[0][0])) <= 1.5e-10 * np.linalg.norm(b)
    ke = min(len(out[0][2]), len(ref["history"]), 8)
    assert np.allclose(out[0][2][:ke], ref["history"][:ke], rtol=1e-8)
    assert out[0][0][0] == 0.0 and np.linalg.norm(out[0][0] - ref["x"]) <= 1e-8 * np.linalg.norm(ref["x"])


@pytest.mark.parametrize("P,n,pc,extra,sweeps", [
    (2, (16, 16, 16), "BLOCK_JACOBI", "", 1),
    (3, (12, 10, 9), "NOSOLVER", "", 1),
    (2, (16, 16, 32), "AMG", "pib_agglomerate_below=10\n", 1),
    (4, (32, 32, 32), "AMG", "pib_agglomerate_below=100\n", 2),
    (3, (128, 8, 48), "AMG", "pib_march_min_cells=0\npib_agglomerate_below=100\n", 2),
    (4, (128, 16, 64), "AMG", "pib_march_min_cells=0\npib_agglomerate_below=100\n", 2),
])
def test_single_reduction_on_slabs_one_allreduce_per_iteration(P, n, pc, extra, sweeps):
    """z-slabs over the loopback transport: the single rank's iteration count and solution, and the counters say what the
    variant is for -- one all-reduce per iteration (the standard recurrence: three with the multigrid, two without), no
    exchange for the Krylov product behind a V-cycle on deep halos (z comes out valid on the ghost plane the matrix reaches)."""
    from petibm_amd import capi
    import slab_plans as partition
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_multirank_loopback import _cfg, _run_ranks, _system
    dt = 0.01
    m, A, xs, b = _system(n, dt)
    w = [m.dL[3][d].true for d in range(m.dim)]
    plans = partition.all_plans(n, P)

    def rank_fn(r, uid, sr=True):
        pl = plans[r]
        s = LinSolverHIP("poisson", config_text=_cfg(pc, extra=extra + (SR if sr else ""), sweeps=sweeps), rank=r, nranks=P, uid=uid, device=0)
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(pl.n_local)
        s.solve(x, np.ascontiguousarray(b[pl.row0:pl.row0 + pl.n_local]))
        out = (x, s.getIters(), s.getResidualHistory(), s.counters().copy())
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    std = _run_ranks(P, lambda r, uid: rank_fn(r, uid, sr=False))
    x = np.concatenate([r[0] for r in res])
    assert len({r[1] for r in res}) == 1
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    s1 = LinSolverHIP("poisson", config_text=_cfg(pc, extra=extra + SR, sweeps=sweeps))
    s1.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    x1 = np.zeros(A.n_rows)
    s1.solve(x1, b)
    assert iters_close(res[0][1], s1.getIters()) and iters_close(res[0][1], std[0][1])
    k = min(len(res[0][2]), len(s1.getResidualHistory()), 8)
    assert np.allclose(res[0][2][:k], s1.getResidualHistory()[:k], rtol=1e-8)
    assert np.linalg.norm((x - x.mean()) - (x1 - x1.mean())) <= 1e-8 * np.linalg.norm(x1)
    s1.destroy()
    its = res[0][1]
    for r, t in zip(res, std):
        reductions, exchanges = int(r[3][2]), int(r[3][3])
        # the set-up's and one per ENQUEUED iteration (the host counts its calls: the last batch may reach a few iterations
        # beyond the one that met the tolerance -- their kernels return at once on the device's `done`)
        assert its + 1 <= reductions <= its + 1 + 8, (reductions, its)
        assert int(t[3][2]) >= 2 * t[1] and 2 * reductions <= int(t[3][2]) + 10   # the standard recurrence on the same ranks: >= 2 per iteration
        assert exchanges <= int(t[3][3]) + 1, (exchanges, int(t[3][3]))  # never more exchanges than the standard recurrence
