"""BiCGStab preconditioned by the multigrid (csrc/krylov.hip: solve_bicgstab with Precond::GMG) against its restatement in the
oracle (oracle/csrc/oracle.c:orc_bcgs_gmg: the KSPBCGS / PBICGSTAB recurrences around oracle/csrc/gmg.c's V-cycle, the mean
removed after every application on the singular Poisson system).  AmgX takes any solver x preconditioner pair from the solver
file (/root/reference/src/linsolver/linsolveramgx.cpp:62-72), PETSc any -ksp_type / -pc_type
(/root/reference/src/linsolver/linsolverksp.cpp:62-66): `solver=PBICGSTAB, preconditioner=AMG` and `-ksp_type bcgs -pc_type gamg`
were a refusal (PIB_ERR_SUP) until round 4.  Bars as in tests/test_gpu_parity.py: iteration counts within 1, residual histories
to 1e-8, the residual contract recomputed with the CSR operator.
"""
import numpy as np
import pytest

from oracle import clib, mesh as omesh
from test_gpu_parity import STRETCHED_2D, gmg_cfg, iters_close, poisson_system, rhs_for, stretched_3d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lin():
    from petibm_amd import linsolver
    return linsolver


CASES = {"2d_stretched": STRETCHED_2D, "3d_uniform": omesh.uniform_config((32, 32, 32)), "3d_stretched": stretched_3d((24, 20, 16)),
         "3d_uniform_odd": omesh.uniform_config((21, 18, 13))}


@pytest.mark.parametrize("case,pre,post", [("2d_stretched", 1, 1), ("3d_uniform", 1, 1), ("3d_stretched", 2, 2), ("3d_uniform_odd", 2, 1)])
def test_pbicgstab_with_amg_matches_oracle_amgx_flavour(lin, case, pre, post):
    """right-preconditioned, true-residual L2 norm, zero guess; constant null space"""
    from petibm_amd import capi
    dt = 0.01
    m, A, _ = poisson_system(CASES[case], dt=dt)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    text = gmg_cfg(pre=pre, post=post).replace("solver(solv)=PCG", "solver(solv)=PBICGSTAB") + "pib_initial_guess_nonzero=0\n"
    s = lin.LinSolverHIP("poisson", config_text=text)
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=1, pre=pre, post=post, omega=0.9, coarsest_sweeps=32)
    ref = g.bcgs(A, b, norm="unpreconditioned", rtol=1e-10, dtol=1e300, maxit=200)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 4)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-7)
    e = (x - x.mean()) - (ref["x"] - ref["x"].mean())
    assert np.linalg.norm(e) <= 1e-7 * np.linalg.norm(ref["x"])
    # two V-cycles per iteration: about half the iterations of multigrid-PCG on the same system
    c = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post))
    c.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    xc = np.zeros(A.n_rows)
    c.solve(xc, b)
    assert s.getIters() < c.getIters()
    s.destroy()
    c.destroy()


@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched"])
def test_bcgs_with_gamg_matches_oracle_ksp_flavour(lin, case):
    """left-preconditioned (the monitored norm is that of M^-1 r), PETSc options"""
    from petibm_amd import capi
    dt = 0.01
    m, A, _ = poisson_system(CASES[case], dt=dt)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    text = ("-poisson_ksp_type bcgs\n-poisson_ksp_rtol 1.0E-10\n-poisson_ksp_atol 1.0E-50\n-poisson_pc_type gamg\n"
            "-poisson_pib_smoother jacobi\n-poisson_pib_presweeps 1\n-poisson_pib_postsweeps 1\n-poisson_pib_sweep_pairs 0\n")
    s = lin.LinSolverHIP("poisson", config_text=text)
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=1, pre=1, post=1, omega=0.9, coarsest_sweeps=32)
    ref = g.bcgs(A, b, norm="preconditioned", rtol=1e-10, atol=1e-50, maxit=200)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1e-8 * np.linalg.norm(b)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 4)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-7)
    s.destroy()


@pytest.mark.parametrize("flavour", ["amgx", "ksp"])
@pytest.mark.parametrize("case,pre,post", [("3d_uniform", 1, 1), ("3d_stretched", 2, 2), ("2d_stretched", 1, 1)])
def test_bicgstab_with_the_multigrid_and_a_pinned_row_matches_oracle(lin, case, pre, post, flavour):
    """Round 5 (a PIB_ERR_SUP until then): `solver=PBICGSTAB, preconditioner=AMG` on the system every `type: GPU` run of PetIBM
    hands over -- row 0 pinned (navierstokes.cpp:414-420).  The preconditioner of the pinned system as the CG path and the
    oracle apply it (oracle.c pcapply, nullspace 2): the cycle's right-hand side made compatible with the sum of ITS input,
    the output shifted by its value at cell 0, the pinned unknown keeping the input's value."""
    from petibm_amd import capi
    dt = 0.01
    m, A, _ = poisson_system(CASES[case], dt=dt, pinned=True)
    xs, b = rhs_for(A, zero_mean=False)
    b[0] = 0.0
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    if flavour == "amgx":
        text = gmg_cfg(pre=pre, post=post).replace("solver(solv)=PCG", "solver(solv)=PBICGSTAB") + "pib_initial_guess_nonzero=0\n"
        norm, tolr = "unpreconditioned", 1.5e-10
    else:
        text = ("-poisson_ksp_type bcgs\n-poisson_ksp_rtol 1.0E-10\n-poisson_ksp_atol 1.0E-50\n-poisson_pc_type gamg\n"
                f"-poisson_pib_smoother jacobi\n-poisson_pib_presweeps {pre}\n-poisson_pib_postsweeps {post}\n-poisson_pib_sweep_pairs 0\n")
        norm, tolr = "preconditioned", 1e-8
    s = lin.LinSolverHIP("poisson", config_text=text)
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_PINNED)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=2, pre=pre, post=post, omega=0.9, coarsest_sweeps=32)
    ref = g.bcgs(A, b, norm=norm, rtol=1e-10, atol=1e-50 if flavour == "ksp" else 0.0, dtol=1e300, maxit=200)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= tolr * np.linalg.norm(b)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 4)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-7)
    assert np.linalg.norm(x - ref["x"]) <= 1e-7 * np.linalg.norm(ref["x"])
    s.destroy()


@pytest.mark.parametrize("P,n,extra", [(2, (16, 16, 32), ""), (3, (32, 32, 36), "pib_agglomerate_below=100\n"),
                                       (2, (128, 16, 64), "pib_march_min_cells=0\npib_agglomerate_below=100\n")])
def test_pbicgstab_with_amg_on_loopback_slabs(lin, P, n, extra):
    """z-slabs over the loopback transport (the V-cycle's deep-halo plan under a second caller, the projection's sum through
    the all-reduce): the single rank's iteration count and solution"""
    from petibm_amd import capi
    import slab_plans as partition
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_multirank_loopback import _cfg, _run_ranks, _system
    dt = 0.01
    m, A, xs, b = _system(n, dt)
    w = [m.dL[3][d].true for d in range(m.dim)]
    plans = partition.all_plans(n, P)
    text = _cfg("AMG", extra=extra, sweeps=2).replace("solver(solv)=PCG", "solver(solv)=PBICGSTAB")

    def rank_fn(r, uid):
        pl = plans[r]
        s = LinSolverHIP("poisson", config_text=text, rank=r, nranks=P, uid=uid, device=0)
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(pl.n_local)
        s.solve(x, np.ascontiguousarray(b[pl.row0:pl.row0 + pl.n_local]))
        out = (x, s.getIters(), s.getResidualHistory())
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    x = np.concatenate([r[0] for r in res])
    assert len({r[1] for r in res}) == 1
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    s1 = LinSolverHIP("poisson", config_text=text)
    s1.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    x1 = np.zeros(A.n_rows)
    s1.solve(x1, b)
    assert iters_close(res[0][1], s1.getIters())
    k = min(len(res[0][2]), len(s1.getResidualHistory()), 4)
    assert np.allclose(res[0][2][:k], s1.getResidualHistory()[:k], rtol=1e-7)
    assert np.linalg.norm((x - x.mean()) - (x1 - x1.mean())) <= 1e-7 * np.linalg.norm(x1)
    s1.destroy()


@pytest.mark.parametrize("solver", ["PBICGSTAB", "PCG_SR"])
@pytest.mark.parametrize("P,n", [(2, (16, 16, 32)), (3, (32, 32, 36))])
def test_pinned_row_pairs_on_loopback_slabs(lin, P, n, solver):
    """The two pairs that refused a pinned row until round 5 -- BiCGStab + multigrid, single-reduction CG + multigrid -- on
    z-slabs: rank 0 owns the pinned row, the sum of the cycle's input and the cycle's value at cell 0 travel through
    all-reduces; every rank stops at the single rank's iteration, the solution is the single rank's."""
    from petibm_amd import capi
    import slab_plans as partition
    from petibm_amd.linsolver import LinSolverHIP
    from oracle import operators as oops
    from test_gpu_multirank_loopback import _cfg, _run_ranks, _system
    dt = 0.01
    m, A, xs, _ = _system(n, dt)
    A = oops.pin_row0(A)
    xs = xs - xs[0]
    b = clib.spmv(A, xs)
    w = [m.dL[3][d].true for d in range(m.dim)]
    plans = partition.all_plans(n, P)
    text = _cfg("AMG", extra="pib_agglomerate_below=100\n", sweeps=2)
    text = text.replace("solver(solv)=PCG", "solver(solv)=PBICGSTAB") if solver == "PBICGSTAB" else text + "pib_cg_single_reduction=1\n"

    def rank_fn(r, uid):
        pl = plans[r]
        s = LinSolverHIP("poisson", config_text=text, rank=r, nranks=P, uid=uid, device=0)
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_PINNED)
        x = np.zeros(pl.n_local)
        s.solve(x, np.ascontiguousarray(b[pl.row0:pl.row0 + pl.n_local]))
        out = (x, s.getIters(), s.getResidualHistory())
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    x = np.concatenate([r[0] for r in res])
    assert len({r[1] for r in res}) == 1 and x[0] == 0.0
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    s1 = LinSolverHIP("poisson", config_text=text)
    s1.assemblePoisson(n, w, dt, capi.NULLSPACE_PINNED)
    x1 = np.zeros(A.n_rows)
    s1.solve(x1, b)
    assert iters_close(res[0][1], s1.getIters())
    k = min(len(res[0][2]), len(s1.getResidualHistory()), 4)
    assert np.allclose(res[0][2][:k], s1.getResidualHistory()[:k], rtol=1e-7)
    assert np.linalg.norm(x - x1) <= 1e-7 * np.linalg.norm(x1)
    s1.destroy()
