"""The Chebyshev iteration for the velocity system (`-velocity_ksp_type chebyshev`, AmgX flavour `solver=CHEBYSHEV`) against
the oracle's restatement of KSPSolve_Chebyshev (oracle/csrc/oracle.c:orc_chebyshev; PETSc is not in /root/reference: parity
unpinned by the reference, pinned here by the residual contract and by scipy in tests/test_oracle_krylov.py).

The reference reaches any KSP type through KSPSetFromOptions (src/linsolver/linsolverksp.cpp:62-66); its example files pick
BiCGStab for the velocity system, which the strongly diagonally dominant operator A = I/dt - c nu L does not need: an
iteration without inner products moves a third of the bytes per product."""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops
from test_gpu_parity import STRETCHED_2D, stretched_3d, poisson_system, amgx_cfg, _a0_table, _outflow_3d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lin():
    from petibm_amd import linsolver
    return linsolver


def velocity_system(case, dt=0.05, cnu=0.5 * 0.2, seed=11):
    cfg = {"2d_stretched": STRETCHED_2D, "3d_stretched": stretched_3d()}[case]
    m, _, L = poisson_system(cfg)
    A = oops.create_velocity_operator(L, dt, cnu)
    us = np.random.default_rng(seed).uniform(-1, 1, A.n_rows)
    return m, A, us, clib.spmv(A, us)


KSP = ("-velocity_ksp_type chebyshev\n-velocity_ksp_atol 1.0E-50\n-velocity_ksp_rtol 1.0E-10\n-velocity_ksp_max_it {maxit}\n"
       "-velocity_pc_type jacobi\n{extra}")


@pytest.mark.parametrize("flavour", ["ksp", "ksp_unpreconditioned", "amgx", "ksp_none"])
@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched"])
def test_chebyshev_velocity_system_matches_oracle(lin, case, flavour):
    """Same iteration count, history and solution as the oracle with the Gershgorin bounds both compute from the matrix;
    zero initial guess (KSP) and the caller's x as the guess (AmgX flavour)."""
    m, A, us, b = velocity_system(case)
    x = np.zeros(A.n_rows)
    kw = dict(rtol=1e-10, atol=1e-50, dtol=1e300, maxit=400)  # (relative: an absolute 1e-12 sits at the rounding level of |b|)
    if flavour == "amgx":
        text = amgx_cfg(solver="CHEBYSHEV", pc="BLOCK_JACOBI", tol=1e-10, conv="RELATIVE_INI", maxit=400)
        x = 0.5 * us + 0.1
        ref = clib.chebyshev(A, b, x0=x.copy(), pc="jacobi", norm="unpreconditioned", **kw)
    elif flavour == "ksp_none":
        text = KSP.format(maxit=400, extra="").replace("pc_type jacobi", "pc_type none")
        rp, cl, vl = A.rowptr, A.col, A.val
        d = A.diagonal()
        off = np.add.reduceat(np.abs(vl), rp[:-1]) - np.abs(d)
        ref = clib.chebyshev(A, b, emin=float((d - off).min()), emax=float((d + off).max()), pc="none", norm="preconditioned", **kw)
    else:
        norm = "unpreconditioned" if flavour == "ksp_unpreconditioned" else "preconditioned"
        text = KSP.format(maxit=400, extra="-velocity_ksp_norm_type unpreconditioned\n" if norm == "unpreconditioned" else "")
        ref = clib.chebyshev(A, b, pc="jacobi", norm=norm, **kw)
    s = lin.LinSolverHIP("velocity", config_text=text)
    s.setMatrix(A)
    s.solve(x, b)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert s.getIters() == ref["iters"]
    h = np.asarray(s.getResidualHistory())
    assert len(h) == len(ref["history"]) and np.allclose(h, ref["history"], rtol=1e-7, atol=1e-13 * h[0])
    assert np.isclose(s.getResidual(), ref["rnorm"], rtol=1e-7, atol=1e-13 * h[0])
    assert np.linalg.norm(x - ref["x"]) <= 1e-12 * np.linalg.norm(us)
    assert np.linalg.norm(clib.spmv(A, x) - b) <= 1e-9 * np.linalg.norm(b)
    s.destroy()


def test_chebyshev_with_explicit_eigenvalues_and_iteration_limit(lin):
    """-ksp_chebyshev_eigenvalues emin,emax (KSPChebyshevSetEigenvalues) / cheby_min_lambda, cheby_max_lambda are used as
    given; running into max_it is KSP_DIVERGED_ITS after the closing residual, an error like every other solver's."""
    from petibm_amd.capi import PibError
    m, A, us, b = velocity_system("3d_stretched")
    rho = clib.gershgorin_jacobi(A)
    lo, hi = 0.9 * (1.0 - rho), 1.1 * (1.0 + rho)
    ref = clib.chebyshev(A, b, emin=lo, emax=hi, pc="jacobi", norm="preconditioned", rtol=1e-10, atol=1e-50, maxit=400)
    s = lin.LinSolverHIP("velocity", config_text=KSP.format(maxit=400, extra=f"-velocity_ksp_chebyshev_eigenvalues {lo!r},{hi!r}\n"))
    s.setMatrix(A)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    assert s.getIters() == ref["iters"] and np.allclose(s.getResidualHistory(), ref["history"], rtol=1e-7, atol=1e-13 * ref["history"][0])
    s.destroy()
    ref = clib.chebyshev(A, b, x0=np.zeros(A.n_rows), emin=lo, emax=hi, pc="jacobi", norm="unpreconditioned", rtol=1e-10, atol=1e-50, dtol=1e300, maxit=400)
    text = amgx_cfg(solver="CHEBYSHEV", pc="BLOCK_JACOBI", tol=1e-10, conv="RELATIVE_INI", maxit=400,
                    extra=f"solv:cheby_min_lambda={lo!r}\nsolv:cheby_max_lambda={hi!r}\n")
    s = lin.LinSolverHIP("velocity", config_text=text)
    s.setMatrix(A)
    x2 = np.zeros(A.n_rows)
    s.solve(x2, b)
    assert s.getIters() == ref["iters"] and np.linalg.norm(x2 - ref["x"]) <= 1e-12 * np.linalg.norm(x)
    s.destroy()
    for maxit in (5, 6):  # odd and even: the closing residual falls on either half of the replayed pair
        ref = clib.chebyshev(A, b, pc="jacobi", norm="preconditioned", rtol=1e-10, atol=1e-50, maxit=maxit)
        assert ref["reason"] == -3 and ref["iters"] == maxit
        s = lin.LinSolverHIP("velocity", config_text=KSP.format(maxit=maxit, extra=""))
        s.setMatrix(A)
        x = np.zeros(A.n_rows)
        with pytest.raises(PibError):
            s.solve(x, b)
        assert s.getIters() == maxit and s.getReason() == -3
        assert np.allclose(s.getResidualHistory(), ref["history"], rtol=1e-8)
        assert np.linalg.norm(x - ref["x"]) <= 1e-12 * np.linalg.norm(us)
        s.destroy()


def test_chebyshev_refuses_what_it_cannot_bound(lin):
    """The Poisson operator is not strictly diagonally dominant: without explicit bounds the solver says so instead of
    iterating on a guess; the multigrid is not a preconditioner of this iteration."""
    from petibm_amd import capi
    from petibm_amd.capi import PibError
    m, DBNG, L = poisson_system(STRETCHED_2D)
    pinned = oops.pin_row0(DBNG)
    s = lin.LinSolverHIP("poisson", config_text="-poisson_ksp_type chebyshev\n-poisson_pc_type jacobi\n")
    s.setMatrix(pinned)
    x = np.zeros(pinned.n_rows)
    with pytest.raises(PibError) as ei:
        s.solve(x, np.ones(pinned.n_rows))
    assert ei.value.code == capi.ERR_SUP and "diagonally dominant" in str(ei.value)
    s.destroy()
    s = lin.LinSolverHIP("poisson", config_text="-poisson_ksp_type chebyshev\n-poisson_pc_type gamg\n")
    with pytest.raises(PibError) as ei:
        s.setMatrix(pinned)
        s.solve(x, np.ones(pinned.n_rows))
    assert ei.value.code == capi.ERR_SUP
    s.destroy()


@pytest.mark.parametrize("case", ["3d_stretched", "3d_outflow"])
def test_chebyshev_on_the_matrix_free_velocity_operator(lin, case):
    """pib_assemble_velocity's operator (products from the mesh tables): the oracle's iteration count and solution."""
    cfg = {"3d_stretched": stretched_3d(), "3d_outflow": _outflow_3d()}[case]
    m = omesh.create_mesh(cfg)
    L = oops.create_laplacian(m)
    dt, cnu = 0.004, 0.5 * 0.01
    A = oops.create_velocity_operator(L, dt, cnu)
    s = lin.LinSolverHIP("velocity", config_text=amgx_cfg(solver="CHEBYSHEV", pc="BLOCK_JACOBI", tol=1e-12, conv="ABSOLUTE", maxit=500))
    n = [int(v) for v in m.n[3][: m.dim]]
    s.assembleVelocity(n, [m.dL[3][d].true for d in range(m.dim)], m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
    us = np.random.default_rng(2).uniform(-1, 1, A.n_rows)
    b = clib.spmv(A, us)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    ref = clib.chebyshev(A, b, x0=np.zeros(A.n_rows), pc="jacobi", norm="unpreconditioned", rtol=0.0, atol=1e-12, dtol=1e300, maxit=500)
    assert s.getIters() == ref["iters"] and s.getIters() < 40
    assert np.linalg.norm(x - ref["x"]) <= 1e-12 * np.linalg.norm(us)
    assert np.linalg.norm(x - us) <= 1e-10 * np.linalg.norm(us)
    # second solve from the first's solution (the time loop's pattern): converged at once, x unchanged
    x2 = x.copy()
    s.solve(x2, b)
    assert s.getIters() <= 2 and np.linalg.norm(x2 - x) <= 1e-12 * np.linalg.norm(x)
    s.destroy()


@pytest.mark.parametrize("pc", ["BLOCK_JACOBI", "NOSOLVER"])
@pytest.mark.parametrize("n,per", [((128, 16, 24), (True, True, True)), ((128, 12, 10), (False, False, False)),
                                   ((256, 19, 9), (False, True, False)), ((128, 8, 40), (True, False, True))])
def test_chebyshev_update_inside_the_velocity_product(lin, n, per, pc):
    """velstencil.hip k_vel_product<2>: the update p[kp1] = (1 - omega) p[km1] + omega p[k] + omega scale M^-1 (b - A p[k]) and
    the two norms in the product's launch (every cell by the expressions of the separate pass: the iterates are the same
    bits; the norms are grouped by tile: equal to rounding) -- periodic box, wall-bounded mesh with partial tiles, mixed;
    with and without the Jacobi sweep."""
    cfg = omesh.periodic_config(n, per)
    m = omesh.create_mesh(cfg)
    dt, cnu = 0.004, 0.5 * 0.01
    b = np.random.default_rng(5).uniform(-1, 1, m.UN)
    out = []
    for extra in ("pib_march_min_cells=0\n", "pib_march_min_cells=0\npib_fuse_chebyshev_update=0\n", "pib_matrix_free_velocity=0\n"):
        s = lin.LinSolverHIP("velocity", config_text=amgx_cfg(solver="CHEBYSHEV", pc=pc, tol=1e-11, conv="RELATIVE_INI", maxit=500, extra=extra))
        s.setPeriodic(per)
        s.assembleVelocity(list(n), [m.dL[3][d].true for d in range(m.dim)], m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
        x = np.zeros(m.UN)
        s.solve(x, b)
        out.append((x, s.getIters(), np.asarray(s.getResidualHistory())))
        s.destroy()
    fused, split, csr = out
    assert split[1] == csr[1] >= 3 and np.array_equal(split[0], csr[0]) and np.array_equal(split[2], csr[2])
    assert fused[1] == split[1] and np.array_equal(fused[0], split[0])
    assert np.allclose(fused[2], split[2], rtol=1e-12, atol=0.0)


@pytest.mark.parametrize("P", [2, 3])
def test_chebyshev_on_slabs(P):
    """The iteration on z-slabs through the loopback transport: one exchange of the iterate's boundary planes and ONE
    all-reduce (the norm) per pass -- BiCGStab: two exchanges, three all-reduces per iteration -- and the single rank's
    iteration count and solution, with the update inside the product's launch and as a separate pass."""
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_multirank_loopback import _run_ranks, _velocity_slab_indices
    m = omesh.create_mesh(stretched_3d((128, 10, 16)))
    dt, cnu = 0.004, 0.5 * 0.01
    A = oops.create_velocity_operator(oops.create_laplacian(m), dt, cnu)
    us = np.random.default_rng(12).uniform(-1, 1, A.n_rows)
    b = clib.spmv(A, us)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    text = amgx_cfg(solver="CHEBYSHEV", pc="BLOCK_JACOBI", tol=1e-11, conv="RELATIVE_INI", maxit=500) + "pib_march_min_cells=0\n"
    own = [_velocity_slab_indices(m, P, r) for r in range(P)]

    def rank_fn(r, uid, extra=""):
        s = LinSolverHIP("velocity", config_text=text + extra, rank=r, nranks=P, uid=uid, device=0)
        s.assembleVelocity(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
        x = np.zeros(own[r].size)
        s.solve(x, np.ascontiguousarray(b[own[r]]))
        out = x, s.getIters(), np.asarray(s.getResidualHistory()), s.counters()
        s.destroy()
        return out

    s1 = LinSolverHIP("velocity", config_text=text)
    s1.assembleVelocity(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
    x1 = np.zeros(A.n_rows)
    s1.solve(x1, b)
    its1, h1 = s1.getIters(), np.asarray(s1.getResidualHistory())
    s1.destroy()
    ref = clib.chebyshev(A, b, x0=np.zeros(A.n_rows), pc="jacobi", norm="unpreconditioned", rtol=1e-11, atol=0.0, dtol=1e300, maxit=500)
    assert its1 == ref["iters"] and np.linalg.norm(x1 - ref["x"]) <= 1e-12 * np.linalg.norm(us)
    for extra in ("", "pib_fuse_chebyshev_update=0\n"):
        res = _run_ranks(P, lambda r, uid: rank_fn(r, uid, extra))
        x = np.empty(A.n_rows)
        for r in range(P):
            x[own[r]] = res[r][0]
            assert res[r][1] == its1 and np.allclose(res[r][2], h1, rtol=1e-10)
        assert np.array_equal(x, x1)


CHEB_VEL = ("config_version=2\nsolver(solv)=CHEBYSHEV\nsolv:max_iters=1000\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
            "solv:tolerance=1e-11\nsolv:norm=L2\nsolv:store_res_history=1\nsolv:preconditioner(prec)=BLOCK_JACOBI\n"
            "prec:relaxation_factor=0.9\n")


@pytest.mark.parametrize("case,P", [("3d_cavity", 1), ("3d_convective_outlet", 1), ("2d_cavity", 1), ("3d_periodic_box", 1),
                                    ("3d_cavity", 2), ("3d_convective_outlet", 3), ("3d_periodic_box", 2), ("2d_periodic_y", 2)])
def test_time_step_with_the_chebyshev_velocity_file(case, P):
    """NavierStokesSolver::advance with `solver=CHEBYSHEV` in velocity_solver.info: three steps land where the BiCGStab file's
    do (both solves to their tolerances), on one rank and on loopback slabs (one exchange + one all-reduce per pass)."""
    from petibm_amd.navierstokes import NavierStokesSolver
    from test_gpu_navierstokes import AMGX_P, KSP_P, VEL
    from test_gpu_navierstokes_slabs import CASES
    from test_gpu_multirank_loopback import _run_ranks
    make, pinned = CASES[case]
    cfg = make()
    pcfg = AMGX_P if pinned else KSP_P
    one = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=pcfg)
    rng = np.random.default_rng(11)
    U0 = 0.1 * rng.uniform(-1, 1, one.UN)
    p0 = 0.1 * rng.uniform(-1, 1, one.pN)
    if "convective" in case:
        U0[: int(np.prod(one._field_shape(0)))] += 1.0
    one.setState(U0, p0)
    one.advance(3)
    U3, p3 = one.getState()
    its_b = one.linSolversInfo()[1]
    one.destroy()

    def rank_fn(r, uid):
        kw = dict(device=0, rank=r, nranks=P, uid=uid) if P > 1 else {}
        s = NavierStokesSolver(cfg, velocity_cfg=CHEB_VEL, poisson_cfg=pcfg, **kw)
        s.setState(s.ownedVelocity(U0) if P > 1 else U0, s.ownedPressure(p0) if P > 1 else p0)
        s.advance(3)
        U, p = s.getState()
        cut = (s.ownedVelocity(U3), s.ownedPressure(p3)) if P > 1 else (U3, p3)
        its = s.linSolversInfo()[1]
        s.destroy()
        return U, p, cut, its

    res = _run_ranks(P, rank_fn) if P > 1 else [rank_fn(0, None)]
    for U, p, (cU, cp), its in res:
        assert 2 <= its < 60 and its_b >= 1
        assert np.allclose(U, cU, rtol=0, atol=1e-9)
    pg, pr = np.concatenate([r[1] for r in res]), np.concatenate([r[2][1] for r in res])
    assert np.allclose(pg - pg.mean(), pr - pr.mean(), rtol=0, atol=1e-7)
