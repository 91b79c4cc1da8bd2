"""setMatrix is all an unchanged PetIBM gives a `type: GPU` solver (LinSolverAmgX::setMatrix -> AmgXSolver::setA,
src/linsolver/linsolveramgx.cpp:84): with an AMG entry in the solver file the backend recovers the mesh structure of the
Poisson operator from the CSR itself (csrc/structure.cpp) and runs its geometric multigrid.

Bars: the recovered structure solves exactly like the registered one (same iteration count, residual contract recomputed
by the oracle); anything that is not such an operator is left without structure -- no error at setMatrix, the reference's
ordering error (PETSC_ERR_ORDER) at solve.
"""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops
from test_gpu_parity import STRETCHED_2D, stretched_3d, poisson_system, rhs_for, amgx_cfg
from test_gpu_multirank_loopback import _run_ranks, _cfg

pytestmark = pytest.mark.gpu

AMG = "prec:cycle=V\nprec:presweeps=1\nprec:postsweeps=1\nprec:smoother(smooth)=BLOCK_JACOBI\nsmooth:relaxation_factor=0.9\n"


def _hint(m, dt, dim):
    w = [m.dL[3][d].true for d in range(dim)]
    g = [dt * (1.0 / (0.5 * (wd[1:] + wd[:-1]))) for wd in w]
    return [int(v) for v in m.n[3][:dim]], w, g


@pytest.mark.parametrize("case", ["2d", "3d", "3d_uniform"])
@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("idx", ["i64", "i32"])
def test_structure_recovered_from_the_matrix_solves_like_the_registered_one(case, pinned, idx):
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    cfg = {"2d": STRETCHED_2D, "3d": stretched_3d((20, 18, 14)), "3d_uniform": omesh.uniform_config((16, 12, 20))}[case]
    dim = 2 if case == "2d" else 3
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt, pinned=pinned)
    xs, b = rhs_for(A, zero_mean=not pinned)
    if pinned:
        b[0] = 0.0
    A64 = A  # the oracle's routines take 64-bit indices
    if idx == "i32":
        A = oops.CSR(A.n_rows, A.n_cols, A.rowptr.astype(np.int32), A.col.astype(np.int32), A.val)
    text = amgx_cfg(pc="AMG", tol=1e-10, extra=AMG + "pib_initial_guess_nonzero=0\n")
    s = LinSolverHIP("poisson", config_text=text)
    s.setMatrix(A)
    st = s.gridStructure()
    n, w, g = _hint(m, dt, dim)
    assert st is not None and st["detected"] and st["dim"] == dim and list(st["n"]) == n
    assert st["nullspace"] == (capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    it_detected = s.getIters()
    assert np.linalg.norm(b - clib.spmv(A64, x)) <= 1.5e-10 * np.linalg.norm(b)
    if pinned:
        assert x[0] == 0.0
    # the same solve with the structure registered by the application (pib_set_grid_hint)
    t = LinSolverHIP("poisson", config_text=text + "pib_detect_structure=0\n")
    t.setMatrix(A)
    assert t.gridStructure() is None
    t.setGridHint(n, w, g, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    assert not t.gridStructure()["detected"]
    y = np.zeros(A.n_rows)
    t.solve(y, b)
    assert t.getIters() == it_detected
    assert np.linalg.norm((x - x.mean()) - (y - y.mean())) <= 1e-8 * np.linalg.norm(y - y.mean())
    s.destroy()
    t.destroy()


@pytest.mark.parametrize("n,per", [((12, 10, 9), (True, True, True)), ((10, 8, 12), (False, False, True)), ((9, 12, 8), (True, False, False)),
                                   ((8, 9, 10), (False, True, True)), ((14, 12), (True, True)), ((12, 10), (False, True)),
                                   ((16, 9), (True, False))])
def test_structure_of_periodic_meshes_recovered_from_the_matrix(n, per):
    """The Taylor-Green cases of the reference are periodic boxes: a periodic direction shows in the matrix as the wrapped
    neighbour of its first cell (one more |column - row| in the first rows) and contributes one more face factor, the one
    across the seam.  Nothing but setMatrix: dimensions, sizes AND the periodic directions come out of the entries, and
    the solve is the one of the registered structure (setPeriodic + grid hint)."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    dt, dim = 0.01, len(n)
    ratios = [1.0 if p else 1.0 + 0.01 * (d + 1) for d, p in enumerate(per)]
    m = omesh.create_mesh(omesh.periodic_config(n, per, ratios=ratios))
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.005)
    xs, b = rhs_for(A)
    text = amgx_cfg(pc="AMG", tol=1e-10, extra=AMG + "pib_initial_guess_nonzero=0\n")
    s = LinSolverHIP("poisson", config_text=text)
    s.setMatrix(A)
    st = s.gridStructure()
    assert st is not None and st["detected"] and st["dim"] == dim and tuple(st["n"]) == tuple(n)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    w = [m.dL[3][d].true for d in range(dim)]
    g = [dt * (1.0 / (0.5 * (wd[1:] + wd[:-1]))) for wd in w]
    g = [np.concatenate([gd, [dt / (0.5 * (wd[0] + wd[-1]))]]) if per[d] else gd for d, (gd, wd) in enumerate(zip(g, w))]
    t = LinSolverHIP("poisson", config_text=text + "pib_detect_structure=0\n")
    t.setPeriodic(per)
    t.setMatrix(A)
    t.setGridHint(list(n), w, g, capi.NULLSPACE_CONSTANT)
    y = np.zeros(A.n_rows)
    t.solve(y, b)
    assert t.getIters() == s.getIters()
    assert np.linalg.norm((x - x.mean()) - (y - y.mean())) <= 1e-8 * np.linalg.norm(y - y.mean())
    s.destroy()
    t.destroy()


@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched", "3d_outflow", "3d_all_periodic", "2d_periodic_y", "3d_periodic_xz",
                                  "3d_big_march"])
def test_velocity_structure_recovered_from_the_matrix(case):
    """vSolver->setMatrix(A) is all an unchanged PetIBM gives the velocity solver (navierstokes.cpp:345).  With a Jacobi
    preconditioner the backend recovers the operator's structure from the CSR -- field sizes and periodic directions from
    the offsets of the first rows, the coefficient tables from one line of entries per field and direction, the walls' ghost
    folds from boundary diagonals -- verifies the matrix-free product against the CSR SpMV on the device, and runs BiCGStab on
    it: same iteration count as the CSR products, solutions equal to rounding (the diagonal is re-formed from the tables)."""
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_parity import _outflow_3d
    extra = ""
    if case == "3d_all_periodic":
        cfg, per = omesh.periodic_config((10, 8, 9), (True, True, True)), (True, True, True)
    elif case == "2d_periodic_y":
        cfg, per = omesh.periodic_config((14, 12), (False, True)), (False, True)
    elif case == "3d_periodic_xz":
        cfg, per = omesh.periodic_config((9, 10, 8), (True, False, True), ratios=(1.0, 1.02, 1.0)), (True, False, True)
    elif case == "3d_big_march":  # the one-launch marching product (and the BiCGStab built on it) with recovered tables
        cfg, per = omesh.periodic_config((128, 16, 24), (False, True, False), ratios=(1.002, 1.0, 1.01)), (False, True, False)
        extra = "pib_march_min_cells=0\n"
    else:
        cfg = {"2d_stretched": STRETCHED_2D, "3d_stretched": stretched_3d((11, 9, 10)), "3d_outflow": _outflow_3d()}[case]
        per = (False,) * len(cfg["mesh"])
    m = omesh.create_mesh(cfg)
    dt, cnu = 0.004, 0.5 * 0.01
    A = oops.create_velocity_operator(oops.create_laplacian(m), dt, cnu)
    b = np.random.default_rng(4).uniform(-1, 1, m.UN)
    out = []
    for mf in (1, 0):
        s = LinSolverHIP("velocity", config_text=amgx_cfg(solver="PBICGSTAB", pc="BLOCK_JACOBI", tol=1e-12, conv="ABSOLUTE", maxit=500,
                                                          extra=extra + f"pib_matrix_free_velocity={mf}\n"))
        s.setMatrix(A)
        st = s.velocityStructure()
        if mf:
            assert st is not None and st["detected"] and st["dim"] == m.dim
            assert st["n"] == tuple(int(v) for v in m.n[3][: m.dim]) and st["periodic"] == tuple(per)
        else:
            assert st is None
        y = np.empty(m.UN)
        s.matMult(b, y)
        x = np.zeros(m.UN)
        s.solve(x, b)
        out.append((x, s.getIters(), y))
        s.destroy()
    assert np.array_equal(out[0][2], out[1][2])  # pib_mat_mult stays the CSR product
    assert out[0][1] == out[1][1] and out[0][1] >= 2
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-11 * np.abs(out[1][0]).max()
    assert np.linalg.norm(b - clib.spmv(A, out[0][0])) <= 1e-11 * np.linalg.norm(b)


def test_a_matrix_that_only_looks_like_the_velocity_operator_keeps_its_csr_products():
    from petibm_amd.linsolver import LinSolverHIP
    m = omesh.create_mesh(stretched_3d((9, 8, 7)))
    A = oops.create_velocity_operator(oops.create_laplacian(m), 0.004, 0.005)
    B = A.copy()
    B.val = B.val.copy()
    rows = np.repeat(np.arange(B.n_rows), np.diff(B.rowptr))
    hit = np.nonzero((rows == 300) & (B.col == 301))[0][0]
    B.val[hit] *= 1.0 + 1e-6  # one entry off the tensor structure: the device check against the CSR refuses the tables
    s = LinSolverHIP("velocity", config_text=amgx_cfg(solver="PBICGSTAB", pc="BLOCK_JACOBI", tol=1e-12, conv="ABSOLUTE", maxit=500))
    s.setMatrix(B)
    assert s.velocityStructure() is None
    b = np.random.default_rng(5).uniform(-1, 1, m.UN)
    x = np.zeros(m.UN)
    s.solve(x, b)
    assert np.linalg.norm(b - clib.spmv(B, x)) <= 1e-11 * np.linalg.norm(b)
    # the Poisson matrix with a Jacobi preconditioner is not mistaken for it either
    _, P, _ = poisson_system(stretched_3d((9, 8, 7)))
    s.setMatrix(P)
    assert s.velocityStructure() is None
    s.destroy()


def test_matrices_that_are_not_the_poisson_operator_stay_without_structure():
    from petibm_amd import capi
    from petibm_amd.capi import PibError
    from petibm_amd.linsolver import LinSolverHIP
    text = amgx_cfg(pc="AMG", tol=1e-10, extra=AMG)
    m, A, L = poisson_system(stretched_3d((10, 9, 8)))
    # (i) one entry (and its diagonal) off: factorises almost, the device check against the CSR refuses it
    B = A.copy()
    B.val = B.val.copy()
    rows = np.repeat(np.arange(B.n_rows), np.diff(B.rowptr))
    hit = np.nonzero((rows == 400) & (B.col == 401))[0][0]
    B.val[hit] *= 1.001
    s = LinSolverHIP("poisson", config_text=text)
    s.setMatrix(B)
    assert s.gridStructure() is None
    with pytest.raises(PibError) as ei:
        s.solve(np.zeros(B.n_rows), np.ones(B.n_rows))
    assert ei.value.code == capi.ERR_ORDER
    # (ii) the velocity operator (packed u,v,w ordering, row-scaled Laplacian): a different pattern altogether
    V = oops.create_velocity_operator(L, 0.01, 0.5 * 0.01) if hasattr(oops, "create_velocity_operator") else None
    if V is not None:
        s.setMatrix(V)
        assert s.gridStructure() is None
    # (iii) with a Jacobi preconditioner nothing is searched for
    t = LinSolverHIP("poisson", config_text=amgx_cfg(pc="BLOCK_JACOBI"))
    t.setMatrix(A)
    assert t.gridStructure() is None
    # (iv) the good matrix right after a bad one on the same solver object
    s.setMatrix(A)
    assert s.gridStructure() is not None and s.gridStructure()["detected"]
    s.destroy()
    t.destroy()


@pytest.mark.parametrize("P,n,pinned", [(2, (16, 12, 16), True), (3, (12, 10, 18), False), (2, (24, 20), True),
                                        (4, (16, 16, 4), False)])
def test_structure_recovered_on_slabs(P, n, pinned):
    """every rank hands over its rows only (MatMPIAIJGetLocalMat layout); the lines of entries are gathered"""
    from petibm_amd import capi
    import slab_plans as partition
    from petibm_amd.linsolver import LinSolverHIP
    dim, dt = len(n), 0.02
    cfg = omesh.uniform_config(n)
    m = omesh.create_mesh(cfg)
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.5e-2)
    if pinned:
        A = oops.pin_row0(A)
    xs = np.random.default_rng(3).uniform(-1, 1, m.pN)
    if pinned:
        xs[0] = 0.0
    else:
        xs -= xs.mean()
    b = clib.spmv(A, xs)
    plans = partition.all_plans(n, P)

    def rank_fn(r, uid):
        pl = plans[r]
        s = LinSolverHIP("poisson", config_text=_cfg("AMG", tol=1e-11), rank=r, nranks=P, uid=uid, device=0)
        r0, r1 = pl.row0, pl.row0 + pl.n_local
        p0, p1 = A.rowptr[r0], A.rowptr[r1]
        local = oops.CSR(pl.n_local, A.n_cols, A.rowptr[r0:r1 + 1] - p0, A.col[p0:p1], A.val[p0:p1])
        s.setMatrix(local, row0=r0, n_global=A.n_rows)
        st = s.gridStructure()
        x = np.zeros(pl.n_local)
        s.solve(x, np.ascontiguousarray(b[r0:r1]))
        its = s.getIters()
        s.destroy()
        return x, its, st

    res = _run_ranks(P, rank_fn)
    for r in res:
        assert r[2] is not None and r[2]["detected"] and list(r[2]["n"]) == list(n)
        assert r[2]["nullspace"] == (capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    x = np.concatenate([r[0] for r in res])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-11 * np.linalg.norm(b)
    assert len({r[1] for r in res}) == 1 and res[0][1] < 40


@pytest.mark.parametrize("case", ["uniform_3d", "stretched_3d", "stretched_2d", "periodic_3d"])
def test_multigrid_hierarchy_is_the_oracles(case):
    """pib_get_multigrid_levels (what AmgX's print_grid_stats shows): the selective coarsening of the device build produces
    the level sizes of the oracle's restatement -- plain halving on a uniform mesh, the finest cells first on a stretched one."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    per = None
    if case == "uniform_3d":
        cfg = omesh.uniform_config((32, 24, 40))  # unit cube: widths 1/32, 1/24, 1/40 -- the finer directions coarsen first
    elif case == "stretched_3d":
        cfg = stretched_3d((40, 36, 28))
    elif case == "periodic_3d":
        cfg, per = omesh.periodic_config((24, 20, 16), (True, False, True)), (True, False, True)
    else:
        cfg = STRETCHED_2D
    m = omesh.create_mesh(cfg)
    dim = m.dim
    n = [int(v) for v in m.n[3][:dim]]
    w = [m.dL[3][d].true for d in range(dim)]
    s = LinSolverHIP("poisson", config_text=amgx_cfg(pc="AMG", tol=1e-10, extra=AMG))
    if per:
        s.setPeriodic(per)
    s.assemblePoisson(n, w, 0.01, capi.NULLSPACE_CONSTANT)
    got = s.multigridLevels()
    g = clib.GMG(n, w, 0.01, nullspace=1, pre=1, post=1, omega=0.9, coarsest_sweeps=2, periodic=per or (False,) * dim)
    want = [tuple(int(v) for v in g.level_size(l)) for l in range(g.num_levels())]
    if dim == 2:  # the oracle keeps a 2-D grid as (nx, 1, ny)
        want = [(t[0], t[2]) if len(t) == 3 else t for t in want]
    assert got[0][:dim] == tuple(n) and len(got) >= 3
    assert [t[:dim] for t in got] == [tuple(t[:dim]) for t in want], (got, want)
    if case == "uniform_3d":
        assert got[1] == (16, 24, 20) and got[2] == (8, 12, 10)
    s.destroy()
    t = LinSolverHIP("velocity", config_text=amgx_cfg(solver="PBICGSTAB", pc="BLOCK_JACOBI"))
    assert t.multigridLevels() == []
    t.destroy()
