"""GPU parity tests: the HIP path (through the C ABI) against the oracle.

Reads like the reference's own unit tests would if it had any for the solver
path (it has none -- SURVEY.md 4): build a mesh from a YAML-shaped config,
create operators, createLinSolver / setMatrix / solve / getIters / getResidual.

Bars:  integer / index work and SpMV: bit-exact.  Krylov iterates: the
reduction order differs from the CPU's, so residual histories are compared to
1e-9 relative and solutions to 1e-8 (north-star tolerance 1e-10 on the
relative residual, asserted against an oracle re-computation of b - A x).
"""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops

pytestmark = pytest.mark.gpu

STRETCHED_2D = {
    "mesh": [
        {"direction": "x", "start": 0.1, "subDomains": [
            {"end": 1.6, "cells": 4, "stretchRatio": 0.5},
            {"end": 1.9, "cells": 3, "stretchRatio": 1},
            {"end": 5.0, "cells": 5, "stretchRatio": 2.0}]},
        {"direction": "y", "start": 0.05, "subDomains": [
            {"end": 0.8625, "cells": 4, "stretchRatio": 0.666666666666666},
            {"end": 1.1625, "cells": 3, "stretchRatio": 1.0},
            {"end": 1.975, "cells": 4, "stretchRatio": 1.5}]}],
    "flow": {"boundaryConditions": [
        {"location": loc, "u": ["DIRICHLET", 0.0], "v": ["DIRICHLET", 0.0]}
        for loc in ("xMinus", "xPlus", "yMinus", "yPlus")]},
}


def stretched_3d(n=(20, 18, 14), r=(1.08, 0.93, 1.05)):
    names = "xyz"
    mesh = [{"direction": names[d], "start": -1.0 + 0.25 * d,
             "subDomains": [{"end": 0.0, "cells": n[d] // 2, "stretchRatio": 1.0 / r[d]},
                            {"end": 1.5 + d, "cells": n[d] - n[d] // 2, "stretchRatio": r[d]}]}
            for d in range(3)]
    cfg = omesh.uniform_config(n)
    cfg["mesh"] = mesh
    return cfg


def poisson_system(cfg, dt=0.01, nu=0.01, pinned=False):
    m = omesh.create_mesh(cfg)
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, DBNG = oops.create_poisson_operator(D, G, L, dt, 0.5 * nu)
    if pinned:
        DBNG = oops.pin_row0(DBNG)
    return m, DBNG, L


def amgx_cfg(solver="PCG", pc="NOSOLVER", tol=1e-10, conv="RELATIVE_INI", maxit=5000, extra=""):
    return (f"config_version=2\nsolver(solv)={solver}\nsolv:max_iters={maxit}\nsolv:monitor_residual=1\n"
            f"solv:convergence={conv}\nsolv:tolerance={tol}\nsolv:norm=L2\nsolv:store_res_history=1\n"
            f"solv:preconditioner(prec)={pc}\nprec:relaxation_factor=1.0\n{extra}")


@pytest.fixture(scope="module")
def lin():
    from petibm_amd import linsolver
    return linsolver


def iters_close(got, ref):
    """Krylov iteration counts: the GPU's reduction tree rounds differently from the CPU's running sums,
    and CG on the ill-conditioned stretched meshes amplifies that: allow 1 iteration or 3 %."""
    return abs(got - ref) <= max(1, int(np.ceil(0.03 * ref)))


def rhs_for(A, seed=20260928, zero_mean=True):
    rng = np.random.default_rng(seed)
    xs = rng.uniform(-1.0, 1.0, A.n_rows)
    if zero_mean:
        xs -= xs.mean()
    return xs, clib.spmv(A, xs)


# ------------------------------------------------------------- assembly
@pytest.mark.parametrize("case", ["2d_stretched", "3d_uniform", "3d_stretched"])
@pytest.mark.parametrize("pinned", [False, True])
def test_device_poisson_assembly_bit_exact(lin, case, pinned):
    from petibm_amd import capi
    cfg = {"2d_stretched": STRETCHED_2D, "3d_uniform": omesh.uniform_config((12, 10, 9)),
           "3d_stretched": stretched_3d()}[case]
    dt = 0.0125
    m, DBNG, _ = poisson_system(cfg, dt=dt, pinned=pinned)
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg())
    n = [int(v) for v in m.n[3][: m.dim]]
    s.assemblePoisson(n, [m.dL[3][d].true for d in range(m.dim)], dt,
                      capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    rp, cl, vl = s.getCSR()
    assert np.array_equal(rp, DBNG.rowptr)
    assert np.array_equal(cl, DBNG.col)
    assert np.array_equal(vl, DBNG.val)  # bit-exact: same floating-point order as D*(dt*G)
    s.destroy()


# ------------------------------------------------------------------ SpMV
@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched"])
def test_spmv_bit_exact(lin, case):
    cfg = {"2d_stretched": STRETCHED_2D, "3d_stretched": stretched_3d()}[case]
    m, DBNG, L = poisson_system(cfg)
    for A in (DBNG, L, oops.create_velocity_operator(L, 0.01, 0.005)):
        s = lin.LinSolverHIP("velocity", config_text=amgx_cfg())
        s.setMatrix(A)
        x = np.random.default_rng(7).uniform(-1, 1, A.n_cols)
        y = np.empty(A.n_rows)
        s.matMult(x, y)
        assert np.array_equal(y, clib.spmv(A, x))
        s.destroy()


def expected_product_format(A, compress):
    """the rules of kernels_spmv.hip restated: 0 (row patterns) when every row has at most 8 entries and every 256-row block at
    most 16 distinct rows-as-lists-of-offsets (tried first); else 1 (column codes) when every block has at most 16 distinct offsets; else 4"""
    if compress == 0:
        return 4
    rp, cl = np.asarray(A.rowptr), np.asarray(A.col)
    pat_ok, code_ok = True, True
    for b0 in range(0, A.n_rows, 256):
        pats, offs = set(), set()
        for r in range(b0, min(b0 + 256, A.n_rows)):
            d = tuple(int(c) - r for c in cl[rp[r]:rp[r + 1]])
            pats.add(d)
            offs.update(d)
            pat_ok = pat_ok and len(d) <= 8
        pat_ok = pat_ok and len(pats) <= 16
        code_ok = code_ok and len(offs) <= 16
    if compress == 2 and pat_ok:
        return 0
    return 1 if code_ok else 4


def test_describe_says_what_runs(lin, capsys):
    """pib_describe / printInfo: the product form chosen at setMatrix, the steps a sweep of the file runs as, the null-space
    convention and where the structure came from -- through the API, not on stderr (getType keeps the reference's string)."""
    m, DBNG, _ = poisson_system(stretched_3d())
    gmg = ("config_version=2\nsolver(s)=PCG\ns:max_iters=100\ns:tolerance=1e-8\ns:convergence=RELATIVE_INI\ns:norm=L2\n"
           "s:monitor_residual=1\ns:store_res_history=1\ns:preconditioner(p)=AMG\np:presweeps=1\np:postsweeps=1\n")
    forms = {2: "csr_row_patterns", 1: "csr_column_codes", 0: "csr_int32_columns"}
    for compress, form in forms.items():
        s = lin.LinSolverHIP("poisson", config_text=gmg + f"pib_compress_columns={compress}\npib_matrix_free_poisson=0\n")
        first = s.describe().splitlines()[0]
        assert "product=none" in first and "method=cg" in first and "pc=gmg" in first
        s.setMatrix(DBNG)
        lines = s.describe().splitlines()
        assert f"product={form}" in lines[0] and "structure=recovered" in lines[0] and "nullspace=constant" in lines[0]
        assert "presteps=2 poststeps=2" in lines[0] and "partition=single ranks=1" in lines[0]
        assert "residual_update=separate_pass" in lines[0]  # (a system this small runs as a captured graph: no fused update)
        assert any(ln.startswith("departure: smoother: a sweep of the file runs as a fused pair") for ln in lines[1:])
        assert s.getType() == "NVIDIA AmgX"
        info = s.printInfo()
        assert "\tType: NVIDIA AmgX\n" in info and f"\tRuns: type=\"NVIDIA AmgX\" method=cg pc=gmg product={form}" in info
        s.destroy()
    s = lin.LinSolverHIP("poisson", config_text=gmg + "pib_sweep_pairs=0\n")
    s.setMatrix(DBNG)
    d = s.describe()
    assert "presteps=1 poststeps=1" in d and "departure" not in d
    s.destroy()
    capsys.readouterr()


def test_spmv_from_row_patterns_and_column_codes_is_the_csr_product(lin):
    """`pib_compress_columns`: at setMatrix every 256-row block of the matrix gets the table of its distinct rows-as-lists-of-
    offsets col - row (2, the default: up to 16 patterns of up to 8 entries, one byte per ROW in the product's stream, blocks
    with the same table sharing it -- kernels_spmv.hip k_spmv_lds_pattern) or the dictionary of its distinct offsets (1: up to
    16, one byte per ENTRY -- k_spmv_lds_coded) in place of the int32 columns and row offsets: the same products in the same
    order, bit for bit.  A matrix that does not fit the one form falls to the next; `pib_get_product_format` says which."""
    rng = np.random.default_rng(23)
    mats = []
    for cfg in (STRETCHED_2D, stretched_3d(), omesh.uniform_config((40, 24, 20)), omesh.uniform_config((256, 6, 5)), omesh.uniform_config((130, 9, 4))):
        m, DBNG, L = poisson_system(cfg)
        mats += [DBNG, oops.create_velocity_operator(L, 0.01, 0.005)]
    n = 1500
    rows, cols, vals = [], [], []
    for r in range(n):
        c = np.sort(rng.choice(n, size=int(rng.integers(1, 9)), replace=False))
        rows += [r] * len(c)
        cols += list(c)
        vals += list(rng.uniform(-1, 1, len(c)))
    mats.append(oops.csr_from_coo(n, n, rows, cols, vals))  # scattered columns: hundreds of offsets per block
    n = 600
    offs = list(range(-8, 9))
    for wide in (offs, offs[:16]):  # 17 offsets in the second block: one too many for the dictionary; 16 fit it, but no pattern holds a row of 16
        rows, cols, vals = [], [], []
        for r in range(n):
            for o in (wide if r == 300 else offs[6:11]):
                if 0 <= r + o < n:
                    rows.append(r), cols.append(r + o), vals.append(float(rng.uniform(-1, 1)))
        mats.append(oops.csr_from_coo(n, n, rows, cols, vals))
    for distinct in (16, 17):  # that many distinct rows in the second block: the last pattern / one too many
        rows, cols, vals = [], [], []
        for r in range(n):
            for o in ((-1 - (r % distinct) if r % distinct < 14 else -1, 0, 1 + (r % distinct) // 14 + (r % distinct) % 14 * 0) if 256 <= r < 512 else (-1, 0, 1)):
                if 0 <= r + o < n:
                    rows.append(r), cols.append(r + o), vals.append(float(rng.uniform(-1, 1)))
        mats.append(oops.csr_from_coo(n, n, rows, cols, vals))
    seen = set()
    for A in mats:
        x = rng.uniform(-1, 1, A.n_cols)
        ref = clib.spmv(A, x)
        for compress in (2, 1, 0):
            s = lin.LinSolverHIP("velocity", config_text=amgx_cfg(extra=f"pib_compress_columns={compress}\n"))
            s.setMatrix(A)
            assert s.productIndexBytes() == expected_product_format(A, compress)
            seen.add((compress, s.productIndexBytes()))
            y = np.empty(A.n_rows)
            s.matMult(x, y)
            assert np.array_equal(y, ref)
            s.destroy()
    assert {(2, 0), (2, 1), (2, 4), (1, 1), (1, 4), (0, 4)} <= seen  # every form and every fall was exercised


@pytest.mark.parametrize("n", [(64, 32, 6), (32, 64, 5), (128, 16, 3), (16, 16, 16)])
def test_spmv_tiled_chunk_order_bit_exact(lin, n):
    """With a 3-D grid registered the SpMV walks its 256-row chunks in plane-interleaved tile order;
    the result must not depend on the order (and the dot-product fusion must see every row once)."""
    from petibm_amd import capi
    m = omesh.create_mesh(omesh.uniform_config(n))
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    dt = 0.01
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.005)
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc="BLOCK_JACOBI", tol=1e-9))
    s.assemblePoisson(n, [m.dL[3][d].true for d in range(3)], dt, capi.NULLSPACE_CONSTANT)
    x = np.random.default_rng(5).uniform(-1, 1, A.n_cols)
    y = np.empty(A.n_rows)
    s.matMult(x, y)
    assert np.array_equal(y, clib.spmv(A, x))
    xs, b = rhs_for(A)
    sol = np.zeros(A.n_rows)
    s.solve(sol, b)  # p.Ap is fused into the SpMV: a wrong chunk map would break CG
    ref = clib.cg(A, b, pc="jacobi", nullspace=1, norm="unpreconditioned", rtol=1e-9, atol=0.0, dtol=1e300, maxit=5000)
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, sol)) <= 1.5e-9 * np.linalg.norm(b)
    s.destroy()


def test_spmv_long_rows_and_empty_rows(lin):
    """Rows longer than one LDS tile (2048 products) and empty rows."""
    rng = np.random.default_rng(3)
    n = 700
    dense_rows = {5: 2500 % n, 300: n, 699: 1}
    rows, cols, vals = [], [], []
    for r in range(n):
        if r in (0, 17, 256, 511):
            continue  # empty rows
        k = dense_rows.get(r, int(rng.integers(1, 9)))
        c = np.sort(rng.choice(n, size=min(k, n), replace=False))
        rows += [r] * len(c)
        cols += list(c)
        vals += list(rng.uniform(-1, 1, len(c)))
    A = oops.csr_from_coo(n, n, rows, cols, vals)
    # a wider matrix to get > 2048 entries in a row
    big = 5000
    r2 = np.concatenate([np.full(big, 3), np.arange(big)])
    c2 = np.concatenate([np.arange(big), np.arange(big)])
    v2 = rng.uniform(-1, 1, 2 * big)
    B = oops.csr_from_coo(big, big, r2, c2, v2)
    for M in (A, B):
        s = lin.LinSolverHIP("forces", config_text=amgx_cfg())
        s.setMatrix(M)
        x = rng.uniform(-1, 1, M.n_cols)
        y = np.empty(M.n_rows)
        s.matMult(x, y)
        assert np.array_equal(y, clib.spmv(M, x))
        s.destroy()


# -------------------------------------------------------------------- CG
@pytest.mark.parametrize("pc", ["NOSOLVER", "BLOCK_JACOBI"])
@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched"])
def test_cg_constant_nullspace_matches_oracle(lin, case, pc):
    from petibm_amd import capi
    cfg = {"2d_stretched": STRETCHED_2D, "3d_stretched": stretched_3d()}[case]
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt)
    xs, b = rhs_for(A)
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc=pc, tol=1e-10))
    s.assemblePoisson([int(v) for v in m.n[3][: m.dim]], [m.dL[3][d].true for d in range(m.dim)], dt,
                      capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    ref = clib.cg(A, b, pc="jacobi" if pc != "NOSOLVER" else "none", nullspace=1, norm="unpreconditioned",
                  rtol=1e-10, atol=0.0, dtol=1e300, maxit=5000)
    assert ref["reason"] > 0
    assert iters_close(s.getIters(), ref["iters"])
    # residual contract, recomputed by the oracle
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    h = s.getResidualHistory()
    k = min(len(h), len(ref["history"])) - 1
    # CG amplifies rounding differences (GPU tree reductions vs CPU sums) exponentially on the stretched,
    # ill-conditioned meshes: tight on the early history, loose (same decade) on the tail.
    ke = min(k, 12)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-9)
    assert np.all(np.abs(np.log10(h[:k] / ref["history"][:k])) < 0.5)
    assert s.getResidual() == h[-1]
    e = (x - x.mean()) - (ref["x"] - ref["x"].mean())
    assert np.linalg.norm(e) <= 1e-8 * np.linalg.norm(ref["x"])
    s.destroy()


@pytest.mark.parametrize("pc", ["NOSOLVER", "BLOCK_JACOBI"])
def test_cg_pinned_pressure_matches_oracle(lin, pc):
    """The "NVIDIA AmgX" convention: row/column 0 zeroed, diag 1, rhs[0] = 0
    (navierstokes.cpp:414-420, :553-558)."""
    m, A, _ = poisson_system(stretched_3d((14, 12, 10)), pinned=True)
    xs, b = rhs_for(A, zero_mean=False)
    b[0] = 0.0
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc=pc, tol=1e-11))
    s.setMatrix(A)
    assert s.getType() == "NVIDIA AmgX"
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    ref = clib.cg(A, b, pc="jacobi" if pc != "NOSOLVER" else "none", nullspace=0, norm="unpreconditioned",
                  rtol=1e-11, atol=0.0, dtol=1e300, maxit=5000)
    assert iters_close(s.getIters(), ref["iters"])
    assert x[0] == 0.0
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-11 * np.linalg.norm(b)
    assert np.linalg.norm(x - ref["x"]) <= 1e-8 * np.linalg.norm(ref["x"])
    s.destroy()


def test_initial_guess_is_used_by_amgx_flavour_and_ignored_by_ksp_flavour(lin):
    m, A, _ = poisson_system(STRETCHED_2D, pinned=True)
    xs, b = rhs_for(A, zero_mean=False)
    b[0] = 0.0
    exact = clib.cg(A, b, pc="jacobi", norm="unpreconditioned", rtol=1e-13, atol=0, dtol=1e300, maxit=5000)["x"]
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc="BLOCK_JACOBI", tol=1e-8, conv="ABSOLUTE"))
    s.setMatrix(A)
    x = exact.copy()
    s.solve(x, b)
    assert s.getIters() == 0  # already converged from the guess
    petsc = "-poisson_ksp_type cg\n-poisson_ksp_atol 1.0E-08\n-poisson_ksp_rtol 0.0\n-poisson_pc_type jacobi\n"
    k = lin.LinSolverHIP("poisson", config_text=petsc)
    assert k.getType() == "PETSc KSP"
    k.setMatrix(A)
    x = exact.copy()
    k.solve(x, b)
    assert k.getIters() > 5  # KSP zeroes the guess
    ref = clib.cg(A, b, pc="jacobi", norm="preconditioned", rtol=0.0, atol=1e-8, maxit=10000)
    assert iters_close(k.getIters(), ref["iters"])
    assert np.isclose(k.getResidual(), ref["rnorm"], rtol=1e-6)
    s.destroy()
    k.destroy()


def test_divergence_is_an_error_like_linsolverksp(lin):
    """linsolverksp.cpp:96-104: reason < 0 -> PETSC_ERR_CONV_FAILED (82)."""
    from petibm_amd.capi import PibError, ERR_CONV_FAILED
    m, A, _ = poisson_system(stretched_3d((12, 12, 12)), pinned=True)
    xs, b = rhs_for(A, zero_mean=False)
    b[0] = 0.0
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc="NOSOLVER", tol=1e-12, maxit=3))
    s.setMatrix(A)
    x = np.zeros(A.n_rows)
    with pytest.raises(PibError) as ei:
        s.solve(x, b)
    assert ei.value.code == ERR_CONV_FAILED
    assert s.getIters() == 3 and s.getReason() == -3
    # AmgX behaviour (no check, linsolveramgx.cpp:90-99) is selectable
    t = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc="NOSOLVER", tol=1e-12, maxit=3,
                                                         extra="solv:error_if_not_converged=0\n"))
    t.setMatrix(A)
    t.solve(x, b)
    assert t.getIters() == 3
    s.destroy()
    t.destroy()


def test_solve_before_setmatrix_and_device_vectors(lin):
    from petibm_amd.capi import PibError, ERR_ORDER
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc="BLOCK_JACOBI"))
    with pytest.raises(PibError) as ei:
        s.solve(np.zeros(4), np.zeros(4))
    assert ei.value.code == ERR_ORDER
    m, A, _ = poisson_system(STRETCHED_2D, pinned=True)
    xs, b = rhs_for(A, zero_mean=False)
    b[0] = 0.0
    s.setMatrix(A)
    xh = np.zeros(A.n_rows)
    s.solve(xh, b)
    it_host = s.getIters()
    xd, bd = s.deviceVec(), s.deviceVec()
    bd.upload(b)
    xd.upload(np.zeros(A.n_rows))
    s.solve(xd, bd)
    assert s.getIters() == it_host
    assert np.array_equal(xd.download(), xh)  # deterministic reductions: same bits
    # setMatrix may be called again (rigidkinematics.cpp:135)
    s.setMatrix(A)
    xh[:] = 0.0  # the AmgX flavour takes x as the initial guess
    s.solve(xh, b)
    assert s.getIters() == it_host
    s.destroy()


def test_degenerate_systems_and_inputs(lin):
    """Edge cases around the boundary: a 1x1 system, a zero right-hand side, a zero diagonal under Jacobi, columns out of
    range, a grid hint that does not describe the matrix, more iterations asked for than needed."""
    from petibm_amd import capi
    from petibm_amd.capi import PibError
    one = oops.CSR(1, 1, np.array([0, 1], dtype=np.int64), np.array([0], dtype=np.int64), np.array([-2.5]))
    for pc in ("NOSOLVER", "BLOCK_JACOBI"):
        s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc=pc, tol=1e-14))
        s.setMatrix(one)
        x = np.zeros(1)
        s.solve(x, np.array([5.0]))
        assert abs(x[0] + 2.0) <= 1e-14 and s.getIters() <= 1
        s.destroy()
    m, A, _ = poisson_system(STRETCHED_2D, pinned=True)
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc="BLOCK_JACOBI", tol=1e-12))
    s.setMatrix(A)
    x = np.zeros(A.n_rows)
    s.solve(x, np.zeros(A.n_rows))  # b = 0: converged before the first iteration, x stays 0
    assert s.getIters() == 0 and s.getReason() > 0 and not x.any()
    # the same solver object takes a different matrix afterwards (KSPReset semantics, linsolverksp.cpp:78)
    m3, A3, _ = poisson_system(stretched_3d((8, 7, 6)), pinned=True)
    xs, b = rhs_for(A3, zero_mean=False)
    b[0] = 0.0
    s.setMatrix(A3)
    x = np.zeros(A3.n_rows)
    s.solve(x, b)
    assert np.linalg.norm(b - clib.spmv(A3, x)) <= 1e-11 * max(np.linalg.norm(b), 1.0)
    s.destroy()
    # zero diagonal + Jacobi: refused at setMatrix, not a NaN later
    Z = A.copy()
    rows = np.repeat(np.arange(Z.n_rows), np.diff(Z.rowptr))
    Z.val[(rows == Z.col) & (rows == 3)] = 0.0
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc="BLOCK_JACOBI"))
    with pytest.raises(PibError) as ei:
        s.setMatrix(Z)
    assert ei.value.code == capi.ERR_ARG_WRONG
    s.destroy()
    # a column index beyond the matrix
    B = A.copy()
    B.col = B.col.copy()
    B.col[5] = B.n_rows + 7
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg(pc="NOSOLVER"))
    with pytest.raises(PibError) as ei:
        s.setMatrix(B)
    assert ei.value.code == capi.ERR_ARG_OUTOFRANGE
    # a grid hint that belongs to another mesh: stencil twin and CSR disagree -> PETSC_ERR_ARG_WRONG
    s.setMatrix(A)
    n = [int(v) for v in m.n[3][:2]]
    w = [m.dL[3][d].true[::-1].copy() for d in range(2)]
    g = [0.01 * (1.0 / (0.5 * (wd[1:] + wd[:-1]))) for wd in w]
    with pytest.raises(PibError) as ei:
        s.setGridHint(n, w, g, capi.NULLSPACE_PINNED)
    assert ei.value.code == capi.ERR_ARG_WRONG
    s.destroy()


# ------------------------------------------------------------ BiCGStab (K10)
@pytest.mark.parametrize("flavour", ["amgx", "ksp"])
@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched"])
def test_bicgstab_velocity_system_matches_oracle(lin, case, flavour):
    """A = I/dt - c nu L (navierstokes.cpp:342-344), non-symmetric on stretched meshes; solved with
    PBICGSTAB + BLOCK_JACOBI (examples/.../taylorgreenvortex3dRe1600_GPU/config/velocity_solver.info) or
    -velocity_ksp_type bcgs -velocity_pc_type jacobi (examples/.../liddrivencavity2dRe100/config/velocity_solver.info)."""
    cfg = {"2d_stretched": STRETCHED_2D, "3d_stretched": stretched_3d()}[case]
    m, _, L = poisson_system(cfg)
    A = oops.create_velocity_operator(L, 0.05, 0.5 * 0.2)
    rng = np.random.default_rng(11)
    us = rng.uniform(-1, 1, A.n_rows)
    b = clib.spmv(A, us)
    if flavour == "amgx":
        text = amgx_cfg(solver="PBICGSTAB", pc="BLOCK_JACOBI", tol=1e-12, conv="ABSOLUTE", maxit=1000)
        ref = clib.bcgs(A, b, pc="jacobi", norm="unpreconditioned", rtol=0.0, atol=1e-12, dtol=1e300, maxit=1000)
    else:
        text = ("-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-12\n-velocity_ksp_rtol 0.0\n"
                "-velocity_ksp_max_it 1000\n-velocity_pc_type jacobi\n-velocity_pc_jacobi_type diagonal\n")
        ref = clib.bcgs(A, b, pc="jacobi", norm="preconditioned", rtol=0.0, atol=1e-12, maxit=1000)
    s = lin.LinSolverHIP("velocity", config_text=text)
    s.setMatrix(A)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(x - us) <= 1e-9 * np.linalg.norm(us)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 4)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-8)
    s.destroy()


# ------------------------------------------------- geometric multigrid (K2/K5/K6/K7)
def gmg_cfg(tol=1e-10, pre=1, post=1, omega=0.9, extra=""):
    return (f"config_version=2\nsolver(solv)=PCG\nsolv:max_iters=200\nsolv:monitor_residual=1\n"
            f"solv:convergence=RELATIVE_INI\nsolv:tolerance={tol}\nsolv:norm=L2\nsolv:store_res_history=1\n"
            f"solv:preconditioner(prec)=AMG\nprec:cycle=V\nprec:presweeps={pre}\nprec:postsweeps={post}\n"
            f"prec:max_levels=100\nprec:coarsest_sweeps=2\nprec:smoother(smooth)=BLOCK_JACOBI\n"
            f"smooth:relaxation_factor={omega}\npib_sweep_pairs=0\n{extra}")


@pytest.mark.parametrize("case,pre,post", [("2d_stretched", 1, 1), ("3d_uniform", 1, 1), ("3d_stretched", 1, 1),
                                           ("3d_uniform_odd", 2, 2), ("2d_uniform", 2, 1)])
def test_gmg_pcg_constant_nullspace_matches_oracle(lin, case, pre, post):
    from petibm_amd import capi
    cfg = {"2d_stretched": STRETCHED_2D, "3d_uniform": omesh.uniform_config((32, 32, 32)),
           "3d_stretched": stretched_3d((24, 20, 16)), "3d_uniform_odd": omesh.uniform_config((21, 18, 13)),
           "2d_uniform": omesh.uniform_config((64, 48))}[case]
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post))
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=1, pre=pre, post=post, omega=0.9, coarsest_sweeps=32)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 8)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-8)
    e = (x - x.mean()) - (ref["x"] - ref["x"].mean())
    assert np.linalg.norm(e) <= 1e-8 * np.linalg.norm(ref["x"])
    s.destroy()


def cylinder_mesh_config(cells=(171, 108, 171)):
    """The mesh of examples/ibpm/cylinder2dRe40/config.yaml (the reference's flagship immersed-boundary case):
    a uniform block around the body, geometric stretching (ratio 1.02) out to +-15; 450 x 450 cells."""
    a, c, e = cells
    sub = [{"end": -0.54, "cells": a, "stretchRatio": 0.980392156}, {"end": 0.54, "cells": c, "stretchRatio": 1.0},
           {"end": 15.0, "cells": e, "stretchRatio": 1.02}]
    cfg = omesh.uniform_config((a + c + e, a + c + e))
    cfg["mesh"] = [{"direction": d, "start": -15.0, "subDomains": sub} for d in "xy"]
    return cfg


@pytest.mark.parametrize("case", ["cylinder2d_450", "3d_strong_stretch", "3d_anisotropic_uniform"])
def test_gmg_is_mesh_independent_on_stretched_meshes(lin, case):
    """Selective coarsening + width-based transfers: the V-cycle keeps its uniform-mesh convergence rate
    (about 15-17 PCG iterations to 1e-10) on the stretched meshes the reference's examples use (cell width
    ratios of 30 and more across the domain).  Device vs oracle: same iteration count and residual history."""
    from petibm_amd import capi
    cfg = {"cylinder2d_450": cylinder_mesh_config(),
           "3d_strong_stretch": stretched_3d((40, 36, 32), r=(1.15, 0.88, 1.12)),
           "3d_anisotropic_uniform": omesh.uniform_config((48, 40, 24))}[case]
    dt = 0.01
    m = omesh.create_mesh(cfg)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    A = oops.CSR.from_csr32(*clib.assemble_poisson32(n, w, dt))
    xs = np.random.default_rng(5).uniform(-1, 1, A.n_rows)
    xs -= xs.mean()
    b = clib.spmv(A, xs)
    s = lin.LinSolverHIP("poisson", config_text=gmg_cfg())
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=1, pre=1, post=1, omega=0.9, coarsest_sweeps=32)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert ref["iters"] <= 22 and s.getIters() <= 22
    assert iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 8)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-8)
    s.destroy()


@pytest.mark.parametrize("case", ["2d_stretched", "3d_uniform", "3d_stretched"])
def test_gmg_chebyshev_smoother_matches_oracle(lin, case):
    """AmgX-style `smoother=CHEBYSHEV_POLY` (degree-2 Chebyshev-Jacobi polynomial per sweep)."""
    from petibm_amd import capi
    cfg = {"2d_stretched": STRETCHED_2D, "3d_uniform": omesh.uniform_config((32, 32, 32)),
           "3d_stretched": stretched_3d((24, 20, 16))}[case]
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    text = gmg_cfg().replace("prec:smoother(smooth)=BLOCK_JACOBI", "prec:smoother(smooth)=CHEBYSHEV_POLY") + \
        "smooth:chebyshev_polynomial_order=2\nsmooth:cheby_max_lambda=2.0\nsmooth:cheby_min_lambda=0.5\n"
    s = lin.LinSolverHIP("poisson", config_text=text)
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=1, pre=2, post=2, coarsest_sweeps=32).set_chebyshev(2.0, 4.0)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200)
    assert ref["reason"] > 0 and iters_close(s.getIters(), ref["iters"])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    h = s.getResidualHistory()
    ke = min(len(h), len(ref["history"]), 6)
    assert np.allclose(h[:ke], ref["history"][:ke], rtol=1e-8)
    # fewer iterations than the Jacobi V(1,1) cycle on the same system
    j = lin.LinSolverHIP("poisson", config_text=gmg_cfg())
    j.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
    xj = np.zeros(A.n_rows)
    j.solve(xj, b)
    assert s.getIters() < j.getIters()
    s.destroy()
    j.destroy()


def test_gmg_coarse_tail_kernel_is_bit_identical(lin):
    """pib_coarse_tail > 0 runs the small levels in one single-workgroup kernel: same arithmetic, same bits."""
    from petibm_amd import capi
    cfg = stretched_3d((24, 20, 16))
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    out = []
    for tail in (0, 512, 4096):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(extra=f"pib_coarse_tail={tail}\n"))
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(A.n_rows)
        s.solve(x, b)
        out.append((x, s.getIters()))
        s.destroy()
    assert out[0][1] == out[1][1] == out[2][1]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][0], out[2][0])


@pytest.mark.parametrize("case", ["3d_stretched", "2d_stretched", "3d_periodic_xz", "2d_periodic", "v22"])
def test_gmg_coarse_tail_forms_are_bit_identical(lin, case):
    """The single-workgroup tail walks its levels with the vectors and the 1-D tables in HBM or staged in LDS
    (pib_coarse_tail_lds), one cell per thread with the cell's row in registers or -- levels of more cells than threads
    (pib_coarse_tail=4096) -- through the tables in every phase: per-level launches, and every form of the tail, give the
    same bits."""
    from petibm_amd import capi
    per = None
    pre = post = 1
    if case == "3d_stretched":
        cfg = stretched_3d((24, 20, 16))
    elif case == "2d_stretched":
        cfg = STRETCHED_2D
    elif case == "3d_periodic_xz":
        per = (True, False, True)
        cfg = omesh.periodic_config((24, 20, 16), per)
    elif case == "2d_periodic":
        per = (True, True)
        cfg = omesh.periodic_config((64, 48), per)
    else:
        cfg = omesh.uniform_config((32, 32, 32))
        pre = post = 2
    dt = 0.01
    m = omesh.create_mesh(cfg)
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.005)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    out = []
    for tail, lds in ((0, 0), (1024, 0), (1024, 1), (4096, 0), (4096, 1), (64, 1)):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post, extra=f"pib_coarse_tail={tail}\npib_coarse_tail_lds={lds}\n"))
        if per is not None:
            s.setPeriodic(per)
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(A.n_rows)
        s.solve(x, b)
        assert s.getReason() > 0
        out.append((x, s.getIters(), s.getResidualHistory().copy()))
        s.destroy()
    for x, its, h in out[1:]:
        assert its == out[0][1]
        assert np.array_equal(x, out[0][0])
        assert np.array_equal(h, out[0][2])


@pytest.mark.parametrize("case,pre,post", [("3d_stretched", 2, 2), ("3d_stretched", 1, 1), ("3d_odd", 2, 1), ("2d_stretched", 2, 2),
                                           ("2d_uniform", 1, 2), ("3d_periodic_xz", 2, 2), ("2d_periodic", 2, 2), ("3d_periodic_all", 1, 1),
                                           ("3d_64", 2, 2)])
def test_gmg_small_level_kernels_are_bit_identical(lin, case, pre, post):
    """k_small_down / k_small_up walk a small level's whole way down / up in one launch each, on boxes of coarse cells with
    recomputed margins (pib_fuse_small_levels): the same bits as one launch per phase, with and without the single-workgroup
    tail underneath."""
    from petibm_amd import capi
    per = None
    if case == "3d_stretched":
        cfg = stretched_3d((40, 36, 28))
    elif case == "3d_odd":
        cfg = omesh.uniform_config((37, 30, 21))
    elif case == "2d_stretched":
        cfg = STRETCHED_2D
    elif case == "2d_uniform":
        cfg = omesh.uniform_config((96, 80))
    elif case == "3d_periodic_xz":
        per = (True, False, True)
        cfg = omesh.periodic_config((40, 24, 32), per)
    elif case == "2d_periodic":
        per = (True, True)
        cfg = omesh.periodic_config((64, 48), per)
    elif case == "3d_periodic_all":
        per = (True, True, True)
        cfg = omesh.periodic_config((32, 32, 32), per)
    else:
        cfg = omesh.uniform_config((64, 64, 64))
    dt = 0.01
    m = omesh.create_mesh(cfg)
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.005)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    out = []
    for fuse, tail in ((0, 0), (1, 0), (1, -1), (0, -1)):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post, extra=f"pib_fuse_small_levels={fuse}\npib_coarse_tail={tail}\n"))
        if per is not None:
            s.setPeriodic(per)
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(A.n_rows)
        s.solve(x, b)
        assert s.getReason() > 0
        out.append((x, s.getIters(), s.getResidualHistory().copy()))
        s.destroy()
    for x, its, h in out[1:]:
        assert its == out[0][1]
        assert np.array_equal(h, out[0][2])
        assert np.array_equal(x, out[0][0])


def test_gmg_pcg_pinned_pressure_matches_oracle(lin):
    from petibm_amd import capi
    cfg = stretched_3d((20, 16, 12))
    dt = 0.02
    m, A, _ = poisson_system(cfg, dt=dt, pinned=True)
    xs, b = rhs_for(A, zero_mean=False)
    b[0] = 0.0
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(tol=1e-11))
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_PINNED)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=2, omega=0.9, coarsest_sweeps=32)
    ref = g.pcg(A, b, rtol=1e-11, maxit=200)
    assert iters_close(s.getIters(), ref["iters"])
    assert x[0] == 0.0
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-11 * np.linalg.norm(b)
    assert np.linalg.norm(x - ref["x"]) <= 1e-8 * np.linalg.norm(ref["x"])
    s.destroy()


def test_gmg_with_setmatrix_and_grid_hint(lin):
    """The PetIBM route: the application assembles DBNG (here: the oracle), hands it over with
    setMatrix, and registers the mesh structure with the grid hint."""
    from petibm_amd import capi
    from petibm_amd.capi import PibError, ERR_ARG_WRONG
    cfg = stretched_3d((16, 14, 12))
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    g = [dt * (1.0 / (0.5 * (wd[1:] + wd[:-1]))) for wd in w]
    s = lin.LinSolverHIP("poisson", config_text=gmg_cfg())
    s.setMatrix(A)
    s.setGridHint(n, w, g, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    # a wrong hint is rejected (verified against the CSR on the device)
    bad = [gi * 1.01 for gi in g]
    with pytest.raises(PibError) as ei:
        s.setGridHint(n, w, bad, capi.NULLSPACE_CONSTANT)
    assert ei.value.code == ERR_ARG_WRONG
    # and a multigrid solve without structure fails loudly
    t = lin.LinSolverHIP("poisson", config_text=gmg_cfg())
    t.setMatrix(A)
    with pytest.raises(PibError):
        t.solve(x, b)
    s.destroy()
    t.destroy()


def test_full_size_properties_256(lin):
    """Size-independent properties at a BASELINE size (config 2, 256^3): row sums of the singular
    operator vanish, A is symmetric (<Ax,y> = <x,Ay>), and the multigrid-PCG solve meets the residual
    contract recomputed with the CSR operator."""
    from petibm_amd import capi
    n = 256
    w = np.full(n, 1.0 / n)
    s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(tol=1e-10) + "pib_initial_guess_nonzero=0\n")
    s.assemblePoisson((n, n, n), [w, w, w], 1e-3, capi.NULLSPACE_CONSTANT)
    N = n ** 3
    rng = np.random.default_rng(20260928)
    one, y = s.deviceVec(), s.deviceVec()
    one.upload(np.ones(N))
    s.matMult(one, y)
    diag_scale = 6.0 * 1e-3 * (1.0 / n)
    assert np.abs(y.download()).max() <= 1e-12 * diag_scale
    u = rng.uniform(-1, 1, N)
    v = rng.uniform(-1, 1, N)
    ud, vd = s.deviceVec().upload(u), s.deviceVec().upload(v)
    s.matMult(ud, y)
    au = y.download()
    s.matMult(vd, y)
    av = y.download()
    assert abs(au @ v - u @ av) <= 1e-12 * np.linalg.norm(au) * np.linalg.norm(v)
    b = au - au.mean()
    bd, xd = s.deviceVec().upload(b), s.deviceVec()
    s.solve(xd, bd)
    s.matMult(xd, y)
    assert np.linalg.norm(b - y.download()) <= 1.5e-10 * np.linalg.norm(b)
    assert s.getIters() < 40
    s.destroy()


def test_full_size_stretched_mesh_config5(lin):
    """BASELINE config 5's mesh class at full size: 384 x 256 x 256 cells, three sub-domains per axis (uniform block
    around the body, geometric stretching outwards: the layout of examples/decoupledibpm/flatplate3dRe100AoA30_GPU),
    25.2 M pressure unknowns.  Size-independent checks: zero row sums, and the multigrid-PCG solve meets the
    residual contract (recomputed with the CSR operator) at the uniform-mesh iteration count."""
    from petibm_amd import capi

    n = [384, 256, 256]
    w = []
    for nd, span in zip(n, (12.0, 8.0, 8.0)):
        a = (nd - nd // 3) // 2
        sub = [{"end": -1.0, "cells": a, "stretchRatio": 1.0 / 1.02}, {"end": 1.0, "cells": nd - 2 * a, "stretchRatio": 1.0},
               {"end": span, "cells": a, "stretchRatio": 1.02}]
        w.append(omesh.parse_subdomains(sub, -span)[2])
    assert [len(a) for a in w] == n and max(a.max() for a in w) / min(a.min() for a in w) > 10
    s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(tol=1e-10) + "pib_initial_guess_nonzero=0\n")
    s.assemblePoisson(n, w, 2e-3, capi.NULLSPACE_CONSTANT)
    N = int(np.prod(n))
    one, y = s.deviceVec(), s.deviceVec()
    one.upload(np.ones(N))
    s.matMult(one, y)
    assert np.abs(y.download()).max() <= 1e-12
    u = np.random.default_rng(20260928).uniform(-1, 1, N)
    ud = s.deviceVec().upload(u)
    s.matMult(ud, y)
    b = y.download()
    b -= b.mean()
    bd, xd = s.deviceVec().upload(b), s.deviceVec()
    s.solve(xd, bd)
    s.matMult(xd, y)
    assert np.linalg.norm(b - y.download()) <= 1.5e-10 * np.linalg.norm(b)
    assert s.getIters() <= 22
    s.destroy()


def test_cpp_host_mirror_demo_runs():
    """include/petibm_amd/linsolver.hpp (the C++ mirror of petibm::linsolver) over the C ABI, from a plain
    g++ program: createLinSolver -> setMatrix -> setGridHint -> solve -> getIters/getResidual."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "cpp", "poisson_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++14", "-I", os.path.join(root, "include"),
                               os.path.join(root, "examples", "cpp", "poisson_demo.cpp"), "-L",
                               os.path.join(root, "petibm_amd", "lib"), "-lpetibm_amd",
                               "-Wl,-rpath,$ORIGIN/../../petibm_amd/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, cwd=root, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Type: NVIDIA AmgX" in out.stdout and "recomputed ||b-Ax||/||b||" in out.stdout


# ------------------------------------------------ velocity operator assembled on the device (K9, a-8, a-9)
def _a0_table(mesh):
    a0 = np.zeros((3, 6))
    for f in range(mesh.dim):
        for loc in range(2 * mesh.dim):
            t = mesh.bc_types.get((f, loc), "NOBC")
            if t not in ("NOBC", "PERIODIC"):
                a0[f, loc] = oops.bc_a0(t, f, loc)
    return a0


def _outflow_3d():
    cfg = stretched_3d((10, 9, 8))
    bcs = cfg["flow"]["boundaryConditions"]
    bcs[1]["u"] = ["CONVECTIVE", 1.0]
    bcs[1]["v"] = ["CONVECTIVE", 1.0]
    bcs[1]["w"] = ["CONVECTIVE", 1.0]
    bcs[5]["u"] = ["NEUMANN", 0.0]
    bcs[5]["v"] = ["NEUMANN", 0.0]
    bcs[5]["w"] = ["NEUMANN", 0.0]
    return cfg


@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched", "3d_outflow"])
def test_device_velocity_operator_bit_exact_and_bicgstab(lin, case):
    cfg = {"2d_stretched": STRETCHED_2D, "3d_stretched": stretched_3d(), "3d_outflow": _outflow_3d()}[case]
    m = omesh.create_mesh(cfg)
    L = oops.create_laplacian(m)
    dt, cnu = 0.004, 0.5 * 0.01
    A = oops.create_velocity_operator(L, dt, cnu)
    s = lin.LinSolverHIP("velocity", config_text=amgx_cfg(solver="PBICGSTAB", pc="BLOCK_JACOBI", tol=1e-12,
                                                          conv="ABSOLUTE", maxit=500))
    n = [int(v) for v in m.n[3][: m.dim]]
    s.assembleVelocity(n, [m.dL[3][d].true for d in range(m.dim)], m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
    rp, cl, vl = s.getCSR()
    assert np.array_equal(rp, A.rowptr) and np.array_equal(cl, A.col)
    assert np.array_equal(vl, A.val)  # same floating-point order as createLaplacian + MatScale + MatShift
    us = np.random.default_rng(2).uniform(-1, 1, A.n_rows)
    b = clib.spmv(A, us)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    ref = clib.bcgs(A, b, pc="jacobi", norm="unpreconditioned", rtol=0.0, atol=1e-12, dtol=1e300, maxit=500)
    assert iters_close(s.getIters(), ref["iters"]) and s.getIters() < 30
    assert np.linalg.norm(x - us) <= 1e-10 * np.linalg.norm(us)
    s.destroy()


@pytest.mark.parametrize("sweeps", [2, 1])
@pytest.mark.parametrize("n,pinned", [((128, 16, 12), False), ((128, 24, 10), True), ((256, 8, 70), False)])
def test_fused_presmoothing_pair_is_bit_identical(lin, n, pinned, sweeps):
    """gmg.hip k_presmooth2: the first two pre-smoothing steps from a zero guess in one LDS-tiled kernel (levels with
    nx % 128 == 0, ny % 8 == 0) against the two streaming kernels -- same iterates, bit for bit -- and the oracle.
    sweeps = 1 (the V(1,1) cycle of the reference's AmgX configurations): the one pre-smoothing step and the residual
    of its result in that kernel (k_presmooth2<1>) against mode 1 + the residual kernel."""
    from petibm_amd import capi
    cfg = stretched_3d(n, r=(1.01, 0.97, 1.04))
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt, pinned=pinned)
    xs, b = rhs_for(A, zero_mean=not pinned)
    if pinned:
        b[0] = 0.0
    w = [m.dL[3][d].true for d in range(m.dim)]
    out = []
    for fuse in (1, 0):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=sweeps, post=sweeps, extra=f"pib_march_min_cells=0\npib_fuse_presmooth={fuse}\n"))
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
        x = np.zeros(A.n_rows)
        s.solve(x, b)
        out.append((x, s.getResidualHistory(), s.getIters()))
        s.destroy()
    assert out[0][2] == out[1][2] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
    g = clib.GMG(list(n), w, dt, nullspace=2 if pinned else 1, pre=sweeps, post=sweeps, omega=0.9, coarsest_sweeps=32)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200)
    assert iters_close(out[0][2], ref["iters"])
    ke = min(len(out[0][1]), len(ref["history"]), 6)
    assert np.allclose(out[0][1][:ke], ref["history"][:ke], rtol=1e-8)


@pytest.mark.parametrize("n,pinned", [((128, 16, 12), False), ((128, 24, 70), True)])
def test_blocked_level_kernel_matches_the_streaming_one(lin, n, pinned):
    """gmg.hip k_level_march (2.5-D blocked Jacobi step / residual on levels with nx % 128 == 0, ny % 8 == 0): the
    same arithmetic as k_level modes 2 and 3; mode 8 groups its sums by tile, so the PCG scalars agree to rounding."""
    from petibm_amd import capi
    cfg = stretched_3d(n, r=(1.01, 0.97, 1.04))
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt, pinned=pinned)
    xs, b = rhs_for(A, zero_mean=not pinned)
    if pinned:
        b[0] = 0.0
    w = [m.dL[3][d].true for d in range(m.dim)]
    out = []
    for march in (1, 0):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=2, post=2, extra=f"pib_march_min_cells=0\npib_march={march}\n"))
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
        x = np.zeros(A.n_rows)
        s.solve(x, b)
        out.append((x, s.getResidualHistory(), s.getIters()))
        s.destroy()
    assert out[0][2] == out[1][2]
    assert np.allclose(out[0][1], out[1][1], rtol=1e-9)
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-11 * np.abs(out[1][0]).max()
    assert np.linalg.norm(b - clib.spmv(A, out[0][0])) <= 1.5e-10 * np.linalg.norm(b)


@pytest.mark.parametrize("key,sweeps", [("pib_fuse_down_march", 2), ("pib_march", 2), ("pib_fuse_residual_restrict", 2), ("pib_fuse_post_pair", 2), ("pib_fuse_prolong", 2), ("pib_fuse_prolong", 1)])
@pytest.mark.parametrize("n,pinned", [((128, 32, 24), False), ((256, 16, 40), True), ((128, 16, 34), False)])
def test_marching_transfers_are_bit_identical(lin, n, pinned, key, sweeps):
    """gmg.hip k_restrict_march (fully paired 3-D levels with nx % 128 == 0, ny % 16 == 0: a fine plane goes through LDS
    once and feeds its two coarse planes) against the row kernel k_restrict_rows, and k_prolong_smooth (prolongation +
    first post-smoothing step in one march, the corrected iterate only on chip) against k_prolong_rows + k_level_march,
    k_resid_restrict_march (residual + restriction in one march, the residual only on chip) against k_level_march<3> +
    k_restrict_march, k_prolong_smooth2 (prolongation + both post-smoothing steps in one march) against k_prolong_smooth +
    k_level_march<2 / 8>:
    the same sums in the same order, so the whole solve is bit-identical; mildly stretched widths keep every aggregate
    a pair and the weights non-trivial.  sweeps = 1: the one post-smoothing step of level 0 also delivers the
    Krylov sums (k_prolong_smooth<1>, grouped like k_level_march<8>)."""
    from petibm_amd import capi
    names, r = "xyz", (1.002, 1.01, 0.99)
    cfg = omesh.uniform_config(n)
    cfg["mesh"] = [{"direction": names[d], "start": 0.0,
                    "subDomains": [{"end": 0.05 * n[d], "cells": n[d], "stretchRatio": r[d]}]} for d in range(3)]
    dt = 0.01
    m, A, _ = poisson_system(cfg, dt=dt, pinned=pinned)
    xs, b = rhs_for(A, zero_mean=not pinned)
    if pinned:
        b[0] = 0.0
    w = [m.dL[3][d].true for d in range(m.dim)]
    out = []
    for march in (1, 0):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=sweeps, post=sweeps, extra=f"pib_march_min_cells=0\n{key}={march}\n"))
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
        x = np.zeros(A.n_rows)
        s.solve(x, b)
        out.append((x, s.getResidualHistory(), s.getIters()))
        s.destroy()
    if key == "pib_march":
        # (one switch for the blocked level kernels AND the marching restriction: the level kernel that delivers the Krylov sums
        # groups them by tile, so the flat forms agree to rounding -- test_blocked_level_kernel_matches_streaming_kernel; the
        # restriction itself is bit for bit, which the fused forms below, that contain it, keep asserting)
        assert out[0][2] == out[1][2] and np.allclose(out[0][1], out[1][1], rtol=1e-9, atol=0.0)
        assert np.abs(out[0][0] - out[1][0]).max() <= 1e-12 * np.abs(out[1][0]).max()
    else:
        assert out[0][2] == out[1][2] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
    g = clib.GMG(list(n), w, dt, nullspace=2 if pinned else 1, pre=sweeps, post=sweeps, omega=0.9, coarsest_sweeps=32)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200)
    assert iters_close(out[0][2], ref["iters"])
    ke = min(len(out[0][1]), len(ref["history"]), 6)
    assert np.allclose(out[0][1][:ke], ref["history"][:ke], rtol=1e-8)


@pytest.mark.parametrize("flavour,sweeps", [("amgx", 2), ("amgx", 1), ("ksp", 2)])
def test_residual_update_inside_the_vcycle_is_bit_identical(lin, flavour, sweeps):
    """krylov.hip / gmg.hip k_presmooth2<., 1>: on systems beyond the captured-graph size (> 2^22 rows, one rank) PCG leaves
    r = r - alpha w to the V-cycle's first march, which reads the residual anyway.  Same expression per cell: the iterates and
    the solution are those of the separate pass bit for bit; only the printed norms (sums grouped by tile) move in the last
    digits.  Both norm conventions (AmgX: |r| checked before the cycle; KSP: the preconditioned norm after it) and both
    pre-smoothing forms (pair; step + residual)."""
    from petibm_amd import capi
    n = (256, 128, 136)
    w = [np.full(n[0], 1.0 / n[0]) * (1.0 + 0.3 * np.sin(np.arange(n[0]) / 17.0)),
         np.full(n[1], 1.0 / n[1]), np.full(n[2], 1.5 / n[2]) * (1.0 + 0.2 * np.cos(np.arange(n[2]) / 11.0))]
    dt = 0.01
    xs = np.random.default_rng(7).uniform(-1, 1, n[0] * n[1] * n[2])
    xs -= xs.mean()
    out = []
    for fuse in (1, 0, 2):  # 2: the solver asks for the fused form, the cycle's launch-site predicate refuses it (forced): fallback pass
        extra = f"pib_fuse_residual_update={fuse}\npib_march_min_cells=0\n"
        if flavour == "amgx":
            text = gmg_cfg(pre=sweeps, post=sweeps, extra=extra)
        else:
            text = ("-poisson_ksp_type cg\n-poisson_ksp_rtol 1.0E-10\n-poisson_ksp_atol 1.0E-50\n-poisson_ksp_max_it 200\n"
                    "-poisson_pc_type gamg\n-poisson_pib_smoother JACOBI\n"
                    f"-poisson_pib_presweeps {sweeps}\n-poisson_pib_postsweeps {sweeps}\n-poisson_pib_sweep_pairs 0\n-poisson_pib_fuse_residual_update {fuse}\n"
                    "-poisson_pib_march_min_cells 0\n")
        s = lin.LinSolverHIP("poisson", config_text=text)
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_CONSTANT)
        b = np.empty_like(xs)
        s.matMult(xs, b)
        x = np.zeros_like(xs)
        s.solve(x, b)
        r = np.empty_like(xs)
        s.matMult(x, r)
        out.append((x, np.array(s.getResidualHistory()), s.getIters(), np.linalg.norm(b - r) / np.linalg.norm(b), int(s.counters()[6])))
        assert ("residual_update=in_vcycle" if fuse == 1 else "residual_update=separate_pass") in s.describe().splitlines()[0]
        s.destroy()
    assert out[0][4] >= out[0][2] and out[1][4] == 0  # every (enqueued) iteration of the first run took the fused form, none of the second
    assert out[0][2] == out[1][2] and 5 <= out[0][2] <= 40
    assert np.array_equal(out[0][0], out[1][0])
    assert np.allclose(out[0][1], out[1][1], rtol=1e-12)
    assert out[0][3] <= 2e-10
    # the fallback (a refusing launch site no longer fails the solve: the update runs as its own pass, into the other buffer)
    assert out[2][2] == out[1][2] and np.array_equal(out[2][0], out[1][0]) and np.array_equal(out[2][1], out[1][1])


def test_way_down_in_one_march_with_the_residual_update_is_bit_identical(lin):
    """gmg.hip k_down_march<1>: the two pre-smoothing steps with PCG's residual update, the residual and the restriction of
    level 0 in ONE march against k_presmooth2<0, 1> + k_resid_restrict_march -- the same expressions cell by cell and the Krylov
    sums in the same grouping: iterates, solution AND the residual history bit for bit."""
    from petibm_amd import capi
    n = (256, 128, 136)
    w = [np.full(n[0], 1.0 / n[0]) * (1.0 + 0.3 * np.sin(np.arange(n[0]) / 17.0)),
         np.full(n[1], 1.0 / n[1]), np.full(n[2], 1.5 / n[2]) * (1.0 + 0.2 * np.cos(np.arange(n[2]) / 11.0))]
    dt = 0.01
    xs = np.random.default_rng(7).uniform(-1, 1, n[0] * n[1] * n[2])
    out = []
    for pinned in (False, True):
        xp = xs - (xs[0] if pinned else xs.mean())
        for down in (1, 0):
            s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=2, post=2, extra=f"pib_fuse_down_march={down}\npib_march_min_cells=0\n"))
            s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
            b = np.empty_like(xp)
            s.matMult(xp, b)
            x = np.zeros_like(xp)
            s.solve(x, b)
            out.append((x, np.array(s.getResidualHistory()), s.getIters(), int(s.counters()[6])))
            s.destroy()
        a, c = out[-2], out[-1]
        assert a[3] >= a[2] > 4 and c[3] >= c[2]  # the residual update ran inside the cycle either way
        assert a[2] == c[2] and np.array_equal(a[1], c[1]) and np.array_equal(a[0], c[0])


@pytest.mark.parametrize("sweeps", [2, 1])
def test_residual_update_inside_the_vcycle_with_a_pinned_row(lin, sweeps):
    """The convention every `type: GPU` run of PetIBM uses (MatZeroRowsColumns on row 0, navierstokes.cpp:414-420) with the
    residual update inside the cycle's first march (round 5).  The cycle's compatible right-hand side needs sum r of the NEW
    residual before the march that forms it: cg_s1 gives it as sum r_old - alpha sum w, with sum w from the entries of p next
    to cell 0 (the columns of the singular operator sum to zero).  Against the separate pass fed the same sum
    (pib_pin_sum_local=1: the same iterates up to the grouping of red[5]'s partial sums, which the recurrence starts from)
    and against the separate pass with its own direct sum (the round-4 path); all the other fused marches (residual +
    restriction, prolongation + two steps) run with the pinned cell too."""
    from petibm_amd import capi
    n = (256, 128, 136)
    w = [np.full(n[0], 1.0 / n[0]) * (1.0 + 0.3 * np.sin(np.arange(n[0]) / 17.0)),
         np.full(n[1], 1.0 / n[1]), np.full(n[2], 1.5 / n[2]) * (1.0 + 0.2 * np.cos(np.arange(n[2]) / 11.0))]
    dt = 0.01
    xs = np.random.default_rng(7).uniform(-1, 1, n[0] * n[1] * n[2])
    xs[0] = 0.0
    out = []
    for extra in ("", "pib_fuse_residual_update=0\npib_pin_sum_local=1\n", "pib_fuse_residual_update=0\n", "pib_pin_sum_local=0\n"):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=sweeps, post=sweeps, extra="pib_march_min_cells=0\n" + extra))
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_PINNED)
        b = np.empty_like(xs)
        s.matMult(xs, b)
        assert b[0] == 0.0  # (the identity row)
        x = np.zeros_like(xs)
        s.solve(x, b)
        r = np.empty_like(xs)
        s.matMult(x, r)
        out.append((x, np.array(s.getResidualHistory()), s.getIters(), np.linalg.norm(b - r) / np.linalg.norm(b), int(s.counters()[6])))
        s.destroy()
    fused, same_sum, direct, off = out
    assert fused[4] >= fused[2] and same_sum[4] == 0 and direct[4] == 0 and off[4] == 0  # pin_sum_local=0: no fused update under a pinned row
    assert 5 <= fused[2] <= 40 and fused[2] == same_sum[2] == direct[2] == off[2]
    assert fused[3] <= 2e-10 and fused[0][0] == 0.0
    scale = np.abs(direct[0]).max()
    assert np.abs(fused[0] - same_sum[0]).max() <= 1e-11 * scale and np.allclose(fused[1], same_sum[1], rtol=1e-9)
    assert np.abs(fused[0] - direct[0]).max() <= 1e-9 * scale and np.allclose(fused[1], direct[1], rtol=1e-7)
    assert np.array_equal(direct[0], off[0]) and np.array_equal(direct[1], off[1])
    assert np.abs(fused[0] - xs).max() <= 1e-6 * np.abs(xs).max()


@pytest.mark.parametrize("pinned", [False, True])
def test_iterations_enqueued_blind_are_no_ops_behind_done(lin, pinned):
    """`pib_check_every=64`: 64 iterations go out before the host looks at the device's scalars; the ones behind convergence are
    no-ops behind the device's `done` flag, and the x update the last live iteration owes is applied exactly once (the owed /
    applied counters of the scalars) -- the same solution, iteration count and history as with a poll after every iteration."""
    from petibm_amd import capi
    n = (256, 128, 136)
    w = [np.full(n[0], 1.0 / n[0]) * (1.0 + 0.3 * np.sin(np.arange(n[0]) / 17.0)),
         np.full(n[1], 1.0 / n[1]), np.full(n[2], 1.5 / n[2]) * (1.0 + 0.2 * np.cos(np.arange(n[2]) / 11.0))]
    dt = 0.01
    xs = np.random.default_rng(7).uniform(-1, 1, n[0] * n[1] * n[2])
    xs -= xs[0] if pinned else xs.mean()
    out = []
    for every in (64, 1):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=2, post=2, extra=f"pib_march_min_cells=0\npib_check_every={every}\n"))
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
        b = np.empty_like(xs)
        s.matMult(xs, b)
        x = np.zeros_like(xs)
        s.solve(x, b)
        out.append((x, s.getIters(), np.array(s.getResidualHistory())))
        s.destroy()
    assert out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][0], out[1][0])
    assert np.abs(out[0][0] - xs).max() <= 1e-6 * np.abs(xs).max()


@pytest.mark.parametrize("sr", [0, 1])
def test_search_direction_placed_against_x_is_bit_identical(lin, sr):
    """`pib_place_update_vector` (CG on one rank, 2^25 rows and more; here pushed down to every size): p moves to an allocation of
    its own, chosen by timing the p-update's access pattern against the caller's x while walking through fresh allocations
    (krylov.hip, place_update_vector; profiles/r05_vector_placement_lab.txt).  The probe computes x + (-0.0) * 0: a guess in x
    comes back bit for bit, and the solve is the one without the search.  One search per x the solver has not seen, three in a
    solver's life."""
    from petibm_amd import capi
    n = (64, 48, 40)
    w = [np.full(n[0], 1.0 / n[0]), np.full(n[1], 1.0 / n[1]) * (1.0 + 0.2 * np.sin(np.arange(n[1]) / 5.0)), np.full(n[2], 1.0 / n[2])]
    dt = 0.01
    rng = np.random.default_rng(11)
    xs = rng.uniform(-1, 1, n[0] * n[1] * n[2])
    xs -= xs.mean()
    guess = xs + 1e-3 * rng.uniform(-1, 1, xs.size)
    guess[::7] = -0.0  # (signed zeros survive the probe too)
    out = []
    for place in (1, 0):
        extra = f"pib_place_update_vector={place}\npib_place_min_rows=1000\npib_cg_single_reduction={sr}\n"
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=2, post=2, extra=extra))
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_CONSTANT)
        xs_d, b_d, x_d, x2_d = s.deviceVec(), s.deviceVec(), s.deviceVec(), s.deviceVec()
        xs_d.upload(xs)
        s.matMult(xs_d, b_d)
        x_d.upload(guess)
        s.solve(x_d, b_d)
        first = (x_d.download(), s.getIters(), np.array(s.getResidualHistory()))
        searches = [s.placement()[0]]
        x_d.upload(guess)
        s.solve(x_d, b_d)  # the same x: no second search
        searches.append(s.placement()[0])
        x2_d.upload(guess)
        s.solve(x2_d, b_d)  # another x: one more
        searches.append(s.placement()[0])
        again = (x2_d.download(), s.getIters(), np.array(s.getResidualHistory()))
        out.append((first, again, searches, s.placement(), s.placementInfo(), s.deviceMemInfo()))
        s.destroy()
    for k in (0, 1):
        assert out[0][k][1] == out[1][k][1] and np.array_equal(out[0][k][2], out[1][k][2]) and np.array_equal(out[0][k][0], out[1][k][0])
    assert np.array_equal(out[0][0][0], out[0][1][0])
    assert out[0][2] == [1, 1, 2] and out[1][2] == [0, 0, 0]
    assert out[0][3][1] >= 1 and out[0][3][2] > 0.0 and out[0][3][3] <= out[0][3][2]
    # the walk is bounded: what it held at one time stays under min(16 GiB, a tenth of the free memory)
    free = out[0][5][0]
    assert 0 <= out[0][4][4] <= min(16 << 30, free // 10 + (64 << 20)) and out[0][4][5] > 0.0
    assert out[1][4][4] == 0 and out[1][4][5] == 0.0
    err = out[0][0][0] - xs
    assert np.abs(err - err.mean()).max() <= 1e-6 * np.abs(xs).max()  # (up to the constant the guess brought)


def test_no_placement_search_on_a_device_whose_memory_is_mostly_taken(lin):
    """The walk allocates transients; on a device that somebody else has filled (another solver of the process, other ranks
    sharing the GPU) they would be the neighbour's to miss.  With 85 % of the HBM taken before the solver is created: no error,
    no candidate timed, nothing held, and the solve is the plain one bit for bit.  (The memory is taken through the library's
    own allocator: the pytest process never loads torch's HIP runtime, tests/conftest.py.)"""
    from petibm_amd import capi
    from petibm_amd.linsolver import DeviceVec
    n = (48, 40, 32)
    w = [np.full(n[0], 1.0 / n[0]), np.full(n[1], 1.0 / n[1]), np.full(n[2], 1.0 / n[2]) * (1.0 + 0.2 * np.cos(np.arange(n[2]) / 4.0))]
    xs = np.random.default_rng(5).uniform(-1, 1, n[0] * n[1] * n[2])
    xs -= xs.mean()
    out = []
    scratch = lin.LinSolverHIP("scratch", config_text=amgx_cfg())
    for fill in (False, True):
        hog = None
        if fill:
            free, total = scratch.deviceMemInfo()
            want = int(0.85 * total) - (total - free)
            assert want > 0
            hog = DeviceVec(scratch, want // 8)  # (allocated, never touched)
            free2, _ = scratch.deviceMemInfo()
            assert free2 < total // 2
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(pre=2, post=2, extra="pib_place_update_vector=1\npib_place_min_rows=1000\n"))
        s.assemblePoisson(list(n), w, 0.01, capi.NULLSPACE_CONSTANT)
        x_d, b_d, xs_d = s.deviceVec(), s.deviceVec(), s.deviceVec()
        xs_d.upload(xs)
        s.matMult(xs_d, b_d)
        x_d.upload(np.zeros_like(xs))
        s.solve(x_d, b_d)
        out.append((x_d.download(), s.getIters(), np.array(s.getResidualHistory()), s.placementInfo()))
        for v in (x_d, b_d, xs_d):
            v.free()
        s.destroy()
        if hog is not None:
            hog.free()
    scratch.destroy()
    assert out[0][3][1] >= 1 and out[0][3][4] > 0          # the free device: a search that timed candidates and held something
    assert out[1][3][1] == 0 and out[1][3][4] == 0          # the filled one: nothing timed, nothing held
    assert out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][0], out[1][0])


@pytest.mark.parametrize("solver,pc", [("PCG", "BLOCK_JACOBI"), ("PCG", "AMG"), ("PBICGSTAB", "BLOCK_JACOBI"), ("PBICGSTAB", "NOSOLVER")])
def test_work_vectors_as_separate_allocations_are_bit_identical(lin, solver, pc):
    """`pib_place_min_rows` (one rank, 2^25 rows and more; here pushed down to every size, -1: never): every work vector of the
    Krylov method is an allocation of its own instead of a slice of one pool.  Where a vector lives changes no bit of the solve."""
    from petibm_amd import capi
    n = (48, 40, 36)
    w = [np.full(n[0], 1.0 / n[0]) * (1.0 + 0.25 * np.sin(np.arange(n[0]) / 7.0)), np.full(n[1], 1.0 / n[1]), np.full(n[2], 1.2 / n[2])]
    dt = 0.01
    xs = np.random.default_rng(31).uniform(-1, 1, n[0] * n[1] * n[2])
    xs -= xs.mean()
    out = []
    for split in (-1, 1000):
        extra = f"pib_place_min_rows={split}\npib_place_update_vector=0\n"
        text = gmg_cfg(pre=2, post=2, extra=extra) if pc == "AMG" else amgx_cfg(solver=solver, pc=pc, tol=1e-9, extra=extra)
        s = lin.LinSolverHIP("poisson", config_text=text)
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_CONSTANT)
        b = np.empty_like(xs)
        s.matMult(xs, b)
        x = np.zeros_like(xs)
        s.solve(x, b)
        out.append((x, s.getIters(), np.array(s.getResidualHistory())))
        s.destroy()
    assert out[1][1] == out[0][1] and np.array_equal(out[1][2], out[0][2]) and np.array_equal(out[1][0], out[0][0])
    assert out[0][1] > 3


def test_one_sweep_of_the_solver_file_is_a_fused_pair_of_steps(lin):
    """`pib_sweep_pairs` (default 1): the reference's files say presweeps = postsweeps = 1 (AmgX's classical AMG); the
    geometric stand-in reads a sweep as one fused pair of damped-Jacobi steps.  The default with V(1,1) in the file is, bit
    for bit, the literal V(2,2); `pib_sweep_pairs=0` is the literal V(1,1) -- the oracle's cycles either way.  Chebyshev
    smoothing is not doubled."""
    from petibm_amd import capi
    dt = 0.01
    m, A, _ = poisson_system(stretched_3d((24, 20, 16)), dt=dt)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    file_v11 = gmg_cfg(pre=1, post=1).replace("pib_sweep_pairs=0\n", "")
    assert "pib_sweep_pairs" not in file_v11

    def run(text):
        s = lin.LinSolverHIP("poisson", config_text=text)
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(A.n_rows)
        s.solve(x, b)
        out = (x, s.getIters(), s.getResidualHistory().copy())
        s.destroy()
        return out

    default, v22, v11 = run(file_v11), run(gmg_cfg(pre=2, post=2)), run(gmg_cfg(pre=1, post=1))
    assert default[1] == v22[1] and np.array_equal(default[2], v22[2]) and np.array_equal(default[0], v22[0])
    assert v11[1] > v22[1]
    for got, (pre, post) in ((v22, (2, 2)), (v11, (1, 1))):
        ref = clib.GMG(n, w, dt, nullspace=1, pre=pre, post=post, omega=0.9, coarsest_sweeps=32).pcg(A, b, rtol=1e-10, maxit=200)
        assert iters_close(got[1], ref["iters"])
        assert np.allclose(got[2][:8], ref["history"][:8], rtol=1e-8)
    cheb = file_v11.replace("BLOCK_JACOBI", "CHEBYSHEV_POLY")
    c1, c0 = run(cheb), run(cheb + "pib_sweep_pairs=0\n")
    assert c1[1] == c0[1] and np.array_equal(c1[2], c0[2])
