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
@pytest.mark.parametrize("variant", [0, 2])
@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched"])
def test_spmv_bit_exact(lin, case, variant):
    cfg = {"2d_stretched": STRETCHED_2D, "3d_stretched": stretched_3d()}[case]
    m, DBNG, L = poisson_system(cfg)
    for A in (DBNG, L, oops.create_velocity_operator(L, 0.01, 0.005)):
        s = lin.LinSolverHIP("velocity", config_text=amgx_cfg(extra=f"pib_spmv_variant={variant}\n"))
        s.setMatrix(A)
        x = np.random.default_rng(7).uniform(-1, 1, A.n_cols)
        y = np.empty(A.n_rows)
        s.matMult(x, y)
        assert np.array_equal(y, clib.spmv(A, x))
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
    assert abs(s.getIters() - ref["iters"]) <= 1
    # residual contract, recomputed by the oracle
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    h = s.getResidualHistory()
    k = min(len(h), len(ref["history"])) - 1
    assert np.allclose(h[:k], ref["history"][:k], rtol=1e-7, atol=1e-14 * h[0])
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
    assert abs(s.getIters() - ref["iters"]) <= 1
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
    assert abs(k.getIters() - ref["iters"]) <= 1
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
    s.solve(xh, b)
    assert s.getIters() == it_host
    s.destroy()
