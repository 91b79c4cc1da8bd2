"""the committed scipy-derived fixtures (tests/golden/scipy_poisson_*.npz, written by
tests/golden/make_scipy_fixtures.py from Kronecker products, scipy's sparse product and scipy's CG -- nothing of
oracle/ or petibm_amd/) against (CPU suite) the oracle's assembly and product, and (-m gpu) the HIP assembly, SpMV and
solves DIRECTLY: on the GPU box the product meets an implementation that is neither itself nor its own oracle."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["ref12x11", "uniform8", "stretched16"]


def load(name):
    f = np.load(os.path.join(HERE, "golden", f"scipy_poisson_{name}.npz"))
    dim = int(f["dim"])
    return f, dim, [f[f"w{d}"] for d in range(dim)], float(f["dt"])


def csr_matvec(rowptr, col, val, x):
    return np.add.reduceat(val * x[col], rowptr[:-1])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_assembly_and_product_match_the_scipy_fixture(name):
    from oracle import clib
    f, dim, w, dt = load(name)
    n = [len(v) for v in w]
    rp, cl, vl = clib.assemble_poisson32(n, w, dt)
    assert np.array_equal(rp, f["A_rowptr"]) and np.array_equal(cl, f["A_col"])
    assert np.allclose(vl, f["A_val"], rtol=1e-14, atol=0.0)
    y = np.empty(len(rp) - 1)
    clib.spmv32(len(rp) - 1, rp, cl, vl, np.ascontiguousarray(f["xr"]), y)
    assert np.abs(y - f["y"]).max() <= 1e-13 * np.abs(f["y"]).max()
    # the fixture's own consistency (numpy only)
    assert np.abs(csr_matvec(f["A_rowptr"], f["A_col"], f["A_val"], f["xs"]) - f["b"]).max() <= 1e-13 * np.abs(f["b"]).max()


def test_fixture_script_reproduces_the_committed_files(tmp_path):
    """(build container only: needs scipy) the committed .npz files are what the committed script writes"""
    pytest.importorskip("scipy.sparse")
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_scipy_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    f, dim, w, dt = load("stretched16")
    G, D, A = mk.operators([mk.stretched(16, 1.25), mk.stretched(16, 1.15), mk.stretched(16, 1.3)], 5e-3)
    assert np.array_equal(A.indptr, f["A_rowptr"]) and np.array_equal(A.indices, f["A_col"]) and np.array_equal(A.data, f["A_val"])
    assert np.array_equal(D.data, f["D_val"]) and np.array_equal(G.data, f["G_val"])
    ref = json.load(open(os.path.join(HERE, "golden", "reference_test_vectors.json")))["cartesianmesh2d_dirichlet"]
    f2 = load("ref12x11")[0]
    assert np.array_equal(f2["w0"], np.array(ref["dLTrue"][3][0])) and np.array_equal(f2["w1"], np.array(ref["dLTrue"][3][1]))


AMG = ("config_version=2\nsolver(solv)=PCG\nsolv:max_iters=500\nsolv:monitor_residual=1\nsolv:convergence=RELATIVE_INI\n"
       "solv:tolerance=1e-10\nsolv:norm=L2\nsolv:store_res_history=1\nsolv:preconditioner(prec)={pc}\nprec:relaxation_factor=1.0\n"
       "prec:cycle=V\nprec:presweeps=1\nprec:postsweeps=1\nprec:smoother(smooth)=BLOCK_JACOBI\nsmooth:relaxation_factor=0.9\n"
       "pib_initial_guess_nonzero=0\n")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_assembly_product_and_solves_match_the_scipy_fixture(name):
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    f, dim, w, dt = load(name)
    n = [len(v) for v in w]
    rp, cl, vl = f["A_rowptr"], f["A_col"], f["A_val"]
    b, N = np.ascontiguousarray(f["b"]), len(f["b"])
    for pc in ("AMG", "BLOCK_JACOBI"):
        s = LinSolverHIP("poisson", config_text=AMG.format(pc=pc))
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        hrp, hcl, hvl = s.getCSR()
        assert np.array_equal(np.asarray(hrp, dtype=np.int64), rp) and np.array_equal(np.asarray(hcl, dtype=np.int64), cl)
        assert np.allclose(hvl, vl, rtol=1e-14, atol=0.0)
        y = np.empty(N)
        s.matMult(np.ascontiguousarray(f["xr"]), y)
        assert np.abs(y - f["y"]).max() <= 1e-13 * np.abs(f["y"]).max()
        x = np.zeros(N)
        s.solve(x, b)
        assert s.getReason() > 0
        assert np.linalg.norm(b - csr_matvec(rp, cl, vl, x)) <= 1.5e-10 * np.linalg.norm(b)       # with the FIXTURE's operator
        assert np.linalg.norm((x - x.mean()) - f["x_cg"]) <= 1e-8 * np.linalg.norm(f["x_cg"])
        assert np.linalg.norm((x - x.mean()) - f["xs"]) <= 1e-8 * np.linalg.norm(f["xs"])
        s.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_the_fixture_matrix_through_setmatrix_only(name):
    """the route of an unchanged PetIBM: scipy's DBNG handed over as a matrix, nothing else -- the mesh structure is
    recovered from its entries, the multigrid solves it"""
    from petibm_amd.linsolver import LinSolverHIP

    class M:
        pass

    f, dim, w, dt = load(name)
    A = M()
    A.rowptr, A.col, A.val = f["A_rowptr"], f["A_col"], f["A_val"]
    s = LinSolverHIP("poisson", config_text=AMG.format(pc="AMG"))
    s.setMatrix(A)
    st = s.gridStructure()
    assert st is not None and st["detected"] and tuple(st["n"]) == tuple(len(v) for v in w)
    b, N = np.ascontiguousarray(f["b"]), len(f["b"])
    x = np.zeros(N)
    s.solve(x, b)
    assert s.getReason() > 0 and s.getIters() <= 40
    assert np.linalg.norm(b - csr_matvec(A.rowptr, A.col, A.val, x)) <= 1.5e-10 * np.linalg.norm(b)
    assert np.linalg.norm((x - x.mean()) - f["x_cg"]) <= 1e-8 * np.linalg.norm(f["x_cg"])
    s.destroy()


# ---- round 5: the velocity operator (tests/golden/scipy_velocity_*.npz: L, A = I/dt - c nu L with the ghost-point folds of
# createlaplacian.cpp:232-243 -- Dirichlet, Neumann and convective faces --, D with the Neumann fold of createdivergence.cpp:231-242;
# Kronecker products + explicit boundary-row edits, scipy's sparse LU solution; nothing of oracle/ or petibm_amd/)
VNAMES = ["ref12x11", "stretched16", "dirichlet8"]


def vload(name):
    f = np.load(os.path.join(HERE, "golden", f"scipy_velocity_{name}.npz"))
    dim = int(f["dim"])
    return f, dim, [f[f"w{d}"] for d in range(dim)], float(f["dt"])


def config_from_fixture(f, dim, w):
    """the reference-shaped config dict of the fixture's mesh (one sub-domain per cell: arbitrary widths) and faces"""
    names = "xyz"
    mesh = []
    for d in range(dim):
        x = float(f["lo"][d]) + np.concatenate([[0.0], np.cumsum(w[d])])
        mesh.append({"direction": names[d], "start": float(x[0]),
                     "subDomains": [{"end": float(x[i + 1]), "cells": 1, "stretchRatio": 1.0} for i in range(len(w[d]))]})
    locs = ["xMinus", "xPlus", "yMinus", "yPlus", "zMinus", "zPlus"]
    bcs = []
    for d in range(dim):
        for side in (0, 1):
            bc = {"location": locs[2 * d + side]}
            for c in "uvw"[:dim]:
                bc[c] = [str(f["bc"][d][side]), 0.0]
            bcs.append(bc)
    return {"mesh": mesh, "flow": {"boundaryConditions": bcs}}


def to_dense_rows(rowptr, col, val, shape):
    import numpy as _np
    M = _np.zeros(tuple(int(v) for v in shape))
    for r in range(len(rowptr) - 1):
        for k in range(int(rowptr[r]), int(rowptr[r + 1])):
            M[r, int(col[k])] += val[k]
    return M


@pytest.mark.parametrize("name", VNAMES)
def test_oracle_velocity_operator_and_divergence_match_the_scipy_fixture(name):
    """the oracle's createLaplacian / createDivergence (a0 folds included) and A = I/dt - c nu L against the independent fixture:
    the same products on a random vector to 1e-13, the same entries where the fixture is small enough to compare densely"""
    from oracle import clib, mesh as omesh, operators as oops
    f, dim, w, dt = vload(name)
    m = omesh.create_mesh(config_from_fixture(f, dim, w))
    for d in range(dim):
        assert np.allclose(m.dL[3][d].true, w[d], rtol=1e-12, atol=0.0)
    L, D = oops.create_laplacian(m), oops.create_divergence(m)
    assert L.n_rows == int(f["L_shape"][0]) and D.n_rows == int(f["D_shape"][0])
    ur = np.ascontiguousarray(f["ur"])
    assert np.abs(clib.spmv(L, ur) - f["yL"]).max() <= 1e-12 * np.abs(f["yL"]).max()
    assert np.abs(clib.spmv(D, ur) - f["yD"]).max() <= 1e-12 * np.abs(f["yD"]).max()
    A = oops.create_velocity_operator(L, dt, float(f["cnu"])) if hasattr(oops, "create_velocity_operator") else None
    if A is None:
        yA = ur / dt - float(f["cnu"]) * clib.spmv(L, ur)
    else:
        yA = clib.spmv(A, ur)
    assert np.abs(yA - f["y"]).max() <= 1e-12 * np.abs(f["y"]).max()
    if L.n_rows <= 2000:
        Ld = to_dense_rows(L.rowptr, L.col, L.val, f["L_shape"])
        Lf = to_dense_rows(f["L_rowptr"], f["L_col"], f["L_val"], f["L_shape"])
        assert np.abs(Ld - Lf).max() <= 1e-12 * np.abs(Lf).max()
        Dd = to_dense_rows(D.rowptr, D.col, D.val, f["D_shape"])
        Df = to_dense_rows(f["D_rowptr"], f["D_col"], f["D_val"], f["D_shape"])
        assert np.abs(Dd - Df).max() <= 1e-13 * np.abs(Df).max()
    # the fixture's own consistency
    assert np.abs(csr_matvec(f["A_rowptr"], f["A_col"], f["A_val"], f["us"]) - f["b"]).max() <= 1e-13 * np.abs(f["b"]).max()
    assert np.linalg.norm(f["x_lu"] - f["us"]) <= 1e-10 * np.linalg.norm(f["us"])


VEL = ("config_version=2\nsolver(solv)={solver}\nsolv:max_iters=500\nsolv:monitor_residual=1\nsolv:convergence=RELATIVE_INI\n"
       "solv:tolerance=1e-11\nsolv:norm=L2\nsolv:store_res_history=1\nsolv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=1.0\n"
       "pib_initial_guess_nonzero=0\n")


def fixture_a0(f, dim):
    """(3, 6) ghost coefficients as pib_assemble_velocity wants them: a0[field][2 * direction + side]"""
    table = {"DIRICHLET": (0.0, -1.0), "CONVECTIVE": (0.0, -1.0), "NEUMANN": (1.0, 1.0)}
    a0 = np.zeros((3, 6))
    for fld in range(dim):
        for d in range(dim):
            for side in (0, 1):
                a0[fld, 2 * d + side] = table[str(f["bc"][d][side])][0 if d == fld else 1]
    return a0


@pytest.mark.gpu
@pytest.mark.parametrize("name", VNAMES)
def test_hip_velocity_assembly_product_and_solves_match_the_scipy_fixture(name):
    """pib_assemble_velocity (csrc/assemble.hip k_assemble_velocity: the reference's createLaplacian + MatScale / MatShift,
    navierstokes.cpp:342-344) against the fixture's A entry by entry, the CSR and matrix-free products on the fixture's random
    vector, and the BiCGStab and Chebyshev solves against scipy's sparse LU solution -- an implementation that is neither the
    product nor its oracle, on the non-symmetric operator with all three kinds of ghost fold."""
    from petibm_amd.linsolver import LinSolverHIP
    f, dim, w, dt = vload(name)
    n = [len(v) for v in w]
    lo = [float(v) for v in f["lo"]]
    hi = [lo[d] + float(np.sum(w[d])) for d in range(dim)]
    rp, cl, vl = f["A_rowptr"], f["A_col"], f["A_val"]
    b, N = np.ascontiguousarray(f["b"]), len(f["b"])
    for solver in ("PBICGSTAB", "CHEBYSHEV"):
        for mf in (1, 0):
            s = LinSolverHIP("velocity", config_text=VEL.format(solver=solver) + f"pib_matrix_free_velocity={mf}\n")
            s.assembleVelocity(n, w, lo, hi, fixture_a0(f, dim), dt, float(f["cnu"]))
            hrp, hcl, hvl = s.getCSR()
            assert np.array_equal(np.asarray(hrp, dtype=np.int64), rp) and np.array_equal(np.asarray(hcl, dtype=np.int64), cl)
            assert np.allclose(hvl, vl, rtol=1e-12, atol=0.0)
            y = np.empty(N)
            s.matMult(np.ascontiguousarray(f["ur"]), y)
            assert np.abs(y - f["y"]).max() <= 1e-13 * np.abs(f["y"]).max()
            x = np.zeros(N)
            s.solve(x, b)
            assert s.getReason() > 0
            assert np.linalg.norm(b - csr_matvec(rp, cl, vl, x)) <= 1e-10 * np.linalg.norm(b)   # with the FIXTURE's operator
            assert np.linalg.norm(x - f["x_lu"]) <= 1e-9 * np.linalg.norm(f["x_lu"])
            s.destroy()
