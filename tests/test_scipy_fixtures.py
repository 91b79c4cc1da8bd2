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
