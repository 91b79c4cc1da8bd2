"""small fixtures from an INDEPENDENT second implementation -- scipy.sparse (Kronecker products of 1-D
difference matrices, the sparse product `D @ (dt G)`, scipy's own CG) -- for the operators of the hot path on

  * the reference's 12 x 11 test mesh (cell widths = the dL values its own test holds:
    tests/mesh/cartesianmesh2d_dirichlet.cpp:220-270, via reference_test_vectors.json),
  * an 8^3 uniform and a 16^3 stretched cavity (three geometric sub-domains per direction, written out here).

Nothing of oracle/ or petibm_amd/ is imported: D, G and DBNG are rebuilt from the definitions in
createdivergence.cpp:135-152 (+- the product of the two perpendicular widths at the cell's two faces),
creategradient.cpp:64-87 (-+ 1 / distance of the two pressure points) and navierstokes.cpp:349-356 (BN = dt I), all walls
Dirichlet (no ghost fold into D).  Build container only (scipy does not travel to the GPU box); the .npz files do:

    python tests/golden/make_scipy_fixtures.py            # writes tests/golden/scipy_poisson_<name>.npz

Each file: widths w0..w2, dt, the CSR triplets of G, D and DBNG (sorted columns), a zero-mean x*, b = DBNG x*, y = DBNG xr for
a second random vector, and x_cg = scipy's CG solution of (-DBNG) x = -b to 1e-13 (mean removed).
"""
import json
import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))


def grad1(w):
    n = len(w)
    h = 0.5 * (w[:-1] + w[1:])
    return sp.diags([-1.0 / h, 1.0 / h], [0, 1], shape=(n - 1, n), format="csr")


def div1(n):
    # cell s: +1 at its + face (face s, s < n-1), -1 at its - face (face s-1, s > 0)
    return sp.diags([np.ones(n - 1), -np.ones(n - 1)], [0, -1], shape=(n, n - 1), format="csr")


def kron3(az, ay, ax):
    return sp.kron(az, sp.kron(ay, ax, format="csr"), format="csr")


def operators(w, dt):
    """w: list of 1-D width arrays (2 or 3).  x fastest: cell (i, j, k) -> i + nx (j + ny k); velocity vector [u | v | w]."""
    dim = len(w)
    w = [np.asarray(v, dtype=np.float64) for v in w] + [np.ones(1)] * (3 - dim)
    n = [len(v) for v in w]
    I = [sp.identity(m, format="csr") for m in n]
    Wd = [sp.diags(v, format="csr") for v in w]
    Gs, Ds = [], []
    if True:
        Gs.append(kron3(I[2], I[1], grad1(w[0])))
        Ds.append(kron3(Wd[2], Wd[1], div1(n[0])))
        Gs.append(kron3(I[2], grad1(w[1]), I[0]))
        Ds.append(kron3(Wd[2], div1(n[1]), Wd[0]))
    if dim == 3:
        Gs.append(kron3(grad1(w[2]), I[1], I[0]))
        Ds.append(kron3(div1(n[2]), Wd[1], Wd[0]))
    G = sp.vstack(Gs, format="csr")
    D = sp.hstack(Ds, format="csr")
    A = (D @ (dt * G)).tocsr()
    for M in (G, D, A):
        M.sum_duplicates()
        M.sort_indices()
    return G, D, A


def stretched(n, r):
    """three sub-domains: geometric refinement towards the middle block (ratio 1/r, 1, r)"""
    a = n // 4
    left = (1.0 / r) ** np.arange(a)
    mid = np.full(n - 2 * a, left[-1])
    right = left[-1] * r ** np.arange(1, a + 1)
    w = np.concatenate([left, mid, right])
    return w / w.sum()


def make(name, w, dt, seed):
    G, D, A = operators(w, dt)
    N = A.shape[0]
    rng = np.random.default_rng(seed)
    xs = rng.uniform(-1.0, 1.0, N)
    xs -= xs.mean()
    b = A @ xs
    xr = rng.uniform(-1.0, 1.0, N)
    y = A @ xr
    assert abs(A.sum(axis=1)).max() < 1e-9 * abs(A.diagonal()).max()      # constants are the null space
    assert abs(A - A.T).max() < 1e-12 * abs(A.diagonal()).max()           # symmetric
    x_cg, info = spla.cg(-A, -b, rtol=1e-13, atol=0.0, maxiter=20000)
    assert info == 0
    x_cg -= x_cg.mean()
    assert np.linalg.norm(x_cg - xs) <= 1e-8 * np.linalg.norm(xs)
    out = {"dt": np.float64(dt), "dim": np.int64(len(w)), "xs": xs, "b": b, "xr": xr, "y": y, "x_cg": x_cg}
    for d, wd in enumerate(w):
        out[f"w{d}"] = np.asarray(wd, dtype=np.float64)
    for tag, M in (("G", G), ("D", D), ("A", A)):
        out[f"{tag}_rowptr"] = M.indptr.astype(np.int64)
        out[f"{tag}_col"] = M.indices.astype(np.int64)
        out[f"{tag}_val"] = M.data.astype(np.float64)
        out[f"{tag}_shape"] = np.array(M.shape, dtype=np.int64)
    path = os.path.join(HERE, f"scipy_poisson_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {N} cells, nnz(A) = {A.nnz}, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    ref = json.load(open(os.path.join(HERE, "reference_test_vectors.json")))["cartesianmesh2d_dirichlet"]
    w2 = [np.array(ref["dLTrue"][3][0], dtype=np.float64), np.array(ref["dLTrue"][3][1], dtype=np.float64)]
    assert len(w2[0]) == 12 and len(w2[1]) == 11
    make("ref12x11", w2, 0.01, 11)
    make("uniform8", [np.full(8, 1.0 / 8)] * 3, 1e-3, 12)
    make("stretched16", [stretched(16, 1.25), stretched(16, 1.15), stretched(16, 1.3)], 5e-3, 13)
