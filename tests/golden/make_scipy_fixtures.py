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


# ---------------------------------------------------------------------------------------------------------------------------
# Round 5: the velocity operator.  L (createlaplacian.cpp:108-162) and A = I/dt - c nu L (navierstokes.cpp:342-344) with the
# ghost-point folds of createlaplacian.cpp:232-243, and D with the Neumann fold of createdivergence.cpp:231-242, rebuilt from the
# definitions:
#   * staggered points (cartesianmesh.cpp:225-330): component f along its own direction sits on the interior vertices (n - 1
#     points, control-volume width = mean of the two adjacent cell widths, the two boundary vertices are its ghost points); along
#     the other directions on the cell centres (n points, width = the cell's, ghost points mirrored half a cell outside);
#   * second difference of point s along a direction: 1 / (dneg dself) towards s - 1, 1 / (dpos dself) towards s + 1
#     (createlaplacian.cpp:134-151), the diagonal minus their sum over all directions;
#   * a ghost value is a0 * (the boundary-adjacent point) + a1, so its coefficient times a0 lands on the row's own diagonal
#     (misc.cpp:226-260: the target of a ghost is the point next to it);  a0 per face and component
#     (singleboundarydirichlet.cpp:34-43, singleboundaryneumann.cpp:27-28, singleboundaryconvective.cpp:20-47):
#     DIRICHLET / CONVECTIVE: 0 for the component normal to the face, -1 for a tangential one;  NEUMANN: 1.
A0 = {"DIRICHLET": (0.0, -1.0), "CONVECTIVE": (0.0, -1.0), "NEUMANN": (1.0, 1.0)}  # (normal, tangential)


def lap1(w, lo, f_is_dir, a0_lo, a0_hi):
    w = np.asarray(w, dtype=np.float64)
    n = len(w)
    vert = lo + np.concatenate([[0.0], np.cumsum(w)])
    if f_is_dir:
        full = vert                                   # ghosts = the two boundary vertices
        dl = 0.5 * (w[:-1] + w[1:])
    else:
        cen = vert[:-1] + 0.5 * w
        full = np.concatenate([[vert[0] - 0.5 * w[0]], cen, [vert[-1] + 0.5 * w[-1]]])
        dl = w
    m = len(full) - 2
    dneg = full[1:-1] - full[:-2]
    dpos = full[2:] - full[1:-1]
    am, ap = 1.0 / (dneg * dl), 1.0 / (dpos * dl)
    T = sp.diags([am[1:], -(am + ap), ap[:-1]], [-1, 0, 1], shape=(m, m), format="lil")
    T[0, 0] += am[0] * a0_lo
    T[m - 1, m - 1] += ap[m - 1] * a0_hi
    return T.tocsr()


def velocity_operators(w, lo, bc, dt, cnu):
    """w: 2 or 3 width arrays; lo: domain start per direction; bc[d] = (type at the minus face, type at the plus face) of direction
    d, the same type for every component.  Returns L, A = I/dt - cnu L (packed [u | v | w], x fastest in every block) and D with the
    Neumann fold."""
    dim = len(w)
    w3 = [np.asarray(v, dtype=np.float64) for v in w] + [np.ones(1)] * (3 - dim)
    lo3 = list(lo) + [0.0] * (3 - dim)
    n = [len(v) for v in w3]
    blocks, dblocks = [], []
    for f in range(dim):
        nf = [n[d] - 1 if d == f else n[d] for d in range(3)]
        I = [sp.identity(m, format="csr") for m in nf]
        Lf = None
        for d in range(dim):
            a0 = [A0[bc[d][side]][0 if d == f else 1] for side in (0, 1)]
            T = lap1(w3[d], lo3[d], d == f, a0[0], a0[1])
            mats = [I[0], I[1], I[2]]
            mats[d] = T
            term = kron3(mats[2], mats[1], mats[0])
            Lf = term if Lf is None else Lf + term
        blocks.append(Lf.tocsr())
        # divergence block of component f: +area at a cell's plus face, -area at its minus face; a NEUMANN face of direction f folds
        # the boundary flux (ghost = 1 * the adjacent interior face + a1) onto that interior face's column
        d1 = div1(n[f]).tolil()
        if A0[bc[f][0]][0] != 0.0:
            d1[0, 0] += -1.0 * A0[bc[f][0]][0]
        if A0[bc[f][1]][0] != 0.0:
            d1[n[f] - 1, n[f] - 2] += 1.0 * A0[bc[f][1]][0]
        Wd = [sp.diags(v, format="csr") for v in w3]
        mats = [Wd[0], Wd[1], Wd[2]]
        mats[f] = d1.tocsr()
        dblocks.append(kron3(mats[2], mats[1], mats[0]))
    L = sp.block_diag(blocks, format="csr")
    N = L.shape[0]
    A = (sp.identity(N, format="csr") / dt - cnu * L).tocsr()
    D = sp.hstack(dblocks, format="csr")
    for M in (L, A, D):
        M.sum_duplicates()
        M.sort_indices()
    return L, A, D


def make_velocity(name, w, lo, bc, dt, nu, seed):
    cnu = 0.5 * nu  # Crank-Nicolson (timeintegration.h: implicit coefficient 0.5)
    L, A, D = velocity_operators(w, lo, bc, dt, cnu)
    N = A.shape[0]
    rng = np.random.default_rng(seed)
    us = rng.uniform(-1.0, 1.0, N)
    b = A @ us
    ur = rng.uniform(-1.0, 1.0, N)
    x_lu = spla.spsolve(A.tocsc(), b)
    assert np.linalg.norm(x_lu - us) <= 1e-10 * np.linalg.norm(us)
    out = {"dt": np.float64(dt), "nu": np.float64(nu), "cnu": np.float64(cnu), "dim": np.int64(len(w)), "us": us, "b": b, "ur": ur, "y": A @ ur,
           "yL": L @ ur, "yD": D @ ur, "x_lu": x_lu, "lo": np.asarray(lo, dtype=np.float64),
           "bc": np.array([[t for t in bc[d]] for d in range(len(w))])}
    for d, wd in enumerate(w):
        out[f"w{d}"] = np.asarray(wd, dtype=np.float64)
    for tag, M in (("L", L), ("A", A), ("D", D)):
        out[f"{tag}_rowptr"] = M.indptr.astype(np.int64)
        out[f"{tag}_col"] = M.indices.astype(np.int64)
        out[f"{tag}_val"] = M.data.astype(np.float64)
        out[f"{tag}_shape"] = np.array(M.shape, dtype=np.int64)
    path = os.path.join(HERE, f"scipy_velocity_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {N} velocity points, nnz(A) = {A.nnz}, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    ref = json.load(open(os.path.join(HERE, "reference_test_vectors.json")))["cartesianmesh2d_dirichlet"]
    w2 = [np.array(ref["dLTrue"][3][0], dtype=np.float64), np.array(ref["dLTrue"][3][1], dtype=np.float64)]
    assert len(w2[0]) == 12 and len(w2[1]) == 11
    make("ref12x11", w2, 0.01, 11)
    make("uniform8", [np.full(8, 1.0 / 8)] * 3, 1e-3, 12)
    make("stretched16", [stretched(16, 1.25), stretched(16, 1.15), stretched(16, 1.3)], 5e-3, 13)
    # the velocity operator: the reference's 12 x 11 mesh (its own domain start, tests/mesh/cartesianmesh2d_dirichlet.cpp) with a
    # convective outlet and a Neumann top, and the 16^3 stretched cavity with one face of every kind
    lo2 = [float(ref["coordTrue"][4][0][0]), float(ref["coordTrue"][4][1][0])] if "coordTrue" in ref else [0.1, 0.05]
    make_velocity("ref12x11", w2, lo2, [("DIRICHLET", "CONVECTIVE"), ("DIRICHLET", "NEUMANN")], 0.01, 0.01, 21)
    make_velocity("stretched16", [stretched(16, 1.25), stretched(16, 1.15), stretched(16, 1.3)], [0.0, -0.5, 0.25],
                  [("DIRICHLET", "CONVECTIVE"), ("NEUMANN", "DIRICHLET"), ("DIRICHLET", "NEUMANN")], 5e-3, 1e-2, 22)
    make_velocity("dirichlet8", [np.full(8, 1.0 / 8)] * 3, [0.0, 0.0, 0.0], [("DIRICHLET", "DIRICHLET")] * 3, 1e-3, 1e-3, 23)
