"""oracle/dmda.py: the DMDA process grid, ownership and orderings restated for the multi-rank drop-in tests.

PETSc (da2.c / da3.c) is a third-party dependency absent from /root/reference and the reference holds no vector for the
grid PETSC_DECIDE picks ("parity unpinned" in the module header); what CAN be pinned is pinned here: the orderings are
permutations, the (1,1,P) grid gives back the natural ordering the z-slab route assumes (cartesianmesh.cpp:673), the
packed ordering follows getPackedGlobalIndex (cartesianmesh.cpp:741-779) restated independently below, and the grids
for the BASELINE configurations are the ones the round-2 review derived by hand ((1,1,2), (1,2,2), (2,2,2) on a cube).
"""
import numpy as np
import pytest

from oracle import dmda, mesh as omesh, operators as oops


def test_petsc_decide_grids_of_the_baseline_configurations():
    assert dmda.decide_process_grid((512, 512, 512), 1) == (1, 1, 1)
    assert dmda.decide_process_grid((512, 512, 512), 2) == (1, 1, 2)
    assert dmda.decide_process_grid((512, 512, 512), 4) == (1, 2, 2)
    assert dmda.decide_process_grid((512, 512, 512), 8) == (2, 2, 2)
    assert dmda.decide_process_grid((384, 256, 256), 8) == (2, 2, 2)      # config 5
    assert dmda.decide_process_grid((128, 128), 4) == (2, 2, 1)
    for dims in ((7, 9, 11), (30, 10, 5), (450, 450), (100, 7)):
        for size in (1, 2, 3, 4, 6, 8, 12):
            g = dmda.decide_process_grid(dims, size)
            assert g[0] * g[1] * g[2] == size and min(g) >= 1


def test_ownership_is_the_dmda_default_split():
    assert dmda.ownership(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert dmda.ownership(512, 8) == [(64 * r, 64) for r in range(8)]
    assert [c for _, c in dmda.ownership(7, 4)] == [2, 2, 2, 1]


@pytest.mark.parametrize("n,P,grid", [((6, 5, 4), 4, None), ((7, 6, 9), 8, None), ((8, 6), 4, None), ((5, 9, 7), 6, (1, 2, 3)),
                                      ((6, 4, 5), 4, (4, 1, 1))])
def test_orderings_are_permutations_and_boxes_are_numbered_in_their_own_natural_order(n, P, grid):
    m = omesh.create_mesh(omesh.uniform_config(n))
    L = dmda.dmda_layout(m, P, grid)
    lay = L.pressure
    assert sorted(lay.petsc_of_natural.tolist()) == list(range(m.pN))
    assert sorted(L.packed_of_natural.tolist()) == list(range(m.UN))
    n3 = lay.n
    for r, (xs, ys, zs, xm, ym, zm) in enumerate(lay.boxes):
        for (i, j, k) in ((0, 0, 0), (xm - 1, 0, 0), (0, ym - 1, 0), (xm - 1, ym - 1, zm - 1)):
            nat = (xs + i) + n3[0] * ((ys + j) + n3[1] * (zs + k))
            assert lay.petsc_of_natural[nat] == lay.offsets[r] + i + xm * (j + ym * k)
            assert lay.rank_of_natural[nat] == r
    # getPackedGlobalIndex, restated point by point: offset of the owner's pack + the fields before f on that rank + the
    # point's place in the owner's box of field f
    dim = m.dim
    base = 0
    for f in range(dim):
        Lf = L.velocity[f]
        nf = int(np.prod(Lf.n))
        for nat in (0, nf // 3, nf - 1):
            p = int(Lf.rank_of_natural[nat])
            before = sum(int(L.velocity[e].offsets[p + 1] - L.velocity[e].offsets[p]) for e in range(f))
            want = int(L.packed_offsets[p]) + before + int(Lf.petsc_of_natural[nat] - Lf.offsets[p])
            assert int(L.packed_of_natural[base + nat]) == want and int(L.packed_rank[base + nat]) == p
        base += nf


def test_slab_grid_is_the_natural_ordering():
    m = omesh.create_mesh(omesh.uniform_config((5, 4, 9)))
    L = dmda.dmda_layout(m, 3, (1, 1, 3))
    assert np.array_equal(L.pressure.petsc_of_natural, np.arange(m.pN))
    assert L.pressure.offsets.tolist() == [0, 60, 120, 180]


def test_permuted_local_rows_is_the_permuted_matrix():
    m = omesh.create_mesh(omesh.uniform_config((6, 5, 4)))
    D, G, Lp = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, Lp, 0.01, 0.005)
    L = dmda.dmda_layout(m, 4)
    Pm = np.zeros((m.pN, m.pN))
    Pm[L.pressure.petsc_of_natural, np.arange(m.pN)] = 1.0
    ref = Pm @ A.to_dense() @ Pm.T
    parts = [dmda.permuted_local_rows(A, L.pressure.petsc_of_natural, L.pressure.offsets, r) for r in range(4)]
    assert np.array_equal(np.vstack([p[0].to_dense() for p in parts]), ref)
    for (csr, row0), r in zip(parts, range(4)):
        assert row0 == L.pressure.offsets[r]
        for l in range(csr.n_rows):
            c = csr.col[csr.rowptr[l]:csr.rowptr[l + 1]]
            assert np.all(np.diff(c) > 0)   # ascending global columns, as MatMPIAIJGetLocalMat delivers them
