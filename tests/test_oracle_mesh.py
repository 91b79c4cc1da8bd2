"""The oracle's mesh arithmetic pinned against the reference's own golden vectors
(tests/golden/reference_test_vectors.json holds the data of
tests/mesh/cartesianmesh2d_dirichlet.cpp, cartesianmesh2d_yperiodic.cpp and
cartesianmesh3d_dirichlet.cpp)."""
import json
import os

import numpy as np
import pytest

from oracle import mesh as omesh

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_test_vectors.json")))


@pytest.mark.parametrize("name", ["cartesianmesh2d_dirichlet", "cartesianmesh2d_yperiodic"])
def test_mesh_2d_golden(name):
    g = G[name]
    m = omesh.create_mesh(g["config"])
    assert m.dim == g["dim"]
    assert np.array_equal(m.n, np.array(g["n"]))
    assert m.UN == g["UN"] and m.pN == g["pN"]
    assert np.array_equal(m.periodic, np.array(g["periodic"]))
    if "min" in g:
        assert np.allclose(m.min, g["min"], rtol=0, atol=1e-15)
        assert np.allclose(m.max, g["max"], rtol=0, atol=1e-15)
    tol = g["tol"]
    # ASSERT_NEAR(coord[f][d][i], mesh->coord[f][d][i], 1e-12) for i in [-1, n] (velocity fields) ...
    for f in range(2):
        for d in range(2):
            exp = np.array(g["coordTrue"][f][d])
            got = m.coord[f][d][np.arange(-1, m.n[f][d] + 1)]
            assert np.abs(got - exp[: len(got)]).max() <= tol, (f, d)
            expd = np.array(g["dLTrue"][f][d])
            gotd = m.dL[f][d][np.arange(-1, m.n[f][d] + 1)]
            assert np.abs(gotd - expd[: len(gotd)]).max() <= tol, (f, d)
        assert abs(m.coord[f][2][0] - g["coordTrue"][f][2][0]) <= tol
        assert abs(m.dL[f][2][0] - g["dLTrue"][f][2][0]) <= tol
    # ... and for i in [0, n) for pressure (3) and vertex (4) fields
    for f in (3, 4):
        for d in range(2):
            exp = np.array(g["coordTrue"][f][d])
            got = m.coord[f][d][np.arange(m.n[f][d])]
            assert np.abs(got - exp).max() <= tol
    for d in range(2):
        assert np.abs(m.dL[3][d][np.arange(m.n[3][d])] - np.array(g["dLTrue"][3][d])).max() <= tol


def test_mesh_3d_golden():
    g = G["cartesianmesh3d_dirichlet"]
    m = omesh.create_mesh(g["config"])
    assert m.dim == 3
    assert np.array_equal(m.min, g["min"]) and np.array_equal(m.max, g["max"])
    assert np.array_equal(m.n, np.array(g["n"]))  # w has 10 points in z: periodic (:70-76)


def test_natural_and_packed_index_tables():
    """cartesianmesh2d_dirichlet.cpp:447-1013 checks natural/global/packed maps programmatically:
    interior = i + j*nx (+ k*nx*ny), every ghost of a non-periodic boundary = -1, packed = block offset."""
    m = omesh.create_mesh(G["cartesianmesh2d_dirichlet"]["config"])
    for f in (0, 1, 3):
        n0, n1 = int(m.n[f][0]), int(m.n[f][1])
        for j in range(n1):
            for i in range(n0):
                assert m.natural_index(f, i, j, 0) == i + j * n0
        for i in range(n0):
            assert m.natural_index(f, i, -1, 0) == -1 and m.natural_index(f, i, n1, 0) == -1
        for j in range(n1):
            assert m.natural_index(f, -1, j, 0) == -1 and m.natural_index(f, n0, j, 0) == -1
        for ii, jj in ((-1, -1), (-1, n1), (n0, -1), (n0, n1)):
            assert m.natural_index(f, ii, jj, 0) == -1
    assert m.packed_index(1, 0, 0, 0) == 11 * 11  # v block starts after the u block
    assert m.packed_index(3, 5, 4, 0) == 5 + 4 * 12  # pressure is not packed
    mp = omesh.create_mesh(G["cartesianmesh2d_yperiodic"]["config"])
    # periodic y: the ghost wraps (cartesianmesh.cpp:636-658)
    assert mp.natural_index(0, 3, -1, 0) == 3 + (11 - 1) * 11
    assert mp.natural_index(0, 3, 11, 0) == 3
    assert mp.natural_index(0, -1, 3, 0) == -1


def test_stretch_grid_formula():
    dL = omesh.stretch_grid(0.1, 1.6, 4, 0.5)
    assert np.allclose(dL, [0.8, 0.4, 0.2, 0.1], rtol=0, atol=1e-15)
    assert abs(dL.sum() - 1.5) < 1e-15


def test_slab_ranges_follow_dmda_rule():
    assert omesh.slab_ranges(512, 8) == [(64 * r, 64 * (r + 1)) for r in range(8)]
    assert omesh.slab_ranges(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
