"""Oracle operators: the reference's createBnHead known answer, and independent cross-checks
against scipy.sparse (a second implementation; scipy is in the image but is not the reference)."""
import json
import os

import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_test_vectors.json")))
CFG2D = G["cartesianmesh2d_dirichlet"]["config"]


def test_create_bn_head_known_answer():
    """tests/operators/createbnhead_test.cpp:17-61: Op = (2/dt) I on a 10x12 grid,
    sum of all entries of BNHat for N = 1..10, |sum - ans| <= 1e-11."""
    g = G["createbnhead"]
    dt, c, nx, ny = g["dt"], g["c"], g["nx"], g["ny"]
    val = 2.0 / dt
    n = nx * ny
    op = oops.CSR(n, n, np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int64), np.full(n, val))
    ans = nx * ny * dt
    for N in g["orders"]:
        bn = oops.create_bn_head(op, dt, c, N)
        assert (bn.n_rows, bn.n_cols) == (n, n)
        if N > 1:
            ans += dt * nx * ny * (c * dt * val) ** (N - 1)
        assert abs(bn.val.sum() - ans) <= g["tol"]
    with pytest.raises(ValueError):
        oops.create_bn_head(op, dt, c, 0)


def _mesh3d():
    cfg = omesh.uniform_config((7, 6, 5))
    cfg["mesh"][0]["subDomains"] = [{"end": 0.4, "cells": 3, "stretchRatio": 0.8}, {"end": 1.0, "cells": 4, "stretchRatio": 1.25}]
    return omesh.create_mesh(cfg)


@pytest.mark.parametrize("mesh", ["2d", "3d"])
def test_operators_against_scipy_and_structure(mesh):
    sp = pytest.importorskip("scipy.sparse")
    m = omesh.create_mesh(CFG2D) if mesh == "2d" else _mesh3d()
    D, Gm, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    assert (D.n_rows, D.n_cols) == (m.pN, m.UN) and (Gm.n_rows, Gm.n_cols) == (m.UN, m.pN)
    assert np.diff(Gm.rowptr).max() == 2 and np.diff(D.rowptr).max() == 2 * m.dim
    assert np.diff(L.rowptr).max() == 1 + 2 * m.dim
    dt, cnu = 0.01, 0.5 * 0.01
    BNG, DBNG = oops.create_poisson_operator(D, Gm, L, dt, cnu)
    S = (D.to_scipy() @ (dt * Gm.to_scipy())).toarray()
    A = DBNG.to_dense()
    assert np.abs(S - A).max() <= 1e-17 + 1e-15 * np.abs(A).max()
    # symmetric, negative semi-definite, constants in the null space, 2*dim+1 per row
    assert np.abs(A - A.T).max() == 0.0
    assert np.abs(A.sum(1)).max() <= 1e-15 * np.abs(A).max() * 8
    w = np.linalg.eigvalsh(A)
    assert w.max() <= 1e-14 * abs(w.min()) and (np.abs(w) < 1e-12 * abs(w.min())).sum() == 1
    assert np.diff(DBNG.rowptr).max() == 1 + 2 * m.dim
    # velocity operator A = I/dt - c nu L
    Av = oops.create_velocity_operator(L, dt, cnu)
    ref = sp.identity(m.UN).toarray() / dt - cnu * L.to_dense()
    assert np.abs(Av.to_dense() - ref).max() <= 1e-12 * np.abs(ref).max()
    # off-diagonal formula of the Poisson operator: dt*dy*dz / (0.5 (dx_i + dx_i+1))
    dx, dy = m.dL[3][0].true, m.dL[3][1].true
    dz = m.dL[3][2].true if m.dim == 3 else np.ones(1)
    i, j, k = 1, 2, (1 if m.dim == 3 else 0)
    row = i + m.n[3][0] * (j + m.n[3][1] * k)
    expect = dt * dy[j] * dz[k] / (0.5 * (dx[i] + dx[i + 1]))
    assert np.isclose(A[row, row + 1], expect, rtol=1e-14)


def test_laplacian_entries_and_bc_fold():
    """createlaplacian.cpp:134-151 + :232-243 on the 12x11 stretched mesh: a tangential Dirichlet wall
    (a0 = -1) subtracts the ghost coefficient from the diagonal, a normal one (a0 = 0) just drops it."""
    m = omesh.create_mesh(CFG2D)
    L = oops.create_laplacian(m).to_dense()
    f, i, j = 0, 4, 5  # interior u point
    r = int(m.packed_index(f, i, j, 0))
    c = m.coord[f]
    dls = m.dL[f]
    vals = [1.0 / ((c[0][i] - c[0][i - 1]) * dls[0][i]), 1.0 / ((c[0][i + 1] - c[0][i]) * dls[0][i]),
            1.0 / ((c[1][j] - c[1][j - 1]) * dls[1][j]), 1.0 / ((c[1][j + 1] - c[1][j]) * dls[1][j])]
    assert L[r, r - 1] == vals[0] and L[r, r + 1] == vals[1]
    assert L[r, r - 11] == vals[2] and L[r, r + 11] == vals[3]
    assert L[r, r] == -(((0.0 + vals[0]) + vals[1]) + vals[2] + vals[3])
    # bottom row of u (j = 0): ghost below, tangential Dirichlet -> diag -= coeff
    j = 0
    r = int(m.packed_index(f, i, j, 0))
    vneg = 1.0 / ((c[1][0] - c[1][-1]) * dls[1][0])
    vx = [1.0 / ((c[0][i] - c[0][i - 1]) * dls[0][i]), 1.0 / ((c[0][i + 1] - c[0][i]) * dls[0][i])]
    vpos = 1.0 / ((c[1][1] - c[1][0]) * dls[1][0])
    diag0 = -((((0.0 + vx[0]) + vx[1]) + vneg) + vpos)
    assert L[r, r] == diag0 + vneg * -1.0
    # left column of u (i = 0): ghost is ON the wall (normal Dirichlet, a0 = 0): nothing added
    r = int(m.packed_index(0, 0, 5, 0))
    assert np.isclose(L[r].sum(), -1.0 / ((c[0][0] - c[0][-1]) * dls[0][0]), rtol=1e-13)


def test_neumann_and_convective_a0_table():
    assert oops.bc_a0("DIRICHLET", 0, 0) == 0.0 and oops.bc_a0("DIRICHLET", 0, 2) == -1.0
    assert oops.bc_a0("NEUMANN", 1, 3) == 1.0 and oops.bc_a0("NEUMANN", 1, 0) == 1.0
    assert oops.bc_a0("CONVECTIVE", 0, 1) == 0.0 and oops.bc_a0("CONVECTIVE", 1, 1) == -1.0
    # a Neumann outlet modifies D (createdivergence.cpp:231-242): the ghost face folds onto the last face
    cfg = json.loads(json.dumps(CFG2D))
    cfg["flow"]["boundaryConditions"][1]["u"] = ["NEUMANN", 0.0]
    m = omesh.create_mesh(cfg)
    D = oops.create_divergence(m).to_dense()
    D0 = oops.create_divergence(omesh.create_mesh(CFG2D)).to_dense()
    changed = np.argwhere(D != D0)
    assert len(changed) == 11 and set(changed[:, 0]) == {11 + 12 * j for j in range(11)}
    for row, col in changed:
        # +area (ghost face) folded onto the -area entry of the last real face: the row's u part cancels
        assert D[row, col] == D0[row, col] + (-D0[row, col])


def test_pin_row0_matches_matzerorowscolumns():
    m = omesh.create_mesh(CFG2D)
    D, Gm, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, Gm, L, 0.01, 0.005)
    P = oops.pin_row0(A).to_dense()
    Ad = A.to_dense()
    assert P[0, 0] == 1.0 and np.all(P[0, 1:] == 0) and np.all(P[1:, 0] == 0)
    assert np.array_equal(P[1:, 1:], Ad[1:, 1:])
    assert np.linalg.matrix_rank(P) == m.pN
