"""Periodic boundaries in the oracle: mesh counts (tests/mesh/cartesianmesh2d_yperiodic.cpp), the wrapped operators,
and the multigrid restatement's level operator against the assembled D*BN*G."""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops


@pytest.mark.parametrize("n,per", [((12, 10), (True, True)), ((12, 10), (False, True)), ((8, 6, 10), (True, False, True)),
                                   ((8, 8, 8), (True, True, True))])
def test_gmg_level_operator_is_dbng(n, per):
    dim = len(n)
    cfg = omesh.periodic_config(n, per, ratios=[1.0] * dim)
    m = omesh.create_mesh(cfg)
    for d in range(dim):
        assert m.n[d][d] == (n[d] if per[d] else n[d] - 1)
    dt = 0.01
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, DBNG = oops.create_poisson_operator(D, G, L, dt, 0.5 * 0.01, 1)
    w = [np.array([m.dL[3][d][i] for i in range(n[d])]) for d in range(dim)]
    g = clib.GMG(n, w, dt, nullspace=1, pre=1, post=1, omega=0.8, periodic=per)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(m.pN)
    y_ref = DBNG.to_scipy() @ x
    y = g.apply_operator(x)
    assert np.abs(y - y_ref).max() <= 1e-12 * np.abs(y_ref).max()
    # symmetric, constants in the null space
    A = DBNG.to_dense()
    assert np.abs(A - A.T).max() <= 1e-12 * np.abs(A).max()
    assert np.abs(A.sum(axis=1)).max() <= 1e-10 * np.abs(A).max()
    # PCG with the periodic V-cycle converges like the wall-bounded case
    b = y_ref
    res = g.pcg(DBNG, b, rtol=1e-10, maxit=60)
    assert res["reason"] > 0 and res["iters"] <= 25, res
    r = b - DBNG.to_scipy() @ res["x"]
    assert np.linalg.norm(r) <= 1.01e-10 * np.linalg.norm(b)


@pytest.mark.parametrize("n,per,ratios", [((14, 10), (False, False), (1.15, 0.9)), ((12, 10), (True, False), (1.1, 1.2)),
                                          ((9, 8, 10), (False, True, False), (1.2, 1.05, 0.85)), ((8, 7, 9), (True, True, True), (1.1, 0.9, 1.3)),
                                          ((10, 9, 8), (False, False, False), (0.8, 1.25, 1.1))])
def test_gmg_scaled_rows_are_the_assembled_operator_on_stretched_meshes(n, per, ratios):
    """oracle/csrc/gmg.c works, like gmg.hip, on the level operator's rows divided by the cell volume (1-D tables cm / cp / 1/w);
    that formulation is checked HERE against an independent one -- the reference's assembled D (dt I) G
    (createdivergence.cpp:135-223, creategradient.cpp:64-128) from oracle/operators.py -- on stretched, periodic and mixed
    meshes: the level-0 operator the V-cycle smooths with is that matrix to rounding, so an error in the scaled tables (wall
    or wrap handling) cannot hide on both sides of the GPU-vs-oracle multigrid parity tests."""
    dim = len(n)
    m = omesh.create_mesh(omesh.periodic_config(n, per, ratios=list(ratios)))
    dt = 0.02
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, DBNG = oops.create_poisson_operator(D, G, L, dt, 0.5 * 0.01, 1)
    w = [np.array([m.dL[3][d][i] for i in range(n[d])]) for d in range(dim)]
    assert max(wd.max() / wd.min() for wd in w) > 1.5  # really stretched
    g = clib.GMG(n, w, dt, nullspace=1, pre=2, post=2, omega=0.9, periodic=per)
    A = DBNG.to_dense()
    for seed in range(3):
        x = np.random.default_rng(seed).standard_normal(m.pN)
        y_ref = A @ x
        assert np.abs(g.apply_operator(x) - y_ref).max() <= 2e-13 * np.abs(A).max() * np.abs(x).max() * (2 * dim + 1)
    # unit vectors: entry by entry (column by column) for the first, a middle and the last cell
    for c in (0, m.pN // 2, m.pN - 1):
        e = np.zeros(m.pN)
        e[c] = 1.0
        assert np.abs(g.apply_operator(e) - A[:, c]).max() <= 1e-13 * np.abs(A).max()


def tgv2d_fields(m, t, nu):
    """Taylor-Green vortex of examples/navierstokes/taylorgreenvortex2dRe100/config.yaml at the velocity points"""
    out = []
    for f in range(2):
        x = np.array([m.coord[f][0][i] for i in range(int(m.n[f][0]))])[None, :]
        y = np.array([m.coord[f][1][j] for j in range(int(m.n[f][1]))])[:, None]
        v = np.cos(x) * np.sin(y) if f == 0 else -np.sin(x) * np.cos(y)
        out.append((v * np.exp(-2.0 * nu * t)).ravel())
    return np.concatenate(out)


def test_taylor_green_vortex_2d_decays_like_the_analytical_solution():
    from oracle import navierstokes as ons
    errs = []
    for n in (16, 32):
        cfg = omesh.periodic_config((n, n), (True, True), lo=-np.pi, hi=np.pi)
        m = omesh.create_mesh(cfg)
        nu, dt, nt = 0.1, 0.02 * 16 / n, 5 * n // 16
        ns = ons.NavierStokes(m, dt, nu)
        ns.set_state(tgv2d_fields(m, 0.0, nu))
        for _ in range(nt):
            ns.advance()
        assert np.abs(ns.last_rhs2).max() < 1.0  # sanity
        div = clib.spmv(ns.D, ns.U)
        assert np.abs(div).max() < 1e-10
        e = ns.U - tgv2d_fields(m, nt * dt, nu)
        errs.append(np.abs(e).max())
    assert errs[0] < 5e-3 and errs[1] < errs[0] / 2.5, errs  # between first and second order at these sizes
