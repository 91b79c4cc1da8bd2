"""CPU checks of the oracle's time step (oracle/navierstokes.py): the restated ghost-point equations against
hand-evaluated values of the reference's kernels, and invariants of the projection step."""
import numpy as np

from oracle import clib, mesh as omesh, navierstokes as ons


def _stream_config(n=(12, 8)):
    cfg = omesh.uniform_config(n)
    for bc in cfg["flow"]["boundaryConditions"]:
        bc["u"] = ["CONVECTIVE", 1.5] if bc["location"] == "xPlus" else ["DIRICHLET", 1.0]
        bc["v"] = ["CONVECTIVE", 1.5] if bc["location"] == "xPlus" else ["DIRICHLET", 0.0]
    return cfg


def test_ghost_point_kernels_match_the_reference_formulas():
    """singleboundarydirichlet.cpp:35-44, singleboundaryneumann.cpp:27-28, singleboundaryconvective.cpp:12-38."""
    cfg = _stream_config()
    cfg["flow"]["boundaryConditions"][2]["u"] = ["NEUMANN", 0.25]  # yMinus, tangential component
    m = omesh.create_mesh(cfg)
    g = ons.make_ghosts(m)
    rng = np.random.default_rng(1)
    U = rng.uniform(-1, 1, m.UN)
    ons.set_ghost_ics(m, g, U)
    fa = ons._field_arrays(m, U)
    # Dirichlet, normal component (u on xMinus): ghost IS the boundary value
    d = g[(0, 0)]
    assert d.a0 == 0.0 and np.all(d.a1 == 1.0) and np.all(d.value == 1.0)
    # Dirichlet, tangential component (v on xMinus): linear extrapolation through the wall value
    d = g[(1, 0)]
    assert d.a0 == -1.0 and np.all(d.a1 == 0.0) and np.array_equal(d.value, -fa[1][:, :, 0])
    # Neumann: ghost = target + normal*dL*value, dL = ghost-target distance
    d = g[(0, 2)]
    dy = 1.0 / 8
    assert d.a0 == 1.0 and np.allclose(d.a1, -1.0 * dy * 0.25, rtol=1e-14)
    assert np.allclose(d.value, fa[0][:, 0, :] - dy * 0.25, rtol=1e-14)
    # convective, t = 0: ghost := target; same direction a0 = 0, a1 = value; other direction a0 = -1, a1 = 2 target
    c = g[(0, 1)]
    assert c.a0 == 0.0 and np.array_equal(c.value, fa[0][:, :, -1]) and np.array_equal(c.a1, c.value)
    c = g[(1, 1)]
    assert c.a0 == -1.0 and np.array_equal(c.value, fa[1][:, :, -1]) and np.array_equal(c.a1, 2.0 * c.value)
    # one update with a moved target: du/dt + c du/dn = 0 in the reference's discrete form
    U2 = U + 0.1
    fa2 = ons._field_arrays(m, U2)
    dt, speed = 0.01, 1.5
    old_u, old_v = g[(0, 1)].value.copy(), g[(1, 1)].value.copy()
    ons.update_eqs(m, g, U2, dt)
    dLu = m.coord[0][0][int(m.n[0][0])] - m.coord[0][0][int(m.n[0][0]) - 1]
    dLv = m.coord[1][0][int(m.n[1][0])] - m.coord[1][0][int(m.n[1][0]) - 1]
    t_u, t_v = fa2[0][:, :, -1], fa2[1][:, :, -1]
    assert np.allclose(g[(0, 1)].a1, old_u - dt * speed * (old_u - t_u) / dLu, rtol=1e-14)
    assert np.allclose(g[(1, 1)].a1, old_v + t_v - 2.0 * dt * speed * (old_v - t_v) / dLv, rtol=1e-14)
    ons.update_ghost_values(m, g, U2)
    assert np.allclose(g[(1, 1)].value, -t_v + g[(1, 1)].a1, rtol=1e-14)


def test_uniform_stream_is_a_fixed_point_with_a_convective_outlet():
    m = omesh.create_mesh(_stream_config())
    ns = ons.NavierStokes(m, 0.01, 0.01, pinned=True)
    U0 = np.zeros(m.UN)
    U0[: int(np.prod(m.n[0]))] = 1.0
    ns.set_state(U0, np.zeros(m.pN))
    for _ in range(3):
        ns.advance()
    assert np.abs(ns.U - U0).max() <= 1e-12 and np.abs(ns.p).max() <= 1e-10


def test_projection_makes_the_velocity_discretely_divergence_free():
    for cfg, pinned in ((omesh.uniform_config((10, 9), lid=1.0), False), (_stream_config(), True)):
        m = omesh.create_mesh(cfg)
        ns = ons.NavierStokes(m, 0.01, 0.02, pinned=pinned)
        rng = np.random.default_rng(4)
        U0 = 0.05 * rng.uniform(-1, 1, m.UN)
        if pinned:
            U0[: int(np.prod(m.n[0]))] += 1.0
        ns.set_state(U0, np.zeros(m.pN))
        for _ in range(4):
            ns.advance()
            div = clib.spmv(ns.D, ns.U) + ons.divergence_correction(m, ns.ghosts)
            if pinned:
                div[0] = 0.0
            assert np.abs(div).max() <= 1e-11
        assert np.isfinite(ns.U).all() and np.abs(ns.U).max() < 2.0
