"""CPU checks of the oracle's immersed-boundary restatement (oracle/ibm.py) against the reference's own known answers
(tests/misc/delta_test.cpp) and against properties the operators must have."""
import json
import os

import numpy as np
import pytest

from oracle import ibm, mesh as omesh

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_test_vectors.json")))


def circle(n, r=0.5, c=(0.0, 0.0)):
    a = 2.0 * np.pi * np.arange(n) / n
    return np.stack([c[0] + r * np.cos(a), c[1] + r * np.sin(a)], axis=1)


def body_mesh(cells=(12, 16, 12), ratio=1.2, span=3.0, core=0.8, dim=2):
    """uniform block [-core, core] around the body, stretched towards +-span (the layout of every cylinder example)"""
    a, c, e = cells
    sub = [{"end": -core, "cells": a, "stretchRatio": 1.0 / ratio}, {"end": core, "cells": c, "stretchRatio": 1.0},
           {"end": span, "cells": e, "stretchRatio": ratio}]
    cfg = omesh.uniform_config((a + c + e,) * dim)
    cfg["mesh"] = [{"direction": "xyz"[d], "start": -span, "subDomains": sub} for d in range(dim)]
    return cfg


def test_delta_kernels_known_answers():
    g = G["delta_roma_et_al_1999"]
    for k in g["known"]:
        assert ibm.roma_et_al_1999(k["r"], g["h"]) == k["value"]
    r = np.sort(np.random.default_rng(0).uniform(*g["decreasing_on"], 10))
    v = [ibm.roma_et_al_1999(x, g["h"]) for x in r]
    assert all(a > b for a, b in zip(v[:-1], v[1:]))
    # both kernels: even, unit integral on a uniform grid for any offset, compact support
    for kern, width in ((ibm.roma_et_al_1999, 1.5), (ibm.peskin_2002, 2.0)):
        for h in (1.0, 0.37):
            for off in (0.0, 0.123, 0.5):
                s = sum(kern((i + off) * h, h) * h for i in range(-4, 5))
                assert abs(s - 1.0) < 1e-14
            assert kern(0.3 * h, h) == kern(-0.3 * h, h) and kern(width * h * 1.0001, h) == 0.0
    with pytest.raises(ValueError):
        ibm.get_kernel("COSINE")


def test_lagrangian_point_files(tmp_path):
    pts = circle(7)
    f = tmp_path / "c.body"
    f.write_text("7\n" + "\n".join(f"{x:.18e} {y:.18e}" for x, y in pts) + "\n")
    assert np.array_equal(ibm.read_lagrangian_points(str(f)), pts)
    f.write_text("7 2\n0 0\n")
    with pytest.raises(ValueError):
        ibm.read_lagrangian_points(str(f))


@pytest.mark.parametrize("kernel", ["ROMA_ET_AL_1999", "PESKIN_2002"])
def test_operators_have_the_properties_of_the_method(kernel):
    m = omesh.create_mesh(body_mesh())
    body = circle(40)
    ops = ibm.create_ib_operators(m, [body], 0.01, kernel)
    E, H, D = ops["E"], ops["H"], ops["delta"]
    h = 1.6 / 16
    # background cells: the point lies inside its cell
    idx = ibm.mesh_index(m, body)
    for d in range(2):
        v = m.coord[4][d].true
        assert np.all(v[idx[:, d]] <= body[:, d]) and np.all(body[:, d] < v[idx[:, d] + 1])
    # interpolation reproduces constants: rows of E sum to one (body inside the uniform block)
    rs = np.add.reduceat(E.val, E.rowptr[:-1])
    assert np.abs(rs - 1.0).max() < 1e-13
    # ... and linear fields, for Roma's kernel (first moment condition): E x_u = X
    xs = []
    for f in range(2):
        n0, n1, _ = (int(v) for v in m.n[f])
        jj, ii = np.meshgrid(np.arange(n1), np.arange(n0), indexing="ij")
        xs.append((m.coord[f][0][ii.ravel()], m.coord[f][1][jj.ravel()]))
    lin = np.concatenate([xs[0][0], xs[1][1]])  # u-points carry x, v-points carry y
    from oracle import clib
    got = clib.spmv(E, lin)
    assert np.abs(got[0::2] - body[:, 0]).max() < 1e-12 and np.abs(got[1::2] - body[:, 1]).max() < 1e-12
    # H is the transpose of Delta, E = Delta * h^2 in the uniform block
    assert np.allclose(H.to_dense().T, D.to_dense(), rtol=0, atol=0)
    assert np.allclose(E.val, D.val * h * h, rtol=1e-14)
    # the force system is symmetric positive definite
    A = ops["EBNH"].to_dense()
    assert np.abs(A - A.T).max() <= 1e-15 * np.abs(A).max()
    assert np.linalg.eigvalsh(0.5 * (A + A.T)).min() > 0.0


def test_two_bodies_and_points_near_a_wall():
    m = omesh.create_mesh(omesh.uniform_config((16, 12)))
    b1 = circle(9, r=0.12, c=(0.3, 0.5))
    b2 = np.array([[0.02, 0.03], [0.97, 0.95], [0.5, 0.01]])  # kernels clipped by the walls
    ops = ibm.create_ib_operators(m, [b1, b2], 0.02)
    D = ops["delta"]
    assert D.n_rows == 2 * (9 + 3) and D.n_cols == m.UN
    n_per_row = np.diff(D.rowptr)
    assert 4 <= n_per_row[:18].min() and n_per_row[:18].max() == 9 and 1 <= n_per_row[18:].min() and n_per_row[18:].max() <= 6
    # rows of the u-component only reference u points, v rows v points
    nu = int(np.prod(m.n[0]))
    for r in range(D.n_rows):
        c = D.col[D.rowptr[r]:D.rowptr[r + 1]]
        assert np.all(c < nu) if r % 2 == 0 else np.all(c >= nu)
        assert np.all(np.diff(c) > 0)
    with pytest.raises(ValueError):
        ibm.create_delta(m, [np.array([[1.2, 0.5]])])


def test_cylinder_drag_follows_koumoutsakos_leonard():
    """A coarse version of the reference's validation (examples/decoupledibpm/cylinder2dRe40_GPU): impulsively started
    cylinder at Re = 40; the drag coefficient 2 fx settles on the vortex-method curve after the start-up transient."""
    cfg = body_mesh(cells=(30, 32, 30), ratio=1.12, span=15.0, core=0.8)
    for bc in cfg["flow"]["boundaryConditions"]:
        bc["u"] = ["CONVECTIVE", 1.0] if bc["location"] == "xPlus" else ["DIRICHLET", 1.0]
        bc["v"] = ["CONVECTIVE", 1.0] if bc["location"] == "xPlus" else ["DIRICHLET", 0.0]
    m = omesh.create_mesh(cfg)
    h = 1.6 / 32
    ns = ibm.DecoupledIBPM(m, 0.02, 0.025, [circle(int(round(np.pi / h)))], pinned=True, vtol=1e-8, ptol=1e-8)
    U0 = np.zeros(m.UN)
    U0[: int(np.prod(m.n[0]))] = 1.0
    ns.set_state(U0, np.zeros(m.pN))
    kl = G["koumoutsakos_leonard_1995_cylinder_re40"]
    t_ref, cd_ref = 0.5 * np.array(kl["t_radius_units"]), np.array(kl["cd"])
    for it in range(1, 151):
        ns.advance()
        if it in (100, 125, 150):
            cd = 2.0 * ns.body_forces()[0][0]
            assert abs(cd - np.interp(it * 0.02, t_ref, cd_ref)) < 0.12 * cd
    assert abs(ns.body_forces()[0][1]) < 1e-6  # symmetric wake: no lift


def test_coupled_ibpm_enforces_continuity_and_no_slip_together():
    """oracle/ibm.py CoupledIBPM (IBPMSolver, applications/ibpm): one step of the stacked system leaves the velocity
    divergence-free AND at rest on the body; the decoupled scheme only gets the first (the projection undoes part of the
    no-slip correction)."""
    from oracle import clib, ibm, navierstokes as ons
    cfg = body_mesh(cells=(6, 12, 6), ratio=1.3, span=2.5, core=0.7)
    for bc in cfg["flow"]["boundaryConditions"]:
        bc["u"] = ["DIRICHLET", 1.0]
    m = omesh.create_mesh(cfg)
    bodies = [circle(24, r=0.4)]
    U0 = np.zeros(m.UN)
    U0[: int(np.prod(m.n[0]))] = 1.0
    slip = {}
    for name, cls in (("coupled", ibm.CoupledIBPM), ("decoupled", ibm.DecoupledIBPM)):
        s = cls(m, 0.01, 0.02, bodies, pinned=True, vtol=1e-14, ptol=1e-13)
        s.set_state(U0, np.zeros(m.pN))
        s.advance()
        div = clib.spmv(s.D, s.U) + ons.divergence_correction(m, s.ghosts)
        div[0] = 0.0
        assert np.abs(div).max() < 1e-9
        slip[name] = np.abs(clib.spmv(s.ops["E"], s.U)).max()
    assert slip["coupled"] < 1e-10 and slip["decoupled"] > 1e-3
