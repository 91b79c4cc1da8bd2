"""IBPMSolver (applications/ibpm): the coupled immersed-boundary projection method on the GPU -- pressure and Lagrangian
forces as one unknown of D_c BN G_c, solved through the Schur complement on the pressure (csrc/ibm.hip) -- against the
oracle's dense solve of the stacked system and against Koumoutsakos & Leonard (1995)."""
import json
import os

import numpy as np
import pytest

from oracle import ibm, mesh as omesh
from test_gpu_ibm import AMGX_P, FORCES, VEL, flow_config, sphere_points
from test_oracle_ibm import body_mesh, circle

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_test_vectors.json")))


@pytest.mark.parametrize("case", ["2d", "3d"])
def test_coupled_ibpm_step_matches_the_stacked_system(case):
    from petibm_amd.navierstokes import IBPMSolver
    if case == "2d":
        cfg = flow_config(body_mesh(cells=(8, 16, 8), ratio=1.25, span=3.0, core=0.8), dt=0.01)
        bodies = [circle(32)]
    else:
        cfg = flow_config(body_mesh(cells=(3, 8, 3), ratio=1.4, span=2.0, core=0.6, dim=3), nu=0.05, dt=0.02)
        bodies = [sphere_points(30, r=0.35)]
    m = omesh.create_mesh(cfg)
    dt, nu = cfg["parameters"]["dt"], cfg["flow"]["nu"]
    ref = ibm.CoupledIBPM(m, dt, nu, bodies, vtol=1e-14, ptol=1e-13)
    U0 = np.zeros(m.UN)
    U0[: int(np.prod(m.n[0]))] = 1.0
    U0 += 0.02 * np.random.default_rng(3).uniform(-1, 1, m.UN)
    ref.set_state(U0, np.zeros(m.pN))
    s = IBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=AMGX_P.format(tol=1e-13), forces_cfg=FORCES)
    s.setState(U0, np.zeros(m.pN))
    for step in range(3):
        ref.advance()
        s.advance()
        U, p = s.getState()
        f, avg = s.getForces()
        assert np.abs(U - ref.U).max() <= 1e-8 * np.abs(ref.U).max()
        assert np.abs(p - ref.p).max() <= 1e-7 * np.abs(ref.p).max()
        assert np.abs(f - ref.f).max() <= 1e-7 * np.abs(ref.f).max()
        # both constraints at once: discrete continuity and no slip at the Lagrangian points
        from oracle import clib, navierstokes as ons
        div = clib.spmv(ref.D, U) + ons.divergence_correction(m, ref.ghosts)
        div[0] = 0.0
        assert np.abs(div).max() <= 1e-9 * np.abs(ref.D.val).max()
        slip = clib.spmv(ref.ops["E"], U)
        assert np.abs(slip).max() <= 1e-9
    assert 0 < s.linSolversInfo()[3] < 80
    s.destroy()


def test_coupled_cylinder_re550_follows_koumoutsakos_leonard_from_the_start():
    """examples/ibpm/cylinder2dRe550[_GPU] verbatim (450^2 stretched mesh, 315 points, dt = 0.0025, 1200 steps).  With
    both constraints enforced together there is no start-up ringing: the drag follows the vortex-method curve over
    0.25 <= t <= 3 (the decoupled scheme needs until t = 1.75, tests/test_gpu_ibm.py)."""
    from petibm_amd.navierstokes import IBPMSolver
    sub = [{"end": -0.54, "cells": 171, "stretchRatio": 0.980392156}, {"end": 0.54, "cells": 108, "stretchRatio": 1.0},
           {"end": 15.0, "cells": 171, "stretchRatio": 1.02}]
    base = omesh.uniform_config((450, 450))
    base["mesh"] = [{"direction": d, "start": -15.0, "subDomains": sub} for d in "xy"]
    cfg = flow_config(base, nu=0.00181818181818, dt=0.0025)
    vel = ("-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n"
           "-velocity_pc_type jacobi\n")
    s = IBPMSolver(cfg, bodies=[circle(315)], velocity_cfg=vel, poisson_cfg=AMGX_P.format(tol="1.0E-06"), forces_cfg=FORCES)
    kl = G["koumoutsakos_leonard_1995_cylinder_re550"]
    t_ref, cd_ref = 0.5 * np.array(kl["t_radius_units"]), np.array(kl["cd"])
    worst = 0.0
    for it in range(100, 1201, 100):
        s.advance(it - s.ite)
        cd = 2.0 * s.getForces()[1][0][0]
        ref = np.interp(it * 0.0025, t_ref, cd_ref)
        worst = max(worst, abs(cd - ref) / ref)
        assert abs(cd - ref) < 0.08 * ref, (it, cd, ref)
    s.destroy()


def test_errors_of_the_newer_engine_entries():
    """order / support errors of pib_ns_set_coupled, pib_ns_set_bn_order, pib_ns_set_time_integration, pib_ns_get_vorticity"""
    import ctypes as C
    from petibm_amd import capi
    from petibm_amd.navierstokes import DecoupledIBPMSolver, NavierStokesSolver
    lib = capi.load()
    cfg = flow_config(body_mesh(cells=(6, 12, 6), ratio=1.3, span=2.5, core=0.7), dt=0.01)
    s = NavierStokesSolver(cfg)
    assert lib.pib_ns_set_coupled(s._h, 1) == capi.ERR_ORDER                      # no bodies
    n3 = (C.c_int64 * 3)()
    assert lib.pib_ns_get_vorticity(s._h, 0, n3, None) == capi.ERR_ARG_OUTOFRANGE  # wx in a 2-D run
    assert lib.pib_ns_get_vorticity(s._h, 2, n3, None) == 0 and list(n3) == [25, 25, 1]
    assert lib.pib_ns_set_bn_order(s._h, 0) == capi.ERR_SUP                        # createbn.cpp:27-29
    assert lib.pib_ns_set_time_integration(s._h, b"ADAMS_BASHFORTH_3", b"CRANK_NICOLSON") == capi.ERR_ARG_OUTOFRANGE
    s.destroy()
    it = ("config_version=2\nsolver(solv)=PCG\nsolv:max_iters=100\nsolv:convergence=ABSOLUTE\nsolv:tolerance=1e-10\n"
          "solv:norm=L2\nsolv:preconditioner(prec)=BLOCK_JACOBI\n")
    d = DecoupledIBPMSolver(cfg, bodies=[circle(24, r=0.4)], forces_cfg=it)       # an iterative forces solver ...
    assert lib.pib_ns_set_coupled(d._h, 1) == capi.ERR_SUP                         # ... has no explicit inverse to eliminate with
    assert lib.pib_ns_set_bn_order(d._h, 2) == capi.ERR_ORDER                      # BN > 1 goes BEFORE the bodies (BNH = BN H is built from it)
    assert lib.pib_ns_set_time_integration(d._h, b"EULER_EXPLICIT", b"EULER_IMPLICIT") == capi.ERR_ORDER
    d.advance(2)                                                                   # the decoupled scheme still runs
    d.destroy()
