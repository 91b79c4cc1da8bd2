"""parameters.convection / parameters.diffusion (createTimeIntegration, src/timeintegration/timeintegration.cpp:41-80;
coefficients include/petibm/timeintegration.h:107-166) in the device time step, against the oracle, and the reference's
convergence study examples/navierstokes/convergence/liddrivencavity2dRe100_20 (EULER_EXPLICIT + EULER_IMPLICIT)."""
import numpy as np
import pytest

from oracle import mesh as omesh, navierstokes as ons
from test_gpu_periodic import KSP_P, VEL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("conv,diff", [("EULER_EXPLICIT", "EULER_IMPLICIT"), ("EULER_EXPLICIT", "CRANK_NICOLSON"),
                                       ("ADAMS_BASHFORTH_2", "EULER_IMPLICIT"), ("ADAMS_BASHFORTH_2", "EULER_EXPLICIT"),
                                       ("EULER_EXPLICIT", "ADAMS_BASHFORTH_2"), ("CRANK_NICOLSON", "CRANK_NICOLSON"),
                                       ("EULER_IMPLICIT", "CRANK_NICOLSON")])
def test_time_integration_schemes_match_oracle(conv, diff):
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = omesh.uniform_config((14, 12), lid=1.0)
    cfg["mesh"][0]["subDomains"] = [{"end": 0.5, "cells": 7, "stretchRatio": 0.9}, {"end": 1.0, "cells": 7, "stretchRatio": 1.1}]
    cfg["flow"]["nu"] = 0.02
    dt = 0.0005 if diff in ("EULER_EXPLICIT", "ADAMS_BASHFORTH_2") else 0.005  # explicit diffusion: dt < h^2 / (4 nu)
    cfg["parameters"] = {"dt": dt, "convection": conv, "diffusion": diff}
    m = omesh.create_mesh(cfg)
    ref = ons.NavierStokes(m, dt, 0.02, vtol=1e-14, ptol=1e-13, convection=conv, diffusion=diff)
    rng = np.random.default_rng(5)
    U0, p0 = 0.1 * rng.uniform(-1, 1, m.UN), 0.1 * rng.uniform(-1, 1, m.pN)
    ref.set_state(U0, p0)
    s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=KSP_P)
    s.setState(U0, p0)
    for step in range(4):
        ref.advance()
        s.advance()
        U, p, r1, r2 = s.getState(rhs=True)
        if step == 0:
            assert np.array_equal(r1, ref.last_rhs1)
        assert np.abs(r1 - ref.last_rhs1).max() <= 1e-9 * np.abs(ref.last_rhs1).max()
        assert np.abs(U - ref.U).max() <= 1e-9 * np.abs(ref.U).max()
        dp = (p - p.mean()) - (ref.p - ref.p.mean())
        assert np.abs(dp).max() <= 1e-8 * np.abs(ref.p - ref.p.mean()).max()
    s.destroy()


def test_unknown_scheme_is_the_reference_error():
    from petibm_amd import capi
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = omesh.uniform_config((8, 8))
    cfg["flow"]["nu"] = 0.01
    cfg["parameters"] = {"dt": 0.01, "convection": "RUNGE_KUTTA_3", "diffusion": "CRANK_NICOLSON"}
    with pytest.raises(capi.PibError) as ei:
        NavierStokesSolver(cfg)
    assert ei.value.code == capi.ERR_ARG_OUTOFRANGE and "does not exist" in ei.value.message


def test_restart_with_a_one_term_scheme(tmp_path):
    from petibm_amd.navierstokes import NavierStokesSolver
    pytest.importorskip("ctypes")
    cfg = omesh.uniform_config((16, 16), lid=1.0)
    cfg["flow"]["nu"] = 0.01
    cfg["parameters"] = {"dt": 0.002, "convection": "EULER_EXPLICIT", "diffusion": "CRANK_NICOLSON"}
    a = NavierStokesSolver(cfg)
    a.advance(5)
    f = str(tmp_path / "0000005.h5")
    a.writeRestartData(f)
    a.advance(5)
    Ua, pa = a.getState()
    b = NavierStokesSolver(cfg)
    b.readRestartData(f)
    b.advance(5)
    Ub, pb = b.getState()
    assert np.array_equal(Ua, Ub) and np.array_equal(pa, pb)
    a.destroy()
    b.destroy()


def test_convergence_study_of_the_reference():
    """examples/navierstokes/convergence/liddrivencavity2dRe100_20: lid-driven cavity, nu = 0.01, dt = 5e-4, 500 steps of
    EULER_EXPLICIT convection + EULER_IMPLICIT diffusion on 20^2, 60^2, 180^2, 540^2 cells; observed orders from the
    solutions restricted to the coarsest grid's points (scripts/getOrderConvergence.py)."""
    from petibm_amd.navierstokes import NavierStokesSolver
    vel = "-velocity_ksp_type bcgs\n-velocity_ksp_rtol 1.0E-08\n-velocity_ksp_atol 0.0\n-velocity_ksp_max_it 10000\n-velocity_pc_type jacobi\n"
    poi = "-poisson_ksp_type cg\n-poisson_ksp_rtol 1.0E-08\n-poisson_ksp_atol 0.0\n-poisson_ksp_max_it 20000\n-poisson_pc_type gamg\n"
    sol = {}
    for n in (20, 60, 180, 540):
        cfg = omesh.uniform_config((n, n), lid=1.0)
        cfg["flow"]["nu"] = 0.01
        cfg["parameters"] = {"dt": 5.0e-4, "convection": "EULER_EXPLICIT", "diffusion": "EULER_IMPLICIT"}
        s = NavierStokesSolver(cfg, velocity_cfg=vel, poisson_cfg=poi)
        s.advance(500)
        U, p = s.getState()
        r = n // 20
        u = U[: (n - 1) * n].reshape(n, n - 1)
        v = U[(n - 1) * n:].reshape(n - 1, n)
        pp = p.reshape(n, n)
        # points of the 20^2 grid inside the finer ones: u at x = (i+1)/20, y = (j+1/2)/20, etc.
        ux = (np.arange(19) + 1) * r - 1
        cc = (np.arange(20) * r + (r - 1) // 2)  # cell centres coincide for odd ratios
        sol[n] = (u[np.ix_(cc, ux)], v[np.ix_(ux, cc)], (pp - pp.mean())[np.ix_(cc, cc)])
        s.destroy()
    for k, name in enumerate("uvp"):
        e1 = np.linalg.norm(sol[60][k] - sol[20][k])
        e2 = np.linalg.norm(sol[180][k] - sol[60][k])
        e3 = np.linalg.norm(sol[540][k] - sol[180][k])
        first, last = np.log(e1 / e2) / np.log(3.0), np.log(e2 / e3) / np.log(3.0)
        assert 1.0 < last < 2.6, (name, first, last)


def test_decoupled_ibpm_with_euler_schemes_matches_oracle():
    from oracle import ibm
    from petibm_amd.navierstokes import DecoupledIBPMSolver
    from test_gpu_ibm import AMGX_P, FORCES, flow_config
    from test_oracle_ibm import body_mesh, circle
    cfg = flow_config(body_mesh(cells=(8, 16, 8), ratio=1.25, span=3.0, core=0.8), dt=0.01)
    cfg["parameters"].update(convection="EULER_EXPLICIT", diffusion="EULER_IMPLICIT")
    bodies = [circle(32)]
    m = omesh.create_mesh(cfg)
    ref = ibm.DecoupledIBPM(m, 0.01, cfg["flow"]["nu"], bodies, pinned=True, vtol=1e-14, ptol=1e-13,
                            convection="EULER_EXPLICIT", diffusion="EULER_IMPLICIT")
    U0 = np.zeros(m.UN)
    U0[: int(np.prod(m.n[0]))] = 1.0
    U0 += 0.02 * np.random.default_rng(3).uniform(-1, 1, m.UN)
    ref.set_state(U0, np.zeros(m.pN))
    s = DecoupledIBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=AMGX_P.format(tol=1e-13), forces_cfg=FORCES)
    s.setState(U0, np.zeros(m.pN))
    for step in range(3):
        ref.advance()
        s.advance()
        U, p, r1, r2 = s.getState(rhs=True)
        f, avg = s.getForces()
        assert np.abs(r1 - ref.last_rhs1).max() <= 1e-9 * np.abs(ref.last_rhs1).max()
        assert np.abs(U - ref.U).max() <= 1e-9 * np.abs(ref.U).max()
        assert np.abs(f - ref.f).max() <= 1e-8 * np.abs(ref.f).max()
    s.destroy()
