"""The reference's examples/ibpm cylinder cases (coupled IBPM) on one MI355X: drag history next to Koumoutsakos & Leonard.
    python tools/ibpm_coupled_cylinder.py 40 | 550 | 3000"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from petibm_amd import cases
from petibm_amd.navierstokes import IBPMSolver
from petibm_amd.cases import uniform_stream as flow_config, AMGX_POISSON as AMGX_P, DIRECT_FORCES as FORCES
from petibm_amd.cases import circle

re = int(sys.argv[1]) if len(sys.argv) > 1 else 550
case = {40: dict(sub=[(-0.6, 69, 0.952380952), (0.6, 48, 1.0), (15.0, 69, 1.05)], nu=0.025, dt=0.01, nt=2000, npts=158, every=200),
        550: dict(sub=[(-0.54, 171, 0.980392156), (0.54, 108, 1.0), (15.0, 171, 1.02)], nu=0.00181818181818, dt=0.0025, nt=1200, npts=315, every=100),
        3000: dict(sub=[(-0.52, 363, 0.9900990099), (0.52, 260, 1.0), (15.0, 363, 1.01)], nu=0.00033333333333, dt=0.001, nt=3000, npts=786, every=250)}[re]
sub = [{"end": e, "cells": c, "stretchRatio": r} for e, c, r in case["sub"]]
n = sum(c for _, c, _ in case["sub"])
base = cases.cavity((n, n))
base["mesh"] = [{"direction": d, "start": -15.0, "subDomains": sub} for d in "xy"]
cfg = flow_config(base, nu=case["nu"], dt=case["dt"])
vel = "-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n-velocity_pc_type jacobi\n"
G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_test_vectors.json")))[f"koumoutsakos_leonard_1995_cylinder_re{re}"]
t_ref, cd_ref = 0.5 * np.array(G["t_radius_units"]), np.array(G["cd"])
t0 = time.perf_counter()
s = IBPMSolver(cfg, bodies=[circle(case["npts"])], velocity_cfg=vel, poisson_cfg=AMGX_P.format(tol="1.0E-06"), forces_cfg=FORCES)
print(f"Re {re}: {n}^2 cells, {s.nf} force unknowns; set-up {time.perf_counter() - t0:.1f} s")
print("  step      t      cd     K&L    v_its p_its")
t0 = time.perf_counter()
for it in range(case["every"], case["nt"] + 1, case["every"]):
    s.advance(it - s.ite)
    cd = 2.0 * s.getForces()[1][0][0]
    info = s.linSolversInfo()
    print(f"{it:6d} {it * case['dt']:6.3f} {cd:7.4f} {np.interp(it * case['dt'], t_ref, cd_ref):7.4f}  {info[1]:4d} {info[3]:4d}")
w = time.perf_counter() - t0
print(f"{case['nt']} steps in {w:.1f} s = {1e3 * w / case['nt']:.2f} ms/step")
