"""The reference's examples/decoupledibpm/flatplate3dRe100_GPU sweep: the inclined flat plate (Re = 100, aspect ratio 2)
at 0, 10, ..., 90 degrees, 2000 steps each, force coefficients averaged over 15 <= t <= 20 next to the measurements
reported by Taira et al. (2007) (tests/golden/reference_test_vectors.json).    python tools/flatplate_sweep.py"""
import json, math, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from petibm_amd import cases
from petibm_amd.navierstokes import DecoupledIBPMSolver
from petibm_amd.cases import uniform_stream as flow_config, AMGX_POISSON as AMGX_P, DIRECT_FORCES as FORCES

G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_test_vectors.json")))["taira_et_al_2007_flatplate_re100_ar2"]


def sub(a, b, c, e1, e2, e3, r1, r3):
    return [{"end": e1, "cells": a, "stretchRatio": r1}, {"end": e2, "cells": b, "stretchRatio": 1.0},
            {"end": e3, "cells": c, "stretchRatio": r3}]


vel = ("config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=1000\nsolv:monitor_residual=1\n"
       "solv:convergence=ABSOLUTE\nsolv:tolerance=1.0E-06\nsolv:norm=L2\nsolv:preconditioner(prec)=NOSOLVER\n")
print(" AoA    C_D     C_L   | Taira et al.: C_D     C_L   | wall s")
total = 0.0
for aoa in range(0, 91, 10):
    base = cases.cavity((127, 56, 84))
    base["mesh"] = [{"direction": "x", "start": -4.0, "subDomains": sub(43, 30, 54, -0.5, 0.7, 6.1, 0.970873786407767, 1.03)},
                    {"direction": "y", "start": -5.0, "subDomains": sub(13, 30, 13, -0.6, 0.6, 5.0, 0.7692307692307692, 1.3)},
                    {"direction": "z", "start": -5.0, "subDomains": sub(12, 60, 12, -1.2, 1.2, 5.0, 0.7692307692307692, 1.3)}]
    cfg = flow_config(base, nu=0.01, dt=0.01)
    for bc in cfg["flow"]["boundaryConditions"]:
        if bc["location"] == "xPlus":
            bc["v"], bc["w"] = ["CONVECTIVE", 0.0], ["CONVECTIVE", 0.0]
    sx = np.linspace(-0.5, 0.5, math.ceil(1.0 / 0.04) + 1)
    x, y = np.cos(np.radians(-aoa)) * sx, np.sin(np.radians(-aoa)) * sx
    z = np.linspace(-1.0, 1.0, math.ceil(2.0 / 0.04) + 1)
    body = np.concatenate([np.stack([x, y, np.full_like(x, zi)], axis=1) for zi in z])
    t0 = time.perf_counter()
    s = DecoupledIBPMSolver(cfg, bodies=[body], velocity_cfg=vel, poisson_cfg=AMGX_P.format(tol="1.0E-06"), forces_cfg=FORCES)
    s.advance(1500)
    acc = np.zeros(3)
    for _ in range(500):
        s.advance()
        acc += s.getForces()[1][0]
    w = time.perf_counter() - t0
    total += w
    cd, cl, _ = 2.0 * acc / 500 / 2.0  # force coefficient = F / (0.5 rho U^2 * chord * span), span = 2
    print(f"{aoa:4d}  {cd:6.3f}  {cl:6.3f}  |             {np.interp(aoa, G['cd_aoa'], G['cd']):6.3f}  {np.interp(aoa, G['cl_aoa'], G['cl']):6.3f}  | {w:5.1f}")
    s.destroy()
print(f"ten angles in {total:.0f} s (the reference's README: about 8 minutes each with 4 CPU processes and a K40)")
