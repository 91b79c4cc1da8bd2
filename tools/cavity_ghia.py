"""The reference's high-Reynolds lid-driven cavity cases (examples/navierstokes/liddrivencavity2dRe3200 / Re5000:
192 x 192 cells, dt = 0.002, 25000 / 60000 steps, linear solvers to 1e-6) against Ghia et al. (1982).
    python tools/cavity_ghia.py 3200 | 5000"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from petibm_amd import cases
from petibm_amd.navierstokes import NavierStokesSolver

re = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
nu, nt = {3200: (0.0003125, 25000), 5000: (0.0002, 60000)}[re]
n = 192
cfg = cases.cavity((n, n), lid=1.0)
cfg["flow"]["nu"] = nu
cfg["parameters"] = {"dt": 0.002, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
vel = "-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n-velocity_pc_type jacobi\n"
poi = "-poisson_ksp_type cg\n-poisson_ksp_atol 1.0E-06\n-poisson_ksp_rtol 0.0\n-poisson_ksp_max_it 1000\n-poisson_pc_type gamg\n"
s = NavierStokesSolver(cfg, velocity_cfg=vel, poisson_cfg=poi)
t0 = time.perf_counter()
s.advance(nt)
U, p = s.getState()
wall = time.perf_counter() - t0
u = U[: (n - 1) * n].reshape(n, n - 1)
v = U[(n - 1) * n:].reshape(n - 1, n)
yc = (np.arange(n) + 0.5) / n
g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_test_vectors.json")))[f"ghia_1982_re{re}_centerlines"]
ui = np.interp(g["y"][1:-1], yc, u[:, n // 2 - 1])
vi = np.interp(g["x"][1:-1], yc, v[n // 2 - 1, :])
du = np.abs(ui - np.array(g["u"][1:-1]))
dv = np.abs(vi - np.array(g["v"][1:-1]))
print(f"Re {re}: {nt} steps in {wall:.1f} s; |u - Ghia| per point", np.round(du, 4), "max |v - Ghia|", dv.max())
