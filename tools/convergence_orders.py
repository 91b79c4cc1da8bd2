import sys, time
sys.path.insert(0, ".")
import numpy as np
from petibm_amd import cases
from petibm_amd.navierstokes import NavierStokesSolver
vel = "-velocity_ksp_type bcgs\n-velocity_ksp_rtol 1.0E-08\n-velocity_ksp_atol 0.0\n-velocity_ksp_max_it 10000\n-velocity_pc_type jacobi\n"
poi = "-poisson_ksp_type cg\n-poisson_ksp_rtol 1.0E-08\n-poisson_ksp_atol 0.0\n-poisson_ksp_max_it 20000\n-poisson_pc_type gamg\n"
sol = {}
t0 = time.perf_counter()
for n in (20, 60, 180, 540):
    cfg = cases.cavity((n, n), lid=1.0)
    cfg["flow"]["nu"] = 0.01
    cfg["parameters"] = {"dt": 5.0e-4, "convection": "EULER_EXPLICIT", "diffusion": "EULER_IMPLICIT"}
    s = NavierStokesSolver(cfg, velocity_cfg=vel, poisson_cfg=poi)
    s.advance(500)
    U, p = s.getState()
    r = n // 20
    u = U[: (n - 1) * n].reshape(n, n - 1); v = U[(n - 1) * n:].reshape(n - 1, n); pp = p.reshape(n, n)
    ux = (np.arange(19) + 1) * r - 1; cc = (np.arange(20) * r + (r - 1) // 2)
    sol[n] = (u[np.ix_(cc, ux)], v[np.ix_(ux, cc)], (pp - pp.mean())[np.ix_(cc, cc)])
    s.destroy()
print("wall", time.perf_counter() - t0)
for k, name in enumerate("uvp"):
    e1 = np.linalg.norm(sol[60][k] - sol[20][k]); e2 = np.linalg.norm(sol[180][k] - sol[60][k]); e3 = np.linalg.norm(sol[540][k] - sol[180][k])
    print(name, "first", np.log(e1 / e2) / np.log(3.0), "last", np.log(e2 / e3) / np.log(3.0))
