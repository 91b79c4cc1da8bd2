import sys, json, time
sys.path.insert(0, ".")
import numpy as np
from petibm_amd import cases
from petibm_amd.navierstokes import DecoupledIBPMSolver
from petibm_amd.cases import uniform_stream as flow_config, AMGX_POISSON as AMGX_P, DIRECT_FORCES as FORCES
from petibm_amd.cases import circle
G = json.load(open("tests/golden/reference_test_vectors.json"))
sub = [{"end": -0.52, "cells": 363, "stretchRatio": 0.9900990099}, {"end": 0.52, "cells": 260, "stretchRatio": 1.0},
       {"end": 15.0, "cells": 363, "stretchRatio": 1.01}]
base = cases.cavity((986, 986))
base["mesh"] = [{"direction": d, "start": -15.0, "subDomains": sub} for d in "xy"]
cfg = flow_config(base, nu=0.00033333333333, dt=0.001)
vel = "-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n-velocity_pc_type jacobi\n"
t0 = time.perf_counter()
s = DecoupledIBPMSolver(cfg, bodies=[circle(786)], velocity_cfg=vel, poisson_cfg=AMGX_P.format(tol="1.0E-06"), forces_cfg=FORCES)
print("setup", time.perf_counter() - t0, s.pN, s.nf)
kl = G["koumoutsakos_leonard_1995_cylinder_re3000"]
t_ref, cd_ref = 0.5 * np.array(kl["t_radius_units"]), np.array(kl["cd"])
t0 = time.perf_counter()
for it in range(250, 3001, 250):
    s.advance(it - s.ite)
    cd = 2.0 * s.getForces()[1][0][0]
    print(it, it * 0.001, cd, np.interp(it * 0.001, t_ref, cd_ref), s.linSolversInfo()[1:6:2])
print("wall", time.perf_counter() - t0)
