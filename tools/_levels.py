import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import mesh as omesh
from petibm_amd import capi, cases
from petibm_amd.linsolver import LinSolverHIP
sub = [{"end": -0.54, "cells": 171, "stretchRatio": 0.980392156}, {"end": 0.54, "cells": 108, "stretchRatio": 1.0},
       {"end": 15.0, "cells": 171, "stretchRatio": 1.02}]
cfg = cases.cavity((450, 450), lid=0.0)
cfg["mesh"] = [{"direction": d, "start": -15.0, "subDomains": [dict(s) for s in sub]} for d in "xy"]
m = omesh.create_mesh(cfg)
n = [int(v) for v in m.n[3][: m.dim]]
w = [m.dL[3][d].true for d in range(m.dim)]
s = LinSolverHIP("poisson", config_text=bench.solver_config("gmg", 1e-6, 200, 0.9, 2, 2, "jacobi"))
s.assemblePoisson(n, w, 0.0025, capi.NULLSPACE_CONSTANT)
print(s.multigridLevels())
