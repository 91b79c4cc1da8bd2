import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, bench
from petibm_amd import capi
from petibm_amd.linsolver import LinSolverHIP
n = 512
s = LinSolverHIP("poisson", config_text=bench.solver_config("gmg", 1e-10, 1000, 0.9, 2, 2))
w = np.full(n, 1.0 / n)
s.assemblePoisson((n, n, n), [w, w, w], 5e-4, capi.NULLSPACE_CONSTANT)
xs_d, b_d, x_d = s.deviceVec(), s.deviceVec(), s.deviceVec()
pass
for k in range(2):
    print({name: round(s.timeKernel(w_, 10), 4) for name, w_ in (("product", 0), ("product+dot", 6), ("update", 8), ("update+product+dot", 7))}, flush=True)
s.destroy()
