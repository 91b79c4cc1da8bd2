"""round 6: which ingredient of test_search_direction_placed_against_x_is_bit_identical leaves a double free at exit
    python tools/r06_exit_crash.py PLACE(0/1) MEMINFO(0/1) SECOND_X(0/1) MIN_ROWS"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

place, meminfo, second, min_rows = (int(v) for v in sys.argv[1:5])
from petibm_amd import capi
from petibm_amd.linsolver import LinSolverHIP
from test_gpu_parity import gmg_cfg

n = (64, 48, 40)
w = [np.full(n[0], 1.0 / n[0]), np.full(n[1], 1.0 / n[1]), np.full(n[2], 1.0 / n[2])]
xs = np.random.default_rng(11).uniform(-1, 1, n[0] * n[1] * n[2])
xs -= xs.mean()
extra = f"pib_place_update_vector={place}\npib_place_min_rows={min_rows}\n"
s = LinSolverHIP("poisson", config_text=gmg_cfg(pre=2, post=2, extra=extra))
s.assemblePoisson(list(n), w, 0.01, capi.NULLSPACE_CONSTANT)
xs_d, b_d, x_d, x2_d = s.deviceVec(), s.deviceVec(), s.deviceVec(), s.deviceVec()
xs_d.upload(xs)
s.matMult(xs_d, b_d)
x_d.upload(np.zeros_like(xs))
s.solve(x_d, b_d)
print("solve 1", s.getIters(), s.placementInfo(), flush=True)
if second:
    x2_d.upload(np.zeros_like(xs))
    s.solve(x2_d, b_d)
    print("solve 2", s.getIters(), s.placementInfo(), flush=True)
s.destroy()
if meminfo:
    import torch
    print(torch.cuda.mem_get_info(), flush=True)
print("done", flush=True)
