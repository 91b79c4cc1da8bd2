#!/usr/bin/env python3
"""The search direction's placement against x at size (round 5): one 512^3 multigrid-PCG solve per process, with the search
(pib_place_update_vector) on and off.  The p-update's rate is a property of the process's allocations, so the comparison needs
SEVERAL processes per setting:

    for k in 1 2 3 4; do python tools/placement_probe.py 1; python tools/placement_probe.py 0; done
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (one HIP runtime in the process: see petibm_amd/capi.py)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from petibm_amd import capi  # noqa: E402
from petibm_amd.linsolver import LinSolverHIP  # noqa: E402

on = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dt = 5e-4 if n == 512 else 1e-3
cfg = bench.solver_config("gmg", 1e-10, 1000, 0.9, 2, 2) + "pib_place_update_vector=%d\n" % on
s = LinSolverHIP("poisson", config_text=cfg)
w = np.full(n, 1.0 / n)
s.assemblePoisson((n, n, n), [w, w, w], dt, capi.NULLSPACE_CONSTANT)
xs_d, b_d, x_d = s.deviceVec(), s.deviceVec(), s.deviceVec()
xs_d.upload(bench.manufactured_solution(n, 0, n))
s.matMult(xs_d, b_d)
ms = []
for k in range(5):
    s.synchronize()
    t0 = time.perf_counter()
    s.solve(x_d, b_d)
    s.synchronize()
    ms.append(1e3 * (time.perf_counter() - t0))
x = x_d.download()
print(json.dumps({"place_update_vector": on, "n": n, "solve_ms": [round(v, 2) for v in ms], "iters": s.getIters(),
                  "placement": dict(zip(("searches", "candidates", "ms_had", "ms_kept"), s.placement())),
                  "x_crc": int(np.frombuffer(x.tobytes(), dtype=np.uint64).sum() & np.uint64(0xFFFFFFFFFFFF))}), flush=True)
s.destroy()
