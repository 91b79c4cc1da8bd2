#!/usr/bin/env python3
"""HIP-event time of one V-cycle (pib_time_kernel 4) and of the stencil twin's product (3) on the 512^3 cavity Poisson system:
the A/B number for changes to the multigrid kernels (the whole solve varies by +-1.5 ms from run to run)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from petibm_amd import capi  # noqa: E402
from petibm_amd.linsolver import LinSolverHIP  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
s = LinSolverHIP("poisson", config_text=bench.solver_config("gmg", 1e-10, 200, 0.9, 2, 2, "jacobi"))
w = np.full(n, 1.0 / n)
s.assemblePoisson((n, n, n), [w, w, w], 5e-4 if n == 512 else 1e-3, capi.NULLSPACE_CONSTANT)
out = []
for _ in range(3):
    out.append((s.timeKernel(4, reps), s.timeKernel(3, reps)))
print(" ".join(f"V {a:.4f} twin {b:.4f}" for a, b in out))
s.destroy()
