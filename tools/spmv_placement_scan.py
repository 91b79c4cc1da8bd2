#!/usr/bin/env python3
"""Which vector's placement class sets the CSR product's mode at 512^3 (round 5)?  pib_time_kernel(100): the product with its
input, then its output, on each of 14 fresh allocations 2 GiB apart; one process per run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import bench  # noqa: E402
from petibm_amd import capi  # noqa: E402
from petibm_amd.linsolver import LinSolverHIP  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
s = LinSolverHIP("poisson", config_text=bench.solver_config("gmg", 1e-10, 1000, 0.9, 2, 2))
w = np.full(n, 1.0 / n)
s.assemblePoisson((n, n, n), [w, w, w], 5e-4, capi.NULLSPACE_CONSTANT)
print("plain product", s.timeKernel(0, 5), "ms", flush=True)
s.timeKernel(100, 1)
s.destroy()
