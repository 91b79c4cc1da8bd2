#!/usr/bin/env python3
"""HIP-event time of the matrix-free velocity product alone (pib_time_kernel 5) on the n^3 cavity: the A/B number for changes to
velstencil.hip that need no solve (timing experiments may compute garbage).  python tools/vel_product_time.py [n=256] [reps=40]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from petibm_amd.linsolver import LinSolverHIP  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = ("config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=10\nsolv:convergence=ABSOLUTE\nsolv:tolerance=1e-10\nsolv:norm=L2\n"
       "solv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=0.9\n" + os.environ.get("PIB_PROBE_CFG", "").replace(";", "\n") + "\n")
s = LinSolverHIP("velocity", config_text=cfg)
w = np.full(n, 1.0 / n)
a0 = np.array([[0.0 if (loc // 2) == f else -1.0 for loc in range(6)] for f in range(3)])
s.assembleVelocity((n, n, n), [w, w, w], (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), a0, 1e-3, 0.5e-3)
out = [s.timeKernel(5, reps) for _ in range(3)]
print("velocity product %d^3: " % n + " ".join("%.1f us" % (1e3 * t) for t in out))
s.destroy()
