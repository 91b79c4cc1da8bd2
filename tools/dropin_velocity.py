#!/usr/bin/env python3
"""The velocity solve as an unchanged PetIBM reaches it -- setMatrix(A) with the assembled CSR, nothing else -- with and
without the structure recovered from the matrix (csrc/structure.cpp): ms per BiCGStab + Jacobi solve on an n^3 cavity."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from petibm_amd import capi  # noqa: E402
from petibm_amd.linsolver import LinSolverHIP  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
cfg = ("config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=1000\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
       "solv:tolerance=1e-10\nsolv:norm=L2\nsolv:store_res_history=1\nsolv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=0.9\n"
       "pib_initial_guess_nonzero=0\n")
w = np.full(n, 1.0 / n)
a0 = np.array([[0.0 if (loc // 2) == f else -1.0 for loc in range(6)] for f in range(3)])
src = LinSolverHIP("velocity", config_text=cfg)
src.assembleVelocity((n, n, n), [w, w, w], (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), a0, 1e-3, 0.5e-3)
rp, cl, vl = src.getCSR()


class M:  # what LinSolverHIP.setMatrix takes
    pass


A = M()
A.n_rows = A.n_cols = len(rp) - 1
A.rowptr, A.col, A.val = rp, cl, vl
b = np.random.default_rng(1).uniform(-1, 1, A.n_rows)
for label, extra in (("assembled on the device (pib_assemble_velocity)", None), ("setMatrix, structure recovered", ""),
                     ("setMatrix, CSR products (pib_detect_structure=0)", "pib_detect_structure=0\n")):
    if extra is None:
        s = src
    else:
        s = LinSolverHIP("velocity", config_text=cfg + extra)
        s.setMatrix(A)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s.solve(x, b)
    s.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{label:52s} {1e3 * dt:8.2f} ms per solve (host vectors), {s.getIters()} iterations, structure {s.velocityStructure()}")
    if s is not src:
        s.destroy()
src.destroy()
