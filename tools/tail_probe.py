"""The single-workgroup coarse tail of the V-cycle (gmg.hip: k_coarse_tail) with its vectors and tables in HBM / in LDS.

    python tools/tail_probe.py [lds=0|1] [tail cells=-1] ["extra config lines; separated by ;"]

Solves a 2-D (448^2) and a 3-D (128^3) cavity pressure system with the multigrid-PCG and prints iterations, ms per solve
and a hash of the solution (the two settings must print the same hash: same arithmetic).  Under
`rocprofv3 --kernel-trace --stats` the k_coarse_tail row gives the kernel's duration per V-cycle.
"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from petibm_amd import capi
from petibm_amd.linsolver import LinSolverHIP


def main():
    lds = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    tail = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    extra = sys.argv[3].replace(";", "\n") if len(sys.argv) > 3 else ""
    for n in ((448, 448), (450, 450), (128, 128, 128)):
        cfg = bench.solver_config("gmg", 1e-10, 200, 0.9, 2, 2, "jacobi") + f"\npib_coarse_tail_lds={lds}\npib_coarse_tail={tail}\n{extra}\n"
        s = LinSolverHIP("poisson", config_text=cfg)
        w = [np.full(k, 1.0 / k) for k in n]
        s.assemblePoisson(list(n), w, 5e-4, capi.NULLSPACE_CONSTANT)
        N = int(np.prod(n))
        rng = np.random.default_rng(5)
        b = rng.uniform(-1, 1, N)
        b -= b.mean()
        x_d, b_d = s.deviceVec(N), s.deviceVec(N)
        b_d.upload(b)
        s.solve(x_d, b_d)
        s.synchronize()
        t = time.perf_counter()
        reps = 5
        for _ in range(reps):
            s.solve(x_d, b_d)
        s.synchronize()
        ms = 1e3 * (time.perf_counter() - t) / reps
        x = x_d.download()
        print(f"n={n} lds={lds} tail={tail}: {s.getIters()} iterations, {ms:.3f} ms per solve, levels {len(s.multigridLevels())}, "
              f"sha {hashlib.sha256(x.tobytes()).hexdigest()[:16]}", flush=True)
        s.destroy()


if __name__ == "__main__":
    main()
