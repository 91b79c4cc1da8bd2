"""Strong-scaling probe on ONE GPU: per-iteration cost of the multigrid-PCG solve on the 1/P-th slab of the 512^3
problem (512 x 512 x 512/P cells, single rank, no communication) -- the compute part of a P-GPU run, launch gaps
included.  python tools/slab_probe.py [P ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from petibm_amd import capi
from petibm_amd.linsolver import LinSolverHIP

EXTRA = os.environ.get("PIB_PROBE_CFG", "").replace(";", "\n")  # e.g. PIB_PROBE_CFG="pib_graph_max_rows=100000000"
for P in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    n = 512
    nz = n // P
    s = LinSolverHIP("poisson", config_text=bench.solver_config("gmg", 1e-10, 200, 0.8, 2, 2, "jacobi") + "\n" + EXTRA + "\n")
    w = np.full(n, 1.0 / n)
    s.assemblePoisson((n, n, nz), [w, w, w[:nz]], 5e-4, capi.NULLSPACE_CONSTANT)
    rng = np.random.default_rng(0)
    xs = rng.standard_normal(n * n * nz)
    xs_d, b_d, x_d = s.deviceVec(), s.deviceVec(), s.deviceVec()
    xs_d.upload(xs)
    s.matMult(xs_d, b_d)
    for _ in range(2):
        s.solve(x_d, b_d)
    s.synchronize()
    t0 = time.perf_counter()
    it = 0
    K = 5
    for _ in range(K):
        s.solve(x_d, b_d)
        it += s.getIters()
    s.synchronize()
    t = (time.perf_counter() - t0) / K
    print(f"P={P}: slab 512x512x{nz}: {1e3*t:.2f} ms/solve, {it/K:.1f} iters, {1e3*t/(it/K):.3f} ms/iter", flush=True)
    s.destroy()
