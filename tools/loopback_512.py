"""The 8-rank code path of `bench.py --gpus 8` (512^3, multigrid-PCG V(2,2), z-slabs of 64 planes) on ONE GPU through the
loopback transport: every kernel, halo plan and collective call site of the real run except RCCL itself; the ranks share
the device, so the time is not a scaling number.   python tools/loopback_512.py [P] [n]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from petibm_amd import capi
from petibm_amd.linsolver import LinSolverHIP
from test_gpu_multirank_loopback import _run_ranks

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w = np.full(n, 1.0 / n)
cfg = bench.solver_config("gmg", 1e-10, 200, 0.9, 2, 2, "jacobi") + "\n"


def rank_fn(r, uid):
    s = LinSolverHIP("poisson", config_text=cfg, rank=r, nranks=P, uid=uid, device=0)
    s.assemblePoisson((n, n, n), [w, w, w], 5e-4, capi.NULLSPACE_CONSTANT)
    k0, k1 = bench.slab(n, P, r)
    xs = bench.manufactured_solution(n, k0, k1)
    xs_d, b_d, x_d, r_d = s.deviceVec(), s.deviceVec(), s.deviceVec(), s.deviceVec()
    xs_d.upload(xs)
    s.matMult(xs_d, b_d)
    s.solve(x_d, b_d)
    t0 = time.perf_counter()
    s.solve(x_d, b_d)
    s.synchronize()
    t = time.perf_counter() - t0
    s.matMult(x_d, r_d)
    bl = b_d.download()
    rl = bl - r_d.download()
    out = (s.getIters(), float(rl @ rl), float(bl @ bl), t, s.counters().copy())
    s.destroy()
    return out


res = _run_ranks(P, rank_fn)
its = {r[0] for r in res}
rel = np.sqrt(sum(r[1] for r in res) / sum(r[2] for r in res))
print(f"P = {P}, {n}^3: iterations {its}, true relative residual {rel:.3e}, {1e3 * max(r[3] for r in res):.1f} ms per solve "
      f"(ranks time-share one GPU), halo exchanges per solve on rank 0: {int(res[0][4][3])}, reductions: {int(res[0][4][2])}")
