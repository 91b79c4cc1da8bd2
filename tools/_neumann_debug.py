import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_navierstokes as T
from test_gpu_parity import _a0_table, amgx_cfg
from oracle import mesh as omesh, operators as oops, clib
from petibm_amd import capi
from petibm_amd.linsolver import LinSolverHIP
cfg = T.neumann_outlet((16, 12))
m = omesh.create_mesh(cfg)
dt, cnu = cfg["parameters"]["dt"], 0.5 * cfg["flow"]["nu"]
D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
_, A = oops.create_poisson_operator(D, G, L, dt, cnu)
A = oops.pin_row0(A)
rng = np.random.default_rng(3)
xs = rng.uniform(-1, 1, m.pN); xs[0] = 0
b = clib.spmv(A, xs)
nn = [int(v) for v in m.n[3][: m.dim]]
for text in (T.AMGX_P.replace("solver(solv)=PCG", "solver(solv)=PBICGSTAB"), T.AMGX_P.replace("solver(solv)=PCG", "solver(solv)=PBICGSTAB").replace("=AMG", "=BLOCK_JACOBI"),
             T.AMGX_P):
    s = LinSolverHIP("poisson", config_text=text)
    s.assemblePoissonBN(nn, [m.dL[3][d].true for d in range(m.dim)], m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu, 1, capi.NULLSPACE_PINNED)
    x = np.zeros(m.pN)
    try:
        s.solve(x, b)
        print("iters", s.getIters(), "reason", s.getReason(), "res", s.getResidual(), "err", np.abs(x - xs).max(), "true res", np.linalg.norm(b - clib.spmv(A, x)) / np.linalg.norm(b))
    except Exception as e:
        print("solve failed:", e)
    s.destroy()
