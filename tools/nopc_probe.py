import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import clib, mesh as omesh, operators as oops
from petibm_amd.linsolver import LinSolverHIP
from test_gpu_parity import _a0_table, amgx_cfg, stretched_3d
from test_gpu_multirank_loopback import _run_ranks, _velocity_slab_indices
cfg = stretched_3d((128, 10, 16))
m = omesh.create_mesh(cfg)
dt, cnu = 0.004, 0.5 * 0.01
A = oops.create_velocity_operator(oops.create_laplacian(m), dt, cnu)
us = np.random.default_rng(12).uniform(-1, 1, A.n_rows)
b = clib.spmv(A, us)
n = [int(v) for v in m.n[3][: m.dim]]
w = [m.dL[3][d].true for d in range(m.dim)]
text = amgx_cfg(solver="PBICGSTAB", pc="NOSOLVER", tol=1e-10, conv="ABSOLUTE", maxit=500) + "pib_march_min_cells=0\n"
for extra in ("", "pib_velocity_tile_edges=0\n", "pib_fuse_bicgstab_dots=0\n", "pib_matrix_free_velocity=0\n"):
    s1 = LinSolverHIP("velocity", config_text=text + extra)
    s1.assembleVelocity(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
    x1 = np.zeros(A.n_rows); s1.solve(x1, b)
    h = s1.getResidualHistory()
    print("1 rank", repr(extra), s1.getIters(), np.linalg.norm(b - clib.spmv(A, x1)), h[5], h[20], h[60])
    s1.destroy()
    for P in (2, 3):
        own = [_velocity_slab_indices(m, P, r) for r in range(P)]
        def rank_fn(r, uid):
            s = LinSolverHIP("velocity", config_text=text + extra, rank=r, nranks=P, uid=uid, device=0)
            s.assembleVelocity(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
            x = np.zeros(own[r].size); s.solve(x, np.ascontiguousarray(b[own[r]]))
            out = x, s.getIters(), s.getResidualHistory(); s.destroy(); return out
        res = _run_ranks(P, rank_fn)
        x = np.empty(A.n_rows)
        for r in range(P): x[own[r]] = res[r][0]
        print(P, "ranks", repr(extra), res[0][1], np.linalg.norm(b - clib.spmv(A, x)), res[0][2][5], res[0][2][20], res[0][2][60])
