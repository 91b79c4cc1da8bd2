"""The multi-rank drop-in on PETSC_DECIDE's boxes, at size, on ONE GPU (P loopback ranks = P host threads).

Every rank hands its DMDA-ordered rows of the n^3 cavity Poisson operator to pib_set_csr_i32 only (what
AmgXSolver::setA receives from an unchanged PetIBM, src/linsolver/linsolveramgx.cpp:84).  Reported: the set-up time of
the box route (structure recovery + the rows moved to natural z-slabs), the cost of moving b / x per solve, the
iteration count and true residual against the z-slab route's, and the halo bytes per Krylov product of the boxes'
general plan against the slabs' planes (Jacobi-PCG, where the boxes iterate on their own partition).
The ranks share one GPU: times are functional evidence and relative costs, not scaling.

    python tools/box_route_probe.py [n=256] [P=8] [n of the velocity part=128; 0: skip]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import bench
from oracle import clib, dmda, operators as oops
from petibm_amd import capi
import slab_plans as partition
from petibm_amd.linsolver import LinSolverHIP
from test_gpu_multirank_loopback import _run_ranks


class _Mesh:  # what dmda_layout reads
    def __init__(self, n):
        self.dim = 3
        self.n = np.array([[n - 1, n, n], [n, n - 1, n], [n, n, n - 1], [n, n, n], [n + 1] * 3])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dt = 5e-4
    w = np.full(n, 1.0 / n)
    t0 = time.perf_counter()
    A = oops.CSR.from_csr32(*clib.assemble_poisson32([n, n, n], [w, w, w], dt))
    L = dmda.dmda_layout(_Mesh(n), P)
    lay = L.pressure
    parts = [dmda.permuted_local_rows(A, lay.petsc_of_natural, lay.offsets, r)[0] for r in range(P)]
    inv = np.empty(A.n_rows, dtype=np.int64)
    inv[lay.petsc_of_natural] = np.arange(A.n_rows)
    xs = bench.manufactured_solution(n, 0, n)
    b = clib.spmv(A, xs)
    b_p = b[inv]
    print(f"# {n}^3 on {P} ranks, process grid {L.grid} (PETSC_DECIDE), boxes {lay.boxes[0][3:]}; host preparation {time.perf_counter() - t0:.1f} s",
          flush=True)
    for pc, label in (("gmg", "multigrid-PCG V(2,2)"), ("jacobi", "Jacobi-PCG"))[: 1 if n >= 512 else 2]:  # (Jacobi-PCG: ~2 n iterations)
        cfg = bench.solver_config(pc, 1e-10 if pc == "gmg" else 1e-6, 3000, 0.9, 2, 2, "jacobi") + "\n"

        def box_rank(r, uid):
            s = LinSolverHIP("poisson", config_text=cfg, rank=r, nranks=P, uid=uid, device=0)
            r0, r1 = int(lay.offsets[r]), int(lay.offsets[r + 1])
            loc = parts[r]
            t = time.perf_counter()
            s.setMatrix(oops.CSR(loc.n_rows, loc.n_cols, loc.rowptr.astype(np.int32), loc.col.astype(np.int32), loc.val), row0=r0,
                        n_global=A.n_rows)
            t_set = time.perf_counter() - t
            x_d, b_d, r_d = s.deviceVec(r1 - r0), s.deviceVec(r1 - r0), s.deviceVec(r1 - r0)
            b_d.upload(np.ascontiguousarray(b_p[r0:r1]))
            s.solve(x_d, b_d)
            s.synchronize()
            t = time.perf_counter()
            s.solve(x_d, b_d)
            s.synchronize()
            t_solve = time.perf_counter() - t
            cnt = s.counters().copy()
            s.matMult(x_d, r_d)
            rl = b_d.download() - r_d.download()
            out = dict(t_set=t_set, t_solve=t_solve, its=s.getIters(), cnt=cnt, rr=float(rl @ rl), st=s.gridStructure())
            s.destroy()
            return out

        def slab_rank(r, uid):
            s = LinSolverHIP("poisson", config_text=cfg, rank=r, nranks=P, uid=uid, device=0)
            pl = partition.all_plans((n, n, n), P)[r]
            r0, r1 = pl.row0, pl.row0 + pl.n_local
            p0, p1 = A.rowptr[r0], A.rowptr[r1]
            t = time.perf_counter()
            s.setMatrix(oops.CSR(pl.n_local, A.n_cols, (A.rowptr[r0:r1 + 1] - p0).astype(np.int32), A.col[p0:p1].astype(np.int32), A.val[p0:p1]),
                        row0=r0, n_global=A.n_rows)
            t_set = time.perf_counter() - t
            x_d, b_d = s.deviceVec(r1 - r0), s.deviceVec(r1 - r0)
            b_d.upload(np.ascontiguousarray(b[r0:r1]))
            s.solve(x_d, b_d)
            s.synchronize()
            t = time.perf_counter()
            s.solve(x_d, b_d)
            s.synchronize()
            t_solve = time.perf_counter() - t
            out = dict(t_set=t_set, t_solve=t_solve, its=s.getIters(), cnt=s.counters().copy())
            s.destroy()
            return out

        box = _run_ranks(P, box_rank)
        slab = _run_ranks(P, slab_rank)
        rel = np.sqrt(sum(q["rr"] for q in box)) / np.linalg.norm(b)
        print(f"{label}:", flush=True)
        for name, res in (("boxes", box), ("z-slabs", slab)):
            its = {q["its"] for q in res}
            ex = max(int(q["cnt"][3]) for q in res)
            sent = max(int(q["cnt"][7]) for q in res)
            prod = max(int(q["cnt"][0]) for q in res)
            print(f"  {name:8s} setMatrix {max(q['t_set'] for q in res):6.2f} s   solve {1e3 * max(q['t_solve'] for q in res):8.1f} ms   "
                  f"iterations {sorted(its)}   exchanges {ex}   sent {sent / 1e6:.2f} MB per solve ({sent / max(ex, 1) / 1e3:.1f} kB per exchange, "
                  f"{prod} products)", flush=True)
        print(f"  boxes: true relative residual {rel:.3e}; structure {box[0]['st']}", flush=True)
    nv = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    if nv > 0:
        velocity_part(nv, P)


def velocity_part(n, P):
    """vSolver->setMatrix(A): per rank [u box | v box | w box]; moved to packed z-slabs for the matrix-free products
    (default) against CSR products with the general packed halo plan on the boxes (pib_detect_structure=0: nothing recovered, nothing moved)."""
    from oracle import mesh as omesh
    dt, cnu = 1e-3, 0.5e-3
    t0 = time.perf_counter()
    m = omesh.create_mesh(omesh.uniform_config((n, n, n)))
    V = oops.create_velocity_operator(oops.create_laplacian(m), dt, cnu)
    L = dmda.dmda_layout(m, P)
    parts = [dmda.permuted_local_rows(V, L.packed_of_natural, L.packed_offsets, r)[0] for r in range(P)]
    inv = np.empty(m.UN, dtype=np.int64)
    inv[L.packed_of_natural] = np.arange(m.UN)
    b_p = clib.spmv(V, np.random.default_rng(3).uniform(-1, 1, m.UN))[inv]
    print(f"# velocity system {n}^3 ({m.UN} rows) on {P} ranks, process grid {L.grid}; host preparation {time.perf_counter() - t0:.1f} s", flush=True)
    base = ("config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=1000\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
            "solv:tolerance=1e-10\nsolv:norm=L2\nsolv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=0.9\npib_initial_guess_nonzero=0\n")
    for extra, label in (("", "boxes -> packed slabs, matrix-free"), ("pib_detect_structure=0\n", "boxes, CSR products (no structure recovery)")):
        def rank_fn(r, uid):
            s = LinSolverHIP("velocity", config_text=base + extra, rank=r, nranks=P, uid=uid, device=0)
            r0, r1 = int(L.packed_offsets[r]), int(L.packed_offsets[r + 1])
            loc = parts[r]
            t = time.perf_counter()
            s.setMatrix(oops.CSR(loc.n_rows, loc.n_cols, loc.rowptr.astype(np.int32), loc.col.astype(np.int32), loc.val), row0=r0, n_global=m.UN)
            t_set = time.perf_counter() - t
            x_d, b_d = s.deviceVec(r1 - r0), s.deviceVec(r1 - r0)
            b_d.upload(np.ascontiguousarray(b_p[r0:r1]))
            s.solve(x_d, b_d)
            s.synchronize()
            t = time.perf_counter()
            s.solve(x_d, b_d)
            s.synchronize()
            out = dict(t_set=t_set, t_solve=time.perf_counter() - t, its=s.getIters(), cnt=s.counters().copy(), sv=s.velocityStructure())
            s.destroy()
            return out
        res = _run_ranks(P, rank_fn)
        print(f"  {label:38s} setMatrix {max(q['t_set'] for q in res):6.2f} s   solve {1e3 * max(q['t_solve'] for q in res):8.1f} ms   iterations "
              f"{sorted({q['its'] for q in res})}   sent {max(int(q['cnt'][7]) for q in res) / 1e6:.2f} MB per solve   structure "
              f"{'recovered' if res[0]['sv'] else 'none'}", flush=True)


if __name__ == "__main__":
    main()
