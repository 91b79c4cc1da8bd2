"""world_size-2 (and 3) CPU test of the N>1 path over the gloo backend.

The HIP kernels cannot run here, so what is exercised is everything of the multi-GPU design that is not a
kernel: the z-slab partition (petibm_amd.partition, the rule libpetibm_amd.so applies), the ghost-shifted
local column layout of the device CSR, the contiguous-plane halo plan (what rank r sends = what r+-1
receives) and the fused scalar all-reduce -- driven by the same Jacobi-PCG recurrences, with numpy standing
in for the kernels.  The distributed result must reproduce the single-rank oracle.
"""
import os
import socket

import numpy as np
import pytest

# torch is imported where it is used, not at collection: a `-m gpu` run of the suite then never imports it, and the library can run
# on /opt/rocm's own HIP runtime instead of the one torch bundles (petibm_amd/capi.py: PIB_TORCH_FIRST)

from oracle import clib, mesh as omesh, operators as oops
import slab_plans as partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _system(n):
    m = omesh.create_mesh(omesh.uniform_config(n))
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, 1e-2, 0.5e-2)
    xs = np.random.default_rng(20260928).uniform(-1, 1, m.pN)
    xs -= xs.mean()
    return m, A, clib.spmv(A, xs)


def _local_csr(A, plan):
    """Rows [row0,row0+n_local), global columns -> ghost-shifted local columns (assemble.hip:upload_csr)."""
    r0, r1 = plan.row0, plan.row0 + plan.n_local
    p0, p1 = A.rowptr[r0], A.rowptr[r1]
    col = A.col[p0:p1]
    lo = max(0, r0 - int(col.min()))
    hi = max(0, int(col.max()) - (r1 - 1))
    assert (lo, hi) == (plan.ghost_lo, plan.ghost_hi)  # the plan predicts what the matrix needs
    return A.rowptr[r0:r1 + 1] - p0, plan.local_col(col), A.val[p0:p1]


def _halo(x_pad, plan):
    """One exchange: [ghost_lo | owned | ghost_hi]; contiguous planes, one send/recv pair per neighbour."""
    import torch
    import torch.distributed as dist
    n = plan.n_local
    owned = x_pad[plan.ghost_lo:plan.ghost_lo + n]
    reqs = []
    bufs = {}
    if plan.rank > 0:
        reqs.append(dist.isend(torch.from_numpy(owned[:plan.send_prev].copy()), plan.rank - 1))
        bufs["lo"] = torch.empty(plan.ghost_lo, dtype=torch.float64)
        reqs.append(dist.irecv(bufs["lo"], plan.rank - 1))
    if plan.rank < plan.nranks - 1:
        reqs.append(dist.isend(torch.from_numpy(owned[n - plan.send_next:].copy()), plan.rank + 1))
        bufs["hi"] = torch.empty(plan.ghost_hi, dtype=torch.float64)
        reqs.append(dist.irecv(bufs["hi"], plan.rank + 1))
    for r in reqs:
        r.wait()
    if "lo" in bufs:
        x_pad[:plan.ghost_lo] = bufs["lo"].numpy()
    if "hi" in bufs:
        x_pad[plan.ghost_lo + n:] = bufs["hi"].numpy()


def _allreduce(*vals):
    import torch
    import torch.distributed as dist
    t = torch.tensor(vals, dtype=torch.float64)
    dist.all_reduce(t)
    return t.tolist()


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, A, b = _system(n)
    plan = partition.slab_plan(n, world, rank)
    rp, cl, vl = _local_csr(A, plan)
    nl, g0 = plan.n_local, plan.ghost_lo
    bl = b[plan.row0:plan.row0 + nl]
    rows = np.repeat(np.arange(nl), np.diff(rp))
    diag = np.zeros(nl)
    isd = cl == rows + g0
    diag[rows[isd]] = vl[isd]
    dinv = 1.0 / diag

    def spmv(p_pad):
        _halo(p_pad, plan)
        return np.bincount(rows, weights=vl * p_pad[cl], minlength=nl)

    # Jacobi-PCG with the constant null space removed lazily (krylov.hip), monitored on ||r||
    x = np.zeros(nl)
    r = bl.copy()
    z = dinv * r
    zr, rr, sz, sr = _allreduce(z @ r, r @ r, z.sum(), r.sum())
    N = float(m.pN)
    mean = sz / N
    beta = zr - mean * sr
    r0 = np.sqrt(rr)
    p_pad = np.zeros(g0 + nl + plan.ghost_hi)
    its = 0
    while its < 2000:
        own = p_pad[g0:g0 + nl]
        own[:] = (z - mean) + ((beta / beta_old) * own if its else 0.0)
        w = spmv(p_pad)
        (pw,) = _allreduce(own @ w)
        a = beta / pw
        x += a * own
        r -= a * w
        z = dinv * r
        beta_old = beta
        zr, rr, sz, sr = _allreduce(z @ r, r @ r, z.sum(), r.sum())
        mean = sz / N
        beta = zr - mean * sr
        its += 1
        if np.sqrt(rr) <= 1e-10 * r0:
            break
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    np.save(os.path.join(out_dir, f"its{rank}.npy"), np.array([its]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, (12, 10, 8)), (3, (9, 8, 10)), (2, (24, 16))])
def test_zslab_pcg_over_gloo_matches_single_rank_oracle(tmp_path, world, n):
    port = _free_port()
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    m, A, b = _system(n)
    ref = clib.cg(A, b, pc="jacobi", nullspace=1, norm="unpreconditioned", rtol=1e-10, atol=0.0, dtol=1e300,
                  maxit=2000)
    x = np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)])
    its = [int(np.load(tmp_path / f"its{r}.npy")[0]) for r in range(world)]
    assert len(set(its)) == 1 and abs(its[0] - ref["iters"]) <= 1
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    e = (x - x.mean()) - (ref["x"] - ref["x"].mean())
    assert np.linalg.norm(e) <= 1e-8 * np.linalg.norm(ref["x"])


SHIFT = 40.0  # A - SHIFT I: the (negative semi-definite) Poisson operator made non-singular, a stand-in for the velocity system


def _worker_bcgs(rank, world, port, n, out_dir):
    """BiCGStab + Jacobi in the lean form of csrc/krylov.hip with the residual update merged (pib_bicgstab_form=3): per iteration two
    halo exchanges (the products' inputs) and TWO all-reduces -- v.r~, then the five sums s.t, t.t, s.s, r~.s, r~.t from which omega,
    |r|^2 and r.r~ follow -- instead of the three of the textbook iteration."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, A, b = _system(n)
    plan = partition.slab_plan(n, world, rank)
    rp_, cl, vl = _local_csr(A, plan)
    nl, g0 = plan.n_local, plan.ghost_lo
    rows = np.repeat(np.arange(nl), np.diff(rp_))
    isd = cl == rows + g0
    vl = vl.copy()
    vl[isd] -= SHIFT
    dinv = np.zeros(nl)
    dinv[rows[isd]] = 1.0 / vl[isd]
    bl = b[plan.row0:plan.row0 + nl]
    pad = np.zeros(g0 + nl + plan.ghost_hi)
    reductions = 0

    def product(vec):  # A M^-1 vec: the sweep applied to the owned entries, the neighbours' planes exchanged
        pad[g0:g0 + nl] = dinv * vec
        _halo(pad, plan)
        return np.bincount(rows, weights=vl * pad[cl], minlength=nl)

    x = np.zeros(nl)
    r = bl.copy()
    rt = r.copy()
    p = np.zeros(nl)
    v = np.zeros(nl)
    (rho,) = _allreduce(r @ rt)
    r0 = np.sqrt(rho)
    rho_old = alpha = omega_old = 1.0
    its = 0
    while its < 500:
        beta = (rho / rho_old) * (alpha / omega_old)
        p = r - (omega_old * beta) * v + beta * p
        v = product(p)
        (vrt,) = _allreduce(v @ rt)
        reductions += 1
        alpha = rho / vrt
        s = r - alpha * v
        t = product(s)
        st, tt, ss, rts, rtt = _allreduce(s @ t, t @ t, s @ s, rt @ s, rt @ t)
        reductions += 1
        omega = st / tt
        x += alpha * (dinv * p) + omega * (dinv * s)
        r = s - omega * t  # (on the device: formed by the NEXT p-update)
        r2 = max(ss - omega * (2.0 * st - omega * tt), 0.0)
        rho_old, omega_old, rho = rho, omega, rts - omega * rtt
        its += 1
        if np.sqrt(r2) <= 1e-10 * r0:
            break
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    np.save(os.path.join(out_dir, f"its{rank}.npy"), np.array([its, reductions]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, (12, 10, 8)), (3, (9, 8, 10))])
def test_zslab_bicgstab_with_two_reductions_over_gloo_matches_single_rank_oracle(tmp_path, world, n):
    import copy
    port = _free_port()
    import torch.multiprocessing as mp
    mp.spawn(_worker_bcgs, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    m, A, b = _system(n)
    A2 = copy.copy(A)
    A2.val = A.val.copy()
    rows = np.repeat(np.arange(A.n_rows), np.diff(A.rowptr))
    A2.val[A.col == rows] -= SHIFT
    ref = clib.bcgs(A2, b, pc="jacobi", norm="unpreconditioned", rtol=1e-10, atol=0.0, dtol=1e300, maxit=500)
    x = np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)])
    rec = [np.load(tmp_path / f"its{r}.npy") for r in range(world)]
    assert len({int(v[0]) for v in rec}) == 1 and abs(int(rec[0][0]) - ref["iters"]) <= 1
    assert all(int(v[1]) == 2 * int(v[0]) for v in rec)  # two all-reduces per iteration
    assert np.linalg.norm(b - clib.spmv(A2, x)) <= 2e-10 * np.linalg.norm(b)
    assert np.linalg.norm(x - ref["x"]) <= 1e-8 * np.linalg.norm(ref["x"])
