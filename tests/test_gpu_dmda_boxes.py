"""Rows as an UNCHANGED PetIBM hands them to setMatrix on P > 1 ranks: DMDA boxes in PETSc ordering.

The reference creates its DMDAs with nProc = PETSC_DECIDE (src/mesh/cartesianmesh.cpp:97,503-519): from 4 ranks up a
3-D mesh is cut into boxes ((1,2,2), (2,2,2)), the unknowns are numbered rank after rank with every box in its own
natural order (cartesianmesh.cpp:700-738), and the velocity unknowns are packed per rank as [u box | v box | w box]
(cartesianmesh.cpp:741-779).  oracle/dmda.py restates the layout; every rank below feeds its DMDA-ordered local rows
through pib_set_csr[_i32] ONLY (what AmgXSolver::setA receives, src/linsolver/linsolveramgx.cpp:84) -- no assembly on
the device, no grid hint -- over the loopback transport (P host threads on the one GPU).

Bars: the product across the box faces is bit-identical to the oracle's product with the same (permuted) CSR; the
Poisson solve with the multigrid (rows moved once to natural z-slabs inside the backend, b / x per solve) and the
velocity solve with PBICGSTAB + Jacobi (general packed halo plan) reproduce the single-rank solutions to the solver
tolerance with the single-rank iteration counts; the halo bytes of the box route are reported.
"""
import numpy as np
import pytest

from oracle import clib, dmda, mesh as omesh, operators as oops
from test_gpu_multirank_loopback import _cfg, _run_ranks

pytestmark = pytest.mark.gpu


def _stretched(n, periodic=None):
    dim = len(n)
    periodic = periodic or (False,) * dim
    cfg = omesh.periodic_config(n, periodic)
    r = (1.08, 0.93, 1.05)
    cfg["mesh"] = [{"direction": "xyz"[d], "start": -1.0,
                    "subDomains": [{"end": 0.0, "cells": n[d] // 2, "stretchRatio": 1.0 / r[d]},
                                   {"end": 1.5 + 0.5 * d, "cells": n[d] - n[d] // 2, "stretchRatio": r[d]}]}
                   for d in range(dim)]
    return cfg


def _poisson(n, dt, pinned=False, periodic=None, stretched=True):
    cfg = _stretched(n, periodic) if stretched else omesh.periodic_config(n, periodic or (False,) * len(n))
    m = omesh.create_mesh(cfg)
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.5e-2)
    if pinned:
        A = oops.pin_row0(A)
    xs = np.random.default_rng(11).uniform(-1, 1, m.pN)
    if pinned:
        xs[0] = 0.0
    else:
        xs -= xs.mean()
    return m, A, xs, clib.spmv(A, xs)


def _global_permuted(A, new_of_old, offsets, P):
    parts = [dmda.permuted_local_rows(A, new_of_old, offsets, r)[0] for r in range(P)]
    rp = np.concatenate([[0]] + [p.rowptr[1:] + sum(q.nnz for q in parts[:i]) for i, p in enumerate(parts)]).astype(np.int64)
    return oops.CSR(A.n_rows, A.n_cols, rp, np.concatenate([p.col for p in parts]), np.concatenate([p.val for p in parts])), parts


CASES = [
    # P, n, process grid (None: PETSC_DECIDE), pinned, periodic, int32 indices
    (4, (12, 12, 12), None, False, None, True),          # (1,2,2)
    (8, (12, 12, 12), None, False, None, True),          # (2,2,2)
    (8, (13, 11, 14), None, True, None, False),          # ragged boxes, pinned pressure (the AmgX convention), 64-bit indices
    (4, (12, 10, 14), (2, 1, 2), False, None, True),
    (4, (16, 9, 8), (4, 1, 1), False, None, True),       # x-slabs: still boxes in this ordering
    (3, (9, 15, 10), (1, 3, 1), False, None, True),
    (6, (12, 12, 18), (1, 2, 3), False, (False, True, False), True),   # periodic along a cut direction
    (4, (12, 12, 12), (2, 2, 1), False, (True, False, True), True),    # periodic across the process seam in x, inside the box in z
    (4, (16, 20), None, False, None, True),              # 2-D (2,2)
    (8, (24, 40), None, True, None, True),               # 2-D (2,4), pinned
    (4, (18, 16), (4, 1), False, (False, True), True),
]


@pytest.mark.parametrize("host_rows", [False, True])
@pytest.mark.parametrize("P,n,grid,pinned,periodic,i32", CASES)
def test_poisson_rows_in_dmda_boxes(P, n, grid, pinned, periodic, i32, host_rows, monkeypatch):
    """(host_rows: the set-up move of the rows through the host loops, PIB_BOX_ROWS_ON_DEVICE=0 -- the path rows wider than 16
    entries take; default since round 4: records, per-row sort and the slab's CSR on the device, redistribute.hip)"""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    if host_rows:
        monkeypatch.setenv("PIB_BOX_ROWS_ON_DEVICE", "0")
    dt = 0.01
    m, A, xs, b = _poisson(n, dt, pinned, periodic)
    L = dmda.dmda_layout(m, P, grid)
    if grid is None:
        assert L.grid == dmda.decide_process_grid(n, P)
    lay = L.pressure
    Ap, parts = _global_permuted(A, lay.petsc_of_natural, lay.offsets, P)
    inv = np.empty(m.pN, dtype=np.int64)
    inv[lay.petsc_of_natural] = np.arange(m.pN)
    xs_p, b_p = xs[inv], b[inv]                     # PETSc ordering
    y_ref = clib.spmv(Ap, xs_p)
    cfg = _cfg("AMG", tol=1e-11) if pinned else _cfg("AMG", tol=1e-11, extra="pib_agglomerate_below=100\n")

    def rank_fn(r, uid):
        s = LinSolverHIP("poisson", config_text=cfg, rank=r, nranks=P, uid=uid, device=0)
        r0, r1 = int(lay.offsets[r]), int(lay.offsets[r + 1])
        loc = parts[r]
        if i32:
            loc = oops.CSR(loc.n_rows, loc.n_cols, loc.rowptr.astype(np.int32), loc.col.astype(np.int32), loc.val)
        s.setMatrix(loc, row0=r0, n_global=m.pN)
        st = s.gridStructure()
        if r == 0:  # an explicit hint speaks of natural z-slabs: refused on boxes with a message that says what to do instead
            wdt = [m.dL[3][d].true for d in range(m.dim)]
            with pytest.raises(capi.PibError) as ei:
                s.setGridHint(n, wdt, [dt / (0.5 * (q[1:] + q[:-1])) for q in wdt], capi.NULLSPACE_CONSTANT)
            assert ei.value.code == capi.ERR_SUP and "leave the hint out" in str(ei.value)
        y = np.empty(r1 - r0)
        s.matMult(np.ascontiguousarray(xs_p[r0:r1]), y)
        rp, cl, vl = s.getCSR()
        assert np.array_equal(cl, parts[r].col) and np.array_equal(vl, parts[r].val)
        x = np.zeros(r1 - r0)
        s.solve(x, np.ascontiguousarray(b_p[r0:r1]))
        out = y, x, s.getIters(), s.getReason(), st, s.counters().copy()
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    y = np.concatenate([q[0] for q in res])
    assert np.array_equal(y, y_ref), "the product across the box faces is not the oracle's"
    for q in res:
        assert q[3] > 0 and q[4] is not None and q[4]["detected"] and tuple(q[4]["n"]) == tuple(n)
        assert q[4]["nullspace"] == (capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    assert len({q[2] for q in res}) == 1
    x = np.empty(m.pN)
    x[inv] = np.concatenate([q[1] for q in res])     # back to natural ordering
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-11 * np.linalg.norm(b)
    # the single rank, same configuration, the same matrix through setMatrix
    s1 = LinSolverHIP("poisson", config_text=cfg)
    s1.setMatrix(A)
    x1 = np.zeros(m.pN)
    s1.solve(x1, b)
    its1 = s1.getIters()
    s1.destroy()
    assert abs(res[0][2] - its1) <= max(1, int(0.03 * its1)), (res[0][2], its1)
    e = (x - x.mean()) - (x1 - x1.mean()) if not pinned else x - x1
    assert np.linalg.norm(e) <= 1e-8 * np.linalg.norm(x1)
    if pinned:
        assert x[0] == 0.0


@pytest.mark.parametrize("P,n,grid,pc,periodic", [
    (4, (10, 9, 8), None, "BLOCK_JACOBI", None),
    (8, (10, 12, 20), None, "BLOCK_JACOBI", None),      # (>= 2 planes of every component per rank once on slabs)
    (8, (10, 12, 20), None, "NOSOLVER", None),
    (4, (12, 14), None, "BLOCK_JACOBI", None),
    (6, (9, 12, 12), (1, 2, 3), "BLOCK_JACOBI", (False, False, True)),
])
def test_velocity_rows_in_packed_dmda_boxes(P, n, grid, pc, periodic):
    """vSolver->setMatrix(A) on P ranks (navierstokes.cpp:163): per rank [u box | v box | w box], PBICGSTAB + Jacobi (the
    reference's velocity_solver.info) on the general halo plan."""
    from petibm_amd.linsolver import LinSolverHIP
    dt, cnu = 0.01, 0.5 * 0.01
    m = omesh.create_mesh(_stretched(n, periodic))
    Lp = oops.create_laplacian(m)
    V = oops.create_velocity_operator(Lp, dt, cnu)
    L = dmda.dmda_layout(m, P, grid)
    Vp, parts = _global_permuted(V, L.packed_of_natural, L.packed_offsets, P)
    inv = np.empty(m.UN, dtype=np.int64)
    inv[L.packed_of_natural] = np.arange(m.UN)
    xs = np.random.default_rng(5).uniform(-1, 1, m.UN)
    b = clib.spmv(V, xs)
    xs_p, b_p = xs[inv], b[inv]
    y_ref = clib.spmv(Vp, xs_p)
    cfg = (f"config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=500\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
           f"solv:tolerance=1e-11\nsolv:norm=L2\nsolv:preconditioner(prec)={pc}\nprec:relaxation_factor=1.0\npib_initial_guess_nonzero=0\n")

    def rank_fn(r, uid):
        s = LinSolverHIP("velocity", config_text=cfg, rank=r, nranks=P, uid=uid, device=0)
        r0, r1 = int(L.packed_offsets[r]), int(L.packed_offsets[r + 1])
        loc = parts[r]
        s.setMatrix(oops.CSR(loc.n_rows, loc.n_cols, loc.rowptr.astype(np.int32), loc.col.astype(np.int32), loc.val), row0=r0, n_global=m.UN)
        sv = s.velocityStructure()
        y = np.empty(r1 - r0)
        s.matMult(np.ascontiguousarray(xs_p[r0:r1]), y)
        x = np.zeros(r1 - r0)
        s.solve(x, np.ascontiguousarray(b_p[r0:r1]))
        out = y, x, s.getIters(), s.getReason(), s.counters().copy(), sv
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    assert np.array_equal(np.concatenate([q[0] for q in res]), y_ref)
    assert all(q[3] > 0 for q in res) and len({q[2] for q in res}) == 1
    # the boxes are moved to packed z-slabs inside the backend and the operator's structure is recovered there: matrix-free
    # products (a periodic slab axis keeps the CSR products on the boxes)
    if periodic and periodic[-1]:
        assert all(q[5] is None for q in res)
    else:
        assert all(q[5] is not None and q[5]["detected"] and q[5]["dim"] == m.dim for q in res)
    x = np.empty(m.UN)
    x[inv] = np.concatenate([q[1] for q in res])
    assert np.linalg.norm(b - clib.spmv(V, x)) <= 2e-11 * np.sqrt(m.UN)
    s1 = LinSolverHIP("velocity", config_text=cfg + "pib_matrix_free_velocity=0\n")
    s1.setMatrix(V)
    x1 = np.zeros(m.UN)
    s1.solve(x1, b)
    its1 = s1.getIters()
    s1.destroy()
    assert abs(res[0][2] - its1) <= 1, (res[0][2], its1)
    assert np.linalg.norm(x - x1) <= 1e-9 * np.linalg.norm(x1)
    # halo traffic of the box route: every rank reports the bytes it packed and sent, one exchange per product
    for q in res:
        exchanges, sent = int(q[4][3]), int(q[4][7])
        assert exchanges >= 2 * q[2] and sent > 0 and sent % 8 == 0


def test_box_halo_bytes_match_the_faces():
    """Bytes a rank sends per Krylov product on (2,2,2) boxes = 8 B x the cells of its three inner faces (7-point operator:
    no edges, no corners) -- the figure DESIGN.md 5 quotes for the box route against the z-slab route's planes."""
    from petibm_amd.linsolver import LinSolverHIP
    P, n, dt = 8, (12, 10, 8), 0.01
    m, A, xs, b = _poisson(n, dt, stretched=False)
    L = dmda.dmda_layout(m, P)
    lay = L.pressure
    _, parts = _global_permuted(A, lay.petsc_of_natural, lay.offsets, P)
    cfg = _cfg("BLOCK_JACOBI", tol=1e-10)

    def rank_fn(r, uid):
        s = LinSolverHIP("poisson", config_text=cfg, rank=r, nranks=P, uid=uid, device=0)
        r0, r1 = int(lay.offsets[r]), int(lay.offsets[r + 1])
        s.setMatrix(parts[r], row0=r0, n_global=m.pN)
        x = np.zeros(r1 - r0)
        inv = np.empty(m.pN, dtype=np.int64)
        inv[lay.petsc_of_natural] = np.arange(m.pN)
        s.solve(x, np.ascontiguousarray(b[inv][r0:r1]))
        out = s.counters().copy(), s.getIters()
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    for r, (cnt, its) in enumerate(res):
        xs_, ys_, zs_, xm, ym, zm = lay.boxes[r]
        faces = ym * zm + xm * zm + xm * ym          # one inner face per direction on a (2,2,2) grid
        assert int(cnt[7]) == 8 * faces * int(cnt[3]), (r, cnt)


@pytest.mark.parametrize("P,n,kind,order,window_mb", [(4, (16, 12, 12), "poisson", None, None), (8, (16, 16, 16), "poisson", None, None),
                                                      (4, (10, 9, 8), "velocity", None, None), (8, (32, 24), "poisson", None, None),
                                                      # windows of 16384 doubles: the rows travel in several rounds (host-ordered
                                                      # exchange: a window's worth of every rank's stream at a time), and the
                                                      # device-ordered flavour falls back to that path for what does not fit a half
                                                      (4, (24, 24, 24), "poisson", "host", "0.125"), (4, (24, 24, 24), "poisson", "device", "0.125")])
def test_dmda_boxes_across_processes(tmp_path, monkeypatch, P, n, kind, order, window_mb):
    """The same through the peer transport: one PROCESS per rank (HIP-IPC windows), P = 4 and 8 on the one GPU -- the general
    exchange a window's worth at a time, the index lists and the rows travelling at set-up, b / x per solve."""
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_peer_transport import run_ranks
    dt = 0.01
    if kind == "poisson":
        m, A, xs, b = _poisson(n, dt)
        L = dmda.dmda_layout(m, P)
        new, offs = L.pressure.petsc_of_natural, L.pressure.offsets
        cfg, name = _cfg("AMG", tol=1e-11, extra="pib_agglomerate_below=100\n"), "poisson"
    else:
        m = omesh.create_mesh(_stretched(n))
        A = oops.create_velocity_operator(oops.create_laplacian(m), dt, 0.005)
        xs = np.random.default_rng(5).uniform(-1, 1, m.UN)
        b = clib.spmv(A, xs)
        L = dmda.dmda_layout(m, P)
        new, offs = L.packed_of_natural, L.packed_offsets
        cfg = ("config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=500\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
               "solv:tolerance=1e-11\nsolv:norm=L2\nsolv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=1.0\n"
               "pib_initial_guess_nonzero=0\n")
        name = "velocity"
    Ap, parts = _global_permuted(A, new, offs, P)
    inv = np.empty(A.n_rows, dtype=np.int64)
    inv[new] = np.arange(A.n_rows)
    job = dict(kind="boxes", name=name, cfg=cfg, n_global=A.n_rows, offsets=offs, xs=xs[inv], b=b[inv],
               parts=[(p.rowptr, p.col, p.val) for p in parts])
    if window_mb:
        monkeypatch.setenv("PIB_PEER_WINDOW_MB", window_mb)
    res = run_ranks(str(tmp_path), job, P, order=order)
    assert np.array_equal(np.concatenate([r["y"] for r in res]), clib.spmv(Ap, xs[inv]))
    assert len({int(r["its"]) for r in res}) == 1 and all(int(r["reason"]) > 0 for r in res)
    assert all(int(r["counters"][5]) == P for r in res)
    x = np.empty(A.n_rows)
    x[inv] = np.concatenate([r["x"] for r in res])
    s1 = LinSolverHIP(name, config_text=cfg + "pib_matrix_free_velocity=0\n")
    s1.setMatrix(A)
    x1 = np.zeros(A.n_rows)
    s1.solve(x1, b)
    assert abs(int(res[0]["its"]) - s1.getIters()) <= 1
    s1.destroy()
    if kind == "poisson":
        assert all(bool(r["detected"]) for r in res)
        x, x1 = x - x.mean(), x1 - x1.mean()
    assert np.linalg.norm(x - x1) <= 1e-8 * np.linalg.norm(x1)


def test_256_cubed_on_8_petsc_decide_boxes():
    """BASELINE config 2's mesh on config 3's rank count, the way an unchanged PetIBM would hand it over: the 256^3 cavity
    Poisson operator on the (2,2,2) boxes PETSC_DECIDE picks for 8 ranks, every rank's DMDA-ordered rows through
    pib_set_csr_i32 only, the bench's solver file (multigrid-PCG V(2,2), rtol 1e-10), vectors resident in HBM.  Bars: the
    structure is recovered on every rank, the single-rank iteration count (11), the residual contract recomputed with the box
    CSR, and the per-solve budget: the slab solve's exchanges plus the three moves of b / x."""
    import bench
    from petibm_amd.linsolver import LinSolverHIP
    P, n, dt = 8, 256, 1e-3
    w = np.full(n, 1.0 / n)
    A = oops.CSR.from_csr32(*clib.assemble_poisson32([n, n, n], [w, w, w], dt))

    from types import SimpleNamespace
    counts = np.array([[n - 1, n, n], [n, n - 1, n], [n, n, n - 1], [n, n, n], [n + 1] * 3])
    L = dmda.dmda_layout(SimpleNamespace(dim=3, n=counts), P)
    assert L.grid == (2, 2, 2)
    lay = L.pressure
    parts = [dmda.permuted_local_rows(A, lay.petsc_of_natural, lay.offsets, r)[0] for r in range(P)]
    inv = np.empty(A.n_rows, dtype=np.int64)
    inv[lay.petsc_of_natural] = np.arange(A.n_rows)
    xs = bench.manufactured_solution(n, 0, n)
    b_p = clib.spmv(A, xs)[inv]
    cfg = bench.solver_config("gmg", 1e-10, 200, 0.9, 2, 2, "jacobi") + "\n"

    def rank_fn(r, uid):
        s = LinSolverHIP("poisson", config_text=cfg, rank=r, nranks=P, uid=uid, device=0)
        r0, r1 = int(lay.offsets[r]), int(lay.offsets[r + 1])
        loc = parts[r]
        s.setMatrix(oops.CSR(loc.n_rows, loc.n_cols, loc.rowptr.astype(np.int32), loc.col.astype(np.int32), loc.val), row0=r0, n_global=A.n_rows)
        st = s.gridStructure()
        x_d, b_d, r_d = s.deviceVec(r1 - r0), s.deviceVec(r1 - r0), s.deviceVec(r1 - r0)
        b_d.upload(np.ascontiguousarray(b_p[r0:r1]))
        s.solve(x_d, b_d)
        cnt = s.counters().copy()
        s.matMult(x_d, r_d)
        bl = b_d.download()
        rl = bl - r_d.download()
        out = s.getIters(), float(rl @ rl), float(bl @ bl), cnt, st
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    assert {q[0] for q in res} == {11}
    assert np.sqrt(sum(q[1] for q in res) / sum(q[2] for q in res)) <= 1.5e-10
    for q in res:
        assert q[4] is not None and q[4]["detected"] and tuple(q[4]["n"]) == (n, n, n)
        its, pc, exch = q[0], int(q[3][1]), int(q[3][3])
        # (pc counts the enqueued V-cycles: a first solve over-enqueues a few guarded no-ops) the slab solve's exchanges + b in, x out
        assert its + 1 <= pc <= its + 8 and exch <= 6 * pc + 2 + 2


@pytest.mark.parametrize("P,n,periodic", [(2, (12, 10, 12), None), (3, (10, 9, 13), None), (2, (12, 11, 12), (True, False, False)),
                                          (3, (14, 15), None), (2, (128, 10, 16), None)])
def test_velocity_rows_in_packed_slabs_get_the_matrix_free_products(P, n, periodic):
    """vSolver->setMatrix(A) from a (1,1,P) process grid -- what PETSC_DECIDE gives an unchanged PetIBM on TWO ranks, and any
    P once `nProc` is set -- hands every rank [u-slab | v-slab | w-slab] with the DMDA's own split of every component's
    planes (nz - 1 planes of w over P ranks are not the pressure split).  The structure is recovered from the entries on all
    ranks together (structure.cpp: detect_velocity_structure_slabs) and verified against the CSR, so the Krylov products run
    matrix-free as they do for rows assembled on the device: same iteration counts and solutions as with the CSR products."""
    from petibm_amd.linsolver import LinSolverHIP
    dt, cnu = 0.01, 0.5 * 0.01
    m = omesh.create_mesh(_stretched(n, periodic))
    V = oops.create_velocity_operator(oops.create_laplacian(m), dt, cnu)
    grid = (1, 1, P) if len(n) == 3 else (1, P)
    L = dmda.dmda_layout(m, P, grid)
    Vp, parts = _global_permuted(V, L.packed_of_natural, L.packed_offsets, P)
    inv = np.empty(m.UN, dtype=np.int64)
    inv[L.packed_of_natural] = np.arange(m.UN)
    xs = np.random.default_rng(5).uniform(-1, 1, m.UN)
    b = clib.spmv(V, xs)
    xs_p, b_p = xs[inv], b[inv]
    y_ref = clib.spmv(Vp, xs_p)
    base = ("config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=500\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
            "solv:tolerance=1e-11\nsolv:norm=L2\nsolv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=1.0\n"
            "pib_initial_guess_nonzero=0\npib_march_min_cells=0\n")

    def run(extra):
        def rank_fn(r, uid):
            s = LinSolverHIP("velocity", config_text=base + extra, rank=r, nranks=P, uid=uid, device=0)
            r0, r1 = int(L.packed_offsets[r]), int(L.packed_offsets[r + 1])
            loc = parts[r]
            s.setMatrix(oops.CSR(loc.n_rows, loc.n_cols, loc.rowptr.astype(np.int32), loc.col.astype(np.int32), loc.val), row0=r0, n_global=m.UN)
            sv = s.velocityStructure()
            y = np.empty(r1 - r0)
            s.matMult(np.ascontiguousarray(xs_p[r0:r1]), y)
            x = np.zeros(r1 - r0)
            s.solve(x, np.ascontiguousarray(b_p[r0:r1]))
            out = y, x, s.getIters(), s.getReason(), sv
            s.destroy()
            return out
        return _run_ranks(P, rank_fn)

    free, csr = run(""), run("pib_matrix_free_velocity=0\n")
    for q in free:
        assert q[4] is not None and q[4]["detected"] and q[4]["dim"] == m.dim and q[3] > 0
        assert tuple(q[4]["periodic"][: m.dim]) == tuple(bool(v) for v in (periodic or (False,) * m.dim))
    assert all(q[4] is None for q in csr)
    assert np.array_equal(np.concatenate([q[0] for q in free]), y_ref)     # pib_mat_mult is the CSR product either way
    xf, xc = np.concatenate([q[1] for q in free]), np.concatenate([q[1] for q in csr])
    # (the marching product sums BiCGStab's dot products by tile: ~115 iterations on the 128-wide mesh move by a few)
    assert abs(free[0][2] - csr[0][2]) <= max(1, csr[0][2] // 20) and len({q[2] for q in free}) == 1
    assert np.abs(xf - xc).max() <= 1e-10 * np.abs(xc).max()
    x = np.empty(m.UN)
    x[inv] = xf
    assert np.linalg.norm(b - clib.spmv(V, x)) <= 2e-11 * np.sqrt(m.UN)
