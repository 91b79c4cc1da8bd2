"""DMDA decomposition and orderings of petibm::mesh::CartesianMesh on P ranks -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(petibm_amd/) never does.

What an UNCHANGED PetIBM hands `vSolver->setMatrix(A)` / `pSolver->setMatrix(DBNG)`
(applications/navierstokes/navierstokes.cpp:163-164) on P > 1 ranks is decided by three things restated here:

 * the process grid.  CartesianMesh creates its DMDAs with `nProc = PETSC_DECIDE` (src/mesh/cartesianmesh.cpp:97,
   503-519) and never calls DMSetFromOptions, so PETSc's own rule picks (m, n, p): DMSetUp_DA_2D / DMSetUp_DA_3D
   (PETSc 3.16.x, src/dm/impls/da/da2.c, da3.c).  PETSc is a third-party dependency absent from /root/reference; the
   rule is restated from its published source.  PARITY UNPINNED: the reference holds no vector for the grid PETSc
   picks -- the tests therefore ALSO run explicit process grids, which do not depend on this restatement.
   The pressure DMDA is created first and its grid is read back (`DMDAGetInfo`, cartesianmesh.cpp:551-553); the
   velocity DMDAs reuse that grid with their own point counts.
 * ownership.  With lx = ly = lz = nullptr rank (px, py, pz) owns M/m + ((M % m) > px) points along x, etc.
 * the orderings.  "PETSc ordering" of one DMDA: ranks one after the other, rank = px + m (py + n pz), every rank's
   box in its own natural order i + xm (j + ym k) -- what DMDAGetAO + AOApplicationToPetsc give getGlobalIndex
   (cartesianmesh.cpp:700-738).  The velocity unknowns live in a DMComposite: per rank [u box | v box | w box]
   (getPackedGlobalIndex, cartesianmesh.cpp:741-779).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from .operators import CSR


def decide_process_grid(dims: Sequence[int], size: int) -> Tuple[int, int, int]:
    """(m, n, p) PETSc picks for PETSC_DECIDE in every direction (da2.c / da3.c, 'try for squarish distribution')."""
    if len(dims) == 2:
        M, N = (int(v) for v in dims)
        m = int(0.5 + np.sqrt(float(M) * float(size) / float(N)))
        if m == 0:
            m = 1
        n = 1
        while m > 0:
            n = size // m
            if m * n == size:
                break
            m -= 1
        if M > N and m < n:
            m, n = n, m
        return m, n, 1
    M, N, P = (int(v) for v in dims)
    n = int(0.5 + (float(N) * float(N) * float(size) / (float(P) * float(M))) ** (1.0 / 3.0))
    if n == 0:
        n = 1
    while n > 0:
        pm = size // n
        if n * pm == size:
            break
        n -= 1
    if n == 0:
        n = 1
    m = int(0.5 + np.sqrt(float(M) * float(size) / (float(P) * float(n))))
    if m == 0:
        m = 1
    p = 1
    while m > 0:
        p = size // (m * n)
        if m * n * p == size:
            break
        m -= 1
    if M > P and m < p:
        m, p = p, m
    return m, n, p


def ownership(npoints: int, nprocs: int) -> List[Tuple[int, int]]:
    """[(start, count)] per process along one direction: the DMDA default (lx == nullptr)."""
    out, b = [], 0
    for r in range(nprocs):
        c = npoints // nprocs + (1 if (npoints % nprocs) > r else 0)
        out.append((b, c))
        b += c
    return out


@dataclass
class FieldLayout:
    """one DMDA: point counts n (nx, ny, nz), process grid, boxes and the natural -> PETSc permutation"""
    n: Tuple[int, int, int]
    grid: Tuple[int, int, int]
    boxes: List[Tuple[int, int, int, int, int, int]]  # per rank (xs, ys, zs, xm, ym, zm)
    offsets: np.ndarray                               # first PETSc index of every rank (+ total at the end)
    petsc_of_natural: np.ndarray                      # [n_total] PETSc index of natural index i + nx (j + ny k)
    rank_of_natural: np.ndarray


def field_layout(n: Sequence[int], grid: Sequence[int]) -> FieldLayout:
    n3 = tuple(int(v) for v in n) + (1,) * (3 - len(n))
    m, nn, p = (int(v) for v in grid)
    ox, oy, oz = ownership(n3[0], m), ownership(n3[1], nn), ownership(n3[2], p)
    boxes, counts = [], []
    for pz in range(p):
        for py in range(nn):
            for px in range(m):
                boxes.append((ox[px][0], oy[py][0], oz[pz][0], ox[px][1], oy[py][1], oz[pz][1]))
                counts.append(ox[px][1] * oy[py][1] * oz[pz][1])
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    total = n3[0] * n3[1] * n3[2]
    pet = np.empty(total, dtype=np.int64)
    rk = np.empty(total, dtype=np.int64)
    for r, (xs, ys, zs, xm, ym, zm) in enumerate(boxes):
        if xm * ym * zm == 0:
            continue
        k, j, i = np.meshgrid(np.arange(zs, zs + zm), np.arange(ys, ys + ym), np.arange(xs, xs + xm), indexing="ij")
        nat = (i + n3[0] * (j + n3[1] * k)).ravel()       # box walked in ITS natural order: i fastest
        pet[nat] = offsets[r] + np.arange(xm * ym * zm)
        rk[nat] = r
    return FieldLayout(n3, (m, nn, p), boxes, offsets, pet, rk)


@dataclass
class DMDALayout:
    """the pressure DMDA and the velocity DMComposite of a mesh on `size` ranks"""
    dim: int
    size: int
    grid: Tuple[int, int, int]
    pressure: FieldLayout
    velocity: List[FieldLayout]
    packed_offsets: np.ndarray      # first packed velocity index of every rank (+ total)
    packed_of_natural: np.ndarray   # [UN] packed global index of the single-rank packed index (oracle.mesh.packed_index)
    packed_rank: np.ndarray


def dmda_layout(mesh, size: int, grid: Sequence[int] = None) -> DMDALayout:
    """mesh: oracle.mesh.CartesianMesh.  grid None: PETSC_DECIDE from the PRESSURE point counts (createPressureDMDA
    comes first, cartesianmesh.cpp:536-553)."""
    dim = int(mesh.dim)
    pn = [int(v) for v in mesh.n[3][:dim]]
    g = tuple(int(v) for v in grid) if grid is not None else decide_process_grid(pn, size)
    g = g + (1,) * (3 - len(g))
    if g[0] * g[1] * g[2] != size:
        raise ValueError(f"process grid {g} does not have {size} ranks")
    pres = field_layout(pn, g)
    vel = [field_layout([int(v) for v in mesh.n[f][:dim]], g) for f in range(dim)]
    # DMComposite: rank r holds [u box | v box | w box]
    per_rank = np.zeros(size, dtype=np.int64)
    for f in range(dim):
        per_rank += np.diff(vel[f].offsets)
    poff = np.concatenate([[0], np.cumsum(per_rank)]).astype(np.int64)
    un = sum(int(np.prod(vel[f].n)) for f in range(dim))
    packed = np.empty(un, dtype=np.int64)
    prank = np.empty(un, dtype=np.int64)
    base = 0
    for f in range(dim):
        L = vel[f]
        nf = int(np.prod(L.n))
        r = L.rank_of_natural
        before = np.zeros(size, dtype=np.int64)           # points of the fields before f on every rank
        for e in range(f):
            before += np.diff(vel[e].offsets)
        packed[base:base + nf] = poff[r] + before[r] + (L.petsc_of_natural - L.offsets[r])
        prank[base:base + nf] = r
        base += nf
    return DMDALayout(dim, size, g, pres, vel, poff, packed, prank)


def permuted_local_rows(A: CSR, new_of_old: np.ndarray, offsets: np.ndarray, rank: int):
    """The rows rank `rank` owns of P A P^T (P: old index -> new index), as a CSR with LOCAL rows and GLOBAL (new)
    columns sorted ascending -- what MatMPIAIJGetLocalMat delivers (include/petibm_amd/petsc_adapter.hpp).
    Returns (CSR, row0)."""
    lo, hi = int(offsets[rank]), int(offsets[rank + 1])
    old_of_new = np.empty_like(new_of_old)
    old_of_new[new_of_old] = np.arange(len(new_of_old), dtype=np.int64)
    rows_old = old_of_new[lo:hi]
    lens = (A.rowptr[rows_old + 1] - A.rowptr[rows_old]).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(rp[-1])
    lrow = np.repeat(np.arange(hi - lo, dtype=np.int64), lens)
    src = np.repeat(A.rowptr[rows_old].astype(np.int64) - rp[:-1], lens) + np.arange(nnz, dtype=np.int64)
    c = new_of_old[A.col[src]]
    o = np.lexsort((c, lrow))                              # by row, then by new column
    return CSR(hi - lo, len(new_of_old), rp, c[o], A.val[src][o]), lo
