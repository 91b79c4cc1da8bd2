"""TEST SUPPORT (moved out of the product package in round 4): host-side z-slab partition and halo plan (one rank per GPU).

The same rule the C library applies (pib_slab_range / assemble.hip:slab_range):
PETSc's DMDA default split of N planes over P ranks, m_r = N/P + ((N % P) > r)
(what DMDACreate3d gives the reference, src/mesh/cartesianmesh.cpp:492-538, when
the process grid is (1,1,P)).  In natural ordering i + nx*(j + ny*k) a rank's
rows are contiguous and its ghosts are the `plane` entries just below its first
row and just above its last row.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple


def slab_range(nplanes: int, nranks: int, rank: int) -> Tuple[int, int]:
    b = 0
    for r in range(rank):
        b += nplanes // nranks + (1 if (nplanes % nranks) > r else 0)
    return b, b + nplanes // nranks + (1 if (nplanes % nranks) > rank else 0)


@dataclass
class SlabPlan:
    rank: int
    nranks: int
    n: Tuple[int, ...]        # global cells per direction
    plane: int                # entries per plane of the slab axis
    k0: int
    k1: int
    row0: int                 # first global row
    n_local: int
    ghost_lo: int             # entries received from rank-1
    ghost_hi: int             # entries received from rank+1
    send_prev: int            # entries sent to rank-1 (its ghost_hi)
    send_next: int            # entries sent to rank+1 (its ghost_lo)

    def local_col(self, global_col):
        """local (ghost-shifted) column index used on the device"""
        return global_col - (self.row0 - self.ghost_lo)


def slab_plan(n, nranks: int, rank: int) -> SlabPlan:
    n = tuple(int(v) for v in n)
    nlast = n[-1]
    plane = 1
    for v in n[:-1]:
        plane *= v
    k0, k1 = slab_range(nlast, nranks, rank)
    lo = plane if rank > 0 else 0
    hi = plane if rank < nranks - 1 else 0
    return SlabPlan(rank, nranks, n, plane, k0, k1, k0 * plane, (k1 - k0) * plane, lo, hi, lo, hi)


def all_plans(n, nranks: int) -> List[SlabPlan]:
    return [slab_plan(n, nranks, r) for r in range(nranks)]
