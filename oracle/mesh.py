"""Oracle restatement of PetIBM's stretched staggered Cartesian mesh arithmetic.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, line by line:
  * src/parser/parser.cpp:298-356   parseSubDomains / parseOneSubDomain
  * include/petibm/misc.h:148-163   stretchGrid
  * src/mesh/cartesianmesh.cpp:136-355  createPressureMesh / createVertexMesh /
                                        createVelocityMesh
  * src/mesh/cartesianmesh.cpp:578-795  natural / global / packed index maps
  * src/misc/misc.cpp:129-267       getGhostPointList / getGhostTargetStencil

Field indices follow the reference: 0,1,2 = u,v,w ; 3 = pressure ; 4 = vertex.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np


class Ghosted:
    """1-D array addressable from index -1 (the reference's `GhostedVec`:
    a raw pointer to element 1 of a vector that stores one ghost each side,
    cartesianmesh.cpp:320-323)."""

    def __init__(self, true_values: Sequence[float], offset: int):
        self.true = np.asarray(true_values, dtype=np.float64)
        self.offset = int(offset)

    def __getitem__(self, i):
        if isinstance(i, slice):
            start = (i.start if i.start is not None else -self.offset) + self.offset
            stop = (i.stop + self.offset) if i.stop is not None else None
            return self.true[start:stop:i.step]
        return self.true[np.asarray(i) + self.offset]

    def __len__(self):
        return len(self.true)


def stretch_grid(bg: float, ed: float, n: int, r: float) -> np.ndarray:
    """include/petibm/misc.h:148-163 -- geometric cell sizes.
    dL[0] = (ed-bg)(r-1)/(r^n-1); dL[i] = dL[i-1]*r (sequential products, as
    the reference does, so rounding matches)."""
    dL = np.empty(n, dtype=np.float64)
    dL[0] = (ed - bg) * (r - 1.0) / (math.pow(r, float(n)) - 1.0)
    for i in range(1, n):
        dL[i] = dL[i - 1] * r
    return dL


def parse_one_subdomain(sub: dict, bg: float):
    """src/parser/parser.cpp:331-356."""
    n = int(sub["cells"])
    ed = float(sub["end"])
    r = float(sub["stretchRatio"])
    if abs(r - 1.0) <= 1e-12:
        dL = np.full(n, (ed - bg) / n, dtype=np.float64)
    else:
        dL = stretch_grid(bg, ed, n, r)
    return n, ed, dL


def parse_subdomains(subs: Sequence[dict], bg: float):
    """src/parser/parser.cpp:298-329 -- returns (nTotal, ed, dL)."""
    n_total = 0
    ed = float(bg)
    parts: List[np.ndarray] = []
    for sub in subs:
        n, ed, dL = parse_one_subdomain(sub, ed)
        n_total += n
        parts.append(dL)
    return n_total, ed, (np.concatenate(parts) if parts else np.zeros(0))


_DIR = {"x": 0, "y": 1, "z": 2}
# src/misc/type.cpp: BCLoc enum order XMINUS, XPLUS, YMINUS, YPLUS, ZMINUS, ZPLUS
BCLOC = {"xMinus": 0, "xPlus": 1, "yMinus": 2, "yPlus": 3, "zMinus": 4, "zPlus": 5}
_FIELD = {"u": 0, "v": 1, "w": 2}


@dataclass
class CartesianMesh:
    """The arithmetic half of petibm::mesh::CartesianMesh (no DMDA).

    Attributes mirror the reference's public members
    (include/petibm/mesh.h): dim, min, max, n[5][3], coord[5][3], dL[5][3],
    periodic[3][3], UN, pN.
    """

    dim: int
    min: np.ndarray
    max: np.ndarray
    n: np.ndarray  # (5,3) ints
    coord: list  # [5][3] Ghosted
    dL: list  # [5][3] Ghosted (dL[4] = None)
    periodic: np.ndarray  # (3,3) bool
    UN: int
    pN: int
    bc_types: dict = field(default_factory=dict)  # (field, loc) -> type string
    bc_values: dict = field(default_factory=dict)

    # ---- index maps, single rank (cartesianmesh.cpp:595-700) --------------
    def natural_index(self, f: int, i, j, k):
        """getNaturalIndex (cartesianmesh.cpp:595-681); -1 for a ghost on a
        non-periodic boundary, wrapped index on a periodic one.  Vectorised."""
        i = np.asarray(i, dtype=np.int64)
        j = np.asarray(j, dtype=np.int64)
        k = np.asarray(k, dtype=np.int64)
        i, j, k = np.broadcast_arrays(i, j, k)
        n0, n1, n2 = (int(v) for v in self.n[f])
        three = self.dim == 3
        ghost = np.zeros(i.shape, dtype=bool)
        ii, jj, kk = i.copy(), j.copy(), k.copy()
        # the reference tests i first, then j, then k and returns at the
        # first hit, so a corner ghost takes the x rule.
        decided = np.zeros(i.shape, dtype=bool)
        for arr, nn, per in ((ii, n0, self.periodic[0][0]),
                             (jj, n1, self.periodic[0][1]),
                             (kk, n2, self.periodic[0][2])):
            lo = (arr == -1) & ~decided
            hi = (arr == nn) & ~decided
            if per:
                arr[lo] = nn - 1
                arr[hi] = 0
            else:
                ghost |= lo | hi
            decided |= lo | hi
        idx = ii + jj * n0 + (kk * n1 * n0 if three else 0)
        idx = np.where(ghost, -1, idx)
        return idx

    def packed_index(self, f: int, i, j, k):
        """getPackedGlobalIndex (cartesianmesh.cpp:741-779) on ONE rank:
        packed = offset of the field's block + natural index; pressure
        (f == 3) is not packed."""
        idx = self.natural_index(f, i, j, k)
        if f == 3:
            return idx
        off = sum(int(np.prod(self.n[g])) for g in range(f))
        return np.where(idx < 0, -1, idx + off)

    def field_size(self, f: int) -> int:
        return int(np.prod(self.n[f]))


def check_periodic(bc_types: dict, dim: int) -> np.ndarray:
    """src/misc/misc.cpp checkPeriodicBC: a direction is periodic for every
    field iff the BCs at both ends are PERIODIC (all fields agree)."""
    per = np.zeros((3, 3), dtype=bool)
    for f in range(dim):
        for d in range(dim):
            lo = bc_types.get((f, 2 * d), "NOBC")
            hi = bc_types.get((f, 2 * d + 1), "NOBC")
            per[f][d] = (lo == "PERIODIC") and (hi == "PERIODIC")
    return per


def create_mesh(config: dict) -> CartesianMesh:
    """CartesianMesh::init (cartesianmesh.cpp:69-133) minus the DMDA part.

    `config` has the reference's YAML shape:
      mesh: [ {direction, start, subDomains:[{end,cells,stretchRatio}]} ... ]
      flow: {boundaryConditions: [ {location, u:[type,val], v:[..], w:[..]} ]}
    """
    mesh_node = config["mesh"]
    dim = len(mesh_node)
    mn = np.zeros(3)
    mx = np.ones(3)
    n = np.ones((5, 3), dtype=np.int64)
    dL3 = [np.ones(1), np.ones(1), np.ones(1)]
    for ax in mesh_node:
        d = _DIR[ax["direction"]]
        bg = float(ax["start"])
        n_tot, ed, dL = parse_subdomains(ax["subDomains"], bg)
        mn[d], mx[d], n[3][d] = bg, ed, n_tot
        dL3[d] = dL

    bc_types, bc_values = {}, {}
    for bc in config.get("flow", {}).get("boundaryConditions", []):
        loc = BCLOC[bc["location"]]
        for name, f in _FIELD.items():
            if name in bc:
                bc_types[(f, loc)] = str(bc[name][0]).upper()
                bc_values[(f, loc)] = float(bc[name][1])
    periodic = check_periodic(bc_types, dim)

    coord = [[None] * 3 for _ in range(5)]
    dLg = [[None] * 3 for _ in range(5)]

    # createPressureMesh (cartesianmesh.cpp:136-176)
    c3 = []
    for d in range(3):
        if d < dim:
            c = np.cumsum(dL3[d])  # std::partial_sum
            c = c + mn[d] - 0.5 * dL3[d]
        else:
            c = np.zeros(1)
        c3.append(c)
        coord[3][d] = Ghosted(c, 0)
        dLg[3][d] = Ghosted(dL3[d], 0)
    pN = int(n[3][0] * n[3][1] * n[3][2])

    # createVertexMesh (cartesianmesh.cpp:179-210)
    c4 = []
    for d in range(3):
        if d < dim:
            n[4][d] = n[3][d] + 1
            c = np.zeros(n[4][d])
            c[1:] = np.cumsum(dL3[d])
            c = c + mn[d]
        else:
            c = np.zeros(1)
        c4.append(c)
        coord[4][d] = Ghosted(c, 0)

    # createVelocityMesh (cartesianmesh.cpp:213-355)
    UN = 0
    for comp in range(dim):
        for d in range(dim):
            if d == comp:
                n[comp][d] = n[3][d] - 1
                cT = list(c4[d])  # coordTrue[comp][dir] = coordTrue[4][dir]
                # std::adjacent_difference with f(x,y)=0.5(x+y): first element
                # copied, then 0.5*(cur+prev); n3 entries written into a
                # vector of n+2 = n3+1 entries.
                dT = np.zeros(n[comp][d] + 2)
                dT[0] = dL3[d][0]
                dT[1:n[3][d]] = 0.5 * (dL3[d][1:] + dL3[d][:-1])
                dT = list(dT)
                if periodic[comp][d]:
                    n[comp][d] += 1
                    dT[-1] = 0.5 * (dL3[d][0] + dL3[d][-1])
                    dT[0] = dT[-1]
                    dT.append(dT[1])
                    cT.append(mx[d] + dL3[d][0])
                else:
                    dT[-1] = dL3[d][-1]
            else:
                n[comp][d] = n[3][d]
                cT = [0.0] * (n[comp][d] + 2)
                dT = [0.0] * (n[comp][d] + 2)
                cT[1:-1] = list(c3[d])
                dT[1:-1] = list(dL3[d])
                if periodic[comp][d]:
                    cT[0] = mn[d] - dL3[d][-1] / 2.0
                    cT[-1] = mx[d] + dL3[d][0] / 2.0
                    dT[0] = dL3[d][-1]
                    dT[-1] = dL3[d][0]
                else:
                    cT[0] = mn[d] - dL3[d][0] / 2.0
                    cT[-1] = mx[d] + dL3[d][-1] / 2.0
                    dT[0] = dL3[d][0]
                    dT[-1] = dL3[d][-1]
            coord[comp][d] = Ghosted(cT, 1)
            dLg[comp][d] = Ghosted(dT, 1)
        UN += int(n[comp][0] * n[comp][1] * n[comp][2])
    # 2D: z entries of the used fields point at the un-ghosted defaults
    for f in range(3):
        for d in range(3):
            if coord[f][d] is None:
                coord[f][d] = Ghosted([0.0], 0)
                dLg[f][d] = Ghosted([1.0], 0)

    return CartesianMesh(dim=dim, min=mn, max=mx, n=n, coord=coord, dL=dLg,
                         periodic=periodic, UN=UN, pN=pN,
                         bc_types=bc_types, bc_values=bc_values)


# --------------------------------------------------------------------------
# z-slab decomposition (the build's choice).  PETSc's DMDA
# gives rank r of P:  m = N/P + ((N % P) > r)  consecutive planes
# (the DMDA default ownership rule, cartesianmesh.cpp:492-538 via
# DMDACreate3d with lz = nullptr).
# --------------------------------------------------------------------------
def slab_ranges(nplanes: int, nranks: int):
    """[(begin, end)] of owned planes per rank, DMDA default split."""
    out, b = [], 0
    for r in range(nranks):
        m = nplanes // nranks + (1 if (nplanes % nranks) > r else 0)
        out.append((b, b + m))
        b += m
    return out


def uniform_config(n: Sequence[int], lo=0.0, hi=1.0, lid: float = 1.0) -> dict:
    """A lid-driven-cavity config dict of the reference's YAML shape
    (examples/navierstokes/liddrivencavity2dRe1000_GPU/config.yaml:1-30),
    all-Dirichlet walls, u = lid on yPlus."""
    dim = len(n)
    names = "xyz"
    mesh = [{"direction": names[d], "start": lo,
             "subDomains": [{"end": hi, "cells": int(n[d]), "stretchRatio": 1.0}]}
            for d in range(dim)]
    locs = ["xMinus", "xPlus", "yMinus", "yPlus", "zMinus", "zPlus"][: 2 * dim]
    bcs = []
    for loc in locs:
        bc = {"location": loc}
        for c in "uvw"[:dim]:
            bc[c] = ["DIRICHLET", 0.0]
        if loc == "yPlus":
            bc["u"] = ["DIRICHLET", lid]
        bcs.append(bc)
    return {"mesh": mesh, "flow": {"boundaryConditions": bcs}}


def periodic_config(n: Sequence[int], periodic: Sequence[bool], lo=0.0, hi=1.0, ratios=None) -> dict:
    """Config dict with PERIODIC boundaries in the flagged directions (both ends, every component --
    examples/navierstokes/taylorgreenvortex2dRe100/config.yaml:1-18) and no-slip Dirichlet walls elsewhere."""
    dim = len(n)
    names = "xyz"
    mesh = [{"direction": names[d], "start": lo,
             "subDomains": [{"end": hi, "cells": int(n[d]), "stretchRatio": float(ratios[d]) if ratios else 1.0}]}
            for d in range(dim)]
    locs = ["xMinus", "xPlus", "yMinus", "yPlus", "zMinus", "zPlus"][: 2 * dim]
    bcs = []
    for q, loc in enumerate(locs):
        bc = {"location": loc}
        for c in "uvw"[:dim]:
            bc[c] = ["PERIODIC", 0.0] if periodic[q // 2] else ["DIRICHLET", 0.0]
        bcs.append(bc)
    return {"mesh": mesh, "flow": {"boundaryConditions": bcs}}
