"""Oracle restatement of petibm::operators (D, G, L, A, BN, DBNG).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows:
  * src/operators/createdivergence.cpp:103-262  createDivergence
  * src/operators/creategradient.cpp:36-135     createGradient
  * src/operators/createlaplacian.cpp:108-263   createLaplacian (+ a0 fold)
  * src/operators/createbn.cpp:19-95            createBnHead
  * applications/navierstokes/navierstokes.cpp:317-365  A, BNG, DBNG
  * applications/navierstokes/navierstokes.cpp:395-429  setNullSpace (pin row 0)
  * src/boundary/singleboundary{dirichlet,neumann,convective}.cpp  a0 table

Matrices are CSR with sorted columns per row, like PETSc AIJ.  Single rank
(packed ordering = [u-block, v-block, w-block], cartesianmesh.cpp:741-779).
Floating-point evaluation order is kept as in the reference so that matrix
entries can be compared bit-for-bit with the device assembly.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .mesh import CartesianMesh


@dataclass
class CSR:
    n_rows: int
    n_cols: int
    rowptr: np.ndarray  # int64, n_rows+1
    col: np.ndarray  # int64
    val: np.ndarray  # float64

    @property
    def nnz(self) -> int:
        return int(self.rowptr[-1])

    @staticmethod
    def from_csr32(rowptr, col, val) -> "CSR":
        """square CSR from the int32 arrays of clib.assemble_poisson32"""
        n = len(rowptr) - 1
        return CSR(n, n, np.asarray(rowptr, dtype=np.int64), np.asarray(col, dtype=np.int64), np.asarray(val))

    def copy(self) -> "CSR":
        return CSR(self.n_rows, self.n_cols, self.rowptr.copy(), self.col.copy(), self.val.copy())

    def diagonal(self) -> np.ndarray:
        d = np.zeros(min(self.n_rows, self.n_cols))
        rows = np.repeat(np.arange(self.n_rows), np.diff(self.rowptr))
        m = rows == self.col
        d[rows[m]] = self.val[m]
        return d

    def to_dense(self) -> np.ndarray:
        a = np.zeros((self.n_rows, self.n_cols))
        rows = np.repeat(np.arange(self.n_rows), np.diff(self.rowptr))
        np.add.at(a, (rows, self.col), self.val)
        return a

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.col, self.rowptr), shape=(self.n_rows, self.n_cols))


def csr_from_coo(n_rows, n_cols, row, col, val, ignore_zero=True, drop_negative_cols=True) -> CSR:
    """MatSetValue(s) semantics of the reference's assembly: entries with a
    negative column are dropped (PETSc ignores them), exact zeros are dropped
    (MAT_IGNORE_ZERO_ENTRIES, createlaplacian.cpp:209), duplicates are summed
    IN INSERTION ORDER (ADD_VALUES), columns end up sorted per row."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float64)
    keep = np.ones(len(row), dtype=bool)
    if drop_negative_cols:
        keep &= (col >= 0) & (row >= 0)
    if ignore_zero:
        keep &= val != 0.0
    row, col, val = row[keep], col[keep], val[keep]
    order = np.lexsort((np.arange(len(row)), col, row))  # stable in insertion order
    row, col, val = row[order], col[order], val[order]
    if len(row):
        new = np.ones(len(row), dtype=bool)
        new[1:] = (row[1:] != row[:-1]) | (col[1:] != col[:-1])
        grp = np.cumsum(new) - 1
        ngrp = int(grp[-1]) + 1
        # sequential (insertion-order) sum per group: groups are tiny (<=2)
        out = np.zeros(ngrp)
        first = np.flatnonzero(new)
        out[:] = val[first]
        rest = np.flatnonzero(~new)
        for p in rest:  # rare: only BC folds create duplicates
            out[grp[p]] = out[grp[p]] + val[p]
        row, col, val = row[first], col[first], out
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, row + 1, 1)
    rowptr = np.cumsum(rowptr)
    return CSR(n_rows, n_cols, rowptr, col, val)


# ---------------------------------------------------------------------------
# a0 table: the only part of petibm::boundary that enters the matrices.
# ---------------------------------------------------------------------------
def bc_a0(bc_type: str, field: int, loc: int) -> float:
    """Ghost-point coefficient a0 (ghost = a0*target + a1):
    Dirichlet  : 0 if the boundary normal is the field's direction else -1
                 (singleboundarydirichlet.cpp:35-44)
    Neumann    : 1 (singleboundaryneumann.cpp:27-28)
    Convective : 0 same direction, -1 otherwise (singleboundaryconvective.cpp:20-36)
    Periodic   : no ghost equation (columns wrap instead)."""
    same = (loc // 2) == field
    t = bc_type.upper()
    if t == "DIRICHLET" or t == "CONVECTIVE":
        return 0.0 if same else -1.0
    if t == "NEUMANN":
        return 1.0
    if t == "PERIODIC":
        return 0.0
    raise ValueError(f"unknown BC type {bc_type}")


def _grid(mesh: CartesianMesh, f: int):
    n0, n1, n2 = (int(v) for v in mesh.n[f])
    k, j, i = np.meshgrid(np.arange(n2), np.arange(n1), np.arange(n0), indexing="ij")
    return i.ravel(), j.ravel(), k.ravel()


def _ghost_points(mesh: CartesianMesh, f: int, loc: int):
    """misc::getGhostPointList / getGhostTargetStencil (misc.cpp:129-267):
    returns (ghost (i,j,k), target (i,j,k)) arrays for boundary `loc` of field f."""
    axis = loc // 2
    n = [int(v) for v in mesh.n[f]]
    p_axes = [a for a in range(3) if a != axis]
    a, b = np.meshgrid(np.arange(n[p_axes[0]]), np.arange(n[p_axes[1]]), indexing="ij")
    a, b = a.ravel(), b.ravel()
    tgt = [None, None, None]
    gst = [None, None, None]
    tgt[p_axes[0]] = gst[p_axes[0]] = a
    tgt[p_axes[1]] = gst[p_axes[1]] = b
    if loc % 2 == 0:
        tgt[axis] = np.zeros_like(a)
        gst[axis] = np.full_like(a, -1)
    else:
        tgt[axis] = np.full_like(a, n[axis] - 1)
        gst[axis] = np.full_like(a, n[axis])
    return gst, tgt


def create_gradient(mesh: CartesianMesh, normalize: bool = False) -> CSR:
    """creategradient.cpp:36-135: row = packed velocity point, columns =
    pressure cell (i,j,k) and its + neighbour; values {-1/dL, +1/dL} with
    dL = mesh->dL[f][f][idx] (:70-86)."""
    rows, cols, vals = [], [], []
    for f in range(mesh.dim):
        i, j, k = _grid(mesh, f)
        r = mesh.packed_index(f, i, j, k)
        idx = (i, j, k)[f]
        v = np.ones(len(i)) if normalize else 1.0 / mesh.dL[f][f][idx]
        c0 = mesh.packed_index(3, i, j, k)
        nb = [i, j, k]
        nb[f] = nb[f] + 1
        c1 = mesh.packed_index(3, *nb)
        rows += [r, r]
        cols += [c0, c1]
        vals += [-v, v]
    return csr_from_coo(mesh.UN, mesh.pN, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))


def create_divergence(mesh: CartesianMesh, normalize: bool = False) -> CSR:
    """createdivergence.cpp:103-262: row = pressure cell; for each direction
    +value at the + face (velocity point with the cell's index) and -value at
    the - face; value = product of the two perpendicular cell widths
    (:140-151).  Then D[row, target] += coeff*a0 for every ghost face
    (:231-242); only Neumann-type a0 != 0 changes D."""
    i, j, k = _grid(mesh, 3)
    self_ = mesh.packed_index(3, i, j, k)
    rows, cols, vals = [], [], []
    ghost_coeff = {}
    for f in range(mesh.dim):
        if normalize:
            value = np.ones(len(i))
        elif f == 0:
            value = mesh.dL[0][1][j] * mesh.dL[0][2][k]
        elif f == 1:
            value = mesh.dL[1][0][i] * mesh.dL[1][2][k]
        else:
            value = mesh.dL[2][0][i] * mesh.dL[2][1][j]
        cp = mesh.packed_index(f, i, j, k)
        nb = [i.copy(), j.copy(), k.copy()]
        nb[f] = nb[f] - 1
        cm = mesh.packed_index(f, *nb)
        rows += [self_, self_]
        cols += [cp, cm]
        vals += [value, -value]
        ghost_coeff[f] = {}
        for idx_, c, v, pos in ((np.flatnonzero(cp < 0), cp, value, (i, j, k)),
                                (np.flatnonzero(cm < 0), cm, -value, tuple(nb))):
            for p in idx_:
                ghost_coeff[f][(int(pos[0][p]), int(pos[1][p]), int(pos[2][p]))] = (int(self_[p]), float(v[p]))
    # BC fold
    for f in range(mesh.dim):
        for loc in range(2 * mesh.dim):
            t = mesh.bc_types.get((f, loc), "NOBC")
            if t in ("NOBC", "PERIODIC"):
                continue
            a0 = bc_a0(t, f, loc)
            if a0 == 0.0:
                continue
            gst, tgt = _ghost_points(mesh, f, loc)
            tcol = mesh.packed_index(f, *tgt)
            r_add, c_add, v_add = [], [], []
            for p in range(len(tcol)):
                key = (int(gst[0][p]), int(gst[1][p]), int(gst[2][p]))
                if key in ghost_coeff[f]:
                    row, coeff = ghost_coeff[f][key]
                    r_add.append(row)
                    c_add.append(int(tcol[p]))
                    v_add.append(coeff * a0)
            rows.append(np.array(r_add, dtype=np.int64))
            cols.append(np.array(c_add, dtype=np.int64))
            vals.append(np.array(v_add, dtype=np.float64))
    return csr_from_coo(mesh.pN, mesh.UN, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))


def create_laplacian(mesh: CartesianMesh) -> CSR:
    """createlaplacian.cpp:108-263.  Per velocity point and direction:
    1/(dLNeg*dLSelf), 1/(dLPos*dLSelf) with dLSelf = dL[f][dir][self] and
    dLNeg/dLPos = coordinate differences incl. ghost coordinates (:134-148);
    diagonal = -sum in stencil order (std::accumulate from 0.0, :151); ghost
    columns dropped; then L[row, target] += coeff*a0 (:232-243)."""
    rows, cols, vals = [], [], []
    for f in range(mesh.dim):
        i, j, k = _grid(mesh, f)
        self_ = mesh.packed_index(f, i, j, k)
        ijk = (i, j, k)
        off_vals = []
        off_cols = []
        for d in range(mesh.dim):
            s = ijk[d]
            dLSelf = mesh.dL[f][d][s]
            dLNeg = mesh.coord[f][d][s] - mesh.coord[f][d][s - 1]
            dLPos = mesh.coord[f][d][s + 1] - mesh.coord[f][d][s]
            vneg = 1.0 / (dLNeg * dLSelf)
            vpos = 1.0 / (dLPos * dLSelf)
            nb_m = [i.copy(), j.copy(), k.copy()]
            nb_m[d] = nb_m[d] - 1
            nb_p = [i.copy(), j.copy(), k.copy()]
            nb_p[d] = nb_p[d] + 1
            off_vals += [vneg, vpos]
            off_cols += [mesh.packed_index(f, *nb_m), mesh.packed_index(f, *nb_p)]
        # diag = -accumulate(values[1:], 0.0) in stencil order x-,x+,y-,y+,z-,z+
        acc = np.zeros(len(i))
        for v in off_vals:
            acc = acc + v
        diag = -acc
        rows.append(self_)
        cols.append(self_)
        vals.append(diag)
        for c, v in zip(off_cols, off_vals):
            rows.append(self_)
            cols.append(c)
            vals.append(v)
        # BC fold: every ghost neighbour of a row folds onto `target` (== the
        # row's own point for a 5/7-point stencil) with coeff*a0, ADD_VALUES.
        for d in range(mesh.dim):
            for side, c, v in ((0, off_cols[2 * d], off_vals[2 * d]), (1, off_cols[2 * d + 1], off_vals[2 * d + 1])):
                loc = 2 * d + side
                t = mesh.bc_types.get((f, loc), "NOBC")
                if t in ("NOBC", "PERIODIC"):
                    continue
                a0 = bc_a0(t, f, loc)
                g = np.flatnonzero(c < 0)
                if len(g) == 0:
                    continue
                # only ghosts that really sit on boundary `loc`
                onb = (ijk[d][g] == 0) if side == 0 else (ijk[d][g] == mesh.n[f][d] - 1)
                g = g[onb]
                rows.append(self_[g])
                cols.append(self_[g])  # targetPackedId == the row itself
                vals.append(v[g] * a0)
    return csr_from_coo(mesh.UN, mesh.UN, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))


# ---------------------------------------------------------------------------
# products, A, BN, DBNG  (C kernels for the sparse products)
# ---------------------------------------------------------------------------
def matmatmult(a: CSR, b: CSR) -> CSR:
    """C = A*B the way PETSc's SeqAIJ MatMatMult does it numerically: row by
    row, walking A's row in column order and accumulating a*B[k,:] into a
    sparse accumulator; result columns sorted."""
    from . import clib
    return clib.spgemm(a, b)


def scale_shift(m: CSR, scale: float, shift: float) -> CSR:
    """MatScale(m, scale); MatShift(m, shift)  (navierstokes.cpp:343-344).
    MatShift adds to the existing diagonal (inserting one if absent)."""
    out = m.copy()
    out.val = out.val * scale
    rows = np.repeat(np.arange(m.n_rows), np.diff(m.rowptr))
    isd = rows == out.col
    has = np.zeros(m.n_rows, dtype=bool)
    has[rows[isd]] = True
    if not has.all():
        miss = np.flatnonzero(~has)
        r = np.concatenate([rows, miss])
        c = np.concatenate([out.col, miss])
        v = np.concatenate([out.val, np.zeros(len(miss))])
        out = csr_from_coo(m.n_rows, m.n_cols, r, c, v, ignore_zero=False)
        rows = np.repeat(np.arange(m.n_rows), np.diff(out.rowptr))
        isd = rows == out.col
    out.val[isd] = out.val[isd] + shift
    return out


def create_velocity_operator(L: CSR, dt: float, coeff_nu: float) -> CSR:
    """A = I/dt - c*nu*L  (navierstokes.cpp:342-344): MatDuplicate(L),
    MatScale(A, -c*nu), MatShift(A, 1/dt)."""
    return scale_shift(L, -coeff_nu, 1.0 / dt)


def create_bn_head(op: CSR, dt: float, coeff: float, n: int) -> CSR:
    """createbn.cpp:19-95: BnHead = sum_{k=1..n} dt^k coeff^(k-1) Op^(k-1);
    n == 1 gives dt*I (:49,:53)."""
    if n < 1:
        raise ValueError("The order of Bn can not be smaller than 1.")
    nr = op.n_rows
    eye = CSR(nr, nr, np.arange(nr + 1, dtype=np.int64), np.arange(nr, dtype=np.int64), np.full(nr, dt))
    if n == 1:
        return eye
    from . import clib
    bn = eye
    for term in range(2, n + 1):
        right = op.copy()
        for _ in range(2, term):
            right = matmatmult(op, right)
        a = (dt ** term) * (coeff ** (term - 1))
        bn = clib.axpy_pattern(bn, a, right)  # MatAXPY DIFFERENT_NONZERO_PATTERN
    return bn


def create_poisson_operator(D: CSR, G: CSR, L: CSR, dt: float, coeff_nu: float, bn_order: int = 1):
    """BNG = BN*G ; DBNG = D*BNG (navierstokes.cpp:349-356)."""
    BN = create_bn_head(L, dt, coeff_nu, bn_order)
    BNG = matmatmult(BN, G)
    DBNG = matmatmult(D, BNG)
    return BNG, DBNG


def pin_row0(m: CSR, diag: float = 1.0) -> CSR:
    """MatZeroRowsColumns(DBNG, {0}, diag=1.0) (navierstokes.cpp:414-420):
    row 0 and column 0 zeroed, entry (0,0) = diag.  PETSc keeps the nonzero
    pattern (explicit zeros stay in the AIJ structure)."""
    out = m.copy()
    rows = np.repeat(np.arange(m.n_rows), np.diff(m.rowptr))
    kill = (rows == 0) | (out.col == 0)
    out.val[kill] = 0.0
    d = (rows == 0) & (out.col == 0)
    out.val[d] = diag
    return out


def poisson_coefficients(mesh: CartesianMesh, dt: float):
    """Separable face coefficients of DBNG for BN order 1:
    off-diagonal toward +d of cell (i,j,k) = area_perp * (dt * (1/dLvel_d)),
    evaluated in the same order as D*(dt*G).  Returns per direction the 1-D
    arrays (inverse velocity-cell width between cell s and s+1, the
    perpendicular width arrays) that the structured path consumes."""
    out = []
    for d in range(mesh.dim):
        n = int(mesh.n[3][d])
        idx = np.arange(n - 1)
        inv = 1.0 / mesh.dL[d][d][idx]  # G value for face between s and s+1
        out.append(inv)
    return out
