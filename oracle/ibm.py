"""Oracle restatement of the immersed-boundary operators and of one DecoupledIBPMSolver time step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows
  src/misc/delta.cpp:17-74                     regularised delta kernels (Roma et al. 1999, Peskin 2002)
  src/io/io.cpp:23-118                         Lagrangian point files (count, then one coordinate set per line)
  src/body/singlebodypoints.cpp:90-113         meshIdx = pressure cell that holds the point (upper_bound - 1)
  src/body/bodypack.cpp:261-283                packed force index: bodies one after the other, point-major, dof-minor
  src/operators/createdelta.cpp:34-208         Delta: rows = Lagrangian dofs, columns = velocity points of the same
                                               component within +-window cells of meshIdx, value = prod_d kernel(X_d - x_d, h_d),
                                               h_d = dL[u][d] at the first point's cell; zero entries are not stored
  src/operators/creatediagmatrix.cpp:88-171    R (face areas), MHat (cell width along the component)
  applications/decoupledibpm/decoupledibpm.cpp:141-216  H = Delta^T, E = Delta R MHat, BNH = BN H, EBNH = E BNH
  applications/decoupledibpm/decoupledibpm.cpp:105-131,232-313  advance(): rhs1 += H f; velocity solve;
                                               rhsf = -E u; EBNH df = rhsf; u += BNH df; Poisson; projection; f += df
  src/body/singlebodypoints.cpp:228-259        body force = -sum of the Lagrangian forces
"""
from __future__ import annotations

import math

import numpy as np

from . import clib, navierstokes as ons, operators as oops
from .mesh import CartesianMesh
from .operators import CSR


def roma_et_al_1999(r: float, dr: float) -> float:
    x = abs(r) / dr
    if x > 1.5:
        return 0.0
    if 0.5 < x <= 1.5:
        return (5 - 3 * x - math.sqrt(-3 * (1 - x) * (1 - x) + 1)) / (6 * dr)
    return (1 + math.sqrt(-3 * x * x + 1)) / (3 * dr)


def peskin_2002(r: float, dr: float) -> float:
    x = abs(r) / dr
    if 0.0 <= x <= 1.0:
        return (3 - 2 * x + math.sqrt(1 + 4 * x - 4 * x * x)) / (8 * dr)
    if 1.0 <= x <= 2.0:
        return (5 - 2 * x - math.sqrt(-7 + 12 * x - 4 * x * x)) / (8 * dr)
    return 0.0


KERNELS = {"ROMA_ET_AL_1999": (roma_et_al_1999, 2), "PESKIN_2002": (peskin_2002, 3)}


def get_kernel(name: str):
    if name not in KERNELS:
        raise ValueError(f"No support for delta kernel `{name}`.")
    return KERNELS[name]


def read_lagrangian_points(path: str) -> np.ndarray:
    with open(path) as fh:
        first = fh.readline().split()
        if len(first) != 1:
            raise ValueError(f"The first line in file {path} contains more than one integer.")
        n = int(first[0])
        pts = [[float(v) for v in line.split()] for line in fh if line.strip()]
    if len(pts) != n:
        raise ValueError(f"The number of coordinate sets in {path} does not match the header.")
    return np.array(pts)


def mesh_index(mesh: CartesianMesh, coords: np.ndarray) -> np.ndarray:
    """updateMeshIdx: the pressure cell (vertex interval) that contains each point"""
    idx = np.zeros(coords.shape, dtype=np.int64)
    for d in range(mesh.dim):
        v = mesh.coord[4][d].true
        if np.any(coords[:, d] <= mesh.min[d]) or np.any(coords[:, d] >= mesh.max[d]):
            raise ValueError("body coordinate is outside the domain")
        idx[:, d] = np.searchsorted(v, coords[:, d], side="right") - 1
    return idx


def create_delta(mesh: CartesianMesh, bodies, kernel_name: str = "ROMA_ET_AL_1999") -> CSR:
    """bodies: list of (npts, dim) coordinate arrays"""
    kernel, window = get_kernel(kernel_name)
    dim = mesh.dim
    offs = [0]
    for f in range(dim):
        offs.append(offs[-1] + int(np.prod(mesh.n[f])))
    rowptr, cols, vals = [0], [], []
    for body in bodies:
        midx = mesh_index(mesh, body)
        widths = [mesh.dL[0][d][int(midx[0][d])] for d in range(dim)]
        for pt in range(body.shape[0]):
            for dof in range(dim):
                nn = [int(v) for v in mesh.n[dof]]
                ids, phis = [], []
                for d in range(dim):
                    s_list, p_list = [], []
                    for s in range(int(midx[pt][d]) - window, int(midx[pt][d]) + window + 1):
                        # a window point outside the field contributes nothing, also on a periodic direction: the
                        # reference wraps the index but evaluates the kernel a domain length away (createdelta.cpp:
                        # 188-203: coord[s] -+ L), which is zero and not stored
                        if 0 <= s < nn[d]:
                            s_list.append(s)
                            p_list.append(kernel(body[pt][d] - mesh.coord[dof][d][s], widths[d]))
                    ids.append(s_list)
                    phis.append(p_list)
                if dim == 2:
                    ids.append([0])
                    phis.append([None])
                for kk, k in enumerate(ids[2]):
                    for jj, j in enumerate(ids[1]):
                        for ii, i in enumerate(ids[0]):
                            # delta.cpp:65-72: phi = 1; phi *= kernel_d for d = 0, 1, (2)
                            v = 1.0
                            v *= phis[0][ii]
                            v *= phis[1][jj]
                            if dim == 3:
                                v *= phis[2][kk]
                            if v != 0.0:  # MAT_IGNORE_ZERO_ENTRIES
                                cols.append(offs[dof] + i + nn[0] * (j + nn[1] * k))
                                vals.append(v)
                rowptr.append(len(cols))
    return CSR(len(rowptr) - 1, offs[-1], np.array(rowptr, dtype=np.int64), np.array(cols, dtype=np.int64),
               np.array(vals, dtype=np.float64))


def diag_r_mhat(mesh: CartesianMesh):
    """diagonals of createR and createMHead in packed velocity order"""
    R, M = [], []
    for f in range(mesh.dim):
        n0, n1, n2 = (int(v) for v in mesh.n[f])
        k, j, i = np.meshgrid(np.arange(n2), np.arange(n1), np.arange(n0), indexing="ij")
        ijk = [i.ravel(), j.ravel(), k.ravel()]
        others = [d for d in range(3) if d != f]
        R.append(mesh.dL[f][others[0]][ijk[others[0]]] * mesh.dL[f][others[1]][ijk[others[1]]])
        M.append(mesh.dL[f][f][ijk[f]] + 0.0)
    return np.concatenate(R), np.concatenate(M)


def transpose(a: CSR) -> CSR:
    """MatTranspose: rows of the result sorted by (column of a, row of a)"""
    rows = np.repeat(np.arange(a.n_rows), np.diff(a.rowptr))
    order = np.lexsort((rows, a.col))
    cnt = np.bincount(a.col, minlength=a.n_cols)
    rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    return CSR(a.n_cols, a.n_rows, rp, rows[order].astype(np.int64), a.val[order].copy())


def create_ib_operators(mesh: CartesianMesh, bodies, dt: float, kernel_name: str = "ROMA_ET_AL_1999", BN: CSR = None):
    """decoupledibpm.cpp:141-216; BN: createBnHead(L, dt, c nu, N) for parameters.BN = N > 1 (:194-197), dt I otherwise"""
    delta = create_delta(mesh, bodies, kernel_name)
    H = transpose(delta)
    R, M = diag_r_mhat(mesh)
    E = delta.copy()
    E.val = E.val * R[E.col]  # MatDiagonalScale(E, nullptr, RDiag)
    E.val = E.val * M[E.col]  # MatDiagonalScale(E, nullptr, MHatDiag)
    nu_ = H.n_rows
    if BN is None:
        BN = CSR(nu_, nu_, np.arange(nu_ + 1, dtype=np.int64), np.arange(nu_, dtype=np.int64), np.full(nu_, dt))
    BNH = oops.matmatmult(BN, H)
    EBNH = oops.matmatmult(E, BNH)
    return {"delta": delta, "E": E, "H": H, "BNH": BNH, "EBNH": EBNH}


def mult_add(a: CSR, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """MatMultAdd(A, x, y, z) of SeqAIJ: every row sum starts from y[i] and adds the products in column order"""
    z = y.copy()
    for r in np.flatnonzero(np.diff(a.rowptr)):
        s = z[r]
        for q in range(int(a.rowptr[r]), int(a.rowptr[r + 1])):
            s = s + a.val[q] * x[a.col[q]]
        z[r] = s
    return z


class DecoupledIBPM(ons.NavierStokes):
    """NavierStokes + immersed bodies, decoupled IBPM (Li et al. 2016): one extra SPD solve per step."""

    def __init__(self, mesh: CartesianMesh, dt: float, nu: float, bodies, kernel_name="ROMA_ET_AL_1999", **kw):
        super().__init__(mesh, dt, nu, **kw)
        self.bodies = [np.asarray(b, dtype=np.float64) for b in bodies]
        bn_order = int(kw.get("bn_order", 1))
        self.BN = oops.create_bn_head(self.L, dt, self.cimpl * nu, bn_order) if bn_order > 1 else None
        self.ops = create_ib_operators(mesh, self.bodies, dt, kernel_name, self.BN)
        self.nf = self.ops["E"].n_rows
        self.f = np.zeros(self.nf)
        self.EBNH_dense = self.ops["EBNH"].to_dense()
        self.kernel_name = kernel_name
        self.UB = None

    def move_bodies(self, bodies, velocities=None):
        """RigidKinematicsSolver::moveBodies (applications/rigidkinematics/rigidkinematics.cpp:118-140): new
        coordinates, prescribed point velocities UB, operators rebuilt, fSolver->setMatrix(EBNH)"""
        self.bodies = [np.asarray(b, dtype=np.float64) for b in bodies]
        self.ops = create_ib_operators(self.mesh, self.bodies, self.dt, self.kernel_name, self.BN)
        self.EBNH_dense = self.ops["EBNH"].to_dense()
        if velocities is not None:
            self.UB = np.concatenate([np.asarray(v, dtype=np.float64) for v in velocities], axis=0).reshape(-1)

    def advance(self):
        rhs1 = self.rhs_velocity()
        rhs1 = mult_add(self.ops["H"], self.f, rhs1)  # MatMultAdd(H, f, rhs1, rhs1)
        self.last_rhs1 = rhs1
        r = clib.bcgs(self.A, rhs1, x0=self.U, pc="jacobi", norm="unpreconditioned", rtol=0.0, atol=self.vtol,
                      dtol=1e300, maxit=2000)
        assert r["reason"] > 0
        self.U = r["x"]
        self.info["vIters"] = r["iters"]
        rhsf = clib.spmv(self.ops["E"], self.U)
        rhsf = -1.0 * rhsf
        if self.UB is not None:
            rhsf = self.UB + 1.0 * rhsf  # VecAYPX(rhsf, 1.0, UB)  (rigidkinematics.cpp:157)
        self.last_rhsf = rhsf
        df = np.linalg.solve(self.EBNH_dense, rhsf)  # -forces_ksp_type preonly -forces_pc_type lu
        self.U = mult_add(self.ops["BNH"], df, self.U)  # MatMultAdd(BNH, df, U, U)
        rhs2 = self.rhs_poisson()
        self.last_rhs2 = rhs2
        rp = self.gmg.pcg(self.DBNG, rhs2, rtol=0.0, atol=self.ptol, maxit=500)
        assert rp["reason"] > 0, rp
        dP = rp["x"]
        if not self.pinned:
            dP = dP - dP.mean()
        self.info["pIters"] = rp["iters"]
        g = clib.spmv(self.BNG, dP)
        self.U = self.U + (-1.0) * g
        self.p = self.p + 1.0 * dP
        self.f = self.f + 1.0 * df
        ons.update_ghost_values(self.mesh, self.ghosts, self.U)

    def body_forces(self):
        """calculateAvgForces: per body, per direction, minus the sum of the Lagrangian forces"""
        out, off = [], 0
        for b in self.bodies:
            n = b.shape[0] * self.mesh.dim
            out.append(-self.f[off:off + n].reshape(-1, self.mesh.dim).sum(axis=0))
            off += n
        return np.array(out)


class CoupledIBPM(DecoupledIBPM):
    """IBPMSolver (applications/ibpm/ibpm.cpp): pressure and Lagrangian forces are one unknown of the modified Poisson
    system D_c BN G_c with G_c = [G, -H], D_c = [D; E] (:110-194); pinned pressure = MatZeroRowsColumns(row 0) of the
    stacked matrix (:264-268).  The restatement assembles the stacked matrix densely and solves it directly (small
    meshes only) -- whatever route a solver takes, this is the solution it has to reach."""

    def __init__(self, mesh, dt, nu, bodies, **kw):
        kw["pinned"] = True
        super().__init__(mesh, dt, nu, bodies, **kw)
        pN, nf = mesh.pN, self.nf
        D, BNG = self.D.to_dense(), self.BNG.to_dense()
        E, BNH = self.ops["E"].to_dense(), self.ops["BNH"].to_dense()
        M = np.zeros((pN + nf, pN + nf))
        M[:pN, :pN] = D @ BNG
        M[:pN, pN:] = -(D @ BNH)
        M[pN:, :pN] = E @ BNG
        M[pN:, pN:] = -(E @ BNH)
        M[0, :] = 0.0
        M[:, 0] = 0.0
        M[0, 0] = 1.0
        self.M = M

    def advance(self):
        rhs1 = self.rhs_velocity()
        rhs1 = mult_add(self.ops["H"], self.f, rhs1)  # -G_c [p; f] = -G p + H f
        self.last_rhs1 = rhs1
        r = clib.bcgs(self.A, rhs1, x0=self.U, pc="jacobi", norm="unpreconditioned", rtol=0.0, atol=self.vtol,
                      dtol=1e300, maxit=2000)
        assert r["reason"] > 0
        self.U = r["x"]
        r1 = self.rhs_poisson()  # D u* + bc, zero at the pinned row
        r2 = clib.spmv(self.ops["E"], self.U)
        x = np.linalg.solve(self.M, np.concatenate([r1, r2]))
        dP, df = x[: self.mesh.pN], x[self.mesh.pN:]
        self.U = self.U - clib.spmv(self.BNG, dP)
        self.U = mult_add(self.ops["BNH"], df, self.U)
        self.p = self.p + dP
        self.f = self.f + df
        ons.update_ghost_values(self.mesh, self.ghosts, self.U)

