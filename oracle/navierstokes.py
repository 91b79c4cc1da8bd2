"""Oracle restatement of one NavierStokesSolver time step (Perot fractional step).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows applications/navierstokes/navierstokes.cpp:
  :240-266  advance()              rhs1 -> velocity solve -> rhs2 -> Poisson solve -> project -> p += dP
                                    -> bc->updateGhostValues
  :432-521  assembleRHSVelocity()   rhs1 = -G p + u/dt + sum c_i conv_i + sum d_i diff_i + c nu Lbc
  :540-563  assembleRHSPoisson()    rhs2 = D u* + Dbc (rhs2[0] = 0 when the pressure is pinned)
  :583-615  applyDivergenceFreeVelocity / updatePressure
and src/operators/createconvection.cpp:40-339 (matrix-free convective term N(u) with ghost values),
src/operators/createlaplacian.cpp:45-78 / createdivergence.cpp:45-78 (BC correction shells: sum coeff*a1),
src/boundary/singleboundary{dirichlet,neumann}.cpp (ghost = a0*target + a1),
include/petibm/timeintegration.h:107-166 (AB2 {1.5,-0.5}; CN implicit 0.5, explicit {0.5}).

Ghost points carry per-point state (a0, a1, value) as in the reference: Dirichlet / Neumann equations are
constant, the convective outlet (singleboundaryconvective.cpp) updates a1 every step.
Every vector operation is done in the reference's order (VecScale / VecAXPY sequence) so that the device
engine can be compared bit-for-bit on the explicit parts.
"""
from __future__ import annotations

import numpy as np

from . import clib, operators as oops
from .mesh import CartesianMesh


def _face_slices(shape, loc):
    """numpy index tuples (ghost layer, target layer) of boundary `loc` on a (k, j, i) array; the target layer of an
    un-padded field array is `_target_slice`."""
    ax = 2 - loc // 2
    sl = [slice(None)] * 3
    sl[ax] = 0 if loc % 2 == 0 else shape[ax] - 1
    return tuple(sl)


class GhostPoints:
    """The ghost points of one (field, boundary): the reference's GhostPointInfo list (include/petibm/type.h,
    misc.cpp:150-196) as arrays over the face, indexed (b, a) with the perpendicular axes in natural order.
    ghost = a0 * target + a1; `value` is the ghost value seen by the convective operator
    (singleboundarybase.cpp:146-182)."""

    def __init__(self, mesh: CartesianMesh, f: int, loc: int):
        self.f, self.loc = f, loc
        self.type = mesh.bc_types.get((f, loc), "NOBC")
        self.bc_value = mesh.bc_values.get((f, loc), 0.0)
        self.axis = loc // 2
        self.normal = 1.0 if loc % 2 == 1 else -1.0
        n = int(mesh.n[f][self.axis])
        c = mesh.coord[f][self.axis]
        self.dL = (c[n] - c[n - 1]) if loc % 2 == 1 else (c[0] - c[-1])  # misc.cpp:187-190
        self.same = self.axis == f
        self.a0 = oops.bc_a0(self.type, f, loc)
        n0, n1, n2 = (int(v) for v in mesh.n[f])
        shape = [n2, n1, n0]
        del shape[2 - self.axis]
        self.a1 = np.zeros(shape)
        self.value = np.zeros(shape)

    def target(self, field_array: np.ndarray) -> np.ndarray:
        return field_array[_face_slices(field_array.shape, self.loc)]

    def _convective(self, target, dt):
        # singleboundaryconvective.cpp:12-38
        if self.same:
            return self.value - self.normal * dt * self.bc_value * (self.value - target) / self.dL
        return self.value + target - 2.0 * self.normal * dt * self.bc_value * (self.value - target) / self.dL

    def set_ics(self, field_array):
        """setGhostICsKernel: singleboundarydirichlet.cpp:22-46, singleboundaryneumann.cpp:22-33,
        singleboundaryconvective.cpp:69-85"""
        t = self.target(field_array)
        if self.type == "DIRICHLET":
            self.a1[...] = self.bc_value if self.same else 2.0 * self.bc_value
            self.value = self.a0 * t + self.a1
        elif self.type == "NEUMANN":
            self.a1[...] = self.normal * self.dL * self.bc_value
            self.value = self.a0 * t + self.a1
        elif self.type == "CONVECTIVE":
            self.value = t.copy()
            self.a1 = self._convective(t, 0.0)
        else:
            raise ValueError(f"BC type {self.type} not restated here")

    def update_eqs(self, field_array, dt):
        """updateEqsKernel: a no-op for the time-independent types, the convective equation otherwise"""
        if self.type == "CONVECTIVE":
            self.a1 = self._convective(self.target(field_array), dt)

    def update_values(self, field_array):
        """singleboundarybase.cpp:146-164"""
        self.value = self.a0 * self.target(field_array) + self.a1


def make_ghosts(mesh: CartesianMesh):
    """a periodic boundary has no ghost points (singleboundaryperiodic.cpp: every kernel is a no-op; the neighbour
    indices wrap instead, cartesianmesh.cpp:595-681)"""
    return {(f, loc): GhostPoints(mesh, f, loc) for f in range(mesh.dim) for loc in range(2 * mesh.dim)
            if not mesh.periodic[f][loc // 2]}


def _field_arrays(mesh: CartesianMesh, U: np.ndarray):
    out, off = [], 0
    for f in range(mesh.dim):
        n0, n1, n2 = (int(v) for v in mesh.n[f])
        sz = n0 * n1 * n2
        out.append(U[off:off + sz].reshape(n2, n1, n0))
        off += sz
    return out


def set_ghost_ics(mesh, ghosts, U):
    fa = _field_arrays(mesh, U)
    for (f, loc), g in ghosts.items():
        g.set_ics(fa[f])


def update_eqs(mesh, ghosts, U, dt):
    fa = _field_arrays(mesh, U)
    for (f, loc), g in ghosts.items():
        g.update_eqs(fa[f], dt)


def update_ghost_values(mesh, ghosts, U):
    fa = _field_arrays(mesh, U)
    for (f, loc), g in ghosts.items():
        g.update_values(fa[f])


def ghost_padded(mesh: CartesianMesh, U: np.ndarray, ghosts):
    """Local arrays with one ghost layer per direction (what DMCompositeScatter + copyValues2LocalVecs give
    createconvection.cpp:213-220): the stored ghost values."""
    res = []
    for f, a in enumerate(_field_arrays(mesh, U)):
        # periodic axes first: DMGlobalToLocal on the BOX-stencil DMDA wraps them, corners between two periodic axes
        # included (cartesianmesh.cpp:507-517); then a zero layer on the wall axes whose interior span receives the
        # stored ghost values.  A corner between a wall ghost and a periodic wrap is written by neither
        # (copyValues2LocalVecs only knows the ghosts facing owned points): it keeps the local vector's initial zero.
        per = [bool(mesh.periodic[f][2 - ax]) if (2 - ax) < mesh.dim else False for ax in range(3)]
        g = np.pad(a, [(1, 1) if per[ax] else (0, 0) for ax in range(3)], mode="wrap")
        g = np.pad(g, [(0, 0) if per[ax] else (1, 1) for ax in range(3)], mode="constant")
        for loc in range(2 * mesh.dim):
            if (f, loc) not in ghosts:
                continue
            ax = 2 - loc // 2  # numpy axis of (k, j, i)
            sl_g = [slice(1, -1)] * 3
            sl_g[ax] = 0 if loc % 2 == 0 else g.shape[ax] - 1
            g[tuple(sl_g)] = ghosts[(f, loc)].value
        res.append(g)
    return res


def convection(mesh: CartesianMesh, U: np.ndarray, ghosts) -> np.ndarray:
    """N(u): createconvection.cpp:40-195 kernels, vectorised; same expression order."""
    q = ghost_padded(mesh, U, ghosts)
    dim = mesh.dim
    out = []

    def sh(a, di=0, dj=0, dk=0, n=None):
        """a[k+dk, j+dj, i+di] for interior (i,j,k) of a field with shape n (k,j,i), on the padded array"""
        n2, n1, n0 = n
        return a[1 + dk:1 + dk + n2, 1 + dj:1 + dj + n1, 1 + di:1 + di + n0]

    for f in range(dim):
        n0, n1, n2 = (int(v) for v in mesh.n[f])
        n = (n2, n1, n0)
        s = q[f]
        self_ = sh(s, n=n)
        dLx = mesh.dL[f][0][np.arange(n0)][None, None, :]
        dLy = mesh.dL[f][1][np.arange(n1)][None, :, None]
        dLz = mesh.dL[f][2][np.arange(n2)][:, None, None] if dim == 3 else None
        # same-component face averages
        W = (self_ + sh(s, di=-1, n=n)) / 2.0
        E = (self_ + sh(s, di=1, n=n)) / 2.0
        S = (self_ + sh(s, dj=-1, n=n)) / 2.0
        N = (self_ + sh(s, dj=1, n=n)) / 2.0
        if dim == 3:
            B = (self_ + sh(s, dk=-1, n=n)) / 2.0
            F = (self_ + sh(s, dk=1, n=n)) / 2.0
        u, v = q[0], q[1]
        w = q[2] if dim == 3 else None
        if f == 0:
            vS = (sh(v, dj=-1, n=n) + sh(v, di=1, dj=-1, n=n)) / 2.0
            vN = (sh(v, n=n) + sh(v, di=1, n=n)) / 2.0
            r = (E * E - W * W) / dLx + (vN * N - vS * S) / dLy
            if dim == 3:
                wB = (sh(w, dk=-1, n=n) + sh(w, di=1, dk=-1, n=n)) / 2.0
                wF = (sh(w, n=n) + sh(w, di=1, n=n)) / 2.0
                r = r + (wF * F - wB * B) / dLz
        elif f == 1:
            uW = (sh(u, di=-1, n=n) + sh(u, di=-1, dj=1, n=n)) / 2.0
            uE = (sh(u, n=n) + sh(u, dj=1, n=n)) / 2.0
            r = (uE * E - uW * W) / dLx + (N * N - S * S) / dLy
            if dim == 3:
                wB = (sh(w, dk=-1, n=n) + sh(w, dj=1, dk=-1, n=n)) / 2.0
                wF = (sh(w, n=n) + sh(w, dj=1, n=n)) / 2.0
                r = r + (wF * F - wB * B) / dLz
        else:
            uW = (sh(u, di=-1, n=n) + sh(u, di=-1, dk=1, n=n)) / 2.0
            uE = (sh(u, n=n) + sh(u, dk=1, n=n)) / 2.0
            vS = (sh(v, dj=-1, n=n) + sh(v, dj=-1, dk=1, n=n)) / 2.0
            vN = (sh(v, n=n) + sh(v, dk=1, n=n)) / 2.0
            r = (uE * E - uW * W) / dLx + (vN * N - vS * S) / dLy + (F * F - B * B) / dLz
        out.append(r.reshape(-1))
    return np.concatenate(out)


def _face_take(g: GhostPoints, ijk, on):
    """a1 of the ghost point facing each selected boundary-adjacent point (flattened (k, j, i) order)"""
    idx = [ijk[2][on], ijk[1][on], ijk[0][on]]
    del idx[2 - g.axis]
    return g.a1[tuple(idx)]


def laplacian_correction(mesh: CartesianMesh, ghosts) -> np.ndarray:
    """LCorrectionMult (createlaplacian.cpp:45-78): y[row] += coeff*a1 for every ghost, in (field, loc) order."""
    y = np.zeros(mesh.UN)
    off = 0
    for f in range(mesh.dim):
        n0, n1, n2 = (int(v) for v in mesh.n[f])
        k, j, i = np.meshgrid(np.arange(n2), np.arange(n1), np.arange(n0), indexing="ij")
        ijk = (i.ravel(), j.ravel(), k.ravel())
        rows = np.arange(n0 * n1 * n2) + off
        for loc in range(2 * mesh.dim):
            if (f, loc) not in ghosts:
                continue
            d = loc // 2
            s = ijk[d]
            on = (s == 0) if loc % 2 == 0 else (s == mesh.n[f][d] - 1)
            dLSelf = mesh.dL[f][d][s[on]]
            if loc % 2 == 0:
                dist = mesh.coord[f][d][s[on]] - mesh.coord[f][d][s[on] - 1]
            else:
                dist = mesh.coord[f][d][s[on] + 1] - mesh.coord[f][d][s[on]]
            coeff = 1.0 / (dist * dLSelf)
            y[rows[on]] = y[rows[on]] + coeff * _face_take(ghosts[(f, loc)], ijk, on)
        off += n0 * n1 * n2
    return y


def divergence_correction(mesh: CartesianMesh, ghosts, normalize: bool = False) -> np.ndarray:
    """DCorrectionMult (createdivergence.cpp:45-78): y[cell] += (+-area)*a1 for the normal-velocity ghost faces."""
    y = np.zeros(mesh.pN)
    n0, n1, n2 = (int(v) for v in mesh.n[3])
    k, j, i = np.meshgrid(np.arange(n2), np.arange(n1), np.arange(n0), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    ijk = (i, j, k)
    for f in range(mesh.dim):
        if f == 0:
            area = mesh.dL[0][1][j] * mesh.dL[0][2][k]
        elif f == 1:
            area = mesh.dL[1][0][i] * mesh.dL[1][2][k]
        else:
            area = mesh.dL[2][0][i] * mesh.dL[2][1][j]
        for loc in (2 * f, 2 * f + 1):
            if (f, loc) not in ghosts:
                continue
            on = (ijk[f] == 0) if loc % 2 == 0 else (ijk[f] == mesh.n[3][f] - 1)
            coeff = (-area[on]) if loc % 2 == 0 else area[on]
            y[on] = y[on] + coeff * _face_take(ghosts[(f, loc)], ijk, on)
    return y


class NavierStokes:
    """State + one-step advance, AB2 convection + CN diffusion, BN order 1."""

    def __init__(self, mesh: CartesianMesh, dt: float, nu: float, pinned: bool = False, vtol=1e-14, ptol=1e-13,
                 bn_order: int = 1, convection: str = "ADAMS_BASHFORTH_2", diffusion: str = "CRANK_NICOLSON"):
        self.mesh, self.dt, self.nu, self.pinned = mesh, dt, nu, pinned
        # include/petibm/timeintegration.h:107-166: (implicit coefficient, explicit coefficients)
        schemes = {"EULER_EXPLICIT": (0.0, [1.0]), "EULER_IMPLICIT": (1.0, []), "ADAMS_BASHFORTH_2": (0.0, [1.5, -0.5]),
                   "CRANK_NICOLSON": (0.5, [0.5])}
        self.conv_c = list(schemes[convection][1])
        self.cimpl, self.diff_c = schemes[diffusion][0], list(schemes[diffusion][1])
        self.D = oops.create_divergence(mesh)
        self.G = oops.create_gradient(mesh)
        self.L = oops.create_laplacian(mesh)
        self.A = oops.create_velocity_operator(self.L, dt, self.cimpl * nu)
        self.BNG, DBNG = oops.create_poisson_operator(self.D, self.G, self.L, dt, self.cimpl * nu, bn_order)
        self.DBNG = oops.pin_row0(DBNG) if pinned else DBNG
        self.nonsymmetric = any(mesh.bc_types.get((f, 2 * f + e), "NOBC") == "NEUMANN" for f in range(mesh.dim) for e in (0, 1))
        self.ghosts = make_ghosts(mesh)
        self.U = np.zeros(mesh.UN)
        self.p = np.zeros(mesh.pN)
        set_ghost_ics(mesh, self.ghosts, self.U)
        self.conv = [np.zeros(mesh.UN) for _ in self.conv_c]
        self.diff = [np.zeros(mesh.UN) for _ in self.diff_c]
        self.vtol, self.ptol = vtol, ptol
        self.info = {}
        w = [mesh.dL[3][d].true for d in range(mesh.dim)]
        self.gmg = clib.GMG([int(v) for v in mesh.n[3][:mesh.dim]], w, dt, nullspace=2 if pinned else 1,
                            periodic=[bool(mesh.periodic[0][d]) for d in range(mesh.dim)])

    def set_state(self, U, p=None):
        """initial data + bc->setGhostICs(solution) (navierstokes.cpp:139-142)"""
        self.U = np.array(U, dtype=np.float64)
        if p is not None:
            self.p = np.array(p, dtype=np.float64)
        set_ghost_ics(self.mesh, self.ghosts, self.U)

    def rhs_velocity(self):
        dt, nu = self.dt, self.nu
        rhs1 = clib.spmv(self.G, self.p)
        rhs1 = -1.0 * rhs1
        rhs1 = rhs1 + (1.0 / dt) * self.U
        for i in range(len(self.conv) - 1, 0, -1):  # VecSwap chain (navierstokes.cpp:452-458)
            self.conv[i], self.conv[i - 1] = self.conv[i - 1], self.conv[i]
        if self.conv:
            self.conv[0] = -1.0 * convection(self.mesh, self.U, self.ghosts)
        for c, v in zip(self.conv_c, self.conv):
            rhs1 = rhs1 + c * v
        for i in range(len(self.diff) - 1, 0, -1):
            self.diff[i], self.diff[i - 1] = self.diff[i - 1], self.diff[i]
        if self.diff:
            diff0 = clib.spmv(self.L, self.U)
            diff0 = diff0 + laplacian_correction(self.mesh, self.ghosts)  # ghost equations of the previous step
            self.diff[0] = nu * diff0
        for c, v in zip(self.diff_c, self.diff):
            rhs1 = rhs1 + c * v
        update_eqs(self.mesh, self.ghosts, self.U, dt)  # navierstokes.cpp:508
        bc1 = nu * laplacian_correction(self.mesh, self.ghosts)
        rhs1 = rhs1 + self.cimpl * bc1
        return rhs1

    def rhs_poisson(self):
        rhs2 = clib.spmv(self.D, self.U)
        rhs2 = rhs2 + divergence_correction(self.mesh, self.ghosts)
        if self.pinned:
            rhs2[0] = 0.0
        return rhs2

    def advance(self, velocity_rhs_only=False):
        rhs1 = self.rhs_velocity()
        self.last_rhs1 = rhs1
        if velocity_rhs_only:
            return
        r = clib.bcgs(self.A, rhs1, x0=self.U, pc="jacobi", norm="unpreconditioned", rtol=0.0, atol=self.vtol,
                      dtol=1e300, maxit=2000)
        assert r["reason"] > 0
        self.U = r["x"]
        self.info["vIters"] = r["iters"]
        rhs2 = self.rhs_poisson()
        self.last_rhs2 = rhs2
        if self.nonsymmetric:
            # a NEUMANN condition on a normal velocity component folds a0 = 1 into D (createdivergence.cpp:231-242): DBNG is no
            # longer symmetric at that face -- BiCGStab (Jacobi) instead of CG
            rp = clib.bcgs(self.DBNG, rhs2, pc="jacobi", norm="unpreconditioned", rtol=0.0, atol=self.ptol, dtol=1e300, maxit=5000)
        else:
            rp = self.gmg.pcg(self.DBNG, rhs2, rtol=0.0, atol=self.ptol, maxit=500)
        assert rp["reason"] > 0, rp
        dP = rp["x"]
        if not self.pinned:
            dP = dP - dP.mean()
        self.info["pIters"] = rp["iters"]
        rhs1 = clib.spmv(self.BNG, dP)
        self.U = self.U + (-1.0) * rhs1
        self.p = self.p + 1.0 * dP
        update_ghost_values(self.mesh, self.ghosts, self.U)  # navierstokes.cpp:263


def vorticity(mesh: CartesianMesh, U: np.ndarray, ghosts) -> dict:
    """petibm-vorticity (applications/vorticity/main.cpp:185-372; fields and point sets :384-470), restated index for
    index on the ghost-padded local arrays: a local index -1 is the ghost layer, corners between two ghost layers keep
    the local vectors' initial zero.  Returns {name: array (nz, ny, nx)}."""
    q = ghost_padded(mesh, U, ghosts)  # [f][k+1, j+1, i+1]
    n3 = [int(v) for v in mesh.n[3]]
    n4 = [n3[d] + 1 if d < mesh.dim else 1 for d in range(3)]

    def at(f, i, j, k):
        return q[f][k + 1, j + 1, i + 1]

    def co(f, d, s):
        return np.array([mesh.coord[f][d][int(t)] for t in np.ravel(s)]).reshape(np.shape(s))

    out = {}
    if mesh.dim == 2:
        j, i = np.meshgrid(np.arange(n4[1]), np.arange(n4[0]), indexing="ij")
        k = np.zeros_like(i)
        wz = (at(1, i, j - 1, k) - at(1, i - 1, j - 1, k)) / (co(1, 0, i) - co(1, 0, i - 1)) - \
             (at(0, i - 1, j, k) - at(0, i - 1, j - 1, k)) / (co(0, 1, j) - co(0, 1, j - 1))
        out["wz"] = wz
        return out
    k, j, i = np.meshgrid(np.arange(n4[2]), np.arange(n4[1]), np.arange(n3[0]), indexing="ij")
    out["wx"] = (at(2, i - 1, j, k - 1) - at(2, i - 1, j - 1, k - 1)) / (co(2, 1, j) - co(2, 1, j - 1)) - \
                (at(1, i - 1, j - 1, k) - at(1, i - 1, j - 1, k - 1)) / (co(1, 2, k) - co(1, 2, k - 1))
    k, j, i = np.meshgrid(np.arange(n4[2]), np.arange(n3[1]), np.arange(n4[0]), indexing="ij")
    out["wy"] = (at(0, i - 1, j - 1, k) - at(0, i - 1, j - 1, k - 1)) / (co(0, 2, k) - co(0, 2, k - 1)) - \
                (at(2, i, j - 1, k - 1) - at(2, i - 1, j - 1, k - 1)) / (co(2, 0, i) - co(2, 0, i - 1))
    k, j, i = np.meshgrid(np.arange(n3[2]), np.arange(n4[1]), np.arange(n4[0]), indexing="ij")
    out["wz"] = (at(1, i, j - 1, k) - at(1, i - 1, j - 1, k)) / (co(1, 0, i) - co(1, 0, i - 1)) - \
                (at(0, i - 1, j, k) - at(0, i - 1, j - 1, k)) / (co(0, 1, j) - co(0, 1, j - 1))
    return out
