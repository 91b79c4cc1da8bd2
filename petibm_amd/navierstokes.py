"""Host-side mirror of the reference's flow-solver step over the C ABI (pib_ns_*).

Mirrors the part of `NavierStokesSolver` (applications/navierstokes/navierstokes.h:47-70) that touches the hot
path: construct from the YAML-shaped config dict (mesh / flow / parameters), `advance()`, and the per-step
solver information the reference logs in iterations-<start>.txt.  Everything runs on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

_DIR = {"x": 0, "y": 1, "z": 2}
_LOC = {"xMinus": 0, "xPlus": 1, "yMinus": 2, "yPlus": 3, "zMinus": 4, "zPlus": 5}
_BCT = {"DIRICHLET": 0, "NEUMANN": 1, "CONVECTIVE": 2, "PERIODIC": 3}

def _eval_expression(text: str, env: dict):
    """flow.initialVelocity / initialPressure expressions (the reference parses them with SymEngine,
    src/parser/parser.cpp): arithmetic on x, y, z, t, nu, pi, e and the elementary functions in `env`.  The text is
    parsed with `ast` and only those node types are evaluated -- a case directory from somebody else must not be able
    to run code."""
    import ast
    import operator
    ops = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
           ast.Pow: operator.pow, ast.USub: operator.neg, ast.UAdd: operator.pos}

    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
            return node.value
        if isinstance(node, ast.Name) and node.id in env and not callable(env[node.id]):
            return env[node.id]
        if isinstance(node, ast.BinOp) and type(node.op) in ops:
            return ops[type(node.op)](ev(node.left), ev(node.right))
        if isinstance(node, ast.UnaryOp) and type(node.op) in ops:
            return ops[type(node.op)](ev(node.operand))
        if (isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and callable(env.get(node.func.id))
                and not node.keywords):
            return env[node.func.id](*[ev(a) for a in node.args])
        raise ValueError(f"unsupported element in expression {text!r}: {ast.dump(node)[:60]}")

    return ev(ast.parse(text.replace("^", "**"), mode="eval"))


DEFAULT_VELOCITY_CFG = ("config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=1000\nsolv:monitor_residual=1\n"
                        "solv:convergence=ABSOLUTE\nsolv:tolerance=1e-12\nsolv:norm=L2\nsolv:store_res_history=1\n"
                        "solv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=1.0\n")
DEFAULT_POISSON_CFG = ("-poisson_ksp_type cg\n-poisson_ksp_atol 1.0E-12\n-poisson_ksp_rtol 0.0\n"
                       "-poisson_ksp_max_it 1000\n-poisson_ksp_norm_type unpreconditioned\n-poisson_pc_type gamg\n"
                       "-poisson_pib_smoother JACOBI\n")


def _widths(axis: dict):
    """parseSubDomains + stretchGrid (src/parser/parser.cpp:298-356, include/petibm/misc.h:148-163)."""
    import math
    bg = float(axis["start"])
    out = []
    for sub in axis["subDomains"]:
        n, ed, r = int(sub["cells"]), float(sub["end"]), float(sub["stretchRatio"])
        if abs(r - 1.0) <= 1e-12:
            out += [(ed - bg) / n] * n
        else:
            d0 = (ed - bg) * (r - 1.0) / (math.pow(r, float(n)) - 1.0)
            seg = [d0]
            for _ in range(1, n):
                seg.append(seg[-1] * r)
            out += seg
        bg = ed
    return np.array(out), float(axis["start"]), bg


class NavierStokesSolver:
    def __init__(self, config: dict, velocity_cfg: str = DEFAULT_VELOCITY_CFG, poisson_cfg: str = DEFAULT_POISSON_CFG,
                 device: int = -1, rank: int = 0, nranks: int = 1, uid: bytes = None):
        """rank / nranks / uid (as LinSolverHIP): the engine on this rank's z-slab (y-slab in 2-D) of the mesh, one rank
        per GPU (pib_ns_create_slab).  UN / pN, setState / getState then speak of this rank's part of the distributed
        vectors; `ownedVelocity` / `ownedPressure` cut it out of a global array."""
        self.rank, self.nranks = int(rank), int(nranks)
        axes = sorted(config["mesh"], key=lambda a: _DIR[a["direction"]])
        self.dim = len(axes)
        ws, lo, hi = [], [], []
        for a in axes:
            w, s, e = _widths(a)
            ws.append(np.ascontiguousarray(w))
            lo.append(s)
            hi.append(e)
        self.n = [len(w) for w in ws]
        self.widths = ws
        self._lo = list(lo)
        bc_t = np.zeros(18, dtype=np.int32)
        bc_v = np.zeros(18)
        for bc in config["flow"]["boundaryConditions"]:
            loc = _LOC[bc["location"]]
            for name, f in (("u", 0), ("v", 1), ("w", 2)):
                if name in bc:
                    t = str(bc[name][0]).upper()
                    if t not in _BCT:
                        raise capi.PibError(capi.ERR_SUP, f"boundary type {t} is not supported by the device time step")
                    bc_t[6 * f + loc] = _BCT[t]
                    bc_v[6 * f + loc] = float(bc[name][1])
        # a direction is periodic when every component is PERIODIC at both of its ends (misc.cpp checkPeriodicBC; the
        # engine reports a partial specification as an error): component d then has n[d] points along d
        self.periodic = [all(bc_t[6 * f + 2 * d + e] == 3 for f in range(self.dim) for e in range(2))
                         for d in range(self.dim)]
        par = config["parameters"]
        self.dt = float(par["dt"])
        self.nu = float(config["flow"]["nu"])
        # parameters.convection / diffusion (createTimeIntegration: both keys are required by the reference; the mirror
        # defaults to the pair every example but the convergence study uses)
        self.convection = str(par.get("convection", "ADAMS_BASHFORTH_2"))
        self.diffusion = str(par.get("diffusion", "CRANK_NICOLSON"))
        n3 = np.array(self.n + [1] * (3 - self.dim), dtype=np.int64)
        lo3 = np.array(lo + [0.0] * (3 - self.dim))
        hi3 = np.array(hi + [1.0] * (3 - self.dim))
        wp = [w.ctypes.data for w in ws] + [None] * (3 - self.dim)
        self._h = C.c_void_p()
        self._keep = (n3, lo3, hi3, bc_t, bc_v)
        if self.nranks > 1:
            uidbuf = C.create_string_buffer(uid, capi.UID_BYTES)
            capi.check(capi.load().pib_ns_create_slab(C.byref(self._h), self.dim, n3.ctypes.data, wp[0], wp[1], wp[2],
                                                      lo3.ctypes.data, hi3.ctypes.data, bc_t.ctypes.data, bc_v.ctypes.data,
                                                      self.dt, self.nu, velocity_cfg.encode(), poisson_cfg.encode(),
                                                      self.rank, self.nranks, uidbuf, device))
        else:
            capi.check(capi.load().pib_ns_create(C.byref(self._h), self.dim, n3.ctypes.data, wp[0], wp[1], wp[2],
                                                 lo3.ctypes.data, hi3.ctypes.data, bc_t.ctypes.data, bc_v.ctypes.data,
                                                 self.dt, self.nu, velocity_cfg.encode(), poisson_cfg.encode(), device))
        if (self.convection, self.diffusion) != ("ADAMS_BASHFORTH_2", "CRANK_NICOLSON"):
            capi.check(capi.load().pib_ns_set_time_integration(self._h, self.convection.encode(), self.diffusion.encode()))
        kept = {"EULER_EXPLICIT": 1, "EULER_IMPLICIT": 0, "ADAMS_BASHFORTH_2": 2, "CRANK_NICOLSON": 1}
        self._nconv, self._ndiff = kept[self.convection], kept[self.diffusion]
        self.bn_order = int(par.get("BN", 1))  # parameters.BN (parser: default 1)
        if self.bn_order != 1:
            capi.check(capi.load().pib_ns_set_bn_order(self._h, self.bn_order))
        un, pn = C.c_int64(), C.c_int64()
        capi.check(capi.load().pib_ns_sizes(self._h, C.byref(un), C.byref(pn)))
        self.UN, self.pN = un.value, pn.value
        self.ite = 0
        self.t = 0.0
        # flow.initialVelocity (src/solution/solutionsimple.cpp:122-226 setInitialConditions, called from
        # NavierStokesSolver::init): one entry per component, a number or an expression in x, y, z, t, nu
        # (the reference parses them with SymEngine) evaluated at the component's points at t = 0
        ic = config["flow"].get("initialVelocity")
        ip = config["flow"].get("initialPressure")  # optional, default 0 (parser.cpp:433-440)
        if ic is not None:
            U0 = self._initial_velocity(ic, lo)
            p0 = self._initial_velocity([ip], lo, pressure=True) if ip is not None else None
            if np.any(U0 != 0.0) or (p0 is not None and np.any(p0 != 0.0)) or self.nranks > 1:  # collective on slabs
                self.setState(self.ownedVelocity(U0), None if p0 is None else self.ownedPressure(p0))

    def _slab(self):
        """owned planes [k0, k1) of the slab axis: the DMDA split (pib_slab_range)"""
        b, e = C.c_int64(), C.c_int64()
        capi.check(capi.load().pib_slab_range(self.n[-1], self.nranks, self.rank, C.byref(b), C.byref(e)))
        return b.value, e.value

    def _field_shape(self, f):
        """points of velocity component f as (slab axis, ..., x)"""
        n = [self.n[d] - (0 if (d != f or self.periodic[d]) else 1) for d in range(self.dim)]
        return tuple(n[::-1])

    def ownedVelocity(self, U):
        """this rank's packed [u-slab | v-slab | w-slab] of a GLOBAL packed velocity array (cartesianmesh.cpp:740-779)"""
        if self.nranks == 1:
            return U
        k0, k1 = self._slab()
        parts, off = [], 0
        for f in range(self.dim):
            shp = self._field_shape(f)
            cnt = int(np.prod(shp))
            blk = np.asarray(U[off:off + cnt]).reshape(shp)
            parts.append(blk[k0:min(k1, shp[0])].reshape(-1))
            off += cnt
        return np.ascontiguousarray(np.concatenate(parts))

    def ownedPressure(self, p):
        if self.nranks == 1:
            return p
        k0, k1 = self._slab()
        pl = int(np.prod(self.n[:-1]))
        return np.ascontiguousarray(np.asarray(p)[k0 * pl:k1 * pl])

    def _points(self, f, d, lo=None):
        """coordinates of field f (0..2 velocity components, 3 pressure) along direction d (cartesianmesh.cpp:136-355)"""
        lo = self._lo if lo is None else lo
        vtx = lo[d] + np.concatenate([[0.0], np.cumsum(self.widths[d])])
        if f == d:
            return vtx[1:] if self.periodic[d] else vtx[1:-1]
        return 0.5 * (vtx[1:] + vtx[:-1])

    def _initial_velocity(self, ic, lo, pressure=False):
        import math
        names = {k: getattr(np, k) for k in ("sin", "cos", "tan", "exp", "log", "sqrt", "tanh", "sinh", "cosh", "abs")}
        names.update(pi=math.pi, e=math.e, t=0.0, nu=self.nu)
        parts = []
        for f in ([3] if pressure else range(self.dim)):
            axes = [self._points(f, d, lo) for d in range(self.dim)]
            grids = np.meshgrid(*axes[::-1], indexing="ij")[::-1]  # arrays indexed (k, j, i)
            env = dict(names, x=grids[0], y=grids[1], z=grids[2] if self.dim == 3 else 0.0)
            v = ic[0 if pressure else f]
            val = float(v) if not isinstance(v, str) else _eval_expression(v, env)
            parts.append(np.broadcast_to(np.asarray(val, dtype=np.float64), grids[0].shape).reshape(-1))
        return np.concatenate(parts)

    def setState(self, U=None, p=None):
        U = None if U is None else np.ascontiguousarray(U, dtype=np.float64)
        p = None if p is None else np.ascontiguousarray(p, dtype=np.float64)
        capi.check(capi.load().pib_ns_set_state(self._h, None if U is None else U.ctypes.data,
                                                None if p is None else p.ctypes.data))

    def advance(self, nsteps: int = 1):
        capi.check(capi.load().pib_ns_advance(self._h, int(nsteps)))
        self.ite += nsteps
        self.t += nsteps * self.dt

    # ---- the reference's PetscLogStage breakdown (navierstokes.cpp:186-199): rhsVelocity / solveVelocity / solveForces /
    # rhsPoisson / solvePoisson / update
    def enableStageTimers(self, on: bool = True) -> None:
        capi.check(capi.load().pib_ns_stage_timers(self._h, 1 if on else 0))

    def stageTimes(self) -> dict:
        """milliseconds accumulated per stage since enableStageTimers(), and the steps they cover"""
        import ctypes as C
        ms = np.zeros(6)
        steps = C.c_int64(0)
        lib = capi.load()
        capi.check(lib.pib_ns_get_stage_times(self._h, ms.ctypes.data, C.byref(steps)))
        out = {lib.pib_ns_stage_name(k).decode(): float(ms[k]) for k in range(6)}
        out["steps"] = int(steps.value)
        return out

    def getState(self, rhs: bool = False):
        U, p = np.empty(self.UN), np.empty(self.pN)
        r1 = np.empty(self.UN) if rhs else None
        r2 = np.empty(self.pN) if rhs else None
        capi.check(capi.load().pib_ns_get_state(self._h, U.ctypes.data, p.ctypes.data,
                                                None if r1 is None else r1.ctypes.data,
                                                None if r2 is None else r2.ctypes.data))
        return (U, p, r1, r2) if rhs else (U, p)

    # ---- files of the reference (HDF5 through petibm_amd.h5io; formats: see that module) -------------------------
    def _field_shapes(self):
        """(name, shape (nz,) ny, nx) of the velocity components, then the pressure"""
        out = []
        for f in range(self.dim):
            nf = [self.n[d] - (1 if (d == f and not self.periodic[d]) else 0) for d in range(self.dim)]
            out.append(("uvw"[f], tuple(nf[::-1])))
        out.append(("p", tuple(self.n[::-1])))
        return out

    def writeGrid(self, path: str) -> None:
        """CartesianMesh::write (src/mesh/cartesianmesh.cpp:798-823): groups u, v, w, p, vertex with the gridline
        coordinates x, y, z of each field (single zeros for the directions / fields a 2-D run does not have)"""
        from . import h5io
        vtx = [self._lo[d] + np.concatenate([[0.0], np.cumsum(self.widths[d])]) for d in range(self.dim)]
        ctr = [0.5 * (v[1:] + v[:-1]) for v in vtx]
        with h5io.File(path, "w") as f:
            for fi, name in enumerate(("u", "v", "w", "p", "vertex")):
                for d, ax in enumerate("xyz"):
                    if d >= self.dim or (fi == 2 and self.dim == 2):
                        c = np.zeros(1)
                    elif fi == 4:
                        c = vtx[d]
                    elif fi == 3 or fi != d:
                        c = ctr[d]
                    else:
                        c = vtx[d][1:] if self.periodic[d] else vtx[d][1:-1]
                    f.write(f"{name}/{ax}", c)

    def write(self, path: str) -> None:
        """NavierStokesSolver::writeSolutionHDF5 (navierstokes.cpp:618-634): u, v[, w], p as (nz,) ny, nx arrays and
        the time as attribute `time` of /p"""
        from . import h5io
        U, p = self.getState()
        with h5io.File(path, "w") as f:
            off = 0
            for name, shape in self._field_shapes():
                sz = int(np.prod(shape))
                if name == "p":
                    f.write(name, p.reshape(shape))
                else:
                    f.write(name, U[off:off + sz].reshape(shape))
                    off += sz
            f.write_attr("p", "time", self.t)

    def vorticity(self):
        """{name: array (nz, ny, nx)} of the reference's petibm-vorticity utility (applications/vorticity/main.cpp): wz at the
        vertices in 2-D; wx, wy, wz in 3-D"""
        out = {}
        for comp in ([2] if self.dim == 2 else [0, 1, 2]):
            n3 = np.zeros(3, dtype=np.int64)
            capi.check(capi.load().pib_ns_get_vorticity(self._h, comp, n3.ctypes.data, None))
            a = np.empty(int(np.prod(n3)))
            capi.check(capi.load().pib_ns_get_vorticity(self._h, comp, n3.ctypes.data, a.ctypes.data))
            shape = tuple(int(v) for v in n3[: self.dim][::-1])
            out["w" + "xyz"[comp]] = a.reshape(shape)
        return out

    def writeVorticity(self, path: str, grid_path: str = None) -> None:
        """append the vorticity datasets to a solution file and (optionally) their gridlines to grid.h5, like
        petibm-vorticity (main.cpp:98-107,160-163)"""
        from . import h5io
        w = self.vorticity()
        with h5io.File(path, "a") as f:
            for name, a in w.items():
                f.write(name, a)
        if grid_path is not None:
            vtx = [self._lo[d] + np.concatenate([[0.0], np.cumsum(self.widths[d])]) for d in range(self.dim)]
            ctr = [0.5 * (v[1:] + v[:-1]) for v in vtx]
            with h5io.File(grid_path, "a") as f:
                for name in w:
                    comp = "xyz".index(name[1])
                    for d, ax in enumerate("xyz"):
                        c = np.zeros(1) if d >= self.dim else (ctr[d] if d == comp else vtx[d])
                        f.write(f"{name}/{ax}", c)

    def _history(self):
        c0, c1, d0 = np.empty(self.UN), np.empty(self.UN), np.empty(self.UN)
        capi.check(capi.load().pib_ns_get_history(self._h, c0.ctypes.data, c1.ctypes.data, d0.ctypes.data))
        return c0, c1, d0

    def _history_term(self, kind: int, index: int, value=None):
        a = np.empty(self.UN) if value is None else np.ascontiguousarray(value, dtype=np.float64)
        capi.check(capi.load().pib_ns_history_term(self._h, kind, index, 0 if value is None else 1, a.ctypes.data))
        return a

    def writeRestartData(self, path: str) -> None:
        """NavierStokesSolver::writeRestartDataHDF5 (navierstokes.cpp:637-686): the solution file plus the explicit
        terms /convection/0, /convection/1, /diffusion/0 (packed velocity ordering)"""
        from . import h5io
        import os
        if not os.path.exists(path):
            self.write(path)
        with h5io.File(path, "a") as f:  # one dataset per explicit term the schemes keep
            for i in range(self._nconv):
                f.write(f"convection/{i}", self._history_term(0, i))
            for i in range(self._ndiff):
                f.write(f"diffusion/{i}", self._history_term(1, i))

    def readRestartData(self, path: str) -> None:
        """NavierStokesSolver::readRestartDataHDF5 (navierstokes.cpp:689-746): fields, time, explicit convective terms,
        then bc->setGhostICs(solution)"""
        from . import h5io
        with h5io.File(path, "r") as f:
            parts = []
            for name, shape in self._field_shapes():
                a = f.read(name)
                if a.shape != shape:
                    raise capi.PibError(capi.ERR_FILE_READ, f"{path}: dataset {name} has shape {a.shape}, the mesh needs {shape}")
                parts.append(a.reshape(-1))
            self.t = f.read_attr("p", "time")
            conv = [f.read(f"convection/{i}") for i in range(self._nconv)]
            diff = [f.read(f"diffusion/{i}") for i in range(self._ndiff)]
        self.setState(np.concatenate(parts[:-1]), parts[-1])  # includes setGhostICs
        for i, a in enumerate(conv):
            self._history_term(0, i, a)
        for i, a in enumerate(diff):
            self._history_term(1, i, a)
        self.ite = int(round(self.t / self.dt))

    def linSolversInfo(self):
        """ite, vIters, vRes, pIters, pRes -- one line of iterations-<start>.txt (navierstokes.cpp:766-794)"""
        vi, pi, vr, pr = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        capi.check(capi.load().pib_ns_get_solver_info(self._h, C.byref(vi), C.byref(vr), C.byref(pi), C.byref(pr)))
        return self.ite, vi.value, vr.value, pi.value, pr.value

    def describeSolver(self, which: str) -> str:
        """what the engine's 'velocity' / 'poisson' solver RUNS (pib_describe), departures from its file included"""
        buf = C.create_string_buffer(4096)
        capi.check(capi.load().pib_ns_describe_solver(self._h, {"velocity": 0, "poisson": 1}[which], buf, 4096))
        return buf.value.decode()

    def destroy(self):
        if self._h:
            capi.load().pib_ns_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


DEFAULT_FORCES_CFG = "-forces_ksp_type preonly\n-forces_pc_type lu\n-forces_pc_factor_mat_solver_type superlu_dist\n"


def read_lagrangian_points(path: str) -> np.ndarray:
    """Body file of the reference (src/io/io.cpp:23-118): the number of points, then one coordinate set per line."""
    with open(path) as fh:
        first = fh.readline().split()
        if len(first) != 1:
            raise capi.PibError(66, f"The first line in file {path} contains more than one integer. Please check the format.")
        n = int(first[0])
        pts = [[float(v) for v in line.split()] for line in fh if line.strip()]
    if len(pts) != n:
        raise capi.PibError(66, f"The number of coordinate sets in {path} does not match the header ({len(pts)} vs {n}).")
    return np.array(pts, dtype=np.float64)


class DecoupledIBPMSolver(NavierStokesSolver):
    """Mirror of `DecoupledIBPMSolver` (applications/decoupledibpm/decoupledibpm.h): the flow solver plus immersed
    bodies given as Lagrangian points (`bodies:` of the YAML file: `type: points`, `file:` relative to the simulation
    directory, src/body/bodypack.cpp; or arrays), the delta kernel `parameters.delta` and a forces solver."""

    def __init__(self, config: dict, bodies=None, velocity_cfg: str = DEFAULT_VELOCITY_CFG,
                 poisson_cfg: str = DEFAULT_POISSON_CFG, forces_cfg: str = DEFAULT_FORCES_CFG, device: int = -1,
                 directory: str = ".", rank: int = 0, nranks: int = 1, uid: bytes = None):
        super().__init__(config, velocity_cfg=velocity_cfg, poisson_cfg=poisson_cfg, device=device, rank=rank,
                         nranks=nranks, uid=uid)
        import os
        if bodies is None:
            bodies = []
            for node in config.get("bodies", []):
                if node.get("type", "points") != "points":
                    raise capi.PibError(capi.ERR_SUP, f"body type {node.get('type')} is not supported (points)")
                f = node["file"]
                bodies.append(read_lagrangian_points(f if os.path.isabs(f) else os.path.join(directory, f)))
        self.bodies = [np.ascontiguousarray(b, dtype=np.float64) for b in bodies]
        for b in self.bodies:
            if b.ndim != 2 or b.shape[1] != self.dim:
                raise capi.PibError(66, "The dimension of Lagrangian points are different than that of the background mesh!")
        npts = np.array([b.shape[0] for b in self.bodies], dtype=np.int64)
        if not self.bodies:
            raise capi.PibError(capi.ERR_ARG_OUTOFRANGE, "DecoupledIBPMSolver: the case has no bodies (use NavierStokesSolver)")
        coords = np.ascontiguousarray(np.concatenate(self.bodies, axis=0))
        kernel = str(config.get("parameters", {}).get("delta", "ROMA_ET_AL_1999"))
        capi.check(capi.load().pib_ns_set_bodies(self._h, len(self.bodies), npts.ctypes.data, coords.ctypes.data,
                                                 kernel.encode(), forces_cfg.encode()))
        nf, nb = C.c_int64(), C.c_int()
        capi.check(capi.load().pib_ns_num_forces(self._h, C.byref(nf), C.byref(nb)))
        self.nf, self.nBodies = nf.value, nb.value

    def moveBodies(self, bodies, velocities=None):
        """RigidKinematicsSolver::moveBodies (applications/rigidkinematics/rigidkinematics.cpp:118-140): new point
        coordinates per body and their prescribed velocities (same shapes); call before the `advance()` of the step."""
        self.bodies = [np.ascontiguousarray(b, dtype=np.float64) for b in bodies]
        coords = np.ascontiguousarray(np.concatenate(self.bodies, axis=0))
        ub = None if velocities is None else np.ascontiguousarray(
            np.concatenate([np.asarray(v, dtype=np.float64) for v in velocities], axis=0)).reshape(-1)
        if coords.shape[0] * self.dim != self.nf or (ub is not None and ub.size != self.nf):
            raise capi.PibError(capi.ERR_ARG_WRONG, "moveBodies: the number of Lagrangian points must not change")
        capi.check(capi.load().pib_ns_move_bodies(self._h, coords.ctypes.data, None if ub is None else ub.ctypes.data))

    def getForces(self):
        """(Lagrangian forces f, per-body forces): the second is one line of forces-<start>.txt (decoupledibpm.cpp:437-465)"""
        f = np.empty(self.nf)
        avg = np.empty(self.nBodies * self.dim)
        capi.check(capi.load().pib_ns_get_forces(self._h, f.ctypes.data, avg.ctypes.data))
        return f, avg.reshape(self.nBodies, self.dim)

    def setForces(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        capi.check(capi.load().pib_ns_set_forces(self._h, f.ctypes.data))

    def linSolversInfo(self):
        """ite, vIters, vRes, pIters, pRes, fIters, fRes (decoupledibpm.cpp:399-434)"""
        base = super().linSolversInfo()
        fi, fr = C.c_int(), C.c_double()
        capi.check(capi.load().pib_ns_get_forces_solver_info(self._h, C.byref(fi), C.byref(fr)))
        return base + (fi.value, fr.value)

    def writeRestartData(self, path: str) -> None:
        """DecoupledIBPMSolver::writeRestartDataHDF5 (decoupledibpm.cpp:316-349): + the Lagrangian forces as /force"""
        from . import h5io
        super().writeRestartData(path)
        with h5io.File(path, "a") as f:
            f.write("force", self.getForces()[0])

    def readRestartData(self, path: str) -> None:
        from . import h5io
        super().readRestartData(path)
        with h5io.File(path, "r") as f:
            self.setForces(f.read("force"))

    def getOperator(self, which: str):
        """'delta' | 'E' | 'H' | 'EBNH' | 'BNH' (assembled for parameters.BN > 1 only) as (n_rows, rowptr, col, val[, row_ids])
        host arrays (inspection / parity)"""
        w = {"delta": 0, "E": 1, "H": 2, "EBNH": 3, "BNH": 4}[which]
        nr, nz = C.c_int64(), C.c_int64()
        lib = capi.load()
        capi.check(lib.pib_ns_get_ib_operator(self._h, w, C.byref(nr), C.byref(nz), None, None, None, None))
        rp = np.empty(nr.value + 1, dtype=np.int32)
        cl = np.empty(nz.value, dtype=np.int32)
        vl = np.empty(nz.value)
        ids = np.empty(nr.value, dtype=np.int32) if w == 2 else None
        capi.check(lib.pib_ns_get_ib_operator(self._h, w, C.byref(nr), C.byref(nz), rp.ctypes.data, cl.ctypes.data,
                                              vl.ctypes.data, None if ids is None else ids.ctypes.data))
        return (nr.value, rp, cl, vl) if ids is None else (nr.value, rp, cl, vl, ids)


class IBPMSolver(DecoupledIBPMSolver):
    """Mirror of IBPMSolver (applications/ibpm/ibpm.h:30-111): the coupled immersed-boundary projection method -- same
    construction as the decoupled solver, pressure and forces solved as one unknown (pib_ns_set_coupled)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        capi.check(capi.load().pib_ns_set_coupled(self._h, 1))

