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
_BCT = {"DIRICHLET": 0, "NEUMANN": 1, "CONVECTIVE": 2}

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
                 device: int = -1):
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
        par = config["parameters"]
        self.dt = float(par["dt"])
        self.nu = float(config["flow"]["nu"])
        for key, want in (("convection", "ADAMS_BASHFORTH_2"), ("diffusion", "CRANK_NICOLSON")):
            if par.get(key, want) != want:
                raise capi.PibError(capi.ERR_SUP, f"{key}: only {want} is provided")
        n3 = np.array(self.n + [1] * (3 - self.dim), dtype=np.int64)
        lo3 = np.array(lo + [0.0] * (3 - self.dim))
        hi3 = np.array(hi + [1.0] * (3 - self.dim))
        wp = [w.ctypes.data for w in ws] + [None] * (3 - self.dim)
        self._h = C.c_void_p()
        self._keep = (n3, lo3, hi3, bc_t, bc_v)
        capi.check(capi.load().pib_ns_create(C.byref(self._h), self.dim, n3.ctypes.data, wp[0], wp[1], wp[2],
                                             lo3.ctypes.data, hi3.ctypes.data, bc_t.ctypes.data, bc_v.ctypes.data,
                                             self.dt, self.nu, velocity_cfg.encode(), poisson_cfg.encode(), device))
        un, pn = C.c_int64(), C.c_int64()
        capi.check(capi.load().pib_ns_sizes(self._h, C.byref(un), C.byref(pn)))
        self.UN, self.pN = un.value, pn.value
        self.ite = 0
        self.t = 0.0

    def setState(self, U=None, p=None):
        U = None if U is None else np.ascontiguousarray(U, dtype=np.float64)
        p = None if p is None else np.ascontiguousarray(p, dtype=np.float64)
        capi.check(capi.load().pib_ns_set_state(self._h, None if U is None else U.ctypes.data,
                                                None if p is None else p.ctypes.data))

    def advance(self, nsteps: int = 1):
        capi.check(capi.load().pib_ns_advance(self._h, int(nsteps)))
        self.ite += nsteps
        self.t += nsteps * self.dt

    def getState(self, rhs: bool = False):
        U, p = np.empty(self.UN), np.empty(self.pN)
        r1 = np.empty(self.UN) if rhs else None
        r2 = np.empty(self.pN) if rhs else None
        capi.check(capi.load().pib_ns_get_state(self._h, U.ctypes.data, p.ctypes.data,
                                                None if r1 is None else r1.ctypes.data,
                                                None if r2 is None else r2.ctypes.data))
        return (U, p, r1, r2) if rhs else (U, p)

    def linSolversInfo(self):
        """ite, vIters, vRes, pIters, pRes -- one line of iterations-<start>.txt (navierstokes.cpp:766-794)"""
        vi, pi, vr, pr = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        capi.check(capi.load().pib_ns_get_solver_info(self._h, C.byref(vi), C.byref(vr), C.byref(pi), C.byref(pr)))
        return self.ite, vi.value, vr.value, pi.value, pr.value

    def destroy(self):
        if self._h:
            capi.load().pib_ns_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
