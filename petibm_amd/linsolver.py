"""Host-side mirror of petibm::linsolver over the C ABI.

Same names, argument meaning and error behaviour as the reference's plugin
interface (include/petibm/linsolver.h:59-147, src/linsolver/linsolver.cpp:57-91):

    solver = createLinSolver("poisson", node)     # node: the YAML tree as a dict
    solver.setMatrix(A)                           # CSR (rowptr, col, val) or scipy-like
    solver.solve(x, b)                            # x in/out, b read-only
    solver.getIters(), solver.getResidual(), solver.getType()
    solver.destroy()

`type: GPU` selects this backend (it takes the place of LinSolverAmgX,
src/linsolver/linsolver.cpp:77-83); `type: CPU` (PETSc KSP on the host) is the
reference's other branch and is NOT provided by this package: asking for it
raises, exactly like the reference built without the backend it needs.
Vectors are numpy arrays (host) or `DeviceVec` (HBM-resident).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import capi
from .capi import PibError


class DeviceVec:
    """n doubles in HBM owned by a solver's device (bench / device-resident callers)."""

    def __init__(self, solver: "LinSolverBase", n: int):
        self._solver = solver
        self.n = int(n)
        p = C.c_void_p()
        capi.check(capi.load().pib_device_alloc(solver._h, self.n * 8, C.byref(p)))
        self.ptr = p.value

    def upload(self, a: np.ndarray) -> "DeviceVec":
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        capi.check(capi.load().pib_memcpy_h2d(self._solver._h, self.ptr, a.ctypes.data, self.n * 8))
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.n)
        capi.check(capi.load().pib_memcpy_d2h(self._solver._h, out.ctypes.data, self.ptr, self.n * 8))
        return out

    def free(self):
        if self.ptr:
            capi.load().pib_device_free(self._solver._h, self.ptr)
            self.ptr = None


def _vptr(v):
    if isinstance(v, DeviceVec):
        return v.ptr
    if isinstance(v, np.ndarray):
        if v.dtype != np.float64 or not v.flags["C_CONTIGUOUS"]:
            raise TypeError("vectors must be contiguous float64 arrays")
        return v.ctypes.data
    raise TypeError(f"unsupported vector type {type(v)}")


class LinSolverBase:
    """petibm::linsolver::LinSolverBase (include/petibm/linsolver.h:59-147)."""

    def __init__(self, solverName: str, file: Optional[str] = None, *, config_text: Optional[str] = None,
                 rank: int = 0, nranks: int = 1, uid: Optional[bytes] = None, device: int = -1):
        self.name = solverName
        self.config = file if file is not None else "None"
        self._h = C.c_void_p()
        lib = capi.load()
        uidbuf = C.create_string_buffer(uid, capi.UID_BYTES) if uid is not None else None
        if config_text is not None:
            code = lib.pib_create_from_string(C.byref(self._h), solverName.encode(), config_text.encode(), rank,
                                              nranks, uidbuf, device)
        else:
            code = lib.pib_create(C.byref(self._h), solverName.encode(),
                                  None if file in (None, "None") else file.encode(), rank, nranks, uidbuf, device)
        capi.check(code)
        buf = C.create_string_buffer(64)
        capi.check(lib.pib_get_type(self._h, buf, 64))
        self.type = buf.value.decode()
        self.n_local = 0

    # -- reference interface ------------------------------------------------
    def getType(self) -> str:
        return self.type

    def describe(self) -> str:
        """pib_describe: what this solver runs -- first line key=value pairs, then one 'departure: ...' line per place where
        the backend departs from the solver file"""
        buf = C.create_string_buffer(4096)
        capi.check(capi.load().pib_describe(self._h, buf, 4096))
        return buf.value.decode()

    def printInfo(self) -> str:
        """the reference's banner (src/linsolver/linsolver.cpp:29-41) + what actually runs"""
        info = "=" * 80 + f"\nLinear Solver {self.name}:\n" + "=" * 80 + "\n"
        info += f"\tType: {self.type}\n\n\tConfig file: {self.config}\n\n"
        if self._h:
            info += "".join(f"\tRuns: {ln}\n" for ln in self.describe().splitlines()) + "\n"
        print(info, end="")
        return info

    def setMatrix(self, A, row0: int = 0, n_global: Optional[int] = None) -> None:
        """A: object with .rowptr/.col/.val (local rows, GLOBAL columns) or a
        scipy.sparse CSR matrix (.indptr/.indices/.data)."""
        if hasattr(A, "indptr"):
            rowptr, col, val = A.indptr, A.indices, A.data
        else:
            rowptr, col, val = A.rowptr, A.col, A.val
        n_local = len(rowptr) - 1
        if n_global is None:
            n_global = n_local
        val = np.ascontiguousarray(val, dtype=np.float64)
        lib = capi.load()
        if np.asarray(rowptr).dtype == np.int32 and np.asarray(col).dtype == np.int32:
            rp = np.ascontiguousarray(rowptr, dtype=np.int32)
            cl = np.ascontiguousarray(col, dtype=np.int32)
            capi.check(lib.pib_set_csr_i32(self._h, n_local, row0, n_global, rp.ctypes.data, cl.ctypes.data,
                                           val.ctypes.data))
        else:
            rp = np.ascontiguousarray(rowptr, dtype=np.int64)
            cl = np.ascontiguousarray(col, dtype=np.int64)
            capi.check(lib.pib_set_csr(self._h, n_local, row0, n_global, rp.ctypes.data, cl.ctypes.data,
                                       val.ctypes.data))
        self.n_local = n_local

    def solve(self, x, b) -> None:
        capi.check(capi.load().pib_solve(self._h, _vptr(x), _vptr(b)))

    def getIters(self) -> int:
        v = C.c_int()
        capi.check(capi.load().pib_get_iters(self._h, C.byref(v)))
        return v.value

    def getResidual(self) -> float:
        v = C.c_double()
        capi.check(capi.load().pib_get_residual(self._h, C.byref(v)))
        return v.value

    def destroy(self) -> None:
        if self._h:
            capi.load().pib_destroy(self._h)
            self._h = C.c_void_p()
        self.name = self.config = self.type = ""

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                capi.load().pib_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # -- extensions behind the same handle -----------------------------------
    def getReason(self) -> int:
        v = C.c_int()
        capi.check(capi.load().pib_get_reason(self._h, C.byref(v)))
        return v.value

    def getResidualAt(self, it: int) -> float:
        v = C.c_double()
        capi.check(capi.load().pib_get_residual_at(self._h, it, C.byref(v)))
        return v.value

    def getResidualHistory(self) -> np.ndarray:
        return np.array([self.getResidualAt(i) for i in range(self.getIters() + 1)])

    def setGridHint(self, n, w, g, nullspace: int) -> None:
        dim = len(n)
        n3 = np.array(list(n) + [1] * (3 - dim), dtype=np.int64)
        ws = [np.ascontiguousarray(a, dtype=np.float64) for a in w]
        gs = [np.ascontiguousarray(a, dtype=np.float64) for a in g]
        while len(ws) < 3:
            ws.append(None)
            gs.append(None)
        p = [a.ctypes.data if a is not None else None for a in ws + gs]
        capi.check(capi.load().pib_set_grid_hint(self._h, dim, n3.ctypes.data, *p, int(nullspace)))

    def multigridLevels(self):
        """pib_get_multigrid_levels: the cells per direction of every level of the geometric multigrid ([] if there is none)"""
        nl = C.c_int()
        capi.check(capi.load().pib_get_multigrid_levels(self._h, C.byref(nl), None, 0))
        n3 = np.zeros(3 * max(nl.value, 1), dtype=np.int64)
        capi.check(capi.load().pib_get_multigrid_levels(self._h, C.byref(nl), n3.ctypes.data, nl.value))
        return [tuple(int(v) for v in n3[3 * l: 3 * l + 3]) for l in range(nl.value)]

    def gridStructure(self):
        """pib_get_grid_structure: None, or dict(dim, n, nullspace, detected)"""
        has, dim, ns, det = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        n3 = np.zeros(3, dtype=np.int64)
        capi.check(capi.load().pib_get_grid_structure(self._h, C.byref(has), C.byref(dim), n3.ctypes.data, C.byref(ns),
                                                      C.byref(det)))
        if not has.value:
            return None
        return {"dim": dim.value, "n": tuple(int(v) for v in n3[:dim.value]), "nullspace": ns.value,
                "detected": bool(det.value)}

    def velocityStructure(self):
        """pib_get_velocity_structure: None, or dict(dim, n, periodic, detected)"""
        has, dim, det = C.c_int(), C.c_int(), C.c_int()
        n3 = np.zeros(3, dtype=np.int64)
        per = (C.c_int * 3)()
        capi.check(capi.load().pib_get_velocity_structure(self._h, C.byref(has), C.byref(dim), n3.ctypes.data, per, C.byref(det)))
        if not has.value:
            return None
        return {"dim": dim.value, "n": tuple(int(v) for v in n3[:dim.value]), "periodic": tuple(bool(per[d]) for d in range(dim.value)),
                "detected": bool(det.value)}

    def setPeriodic(self, periodic) -> None:
        """periodic directions of the mesh (pib_set_periodic); call before the assembly / the grid hint"""
        per = (C.c_int * 3)(*[int(bool(periodic[d])) if d < len(periodic) else 0 for d in range(3)])
        capi.check(capi.load().pib_set_periodic(self._h, per))

    def assemblePoisson(self, n, widths, dt: float, nullspace: int) -> None:
        """DBNG assembled in HBM from the pressure-cell widths (pib_assemble_poisson)."""
        dim = len(n)
        n3 = np.array(list(n) + [1] * (3 - dim), dtype=np.int64)
        ws = [np.ascontiguousarray(a, dtype=np.float64) for a in widths]
        while len(ws) < 3:
            ws.append(None)
        p = [a.ctypes.data if a is not None else None for a in ws]
        capi.check(capi.load().pib_assemble_poisson(self._h, dim, n3.ctypes.data, *p, float(dt), int(nullspace)))
        nl, nnz = C.c_int64(), C.c_int64()
        capi.check(capi.load().pib_get_csr(self._h, C.byref(nl), C.byref(nnz), None, None, None))
        self.n_local, self.nnz = nl.value, nnz.value

    def assembleVelocity(self, n, widths, lo, hi, a0, dt: float, coeff_nu: float) -> None:
        """A = I/dt - c nu L assembled in HBM (pib_assemble_velocity); a0: (3,6) ghost coefficients."""
        dim = len(n)
        n3 = np.array(list(n) + [1] * (3 - dim), dtype=np.int64)
        ws = [np.ascontiguousarray(a, dtype=np.float64) for a in widths]
        while len(ws) < 3:
            ws.append(None)
        p = [a.ctypes.data if a is not None else None for a in ws]
        lo3 = np.array(list(lo) + [0.0] * (3 - len(lo)), dtype=np.float64)
        hi3 = np.array(list(hi) + [1.0] * (3 - len(hi)), dtype=np.float64)
        a = np.ascontiguousarray(np.asarray(a0, dtype=np.float64).reshape(18))
        capi.check(capi.load().pib_assemble_velocity(self._h, dim, n3.ctypes.data, *p, lo3.ctypes.data, hi3.ctypes.data,
                                                     a.ctypes.data, float(dt), float(coeff_nu)))
        nl, nnz = C.c_int64(), C.c_int64()
        capi.check(capi.load().pib_get_csr(self._h, C.byref(nl), C.byref(nnz), None, None, None))
        self.n_local, self.nnz = nl.value, nnz.value

    def assemblePoissonBN(self, n, widths, lo, hi, a0, dt: float, coeff_nu: float, bn_order: int, nullspace: int) -> None:
        """DBNG = D * BN(bn_order) * G assembled in HBM through the reference's chain of sparse products
        (pib_assemble_poisson_bn); arguments as assembleVelocity."""
        dim = len(n)
        n3 = np.array(list(n) + [1] * (3 - dim), dtype=np.int64)
        ws = [np.ascontiguousarray(a, dtype=np.float64) for a in widths]
        while len(ws) < 3:
            ws.append(None)
        p = [a.ctypes.data if a is not None else None for a in ws]
        lo3 = np.array(list(lo) + [0.0] * (3 - len(lo)), dtype=np.float64)
        hi3 = np.array(list(hi) + [1.0] * (3 - len(hi)), dtype=np.float64)
        a = np.ascontiguousarray(np.asarray(a0, dtype=np.float64).reshape(18))
        capi.check(capi.load().pib_assemble_poisson_bn(self._h, dim, n3.ctypes.data, *p, lo3.ctypes.data, hi3.ctypes.data,
                                                       a.ctypes.data, float(dt), float(coeff_nu), int(bn_order), int(nullspace)))
        nl, nnz = C.c_int64(), C.c_int64()
        capi.check(capi.load().pib_get_csr(self._h, C.byref(nl), C.byref(nnz), None, None, None))
        self.n_local, self.nnz = nl.value, nnz.value

    def getCSR(self):
        nl, nnz = C.c_int64(), C.c_int64()
        lib = capi.load()
        capi.check(lib.pib_get_csr(self._h, C.byref(nl), C.byref(nnz), None, None, None))
        rp = np.zeros(nl.value + 1, dtype=np.int64)
        cl = np.zeros(max(nnz.value, 1), dtype=np.int64)
        vl = np.zeros(max(nnz.value, 1))
        capi.check(lib.pib_get_csr(self._h, C.byref(nl), C.byref(nnz), rp.ctypes.data, cl.ctypes.data, vl.ctypes.data))
        return rp, cl[: nnz.value], vl[: nnz.value]

    def matMult(self, x, y) -> None:
        capi.check(capi.load().pib_mat_mult(self._h, _vptr(x), _vptr(y)))

    def timeKernel(self, which: int, reps: int) -> float:
        v = C.c_double()
        capi.check(capi.load().pib_time_kernel(self._h, which, reps, C.byref(v)))
        return v.value

    def counters(self) -> np.ndarray:
        c = np.zeros(8, dtype=np.int64)
        capi.check(capi.load().pib_get_counters(self._h, c.ctypes.data))
        return c

    def stagingMs(self):
        """(h2d_ms, d2h_ms) the last solve spent copying host vectors in / out (pib_get_staging_ms); zeros for device vectors"""
        a, b = C.c_double(), C.c_double()
        capi.check(capi.load().pib_get_staging_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def productIndexBytes(self) -> int:
        """bytes per matrix entry the CSR product reads besides the value: 4 (int32 column), 1 (pib_compress_columns=1: a column
        code per entry) or 0 (pib_compress_columns=2, the default: one pattern byte per ROW)"""
        v = C.c_int()
        capi.check(capi.load().pib_get_product_format(self._h, C.byref(v)))
        return v.value

    def placement(self):
        """(searches, candidates, ms_had, ms_kept) of the search direction's placement against x (pib_get_placement); zeros when none ran"""
        return self.placementInfo()[:4]

    def placementInfo(self):
        """(searches, candidates, ms_had, ms_kept, held_bytes, search_ms): ... plus the most bytes a search held at one time and
        the wall time of all searches"""
        a, b, c, d, e, f = C.c_int(), C.c_int(), C.c_double(), C.c_double(), C.c_int64(), C.c_double()
        capi.check(capi.load().pib_get_placement(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e), C.byref(f)))
        return a.value, b.value, c.value, d.value, e.value, f.value

    def deviceMemInfo(self):
        """(free, total) bytes of the solver's device"""
        a, b = C.c_int64(), C.c_int64()
        capi.check(capi.load().pib_device_mem_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def deviceVec(self, n: Optional[int] = None) -> DeviceVec:
        return DeviceVec(self, self.n_local if n is None else n)

    def synchronize(self) -> None:
        capi.check(capi.load().pib_synchronize(self._h))


class LinSolverHIP(LinSolverBase):
    """Takes the place of LinSolverAmgX (src/linsolver/linsolveramgx.cpp)."""


def createLinSolver(solverName: str, node: dict, **kw) -> LinSolverBase:
    """petibm::linsolver::createLinSolver (src/linsolver/linsolver.cpp:57-91).

    node["parameters"][f"{solverName}Solver"] holds `type` (default "CPU") and
    `config` (default "None"); a relative config path is taken relative to
    node["directory"] (linsolver.cpp:70-72)."""
    key = solverName + "Solver"
    sub = node.get("parameters", {}).get(key, {}) or {}
    type_ = sub.get("type", "CPU")
    config = sub.get("config", "None")
    if config != "None" and not config.startswith("/"):
        config = os.path.join(str(node["directory"]), config)
    if type_ == "GPU":
        return LinSolverHIP(solverName, config, **kw)
    if type_ == "CPU":
        raise PibError(capi.ERR_ARG_WRONG,
                       "PETSc KSP (type: CPU) solver is used, while this package only provides the GPU backend.")
    raise PibError(capi.ERR_ARG_WRONG,
                   f"Unrecognized value \"{type_}\" of the type of the linear solver \"{solverName}\"\n")
