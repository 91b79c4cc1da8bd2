"""ctypes binding of oracle/_build/liboracle.so (TEST INFRASTRUCTURE ONLY).

The shared object is built by `make -C oracle` (also called from
__graft_entry__.build()).  If it is missing it is built on first use.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "csrc", f) for f in ("oracle.c", "gmg.c")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_set_threads.argtypes = [C.c_int]
        _lib.orc_spmv.argtypes = [C.c_int64, _i64p, _i64p, _f64p, _f64p, _f64p]
        _lib.orc_spmv32.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p]
        sig = [C.c_int64, _i64p, _i64p, _f64p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
               C.c_double, C.c_int, C.c_int, _f64p, _f64p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p]
        _lib.orc_cg.argtypes = sig
        _lib.orc_cg.restype = C.c_int
        _lib.orc_cg_single_reduction.argtypes = sig
        _lib.orc_cg_single_reduction.restype = C.c_int
        _lib.orc_bcgs.argtypes = sig
        _lib.orc_bcgs.restype = C.c_int
        _lib.orc_chebyshev.argtypes = sig + [C.c_double, C.c_double]
        _lib.orc_chebyshev.restype = C.c_int
        _lib.orc_gershgorin_jacobi.argtypes = [C.c_int64, _i64p, _i64p, _f64p]
        _lib.orc_gershgorin_jacobi.restype = C.c_double
        _lib.orc_spgemm_symbolic.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, _i64p, _i64p, _i64p]
        _lib.orc_spgemm_symbolic.restype = C.c_int64
        _lib.orc_spgemm_numeric.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p, _i64p,
                                            _i64p, _f64p]
        _lib.orc_axpy_pattern_count.argtypes = [C.c_int64, _i64p, _i64p, _i64p, _i64p, _i64p]
        _lib.orc_axpy_pattern_count.restype = C.c_int64
        _lib.orc_axpy_pattern_fill.argtypes = [C.c_int64, C.c_double, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p, _i64p,
                                               _i64p, _f64p]
        # the parity tests run tiny systems: a 128-thread OpenMP team on 132-cell grids spends minutes in
        # fork/join.  Default to <= 8 threads; the cpu_baseline leg of bench.py asks for all cores explicitly.
        _lib.orc_set_threads(max(1, min(8, os.cpu_count() or 1)))
    return _lib


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def usable_cores() -> int:
    """Cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (a container that sees
    256 hardware threads but is limited to 16 CPUs runs a 256-thread OpenMP team a thousand times slower than a
    16-thread one: every barrier spins on descheduled threads)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / p_)))
        except Exception:
            pass
    return max(1, n)


def spmv(m, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(m.n_rows)
    lib().orc_spmv(m.n_rows, m.rowptr, m.col, m.val, x, y)
    return y


def spmv32(n, rowptr32, col32, val, x, y) -> None:
    lib().orc_spmv32(n, rowptr32, col32, val, x, y)


def spgemm(a, b):
    from .operators import CSR
    L = lib()
    crp = np.zeros(a.n_rows + 1, dtype=np.int64)
    nnz = L.orc_spgemm_symbolic(a.n_rows, b.n_cols, a.rowptr, a.col, b.rowptr, b.col, crp)
    ccol = np.zeros(max(nnz, 1), dtype=np.int64)
    cval = np.zeros(max(nnz, 1))
    L.orc_spgemm_numeric(a.n_rows, b.n_cols, a.rowptr, a.col, a.val, b.rowptr, b.col, b.val, crp, ccol, cval)
    return CSR(a.n_rows, b.n_cols, crp, ccol[:nnz], cval[:nnz])


def axpy_pattern(y, a: float, x):
    from .operators import CSR
    L = lib()
    zrp = np.zeros(y.n_rows + 1, dtype=np.int64)
    nnz = L.orc_axpy_pattern_count(y.n_rows, y.rowptr, y.col, x.rowptr, x.col, zrp)
    zcol = np.zeros(max(nnz, 1), dtype=np.int64)
    zval = np.zeros(max(nnz, 1))
    L.orc_axpy_pattern_fill(y.n_rows, a, y.rowptr, y.col, y.val, x.rowptr, x.col, x.val, zrp, zcol, zval)
    return CSR(y.n_rows, y.n_cols, zrp, zcol[:nnz], zval[:nnz])


PC = {"none": 0, "jacobi": 1}
NORM = {"preconditioned": 0, "unpreconditioned": 1}


def _krylov(fn, m, b, x0=None, pc="none", nullspace=0, norm="preconditioned", rtol=1e-5, atol=1e-50, dtol=1e4,
            maxit=10000):
    n = m.n_rows
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    dinv = None
    if pc == "jacobi":
        dinv = np.ascontiguousarray(1.0 / m.diagonal())
    hist = np.full(maxit + 2, np.nan)
    its = C.c_int(0)
    rn = C.c_double(0)
    reason = fn(n, m.rowptr, m.col, m.val, dinv.ctypes.data if dinv is not None else None, PC[pc], int(nullspace),
                NORM[norm], rtol, atol, dtol, int(maxit), int(x0 is not None), b, x, C.byref(its), C.byref(rn),
                hist.ctypes.data)
    return {"x": x, "iters": its.value, "rnorm": rn.value, "reason": int(reason),
            "history": hist[: its.value + 1].copy()}


def cg(m, b, single_reduction=False, **kw):
    """KSPCG restatement -- see oracle/csrc/oracle.c:orc_cg (single_reduction: orc_cg_single_reduction, the
    KSPCGUseSingleReduction recurrences)."""
    return _krylov(lib().orc_cg_single_reduction if single_reduction else lib().orc_cg, m, b, **kw)


def bcgs(m, b, **kw):
    """KSPBCGS restatement -- see oracle/csrc/oracle.c:orc_bcgs."""
    return _krylov(lib().orc_bcgs, m, b, **kw)


def chebyshev(m, b, emin=None, emax=None, **kw):
    """KSPCHEBYSHEV restatement -- see oracle/csrc/oracle.c:orc_chebyshev.  Without explicit bounds: the Gershgorin interval
    [1 - rho, 1 + rho] of the Jacobi-preconditioned operator (orc_gershgorin_jacobi), the build's default."""
    if emin is None or emax is None:
        rho = gershgorin_jacobi(m)
        emin, emax = 1.0 - rho, 1.0 + rho
    return _krylov(lambda *a: lib().orc_chebyshev(*a, float(emin), float(emax)), m, b, **kw)


def gershgorin_jacobi(m) -> float:
    return float(lib().orc_gershgorin_jacobi(m.n_rows, m.rowptr, m.col, m.val))


# ------------------------------------------------------------------ GMG oracle
class GMG:
    """CPU restatement of the build's geometric V-cycle (oracle/csrc/gmg.c)."""

    def __init__(self, n, widths, dt, nullspace=1, pre=1, post=1, omega=0.9, coarsest_sweeps=32, max_levels=100,
                 periodic=(False, False, False)):
        L = lib()
        L.orc_gmg_create_periodic.restype = C.c_void_p
        L.orc_gmg_create_periodic.argtypes = [C.c_int, _i64p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                              C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_gmg_destroy.argtypes = [C.c_void_p]
        L.orc_gmg_apply.argtypes = [C.c_void_p, _f64p, _f64p]
        L.orc_gmg_apply_operator.argtypes = [C.c_void_p, C.c_int, _f64p, _f64p]
        L.orc_gmg_num_levels.argtypes = [C.c_void_p]
        L.orc_gmg_level_size.argtypes = [C.c_void_p, C.c_int, _i64p]
        L.orc_pcg_gmg.restype = C.c_int
        L.orc_pcg_gmg.argtypes = [C.c_void_p, C.c_int64, _i64p, _i64p, _f64p, C.c_int, C.c_double, C.c_double, C.c_int,
                                  C.c_int, _f64p, _f64p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p]
        L.orc_pcg_gmg_single_reduction.restype = C.c_int
        L.orc_pcg_gmg_single_reduction.argtypes = L.orc_pcg_gmg.argtypes
        L.orc_bcgs_gmg.restype = C.c_int
        L.orc_bcgs_gmg.argtypes = [C.c_void_p, C.c_int64, _i64p, _i64p, _f64p, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.c_double, C.c_int, C.c_int, _f64p, _f64p, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                   C.c_void_p]
        self.dim = len(n)
        self.n = np.array(list(n), dtype=np.int64)
        self._w = [np.ascontiguousarray(w, dtype=np.float64) for w in widths]
        p = [w.ctypes.data for w in self._w] + [None] * (3 - self.dim)
        per = (C.c_int * 3)(*[int(bool(periodic[d])) if d < len(periodic) else 0 for d in range(3)])
        self._h = L.orc_gmg_create_periodic(self.dim, self.n, p[0], p[1], p[2], float(dt), int(nullspace), int(pre),
                                            int(post), float(omega), int(coarsest_sweeps), int(max_levels), per)
        self.N = int(np.prod(self.n))
        self.nullspace = nullspace
        L.orc_gmg_set_chebyshev.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]

    def set_chebyshev(self, lmax=2.0, ratio=4.0, on=True):
        """Chebyshev-Jacobi smoothing: pre/post given at construction are then the number of recurrence steps."""
        lib().orc_gmg_set_chebyshev(self._h, 1 if on else 0, float(lmax), float(ratio))
        return self

    def num_levels(self):
        return int(lib().orc_gmg_num_levels(self._h))

    def level_size(self, lev):
        out = np.zeros(3, dtype=np.int64)
        lib().orc_gmg_level_size(self._h, lev, out)
        return out

    def apply(self, r):
        z = np.empty(self.N)
        lib().orc_gmg_apply(self._h, np.ascontiguousarray(r, dtype=np.float64), z)
        return z

    def apply_operator(self, x, lev=0):
        n = int(np.prod(self.level_size(lev)))
        y = np.empty(n)
        lib().orc_gmg_apply_operator(self._h, lev, np.ascontiguousarray(x, dtype=np.float64), y)
        return y

    def pcg(self, m, b, x0=None, norm="unpreconditioned", rtol=1e-10, atol=0.0, maxit=1000, single_reduction=False):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(m.n_rows) if x0 is None else np.array(x0, dtype=np.float64)
        hist = np.full(maxit + 2, np.nan)
        its, rn = C.c_int(0), C.c_double(0)
        fn = lib().orc_pcg_gmg_single_reduction if single_reduction else lib().orc_pcg_gmg
        reason = fn(self._h, m.n_rows, m.rowptr, m.col, m.val, NORM[norm], rtol, atol, int(maxit),
                    int(x0 is not None), b, x, C.byref(its), C.byref(rn), hist.ctypes.data)
        return {"x": x, "iters": its.value, "rnorm": rn.value, "reason": int(reason),
                "history": hist[: its.value + 1].copy()}

    def bcgs(self, m, b, x0=None, norm="unpreconditioned", rtol=1e-10, atol=0.0, dtol=1e4, maxit=1000):
        """KSPBCGS / PBICGSTAB preconditioned by the V-cycle (oracle/csrc/oracle.c:orc_bcgs_gmg): left-preconditioned with
        norm="preconditioned" (the KSP flavour), right-preconditioned otherwise (the AmgX flavour); the mean is removed after
        every application when the multigrid was created with nullspace=1; nullspace=2 (pinned row 0): the output is shifted
        by its value at cell 0 and the pinned unknown keeps the input's value."""
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(m.n_rows) if x0 is None else np.array(x0, dtype=np.float64)
        hist = np.full(maxit + 2, np.nan)
        its, rn = C.c_int(0), C.c_double(0)
        reason = lib().orc_bcgs_gmg(self._h, m.n_rows, m.rowptr, m.col, m.val, self.nullspace if self.nullspace in (1, 2) else 0, NORM[norm],
                                    rtol, atol, dtol, int(maxit), int(x0 is not None), b, x, C.byref(its), C.byref(rn),
                                    hist.ctypes.data)
        return {"x": x, "iters": its.value, "rnorm": rn.value, "reason": int(reason),
                "history": hist[: its.value + 1].copy()}

    def __del__(self):
        try:
            if self._h:
                lib().orc_gmg_destroy(self._h)
                self._h = None
        except Exception:
            pass


def assemble_poisson32(n, widths, dt):
    """DBNG as int32 CSR straight from the cell widths (oracle/csrc/gmg.c:orc_assemble_poisson32)."""
    L = lib()
    dim = len(n)
    nx, ny = int(n[0]), int(n[1])
    nz = int(n[2]) if dim == 3 else 1
    L.orc_poisson_nnz.restype = C.c_int64
    L.orc_poisson_nnz.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64]
    nnz = int(L.orc_poisson_nnz(dim, nx, ny, nz))
    N = nx * ny * nz
    rp = np.empty(N + 1, dtype=np.int32)
    cl = np.empty(nnz, dtype=np.int32)
    vl = np.empty(nnz)
    w = [np.ascontiguousarray(a, dtype=np.float64) for a in widths]
    L.orc_assemble_poisson32.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, _f64p, _f64p, C.c_void_p, C.c_double,
                                         _i32p, _i32p, _f64p]
    L.orc_assemble_poisson32(dim, nx, ny, nz, w[0], w[1], w[2].ctypes.data if dim == 3 else None, float(dt), rp, cl, vl)
    return rp, cl, vl


def pcg_gmg32(gmg: "GMG", rp, cl, vl, b, rtol=1e-10, maxit=200):
    L = lib()
    L.orc_pcg_gmg32.restype = C.c_int
    L.orc_pcg_gmg32.argtypes = [C.c_void_p, C.c_int64, _i32p, _i32p, _f64p, C.c_double, C.c_int, _f64p, _f64p,
                                C.POINTER(C.c_int), C.POINTER(C.c_double)]
    n = len(rp) - 1
    x = np.empty(n)
    its, rn = C.c_int(0), C.c_double(0)
    reason = L.orc_pcg_gmg32(gmg._h, n, rp, cl, vl, float(rtol), int(maxit), np.ascontiguousarray(b), x, C.byref(its),
                             C.byref(rn))
    return {"x": x, "iters": its.value, "rnorm": rn.value, "reason": int(reason)}
