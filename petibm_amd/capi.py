"""ctypes binding of the C ABI in include/petibm_amd.h.

`import torch` happens BEFORE the library is loaded on purpose: torch ships its
own libamdhip64.so / librccl.so with the same SONAMEs as /opt/rocm's; loading
torch first makes the dynamic loader hand the already-loaded copies to
libpetibm_amd.so, so the process has ONE HIP runtime and ONE RCCL (bench.py
needs torch.distributed + torch.cuda.synchronize around the timed region).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from .build import library_path

_lib = None


class PibError(RuntimeError):
    """Non-zero return of a pib_* call; `.code` is the PETSc-numbered code."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[pib error {code}] {msg}")
        self.code = code
        self.message = msg


# error / reason constants mirrored from the header
ERR_MEM, ERR_SUP, ERR_ORDER, ERR_ARG_WRONG, ERR_ARG_OUTOFRANGE = 55, 56, 58, 62, 63
ERR_FILE_OPEN, ERR_LIB, ERR_CONV_FAILED, ERR_ARG_NULL = 65, 76, 82, 85
ERR_FILE_READ, ERR_MAT_LU_ZRPVT, ERR_ARG_UNKNOWN_TYPE, ERR_MAX_VALUE = 66, 71, 86, 99
NULLSPACE_NONE, NULLSPACE_CONSTANT, NULLSPACE_PINNED = 0, 1, 2
UID_BYTES = 128

_i64 = C.c_int64
_vp = C.c_void_p
_PROTOS = {
    "pib_last_error": (C.c_char_p, []),
    "pib_version": (C.c_int, []),
    "pib_comm_unique_id": (C.c_int, [_vp]),
    "pib_comm_peer_id": (C.c_int, [_vp]),
    "pib_comm_peer_id_ordered": (C.c_int, [_vp, C.c_int]),
    "pib_get_graph_replays": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "pib_comm_latency": (C.c_int, [_vp, _i64, C.c_int, C.POINTER(C.c_double)]),
    "pib_comm_loopback_create": (C.c_int, [C.c_int, _vp]),
    "pib_comm_loopback_destroy": (C.c_int, [_vp]),
    "pib_comm_selftest": (C.c_int, [C.c_int, _i64, _i64, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "pib_create": (C.c_int, [C.POINTER(_vp), C.c_char_p, C.c_char_p, C.c_int, C.c_int, _vp, C.c_int]),
    "pib_create_from_string": (C.c_int, [C.POINTER(_vp), C.c_char_p, C.c_char_p, C.c_int, C.c_int, _vp, C.c_int]),
    "pib_config_describe": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    "pib_slab_range": (C.c_int, [_i64, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "pib_destroy": (C.c_int, [_vp]),
    "pib_get_type": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "pib_describe": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "pib_set_csr": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "pib_set_csr_i32": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "pib_set_grid_hint": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "pib_set_periodic": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "pib_get_multigrid_levels": (C.c_int, [_vp, C.POINTER(C.c_int), _vp, C.c_int]),
    "pib_get_grid_structure": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp, C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]),
    "pib_get_velocity_structure": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp, _vp, C.POINTER(C.c_int)]),
    "pib_assemble_poisson": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_double, C.c_int]),
    "pib_assemble_poisson_bn": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int,
                                          C.c_int]),
    "pib_assemble_velocity": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double]),
    "pib_solve": (C.c_int, [_vp, _vp, _vp]),
    "pib_get_iters": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "pib_get_residual": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "pib_get_residual_at": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    "pib_get_reason": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "pib_mat_mult": (C.c_int, [_vp, _vp, _vp]),
    "pib_device_alloc": (C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    "pib_device_mem_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "pib_device_free": (C.c_int, [_vp, _vp]),
    "pib_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, _i64]),
    "pib_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, _i64]),
    "pib_synchronize": (C.c_int, [_vp]),
    "pib_get_csr": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), _vp, _vp, _vp]),
    "pib_ns_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double,
                                C.c_char_p, C.c_char_p, C.c_int]),
    "pib_ns_create_slab": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double,
                                     C.c_char_p, C.c_char_p, C.c_int, C.c_int, _vp, C.c_int]),
    "pib_ns_set_bn_order": (C.c_int, [_vp, C.c_int]),
    "pib_ns_get_vorticity": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "pib_ns_set_coupled": (C.c_int, [_vp, C.c_int]),
    "pib_ns_set_time_integration": (C.c_int, [_vp, C.c_char_p, C.c_char_p]),
    "pib_ns_history_term": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "pib_ns_sizes": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "pib_ns_set_state": (C.c_int, [_vp, _vp, _vp]),
    "pib_ns_get_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "pib_ns_advance": (C.c_int, [_vp, C.c_int]),
    "pib_ns_get_history": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pib_ns_set_history": (C.c_int, [_vp, _vp, _vp]),
    "pib_ns_describe_solver": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int]),
    "pib_ns_get_solver_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                         C.POINTER(C.c_double)]),
    "pib_ns_stage_timers": (C.c_int, [_vp, C.c_int]),
    "pib_ns_get_stage_times": (C.c_int, [_vp, _vp, C.POINTER(_i64)]),
    "pib_ns_stage_name": (C.c_char_p, [C.c_int]),
    "pib_ns_destroy": (C.c_int, [_vp]),
    "pib_ns_set_bodies": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_char_p, C.c_char_p]),
    "pib_ns_move_bodies": (C.c_int, [_vp, _vp, _vp]),
    "pib_ns_num_forces": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(C.c_int)]),
    "pib_ns_get_forces": (C.c_int, [_vp, _vp, _vp]),
    "pib_ns_set_forces": (C.c_int, [_vp, _vp]),
    "pib_ns_get_forces_solver_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "pib_ns_get_ib_operator": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64), _vp, _vp, _vp, _vp]),
    "pib_time_kernel": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "pib_get_counters": (C.c_int, [_vp, _vp]),
    "pib_get_staging_ms": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pib_get_product_format": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "pib_get_placement": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)


def load():
    """Load libpetibm_amd.so (fails loudly if it was not built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for this backend)")
    if os.environ.get("PIB_TORCH_FIRST", "1") != "0" or "torch" in sys.modules:
        import torch  # noqa: F401  (see module docstring: one HIP runtime per process)
    # (PIB_TORCH_FIRST=0 in a process that never imports torch: the library runs on /opt/rocm's own HIP / ROCr -- what a C or C++
    # application linking libpetibm_amd.so gets -- instead of the older runtime torch bundles; tools/abort_hunt.sh compares the two)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        raise PibError(code, load().pib_last_error().decode("utf-8", "replace"))


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a)  # raw device pointer


def config_describe(name: str, text: str) -> dict:
    """pib_config_describe as a dict (host-only; works without a GPU)."""
    buf = C.create_string_buffer(2048)
    check(load().pib_config_describe(name.encode(), text.encode(), buf, 2048))
    out = {}
    import shlex
    for tok in shlex.split(buf.value.decode()):
        k, _, v = tok.partition("=")
        out[k] = v
    return out
