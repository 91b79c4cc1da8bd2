"""Build recipe of libpetibm_amd.so (hipcc, gfx950 only, in-tree)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc")
_LIBDIR = os.path.join(_HERE, "lib")
SOURCES = ["config.cpp", "capi.cpp", "halo.hip", "kernels_spmv.hip", "krylov.hip", "assemble.hip", "gmg.hip", "navierstokes.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


def library_path() -> str:
    return os.path.join(_LIBDIR, "libpetibm_amd.so")


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(_SRC, f) for f in os.listdir(_SRC)]
    deps.append(os.path.join(_HERE, "..", "include", "petibm_amd.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source of the backend into one shared library."""
    out = library_path()
    if not force and not _stale(out):
        return out
    os.makedirs(_LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", out] + [os.path.join(_SRC, f) for f in SOURCES] + ["-lrccl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out
