"""Build recipe of libpetibm_amd.so (hipcc, gfx950 only, in-tree)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc")
_LIBDIR = os.path.join(_HERE, "lib")
SOURCES = ["config.cpp", "capi.cpp", "structure.cpp", "partition.cpp", "redistribute.hip", "halo.hip", "kernels_spmv.hip", "krylov.hip", "assemble.hip", "gmg.hip", "dense.hip", "navierstokes.hip", "ibm.hip", "bn.hip", "velstencil.hip"]
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


def _headers_mtime() -> float:
    hs = [os.path.join(_SRC, f) for f in os.listdir(_SRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(_HERE, "..", "include", "petibm_amd.h"))
    return max(os.path.getmtime(h) for h in hs if os.path.exists(h))


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source of the backend (one object per source, stale ones only, in parallel) and link
    them into one shared library."""
    from concurrent.futures import ThreadPoolExecutor
    out = library_path()
    if not force and not _stale(out):
        return out
    objdir = os.path.join(_LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    hdr_t = _headers_mtime()
    jobs, objs = [], []
    for f in SOURCES:
        src = os.path.join(_SRC, f)
        obj = os.path.join(objdir, os.path.splitext(f)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc] + cflags + ["-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lrccl"])
    return out
