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


MPI_PREFIX = os.environ.get("PIB_MPI_PREFIX", "/opt/conda")  # MPICH 3.3.2 in this image


def build_mpi_example(verbose: bool = False):
    """examples/mpi/poisson_boxes_mpi: the PETSc-free `mpiexec -n P` launch through the C ABI.  Returns the
    binary's path, or None when the image has no MPI.  The MPI runtime's directory also holds an OLD libstdc++, so it must not
    get onto the binary's search path: the four libraries MPICH needs are reached through symlinks in examples/mpi/lib."""
    lib = os.path.join(MPI_PREFIX, "lib", "libmpi.so.12")
    if not (os.path.exists(lib) and os.path.exists(os.path.join(MPI_PREFIX, "include", "mpi.h"))):
        return None
    root = os.path.join(_HERE, "..")
    d = os.path.join(root, "examples", "mpi")
    priv = os.path.join(d, "lib")
    os.makedirs(priv, exist_ok=True)
    for name in ("libmpi.so.12", "libgfortran.so.4", "libgomp.so.1", "libquadmath.so.0"):
        src, dst = os.path.join(MPI_PREFIX, "lib", name), os.path.join(priv, name)
        if os.path.exists(src) and not os.path.lexists(dst):
            os.symlink(src, dst)
    out = os.path.join(d, "poisson_boxes_mpi")
    cmd = ["g++", "-std=c++14", "-Wall", "-I", os.path.join(root, "include"), "-I", os.path.join(MPI_PREFIX, "include"),
           os.path.join(d, "poisson_boxes_mpi.cpp"), "-L", _LIBDIR, "-lpetibm_amd", os.path.join(priv, "libmpi.so.12"),
           "-Wl,-rpath,$ORIGIN/lib", "-Wl,-rpath,$ORIGIN/../../petibm_amd/lib", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def mpiexec_path():
    p = os.path.join(MPI_PREFIX, "bin", "mpiexec")
    return p if os.path.exists(p) else None
