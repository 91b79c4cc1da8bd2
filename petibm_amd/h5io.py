"""Minimal HDF5 reader / writer over the C library (ctypes), for the files the reference writes with PetscViewerHDF5:

  grid.h5        groups u, v, [w,] p  with the 1-D gridline coordinates x, y[, z]     (src/mesh/cartesianmesh.cpp write())
  NNNNNNN.h5     datasets u, v, [w,] p shaped (nz,) ny, nx + attribute `time` on /p      (solutionsimple.cpp:229-260,
                                                                                          navierstokes.cpp:618-634,797-814)
  restart data   groups /convection/<i>, /diffusion/<i> and dataset /force               (navierstokes.cpp:637-686,
                                                                                          decoupledibpm.cpp:316-349)

h5py is not part of the image; libhdf5 (1.10) is.  Only what those files need: float64 datasets of any rank in nested
groups, float64 scalar attributes, create / append / read.  No HDF5 library -> ImportError with the reason (there is no
silent fallback to another format)."""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os

import numpy as np

_lib = None
_CANDIDATES = ("libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so",
               "/opt/conda/lib/libhdf5.so.103")
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL = 0, 0
hid_t, hsize_t, herr_t = C.c_int64, C.c_uint64, C.c_int


def lib():
    global _lib
    if _lib is not None:
        return _lib
    names = list(_CANDIDATES)
    found = C.util.find_library("hdf5")
    if found:
        names.insert(0, found)
    if os.environ.get("PIB_HDF5_LIBRARY"):
        names.insert(0, os.environ["PIB_HDF5_LIBRARY"])
    err = None
    for n in names:
        try:
            L = C.CDLL(n)
            break
        except OSError as e:  # noqa: PERF203
            err = e
    else:
        raise ImportError(f"no HDF5 C library found (tried {names}): {err}; set PIB_HDF5_LIBRARY")
    L.H5open()
    for f, res, args in (
        ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]),
        ("H5Fclose", herr_t, [hid_t]), ("H5Gcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
        ("H5Gopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Gclose", herr_t, [hid_t]),
        ("H5Lexists", C.c_int, [hid_t, C.c_char_p, hid_t]), ("H5Ldelete", herr_t, [hid_t, C.c_char_p, hid_t]),
        ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), ("H5Screate", hid_t, [C.c_int]),
        ("H5Sclose", herr_t, [hid_t]),
        ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dclose", herr_t, [hid_t]),
        ("H5Dwrite", herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        ("H5Dread", herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), ("H5Dget_space", hid_t, [hid_t]),
        ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]),
        ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        ("H5Acreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), ("H5Aopen", hid_t, [hid_t, C.c_char_p, hid_t]),
        ("H5Aexists", C.c_int, [hid_t, C.c_char_p]), ("H5Adelete", herr_t, [hid_t, C.c_char_p]),
        ("H5Awrite", herr_t, [hid_t, hid_t, C.c_void_p]), ("H5Aread", herr_t, [hid_t, hid_t, C.c_void_p]),
        ("H5Aclose", herr_t, [hid_t]), ("H5Eset_auto2", herr_t, [hid_t, C.c_void_p, C.c_void_p]),
    ):
        fn = getattr(L, f)
        fn.restype, fn.argtypes = res, args
    L.H5Eset_auto2(0, None, None)  # errors are reported through return codes -> Python exceptions
    L.NATIVE_DOUBLE = hid_t.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
    _lib = L
    return L


class H5Error(OSError):
    pass


def _chk(v, what):
    if v < 0:
        raise H5Error(f"HDF5: {what} failed")
    return v


class File:
    """mode 'w' (truncate), 'a' (create or append, replacing datasets of the same name), 'r'"""

    def __init__(self, path: str, mode: str = "r"):
        L = lib()
        p = os.fsencode(path)
        if mode == "w" or (mode == "a" and not os.path.exists(path)):
            self.id = _chk(L.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"creating {path}")
        elif mode == "a":
            self.id = _chk(L.H5Fopen(p, H5F_ACC_RDWR, H5P_DEFAULT), f"opening {path}")
        elif mode == "r":
            if not os.path.exists(path):
                raise H5Error(f'Could not find file "{path}"')
            self.id = _chk(L.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT), f"opening {path}")
        else:
            raise ValueError(mode)
        self.path = path

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        if self.id:
            lib().H5Fclose(self.id)
            self.id = 0

    def _parent(self, name: str, create: bool):
        """open (create) the groups leading to `name`; returns (group id or file id, leaf, ids to close)"""
        L = lib()
        parts = [q for q in name.split("/") if q]
        loc, opened = self.id, []
        for g in parts[:-1]:
            gb = g.encode()
            if L.H5Lexists(loc, gb, H5P_DEFAULT) > 0:
                loc = _chk(L.H5Gopen2(loc, gb, H5P_DEFAULT), f"opening group {g}")
            elif create:
                loc = _chk(L.H5Gcreate2(loc, gb, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"creating group {g}")
            else:
                for o in reversed(opened):
                    L.H5Gclose(o)
                raise H5Error(f"{self.path}: no group {g} on the way to {name}")
            opened.append(loc)
        return loc, parts[-1].encode(), opened

    def write(self, name: str, array) -> None:
        L = lib()
        a = np.ascontiguousarray(array, dtype=np.float64)
        loc, leaf, opened = self._parent(name, True)
        try:
            if L.H5Lexists(loc, leaf, H5P_DEFAULT) > 0:
                _chk(L.H5Ldelete(loc, leaf, H5P_DEFAULT), f"replacing {name}")
            dims = (hsize_t * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
            sp = _chk(L.H5Screate_simple(max(a.ndim, 1), dims, None), "dataspace")
            ds = _chk(L.H5Dcreate2(loc, leaf, L.NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"creating {name}")
            _chk(L.H5Dwrite(ds, L.NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data), f"writing {name}")
            L.H5Dclose(ds)
            L.H5Sclose(sp)
        finally:
            for o in reversed(opened):
                L.H5Gclose(o)

    def read(self, name: str) -> np.ndarray:
        L = lib()
        loc, leaf, opened = self._parent(name, False)
        try:
            if L.H5Lexists(loc, leaf, H5P_DEFAULT) <= 0:
                raise H5Error(f"{self.path}: no dataset {name}")
            ds = _chk(L.H5Dopen2(loc, leaf, H5P_DEFAULT), f"opening {name}")
            sp = L.H5Dget_space(ds)
            nd = L.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            L.H5Sget_simple_extent_dims(sp, dims, None)
            out = np.empty(tuple(int(d) for d in dims[:nd]) if nd else (), dtype=np.float64)
            _chk(L.H5Dread(ds, L.NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), f"reading {name}")
            L.H5Sclose(sp)
            L.H5Dclose(ds)
            return out
        finally:
            for o in reversed(opened):
                L.H5Gclose(o)

    def exists(self, name: str) -> bool:
        try:
            loc, leaf, opened = self._parent(name, False)
        except H5Error:
            return False
        ok = lib().H5Lexists(loc, leaf, H5P_DEFAULT) > 0
        for o in reversed(opened):
            lib().H5Gclose(o)
        return ok

    def write_attr(self, dataset: str, attr: str, value: float) -> None:
        """PetscViewerHDF5WriteAttribute(viewer, "/p", "time", PETSC_DOUBLE, &t)"""
        L = lib()
        loc, leaf, opened = self._parent(dataset, False)
        try:
            ds = _chk(L.H5Dopen2(loc, leaf, H5P_DEFAULT), f"opening {dataset}")
            ab = attr.encode()
            if L.H5Aexists(ds, ab) > 0:
                L.H5Adelete(ds, ab)
            sp = _chk(L.H5Screate(0), "scalar dataspace")  # H5S_SCALAR
            at = _chk(L.H5Acreate2(ds, ab, L.NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT), f"creating attribute {attr}")
            v = C.c_double(float(value))
            _chk(L.H5Awrite(at, L.NATIVE_DOUBLE, C.byref(v)), f"writing attribute {attr}")
            L.H5Aclose(at)
            L.H5Sclose(sp)
            L.H5Dclose(ds)
        finally:
            for o in reversed(opened):
                L.H5Gclose(o)

    def read_attr(self, dataset: str, attr: str) -> float:
        L = lib()
        loc, leaf, opened = self._parent(dataset, False)
        try:
            ds = _chk(L.H5Dopen2(loc, leaf, H5P_DEFAULT), f"opening {dataset}")
            at = _chk(L.H5Aopen(ds, attr.encode(), H5P_DEFAULT), f"opening attribute {attr}")
            v = C.c_double()
            _chk(L.H5Aread(at, L.NATIVE_DOUBLE, C.byref(v)), f"reading attribute {attr}")
            L.H5Aclose(at)
            L.H5Dclose(ds)
            return v.value
        finally:
            for o in reversed(opened):
                L.H5Gclose(o)
