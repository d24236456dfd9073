"""ctypes front-end of the CPU oracle (oracle/sdf_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported from tests/, from
``__graft_entry__.smoke()`` and from ``bench.py``'s ``cpu_baseline`` leg, and
from nowhere else.  It restates the reference's
``sdf_generation::ExtractSignedDistanceField`` (include/sdf_tools/sdf_generation.hpp:209-420)
on the CPU; it is the checker, never the product.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsdf_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/sdf_oracle.c with gcc (plain C, no dependencies)."""
    src = os.path.join(_HERE, "sdf_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["gcc", "-O3", "-std=c11", "-fPIC", "-shared", "-o", _LIB_PATH, src, "-lm"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, dbl, ci = ctypes.c_int64, ctypes.c_double, ctypes.c_int
        vp = ctypes.c_void_p
        L.sdf_oracle_extract.argtypes = [vp, i64, i64, i64, dbl, vp, vp, vp, vp]
        L.sdf_oracle_extract.restype = ci
        L.sdf_oracle_extract_vb.argtypes = [vp, i64, i64, i64, dbl, ci, vp, vp]
        L.sdf_oracle_extract_vb.restype = ci
        L.sdf_oracle_classify_cells.argtypes = [vp, i64, ci, vp]
        L.sdf_oracle_classify_cells.restype = None
        L.sdf_oracle_exact_edt.argtypes = [vp, ci, i64, i64, i64, vp]
        L.sdf_oracle_exact_edt.restype = ci
        L.sdf_oracle_brute_edt.argtypes = [vp, ci, i64, i64, i64, vp]
        L.sdf_oracle_brute_edt.restype = ci
        L.sdf_oracle_exact_sdf.argtypes = [vp, i64, i64, i64, dbl, ci, vp, vp, vp]
        L.sdf_oracle_exact_sdf.restype = ci
        _lib = L
    return _lib


def _mask(filled):
    m = np.ascontiguousarray(filled, dtype=np.uint8)
    if m.ndim != 3:
        raise ValueError("mask must be [nx, ny, nz] (z fastest)")
    return m


def reference_sdf(filled, resolution=1.0, add_virtual_border=False, want_dsq=False):
    """The reference algorithm (bucket-queue propagation, inexact on sparse scenes).

    filled: uint8/bool [nx, ny, nz].  Returns (sdf float32 [nx,ny,nz], (max, min))
    and, with want_dsq (no virtual border only), the two float64 d^2 fields
    (to-filled, to-free) of sdf_generation.hpp:242-243.
    """
    m = _mask(filled)
    nx, ny, nz = m.shape
    out = np.empty(m.shape, dtype=np.float32)
    ext = np.empty(2, dtype=np.float64)
    if want_dsq and not add_virtual_border:
        df = np.empty(m.shape, dtype=np.float64)
        de = np.empty(m.shape, dtype=np.float64)
        rc = lib().sdf_oracle_extract(m.ctypes.data, nx, ny, nz, float(resolution), out.ctypes.data,
                                      ext.ctypes.data, df.ctypes.data, de.ctypes.data)
        if rc:
            raise MemoryError("oracle allocation failed")
        return out, (float(ext[0]), float(ext[1])), df, de
    rc = lib().sdf_oracle_extract_vb(m.ctypes.data, nx, ny, nz, float(resolution),
                                     1 if add_virtual_border else 0, out.ctypes.data, ext.ctypes.data)
    if rc:
        raise MemoryError("oracle allocation failed")
    return out, (float(ext[0]), float(ext[1]))


def classify_cells(cells, unknown_is_filled=False):
    """collision_map.hpp:680-712 predicate over COLLISION_CELL records.

    cells: structured/2-column array of 8-byte records {float32 occupancy; uint32 component},
    shape [nx, ny, nz] (+ record).  Returns the uint8 mask."""
    c = np.ascontiguousarray(cells)
    n = c.size if c.dtype.itemsize == 8 else c.size // 2
    shape = c.shape if c.dtype.itemsize == 8 else c.shape[:-1]
    out = np.empty(shape, dtype=np.uint8)
    lib().sdf_oracle_classify_cells(c.ctypes.data, n, 1 if unknown_is_filled else 0, out.ctypes.data)
    return out


def exact_edt(seed_mask, seed_value=1):
    """Exact squared EDT to voxels with (mask != 0) == seed_value; -1 where no seed exists."""
    m = _mask(seed_mask)
    out = np.empty(m.shape, dtype=np.int64)
    rc = lib().sdf_oracle_exact_edt(m.ctypes.data, int(seed_value), *m.shape, out.ctypes.data)
    if rc:
        raise MemoryError
    return out


def brute_edt(seed_mask, seed_value=1):
    m = _mask(seed_mask)
    out = np.empty(m.shape, dtype=np.int64)
    lib().sdf_oracle_brute_edt(m.ctypes.data, int(seed_value), *m.shape, out.ctypes.data)
    return out


def exact_sdf(filled, resolution=1.0, add_virtual_border=False):
    """Exact signed field with the reference's merge arithmetic.  Returns
    (sdf float32, (max, min), signed int64 d^2 [+free, -filled])."""
    m = _mask(filled)
    out = np.empty(m.shape, dtype=np.float32)
    ext = np.empty(2, dtype=np.float64)
    dsq = np.empty(m.shape, dtype=np.int64)
    rc = lib().sdf_oracle_exact_sdf(m.ctypes.data, *m.shape, float(resolution),
                                    1 if add_virtual_border else 0,
                                    out.ctypes.data, ext.ctypes.data, dsq.ctypes.data)
    if rc:
        raise MemoryError
    return out, (float(ext[0]), float(ext[1])), dsq
