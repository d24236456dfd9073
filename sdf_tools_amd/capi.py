"""ctypes binding of the C ABI in include/sdfgpu.h (libsdfgpu.so).

This is the thin Python face of the drop-in boundary; it carries no compute.
If the HIP library is missing or no GPU is usable every entry point raises --
there is deliberately no CPU fallback (the CPU oracle lives in oracle/ and is
test infrastructure only).
"""
import ctypes
import os

import numpy as np

from . import build as _build

SDFGPU_DSQ_INF = 1 << 30
_STATUS = {0: "OK", -1: "INVALID_ARGUMENT", -2: "HIP", -3: "UNSUPPORTED_SIZE", -4: "NO_DEVICE", -5: "UNRESOLVED", -6: "REDZONE"}

# every symbol include/sdfgpu.h declares (tests check that the library exports them all)
EXPORTS = [
    "sdfgpu_version", "sdfgpu_device_count", "sdfgpu_create", "sdfgpu_destroy", "sdfgpu_last_error",
    "sdfgpu_build", "sdfgpu_build_cells", "sdfgpu_build_device", "sdfgpu_build_cells_device",
    "sdfgpu_get_extrema", "sdfgpu_sweep_zy_device", "sdfgpu_sweep_x_device", "sdfgpu_extrema_from_dsq",
    "sdfgpu_gradient_device", "sdfgpu_debug_copy_zsweep", "sdfgpu_debug_copy_yzsweep", "sdfgpu_debug_flat_habit", "sdfgpu_set_tuning",
    "sdfgpu_set_profiling", "sdfgpu_get_stage_times", "sdfgpu_set_option", "sdfgpu_last_build_info", "sdfgpu_last_dense_certified",
    "sdfgpu_pack_bits_device", "sdfgpu_dense_ball_device", "sdfgpu_voxelize_points_device", "sdfgpu_build_tagged_cells", "sdfgpu_query_points_device", "sdfgpu_fold_extrema_device", "sdfgpu_slab_dense_phase",
    "sdfgpu_gradient", "sdfgpu_sweep_zy_tiered_device", "sdfgpu_sweep_x_lines_device", "sdfgpu_classify_cells_device",
    "sdfgpu_copy_to_host", "sdfgpu_copy_from_host", "sdfgpu_query_points", "sdfgpu_device_malloc", "sdfgpu_device_free",
    "sdfgpu_build_to_device", "sdfgpu_build_cells_to_device", "sdfgpu_upload_classified",
    "sdfgpu_build_bits_device", "sdfgpu_build_bits", "sdfgpu_voxelize_points_bits_device", "sdfgpu_debug_finish_table", "sdfgpu_redzone_check",
]


class SdfGpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("sdfgpu %s (%d): %s" % (_STATUS.get(code, "?"), code, message))
        self.code = code


_lib = None


def load_library():
    """dlopen sdf_tools_amd/libsdfgpu.so (must have been built: see build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SDFGPU_LIB") or _build.LIB        # (SDFGPU_LIB: a profiling build of the same library, tools/ only)
    if not os.path.exists(path):
        raise ImportError("libsdfgpu.so is not built (run `python -m sdf_tools_amd.build`); "
                          "there is no CPU fallback for the SDF build path")
    # torch bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  Import it first so the
    # dynamic loader binds libsdfgpu.so to the HIP runtime torch uses: streams and device pointers
    # are then shared by both.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    L = ctypes.CDLL(path)
    i64, dbl, ci, vp, sz = ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    u32 = ctypes.c_uint32
    L.sdfgpu_version.restype = ctypes.c_char_p
    L.sdfgpu_device_count.restype = ci
    L.sdfgpu_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.sdfgpu_destroy.argtypes = [vp]
    L.sdfgpu_last_error.argtypes = [vp]
    L.sdfgpu_last_error.restype = ctypes.c_char_p
    L.sdfgpu_build.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_build_cells.argtypes = [vp, vp, sz, sz, ci, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_build_tagged_cells.argtypes = [vp, vp, sz, sz, sz, ci, vp, i64, ci, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_query_points_device.argtypes = [vp, vp, i64, i64, i64, dbl, vp, vp, ctypes.c_float, vp, i64, ci, vp, vp, vp, vp]
    L.sdfgpu_fold_extrema_device.argtypes = [vp, vp, vp]
    L.sdfgpu_query_points.argtypes = [vp, vp, i64, i64, i64, dbl, vp, vp, ctypes.c_float, vp, i64, ci, vp, vp, vp]
    L.sdfgpu_device_malloc.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.sdfgpu_device_free.argtypes = [vp, vp]
    L.sdfgpu_build_to_device.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_build_cells_to_device.argtypes = [vp, vp, sz, sz, ci, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_slab_dense_phase.argtypes = [vp, ci, vp, i64, i64, i64, vp, i64, i64, dbl, vp, vp, vp]
    L.sdfgpu_build_device.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, vp]
    L.sdfgpu_build_cells_device.argtypes = [vp, vp, sz, sz, ci, i64, i64, i64, dbl, ci, vp, vp]
    L.sdfgpu_get_extrema.argtypes = [vp, vp, vp]
    L.sdfgpu_build_bits_device.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, vp]
    L.sdfgpu_build_bits.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_voxelize_points_bits_device.argtypes = [vp, vp, i64, vp, dbl, i64, i64, i64, vp, ci, vp]
    L.sdfgpu_sweep_zy_device.argtypes = [vp, vp, i64, i64, i64, vp, vp]
    L.sdfgpu_classify_cells_device.argtypes = [vp, vp, sz, sz, ci, i64, vp, vp]
    L.sdfgpu_sweep_zy_tiered_device.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp]
    L.sdfgpu_sweep_x_lines_device.argtypes = [vp, vp, i64, i64, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_sweep_x_device.argtypes = [vp, vp, i64, i64, i64, i64, i64, ci, ci, i64, i64, dbl, ci, vp, vp, vp, vp]
    L.sdfgpu_pack_bits_device.argtypes = [vp, vp, i64, i64, vp, vp]
    L.sdfgpu_dense_ball_device.argtypes = [vp, vp, i64, i64, i64, i64, i64, dbl, vp, vp, vp, vp]
    L.sdfgpu_voxelize_points_device.argtypes = [vp, vp, i64, vp, dbl, i64, i64, i64, vp, ci, vp]
    L.sdfgpu_extrema_from_dsq.argtypes = [u32, u32, dbl, vp, vp]
    L.sdfgpu_gradient_device.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, ci, vp]
    L.sdfgpu_gradient.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, ci]
    L.sdfgpu_debug_finish_table.argtypes = [vp, vp, i64, dbl, ci, vp]
    L.sdfgpu_redzone_check.argtypes = [vp, vp]
    L.sdfgpu_debug_copy_zsweep.argtypes = [vp, vp, i64]
    L.sdfgpu_debug_copy_yzsweep.argtypes = [vp, vp, i64]
    L.sdfgpu_debug_flat_habit.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.sdfgpu_set_tuning.argtypes = [vp, ci, ci]
    L.sdfgpu_set_option.argtypes = [vp, ctypes.c_char_p, ci]
    L.sdfgpu_last_build_info.argtypes = [vp, vp]
    L.sdfgpu_last_dense_certified.argtypes = [vp, vp]
    L.sdfgpu_set_profiling.argtypes = [vp, ci]
    L.sdfgpu_copy_to_host.argtypes = [vp, vp, vp, ctypes.c_size_t, vp]
    L.sdfgpu_copy_from_host.argtypes = [vp, vp, vp, ctypes.c_size_t, vp]
    L.sdfgpu_upload_classified.argtypes = [vp, vp, vp, sz, sz, ci, i64, vp, vp]
    L.sdfgpu_get_stage_times.argtypes = [vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int or name not in ("sdfgpu_version", "sdfgpu_last_error"):
            fn.restype = ci
    _lib = L
    return L


def pack_bits_host(filled):
    """uint8 / bool occupancy (any shape) -> the linear bit field of sdfgpu_build_bits*: uint32[ceil(n / 32)], bit (v & 31) of
    word (v >> 5) = voxel v in C order."""
    m = np.ascontiguousarray(filled).reshape(-1) != 0
    b = np.packbits(m, bitorder="little")
    b = np.concatenate([b, np.zeros((-b.size) % 4, np.uint8)])
    return b.view("<u4").astype(np.uint32, copy=False)


def device_count():
    return int(load_library().sdfgpu_device_count())


def extrema_from_dsq(max_dsq_free, max_dsq_filled, resolution):
    """Host helper (no GPU needed): (max, min) from the integer maxima, as sdf_generation.hpp:246-269."""
    out = (ctypes.c_double * 2)()
    load_library().sdfgpu_extrema_from_dsq(int(max_dsq_free), int(max_dsq_filled), float(resolution),
                                           ctypes.byref(out, 0), ctypes.byref(out, 8))
    return float(out[0]), float(out[1])


class SdfGpu:
    """One context on one GPU (wraps sdfgpu_create / sdfgpu_destroy)."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.sdfgpu_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise SdfGpuError(rc, self._lib.sdfgpu_last_error(None).decode())
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdfgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SdfGpuError(rc, self._lib.sdfgpu_last_error(self._h).decode())

    # ---- host-buffer API -------------------------------------------------
    def build(self, filled, resolution=1.0, add_virtual_border=False):
        """filled: uint8/bool [nx,ny,nz].  Returns (sdf float32 [nx,ny,nz], (max, min))."""
        m = np.ascontiguousarray(filled, dtype=np.uint8)
        if m.ndim != 3:
            raise ValueError("mask must be [nx, ny, nz]")
        out = np.empty(m.shape, dtype=np.float32)
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_build(self._h, m.ctypes.data, *m.shape, float(resolution),
                                           int(bool(add_virtual_border)), out.ctypes.data,
                                           ctypes.byref(ext, 0), ctypes.byref(ext, 8)))
        return out, (float(ext[0]), float(ext[1]))

    def build_cells(self, cells, shape, cell_stride=8, occupancy_offset=0, unknown_is_filled=False,
                    resolution=1.0, add_virtual_border=False):
        """cells: raw COLLISION_CELL records (any contiguous array of nx*ny*nz*cell_stride bytes)."""
        c = np.ascontiguousarray(cells)
        nx, ny, nz = (int(s) for s in shape)
        if c.nbytes != nx * ny * nz * cell_stride:
            raise ValueError("cells buffer size does not match shape * cell_stride")
        out = np.empty((nx, ny, nz), dtype=np.float32)
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_build_cells(self._h, c.ctypes.data, cell_stride, occupancy_offset,
                                                 int(bool(unknown_is_filled)), nx, ny, nz, float(resolution),
                                                 int(bool(add_virtual_border)), out.ctypes.data,
                                                 ctypes.byref(ext, 0), ctypes.byref(ext, 8)))
        return out, (float(ext[0]), float(ext[1]))

    def build_tagged_cells(self, cells, shape, object_mode=0, object_ids=(), unknown_is_filled=False, resolution=1.0,
                           add_virtual_border=False, cell_stride=16, occupancy_offset=0, object_id_offset=8):
        """cells: raw TAGGED_OBJECT_COLLISION_CELL records; object_mode 0 any / 1 named (id > 0) / 2 id list.
        cells=None re-uses the records the previous call on this context uploaded (one field per object: upload once)."""
        nx, ny, nz = (int(s) for s in shape)
        c = None
        if cells is not None:
            c = np.ascontiguousarray(cells)
            if c.nbytes != nx * ny * nz * cell_stride:
                raise ValueError("cells buffer size does not match shape * cell_stride")
        ids = np.ascontiguousarray(np.asarray(object_ids, dtype=np.uint32))
        out = np.empty((nx, ny, nz), dtype=np.float32)
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_build_tagged_cells(
            self._h, c.ctypes.data if c is not None else None, cell_stride, occupancy_offset, object_id_offset, int(object_mode),
            ids.ctypes.data if ids.size else None, int(ids.size), int(bool(unknown_is_filled)), nx, ny, nz,
            float(resolution), int(bool(add_virtual_border)), out.ctypes.data, ctypes.byref(ext, 0), ctypes.byref(ext, 8)))
        return out, (float(ext[0]), float(ext[1]))

    def query_points_device(self, d_sdf, shape, resolution, d_points, n_points, d_distance=0, d_gradient=0, d_flags=0,
                            world_to_grid=None, rotation=None, oob_value=float("inf"), enable_edge_gradients=False,
                            stream=0):
        """Batched EstimateDistance / GetGradient at n world-frame points (device pointers as ints)."""
        nx, ny, nz = (int(s) for s in shape)
        w = None if world_to_grid is None else (ctypes.c_double * 12)(*np.asarray(world_to_grid, np.float64).reshape(-1)[:12])
        r = None if rotation is None else (ctypes.c_double * 9)(*np.asarray(rotation, np.float64).reshape(-1)[:9])
        self._check(self._lib.sdfgpu_query_points_device(
            self._h, d_sdf, nx, ny, nz, float(resolution), w, r, float(oob_value), d_points, int(n_points),
            int(bool(enable_edge_gradients)), d_distance or None, d_gradient or None, d_flags or None, stream or None))

    # ---- host input -> device-resident field; host points -> host answers (what the C++ mirror's DeviceSignedDistanceField uses)
    def device_malloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self._lib.sdfgpu_device_malloc(self._h, int(nbytes), ctypes.byref(p)))
        return int(p.value)

    def device_free(self, ptr):
        self._check(self._lib.sdfgpu_device_free(self._h, ctypes.c_void_p(int(ptr))))

    def build_to_device(self, filled, d_out, resolution=1.0, add_virtual_border=False):
        """Host mask [nx, ny, nz] -> fp32 field at device address d_out (no download).  Returns (max, min)."""
        m = np.ascontiguousarray(filled, dtype=np.uint8)
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_build_to_device(self._h, m.ctypes.data, *m.shape, float(resolution),
                                                     int(bool(add_virtual_border)), ctypes.c_void_p(int(d_out)),
                                                     ctypes.byref(ext, 0), ctypes.byref(ext, 8)))
        return float(ext[0]), float(ext[1])

    def query_points(self, d_sdf, shape, resolution, points, world_to_grid=None, rotation=None, oob_value=float("inf"),
                     enable_edge_gradients=False):
        """Batched EstimateDistance + GetGradient at host points [n, 3] float64 against a field in HBM.
        Returns (distance [n], gradient [n, 3], flags [n]) as numpy arrays."""
        nx, ny, nz = (int(v) for v in shape)
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        n = pts.shape[0]
        dist, grad, flags = np.empty(n, np.float64), np.empty((n, 3), np.float64), np.empty(n, np.uint8)
        w = None if world_to_grid is None else (ctypes.c_double * 12)(*np.asarray(world_to_grid, np.float64).reshape(-1)[:12])
        r = None if rotation is None else (ctypes.c_double * 9)(*np.asarray(rotation, np.float64).reshape(-1)[:9])
        self._check(self._lib.sdfgpu_query_points(self._h, ctypes.c_void_p(int(d_sdf)), nx, ny, nz, float(resolution), w, r,
                                                  float(oob_value), pts.ctypes.data, n, int(bool(enable_edge_gradients)),
                                                  dist.ctypes.data, grad.ctypes.data, flags.ctypes.data))
        return dist, grad, flags

    def slab_dense_phase(self, phase, d_mask_slab, nxs, ny, nz, d_bits_ext, halo_lo, halo_hi, resolution, d_out, d_small,
                         stream=0):
        self._check(self._lib.sdfgpu_slab_dense_phase(self._h, int(phase), d_mask_slab, int(nxs), int(ny), int(nz), d_bits_ext,
                                                      int(halo_lo), int(halo_hi), float(resolution), d_out, d_small,
                                                      stream or None))

    def fold_extrema_device(self, d_maxdsq, stream=0):
        self._check(self._lib.sdfgpu_fold_extrema_device(self._h, d_maxdsq, stream or None))

    # ---- device-pointer API (raw integers: tensor.data_ptr(), stream.cuda_stream) -------------
    def build_device(self, d_filled, shape, d_out, resolution=1.0, add_virtual_border=False, stream=0):
        nx, ny, nz = (int(s) for s in shape)
        self._check(self._lib.sdfgpu_build_device(self._h, d_filled, nx, ny, nz, float(resolution),
                                                  int(bool(add_virtual_border)), d_out, stream or None))

    def build_bits_device(self, d_bits, shape, d_out, resolution=1.0, add_virtual_border=False, stream=0):
        """d_bits: device pointer of the linear bit field (bit v & 31 of word v >> 5 = voxel v), ceil(n / 32) uint32 words."""
        nx, ny, nz = (int(s) for s in shape)
        self._check(self._lib.sdfgpu_build_bits_device(self._h, d_bits, nx, ny, nz, float(resolution),
                                                       int(bool(add_virtual_border)), d_out, stream or None))

    def build_bits(self, bits, shape, resolution=1.0, add_virtual_border=False):
        """bits: host uint32 array of ceil(n / 32) words (see pack_bits_host).  Returns (sdf float32 [nx,ny,nz], (max, min))."""
        nx, ny, nz = (int(s) for s in shape)
        b = np.ascontiguousarray(bits, dtype=np.uint32)
        if b.size != (nx * ny * nz + 31) // 32:
            raise ValueError("bit field must hold ceil(nx*ny*nz / 32) words")
        out = np.empty((nx, ny, nz), dtype=np.float32)
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_build_bits(self._h, b.ctypes.data, nx, ny, nz, float(resolution),
                                                int(bool(add_virtual_border)), out.ctypes.data,
                                                ctypes.byref(ext, 0), ctypes.byref(ext, 8)))
        return out, (float(ext[0]), float(ext[1]))

    def voxelize_points_bits_device(self, d_points, n_points, origin, resolution, shape, d_bits, clear_first=True, stream=0):
        nx, ny, nz = (int(s) for s in shape)
        o = (ctypes.c_double * 3)(*[float(v) for v in origin])
        self._check(self._lib.sdfgpu_voxelize_points_bits_device(self._h, d_points, int(n_points), o, float(resolution),
                                                                 nx, ny, nz, d_bits, int(bool(clear_first)), stream or None))

    def build_cells_device(self, d_cells, shape, d_out, cell_stride=8, occupancy_offset=0,
                           unknown_is_filled=False, resolution=1.0, add_virtual_border=False, stream=0):
        nx, ny, nz = (int(s) for s in shape)
        self._check(self._lib.sdfgpu_build_cells_device(self._h, d_cells, cell_stride, occupancy_offset,
                                                        int(bool(unknown_is_filled)), nx, ny, nz,
                                                        float(resolution), int(bool(add_virtual_border)),
                                                        d_out, stream or None))

    def get_extrema(self):
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_get_extrema(self._h, ctypes.byref(ext, 0), ctypes.byref(ext, 8)))
        return float(ext[0]), float(ext[1])

    def sweep_zy_device(self, d_filled, slab_shape, d_plane_dsq, stream=0):
        nxs, ny, nz = (int(s) for s in slab_shape)
        self._check(self._lib.sdfgpu_sweep_zy_device(self._h, d_filled, nxs, ny, nz, d_plane_dsq, stream or None))

    def sweep_zy_tiered_device(self, d_filled, slab_shape, d_plane_dsq, d_far=0, stream=0):
        nxs, ny, nz = (int(s) for s in slab_shape)
        self._check(self._lib.sdfgpu_sweep_zy_tiered_device(self._h, d_filled, nxs, ny, nz, d_plane_dsq, d_far or None,
                                                            stream or None))

    def sweep_x_lines_device(self, d_plane_dsq, nx, nys, nz, y_global, ny_global, resolution, add_virtual_border, d_out,
                             d_maxdsq, stream=0):
        self._check(self._lib.sdfgpu_sweep_x_lines_device(self._h, d_plane_dsq, int(nx), int(nys), int(nz), int(y_global),
                                                          int(ny_global), float(resolution), int(bool(add_virtual_border)),
                                                          d_out, d_maxdsq, stream or None))

    def sweep_x_device(self, d_plane_dsq, halo_lo, nxs, halo_hi, ny, nz, lo_truncated, hi_truncated,
                       x_global, nx_global, resolution, add_virtual_border, d_out, d_maxdsq, d_status, stream=0):
        self._check(self._lib.sdfgpu_sweep_x_device(self._h, d_plane_dsq, int(halo_lo), int(nxs), int(halo_hi),
                                                    int(ny), int(nz), int(bool(lo_truncated)),
                                                    int(bool(hi_truncated)), int(x_global), int(nx_global),
                                                    float(resolution), int(bool(add_virtual_border)),
                                                    d_out, d_maxdsq, d_status or None, stream or None))

    def pack_bits_device(self, d_filled, n_rows, nz, d_bits, stream=0):
        self._check(self._lib.sdfgpu_pack_bits_device(self._h, d_filled, int(n_rows), int(nz), d_bits, stream or None))

    def dense_ball_device(self, d_bits, rows_x, out_lo, out_hi, ny, nz, resolution, d_out, d_maxdsq, d_uncertified,
                          stream=0):
        self._check(self._lib.sdfgpu_dense_ball_device(self._h, d_bits, int(rows_x), int(out_lo), int(out_hi), int(ny),
                                                       int(nz), float(resolution), d_out, d_maxdsq, d_uncertified,
                                                       stream or None))

    def voxelize_points_device(self, d_points, n_points, origin, resolution, shape, d_mask, clear_first=True, stream=0):
        nx, ny, nz = (int(s) for s in shape)
        o = (ctypes.c_double * 3)(*[float(v) for v in origin])
        self._check(self._lib.sdfgpu_voxelize_points_device(self._h, d_points, int(n_points), o, float(resolution),
                                                            nx, ny, nz, d_mask, int(bool(clear_first)), stream or None))

    def gradient_device(self, d_sdf, shape, d_out, resolution=1.0, enable_edge_gradients=True, f64=True, stream=0):
        nx, ny, nz = (int(s) for s in shape)
        self._check(self._lib.sdfgpu_gradient_device(self._h, d_sdf, nx, ny, nz, float(resolution),
                                                     int(bool(enable_edge_gradients)), d_out, int(bool(f64)),
                                                     stream or None))

    def copy_to_host(self, dst, d_src, stream=0):
        """device pointer -> numpy array (may be untouched memory) at link rate; returns dst."""
        self._check(self._lib.sdfgpu_copy_to_host(self._h, dst.ctypes.data, int(d_src), dst.nbytes, int(stream)))
        return dst

    def copy_from_host(self, d_dst, src, stream=0):
        """numpy array -> device pointer at link rate."""
        a = np.ascontiguousarray(src)
        self._check(self._lib.sdfgpu_copy_from_host(self._h, int(d_dst), a.ctypes.data, a.nbytes, int(stream)))

    def upload_classified(self, d_mask, filled=None, cells=None, cell_stride=8, occupancy_offset=0, unknown_is_filled=False,
                          stream=0):
        """Host mask (uint8 array) or raw cell records -> device byte mask (0 / 1), classified on the host, 1 bit per voxel
        over PCIe (sdfgpu_upload_classified)."""
        if (filled is None) == (cells is None):
            raise ValueError("exactly one of filled / cells")
        if filled is not None:
            a = np.ascontiguousarray(filled, dtype=np.uint8)
            n = a.size
            self._check(self._lib.sdfgpu_upload_classified(self._h, a.ctypes.data, None, 0, 0, 0, n, int(d_mask), int(stream) or None))
        else:
            a = np.ascontiguousarray(cells)
            n = a.nbytes // cell_stride
            self._check(self._lib.sdfgpu_upload_classified(self._h, None, a.ctypes.data, cell_stride, occupancy_offset,
                                                           int(bool(unknown_is_filled)), n, int(d_mask), int(stream) or None))
        return n

    def gradient(self, sdf, resolution=1.0, enable_edge_gradients=True, f64=True):
        """Host-buffer full-grid gradient: sdf float32 [nx,ny,nz] -> [nx,ny,nz,3] (NaN where the reference has none)."""
        f = np.ascontiguousarray(sdf, dtype=np.float32)
        out = np.empty(f.shape + (3,), dtype=np.float64 if f64 else np.float32)
        self._check(self._lib.sdfgpu_gradient(self._h, f.ctypes.data, *f.shape, float(resolution),
                                              int(bool(enable_edge_gradients)), out.ctypes.data, int(bool(f64))))
        return out

    def redzone_check(self, stream=0):
        """Red-zone mode (SDFGPU_REDZONE=1 / option "redzone"): check every canary now; raises SdfGpuError(REDZONE) naming the buffer."""
        self._check(self._lib.sdfgpu_redzone_check(self._h, stream or None))

    def debug_finish_table(self, d_out, n, resolution, fast=True):
        """float(sqrt(double(D)) * resolution) for D = 0 .. n - 1 into the device buffer d_out: the fp32 form of sdfgpu_finish.hpp
        (fast) or the x sweep's fp64 sequence; returns the number of lanes of the fp32 form that took the fp64 sequence."""
        c = ctypes.c_uint32()
        self._check(self._lib.sdfgpu_debug_finish_table(self._h, d_out, int(n), float(resolution), int(bool(fast)), ctypes.byref(c)))
        return int(c.value)

    def debug_zsweep(self, shape):
        out = np.empty(shape, dtype=np.int16)
        self._check(self._lib.sdfgpu_debug_copy_zsweep(self._h, out.ctypes.data, out.size))
        return out

    def debug_yzsweep(self, shape):
        out = np.empty(shape, dtype=np.int32)
        self._check(self._lib.sdfgpu_debug_copy_yzsweep(self._h, out.ctypes.data, out.size))
        return out

    def debug_flat_habit(self):
        """(score, gate) of the two-valued tiles' device-side habit (include/sdfgpu.h); synchronises."""
        sc, g = ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.sdfgpu_debug_flat_habit(self._h, ctypes.byref(sc), ctypes.byref(g)))
        return sc.value, g.value

    def set_option(self, name, value):
        self._check(self._lib.sdfgpu_set_option(self._h, name.encode(), int(value)))

    def last_build_info(self):
        v = ctypes.c_int()
        self._check(self._lib.sdfgpu_last_build_info(self._h, ctypes.byref(v)))
        return {"fused_zy": bool(v.value & 1), "plane16": bool(v.value & 2), "dense": bool(v.value & 4),
                "standby_far": bool(v.value & 8), "dense3": bool(v.value & 16), "dense3_staged": bool(v.value & 32),
                "far_predicted": bool(v.value & 64)}

    def last_path(self):
        """{'dense_certified', 'far_y', 'far_x'} of the last build (synchronises)."""
        v = ctypes.c_int()
        self._check(self._lib.sdfgpu_last_dense_certified(self._h, ctypes.byref(v)))
        why = (v.value >> 8) & 0xff
        names = ("one_class_tile", "wave_all_undecided", "wave_too_many_undecided", "tile_over_fixup_cap", "beyond_fixup_reach", "beyond_ball", "too_sparse_for_the_shell_pass")
        out = {"dense_certified": bool(v.value & 1), "far_y": bool(v.value & 2), "far_x": bool(v.value & 4)}
        if why:
            out["dense_gave_up"] = [n for k, n in enumerate(names) if why & (1 << k)]
        return out

    def last_dense_certified(self):
        return self.last_path()["dense_certified"]

    def last_build_fused_zy(self):
        return self.last_build_info()["fused_zy"]

    def set_profiling(self, enable=True):
        """False/0 off, True/1 every stage, 2 only the dense ball kernel (two events per build), 3 = 2 on every
        4th build."""
        self._check(self._lib.sdfgpu_set_profiling(self._h, int(enable)))

    def get_stage_times(self):
        """(ms_sum[7] for pack / dense ball / z / y-or-zy / envelope y / x / envelope x, builds) since the
        last call; synchronises."""
        ms = (ctypes.c_double * 7)()
        n = ctypes.c_int64()
        self._check(self._lib.sdfgpu_get_stage_times(self._h, ms, ctypes.byref(n)))
        return [float(v) for v in ms], int(n.value)

    def set_tuning(self, rows_per_chunk_y=0, rows_per_chunk_x=0):
        self._check(self._lib.sdfgpu_set_tuning(self._h, int(rows_per_chunk_y), int(rows_per_chunk_x)))


# ---- multi-GPU C ABI (include/sdfgpu_multi.h, libsdfgpu_multi.so) -------------------------------------------------
MULTI_EXPORTS = [
    "sdfgpu_multi_create", "sdfgpu_multi_destroy", "sdfgpu_multi_last_error", "sdfgpu_multi_ranks",
    "sdfgpu_multi_slab_range", "sdfgpu_multi_build", "sdfgpu_multi_build_cells", "sdfgpu_multi_build_device",
    "sdfgpu_multi_last_path", "sdfgpu_multi_set_option", "sdfgpu_multi_last_stats", "sdfgpu_multi_last_host_us",
]
_multi_lib = None


def load_multi_library():
    """dlopen sdf_tools_amd/libsdfgpu_multi.so (x-slab multi-GPU build over RCCL; no CPU fallback)."""
    global _multi_lib
    if _multi_lib is not None:
        return _multi_lib
    load_library()                               # libsdfgpu.so (and torch's HIP / RCCL runtimes) first
    path = _build.LIB_MULTI
    if not os.path.exists(path):
        raise ImportError("libsdfgpu_multi.so is not built (run `python -m sdf_tools_amd.build`)")
    L = ctypes.CDLL(path)
    i64, dbl, ci, vp, sz = ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    L.sdfgpu_multi_create.argtypes = [ci, vp, ctypes.POINTER(vp)]
    L.sdfgpu_multi_destroy.argtypes = [vp]
    L.sdfgpu_multi_last_error.argtypes = [vp]
    L.sdfgpu_multi_last_error.restype = ctypes.c_char_p
    L.sdfgpu_multi_ranks.argtypes = [vp]
    L.sdfgpu_multi_slab_range.argtypes = [vp, i64, ci, vp, vp]
    L.sdfgpu_multi_build.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_multi_build_cells.argtypes = [vp, vp, sz, sz, ci, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_multi_build_device.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, vp, vp]
    L.sdfgpu_multi_last_path.argtypes = [vp, vp]
    L.sdfgpu_multi_last_stats.argtypes = [vp, vp, vp]
    L.sdfgpu_multi_last_host_us.argtypes = [vp, vp, vp]
    L.sdfgpu_multi_set_option.argtypes = [vp, ctypes.c_char_p, ci]
    for name in MULTI_EXPORTS:
        if name != "sdfgpu_multi_last_error":
            getattr(L, name).restype = ci
    _multi_lib = L
    return L


class MultiSdfGpu:
    """n ranks (one per GPU; the same GPU may be named several times for single-GPU testing) behind sdfgpu_multi_*."""

    def __init__(self, n_ranks, devices=None):
        self._lib = load_multi_library()
        h = ctypes.c_void_p()
        devs = None
        if devices is not None:
            devs = (ctypes.c_int * n_ranks)(*[int(d) for d in devices])
        rc = self._lib.sdfgpu_multi_create(int(n_ranks), devs, ctypes.byref(h))
        if rc != 0:
            raise SdfGpuError(rc, self._lib.sdfgpu_multi_last_error(None).decode())
        self._h = h
        self.n_ranks = int(n_ranks)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdfgpu_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SdfGpuError(rc, self._lib.sdfgpu_multi_last_error(self._h).decode())

    def slab_range(self, nx, rank):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.sdfgpu_multi_slab_range(self._h, int(nx), int(rank), ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def set_option(self, name, value):
        self._check(self._lib.sdfgpu_multi_set_option(self._h, name.encode(), int(value)))

    def last_path(self):
        v = ctypes.c_int()
        self._check(self._lib.sdfgpu_multi_last_path(self._h, ctypes.byref(v)))
        return {"dense_certified": bool(v.value & 1), "whole_lines": bool(v.value & 2), "rccl": bool(v.value & 4)}

    def last_stats(self):
        """{'host_reads': status-block round trips of the last build, 'mispredictions': general builds since creation
        whose predicted x sweep had to be redone, 'host_us_max_rank' / 'host_us_sum': host time of the last build inside API
        calls -- the slowest rank thread / the sum over the rank threads (sdfgpu_multi_last_host_us)}."""
        a, b = ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.sdfgpu_multi_last_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        mx, sm = ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.sdfgpu_multi_last_host_us(self._h, ctypes.byref(mx), ctypes.byref(sm)))
        return {"host_reads": int(a.value), "mispredictions": int(b.value),
                "host_us_max_rank": float(mx.value), "host_us_sum": float(sm.value)}

    def build(self, filled, resolution=1.0, add_virtual_border=False):
        m = np.ascontiguousarray(filled, dtype=np.uint8)
        out = np.empty(m.shape, dtype=np.float32)
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_multi_build(self._h, m.ctypes.data, *m.shape, float(resolution),
                                                 int(bool(add_virtual_border)), out.ctypes.data,
                                                 ctypes.byref(ext, 0), ctypes.byref(ext, 8)))
        return out, (float(ext[0]), float(ext[1]))

    def build_cells(self, cells, shape, cell_stride=8, occupancy_offset=0, unknown_is_filled=False, resolution=1.0,
                    add_virtual_border=False):
        c = np.ascontiguousarray(cells)
        nx, ny, nz = (int(s) for s in shape)
        out = np.empty((nx, ny, nz), dtype=np.float32)
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_multi_build_cells(self._h, c.ctypes.data, cell_stride, occupancy_offset,
                                                       int(bool(unknown_is_filled)), nx, ny, nz, float(resolution),
                                                       int(bool(add_virtual_border)), out.ctypes.data,
                                                       ctypes.byref(ext, 0), ctypes.byref(ext, 8)))
        return out, (float(ext[0]), float(ext[1]))

    def build_device(self, d_mask_slabs, shape, d_out_slabs, resolution=1.0, add_virtual_border=False):
        """d_mask_slabs / d_out_slabs: per-rank device pointers (ints) of the x slabs."""
        nx, ny, nz = (int(s) for s in shape)
        pm = (ctypes.c_void_p * self.n_ranks)(*[int(p) for p in d_mask_slabs])
        po = (ctypes.c_void_p * self.n_ranks)(*[int(p) for p in d_out_slabs])
        ext = (ctypes.c_double * 2)()
        self._check(self._lib.sdfgpu_multi_build_device(self._h, pm, nx, ny, nz, float(resolution),
                                                        int(bool(add_virtual_border)), po, ctypes.byref(ext, 0),
                                                        ctypes.byref(ext, 8)))
        return float(ext[0]), float(ext[1])
