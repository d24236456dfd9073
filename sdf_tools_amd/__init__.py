"""sdf_tools_amd -- MI355X-native build of sdf_tools' signed-distance-field hot path.

  capi        ctypes binding of the C ABI (include/sdfgpu.h, libsdfgpu.so: HIP kernels for gfx950)
  pysdf_tools pybind11 module with the reference's Python surface (src/sdf_tools/bindings.cpp)
  utils_2d / utils_3d   numpy helpers with the reference's [y, x, z] conventions
  slab        x-slab multi-GPU build with an RCCL halo exchange
  synth       counter-based synthetic occupancy generators for tests and bench.py
"""
__all__ = ["capi", "slab", "synth", "utils_2d", "utils_3d", "build"]
