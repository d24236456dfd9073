// sdfgpu_envelope_tu.hip -- the far-field kernel's instantiations (k_envelope_dc, sdfgpu_envelope_dc.hpp) as a translation unit
// of their own: compiled beside sdfgpu.hip and linked into the same libsdfgpu.so (sdf_tools_amd/build.py).  Nothing but the
// explicit instantiations lives here; the launcher is launch_envelope in sdfgpu.hip.
#define SDFGPU_ENVELOPE_TU
#include "sdfgpu_envelope_dc.hpp"

namespace sdfgpu {
#define SDFGPU_ENVELOPE_DEFINE(...) template __global__ void k_envelope_dc<__VA_ARGS__>(const EnvDcArgs);
SDFGPU_ENVELOPE_INSTANCES(SDFGPU_ENVELOPE_DEFINE)
#undef SDFGPU_ENVELOPE_DEFINE
}  // namespace sdfgpu
