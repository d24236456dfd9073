// sdfgpu_dense6_tu.hip -- the shell pass of the dense tier's fix-up stage (k_ball_shell, sdfgpu_dense6.hpp) as a translation unit
// of its own: 668 fully unrolled lattice offsets take a while to compile; linked into libsdfgpu.so (sdf_tools_amd/build.py).
// The launcher is launch_ball_dense in sdfgpu.hip.
#define SDFGPU_AUX_TU
#define SDFGPU_DENSE6_TU
#include "sdfgpu_dense6.hpp"

namespace sdfgpu {
#define SDFGPU_SHELL_DEFINE(BD) template __global__ void k_ball_shell<BD>(const ShellArgs);
SDFGPU_SHELL_INSTANCES(SDFGPU_SHELL_DEFINE)
#undef SDFGPU_SHELL_DEFINE
}  // namespace sdfgpu
