// Micro-benchmark (development aid, not part of the library): the z sweep K1 at 512^3, barrier form (k_sweep_z_vec16)
// against the wave-private form (k_sweep_z_wave16), same inputs, outputs compared word for word.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe/k1_probe.hip -o tools/probe/k1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../sdf_tools_amd/csrc/sdfgpu_kernels.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)
using namespace sdfgpu;

int main(int argc, char** argv) {
    const int nz = argc > 1 ? atoi(argv[1]) : 512;
    const int64_t nrows = argc > 2 ? atoll(argv[2]) : 512 * 512;
    const int64_t n = nrows * nz;
    uint8_t* d_mask; int16_t *d_a, *d_b;
    CK(hipMalloc(&d_mask, n)); CK(hipMalloc(&d_a, n * 2)); CK(hipMalloc(&d_b, n * 2));
    std::vector<uint8_t> hm(n);
    std::vector<int16_t> ha(n), hb(n);
    const double ps[] = {0.5, 0.05, 0.001, -1.0};
    for (double p : ps) {
        uint64_t st = 88172645463325252ull;
        for (int64_t i = 0; i < n; ++i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            if (p < 0) { const int64_t z = i % nz, r = i / nz; hm[i] = (z > 100 && z < 180 && (r % 512) > 50 && (r % 512) < 200 && (r / 512) > 60 && (r / 512) < 300) ? 1 : 0; }
            else hm[i] = ((st >> 11) * (1.0 / 9007199254740992.0)) < p ? 255 : 0;
        }
        CK(hipMemcpy(d_mask, hm.data(), n, hipMemcpyHostToDevice));
        const int W = (nz + 63) / 64, rpb = std::max(1, 4096 / (W * 64));
        const size_t lds = (size_t)rpb * W * 8 + (size_t)rpb * 4;
        int per_cu = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sweep_z_vec16, kBlock, lds));
        const int64_t nblocks = (nrows + rpb - 1) / rpb;
        const unsigned gA = (unsigned)std::min<int64_t>(nblocks, (int64_t)per_cu * 256);
        int per_cu_b = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, k_sweep_z_wave16<32>, kBlock, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time = [&](auto&& f) { for (int i = 0; i < 3; ++i) f(); CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 20; };
        const float tA = time([&] { hipLaunchKernelGGL(k_sweep_z_vec16, dim3(gA), dim3(kBlock), lds, 0, d_mask, d_a, nrows, nz, rpb, (const uint32_t*)nullptr); });
        printf("p=%g  barrier form %.4f ms (grid %u, %d/CU)", p, tA, gA, per_cu);
        for (int mult : {8, 16, 32}) {
            const unsigned gB = (unsigned)(256 * mult);
            float tB = 0;
            if (nz == 512) tB = time([&] { hipLaunchKernelGGL(k_sweep_z_wave16<32>, dim3(gB), dim3(kBlock), 0, 0, d_mask, d_b, nrows, (const uint32_t*)nullptr); });
            else if (nz == 256) tB = time([&] { hipLaunchKernelGGL(k_sweep_z_wave16<16>, dim3(gB), dim3(kBlock), 0, 0, d_mask, d_b, nrows, (const uint32_t*)nullptr); });
            else if (nz == 1024) tB = time([&] { hipLaunchKernelGGL(k_sweep_z_wave16<64>, dim3(gB), dim3(kBlock), 0, 0, d_mask, d_b, nrows, (const uint32_t*)nullptr); });
            else if (nz == 128) tB = time([&] { hipLaunchKernelGGL(k_sweep_z_wave16<8>, dim3(gB), dim3(kBlock), 0, 0, d_mask, d_b, nrows, (const uint32_t*)nullptr); });
            else if (nz == 64) tB = time([&] { hipLaunchKernelGGL(k_sweep_z_wave16<4>, dim3(gB), dim3(kBlock), 0, 0, d_mask, d_b, nrows, (const uint32_t*)nullptr); });
            printf("   wave form x%d %.4f ms", mult, tB);
        }
        printf("  (occupancy %d/CU)\n", per_cu_b);
        CK(hipMemcpy(ha.data(), d_a, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, n * 2, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int64_t i = 0; i < n; ++i) bad += ha[i] != hb[i];
        printf("      differing words: %lld\n", (long long)bad);
    }
    return 0;
}
