// Micro-benchmark for the full-grid gradient stencil (development aid, not part of the library): variants of the memory
// access pattern at 512^3, timed with HIP events.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe/grad_probe.hip -o tools/probe/grad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)
constexpr int kBlock = 256;

// V0: the library's structure (4 voxels per lane, LDS transposition, 3 x 16-B stores per lane)
template <int MODE>   // 0 full, 1 no stores (one dummy), 2 centre row only (no neighbour loads), 3 stores only
__global__ __launch_bounds__(kBlock) void k_v0(const float* __restrict__ f, float* __restrict__ g, int64_t nx, int64_t ny, int64_t nz, float inv2) {
    __shared__ __attribute__((aligned(16))) float stage[(kBlock / 64) * 64 * 12];
    const int64_t n4 = nx * ny * nz / 4;
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float* st = stage + (threadIdx.x >> 6) * (64 * 12);
    if (q < n4) {
        const int64_t i = 4 * q;
        const int64_t z = i % nz, y = (i / nz) % ny, x = i / (nz * ny);
        const int64_t sx = ny * nz, sy = nz;
        float o[12];
        if (MODE == 3) {
            for (int k = 0; k < 12; ++k) o[k] = (float)k + inv2;
        } else if (x > 0 && x < nx - 1 && y > 0 && y < ny - 1 && z > 0 && z + 4 < nz) {
            const float4 c = *reinterpret_cast<const float4*>(f + i);
            float4 xp = c, xm = c, yp = c, ym = c; float zm = c.x, zp = c.w;
            if (MODE != 2) {
                xp = *reinterpret_cast<const float4*>(f + i + sx); xm = *reinterpret_cast<const float4*>(f + i - sx);
                yp = *reinterpret_cast<const float4*>(f + i + sy); ym = *reinterpret_cast<const float4*>(f + i - sy);
                zm = f[i - 1]; zp = f[i + 4];
            }
            const float cz[6] = {zm, c.x, c.y, c.z, c.w, zp};
            const float xpv[4] = {xp.x, xp.y, xp.z, xp.w}, xmv[4] = {xm.x, xm.y, xm.z, xm.w};
            const float ypv[4] = {yp.x, yp.y, yp.z, yp.w}, ymv[4] = {ym.x, ym.y, ym.z, ym.w};
            for (int k = 0; k < 4; ++k) {
                o[3 * k + 0] = (xpv[k] - xmv[k]) * inv2; o[3 * k + 1] = (ypv[k] - ymv[k]) * inv2; o[3 * k + 2] = (cz[k + 2] - cz[k]) * inv2;
            }
        } else { for (int k = 0; k < 12; ++k) o[k] = 0.f; }
        float4* d = reinterpret_cast<float4*>(st + lane * 12);
        d[0] = make_float4(o[0], o[1], o[2], o[3]); d[1] = make_float4(o[4], o[5], o[6], o[7]); d[2] = make_float4(o[8], o[9], o[10], o[11]);
    }
    __syncthreads();
    const int64_t q0 = q - lane;
    if (q0 < n4) {
        float* dst = g + 12 * q0;
        if (MODE == 1) { if (lane == 0 && st[5] == 1234.5f) dst[0] = st[0]; return; }
        for (int k = 0; k < 3; ++k) {
            const int idx = k * 256 + lane * 4;
            *reinterpret_cast<float4*>(dst + idx) = *reinterpret_cast<const float4*>(st + idx);
        }
    }
}

// V1: a workgroup owns a (TY rows x 512 z) strip of one x plane and marches along x keeping the previous / current / next
// plane rows in registers: every input row is loaded once per strip (x neighbours come from registers, not from L2)
template <int TX>
__global__ __launch_bounds__(kBlock) void k_v1(const float* __restrict__ f, float* __restrict__ g, int64_t nx, int64_t ny, int64_t nz, float inv2) {
    __shared__ __attribute__((aligned(16))) float stage[(kBlock / 64) * 64 * 12];
    // lane -> 4 consecutive z of row (y0 + ry); block covers 256 * 4 = 1024 voxels = 2 rows of 512
    const int64_t rows_per_block = (kBlock * 4) / nz;
    const int64_t yb = (int64_t)blockIdx.x * rows_per_block;
    const int ry = (threadIdx.x * 4) / nz, z = (threadIdx.x * 4) % nz;
    const int64_t y = yb + ry;
    const int64_t x0 = (int64_t)blockIdx.y * TX;
    const int64_t sx = ny * nz, sy = nz;
    const int lane = threadIdx.x & 63;
    float* st = stage + (threadIdx.x >> 6) * (64 * 12);
    if (y >= ny) return;
    auto ld = [&](int64_t x) { x = x < 0 ? 0 : (x >= nx ? nx - 1 : x); return *reinterpret_cast<const float4*>(f + x * sx + y * sy + z); };
    float4 prev = ld(x0 - 1), cur = ld(x0);
    for (int64_t x = x0; x < x0 + TX && x < nx; ++x) {
        const float4 next = ld(x + 1);
        const int64_t i = x * sx + y * sy + z;
        const int64_t yu = y + 1 < ny ? y + 1 : y, yd = y > 0 ? y - 1 : y;
        const float4 yp = *reinterpret_cast<const float4*>(f + x * sx + yu * sy + z), ym = *reinterpret_cast<const float4*>(f + x * sx + yd * sy + z);
        const float zm = z > 0 ? f[i - 1] : cur.x, zp = z + 4 < nz ? f[i + 4] : cur.w;
        const float cz[6] = {zm, cur.x, cur.y, cur.z, cur.w, zp};
        const float xpv[4] = {next.x, next.y, next.z, next.w}, xmv[4] = {prev.x, prev.y, prev.z, prev.w};
        const float ypv[4] = {yp.x, yp.y, yp.z, yp.w}, ymv[4] = {ym.x, ym.y, ym.z, ym.w};
        float o[12];
        for (int k = 0; k < 4; ++k) {
            o[3 * k + 0] = (xpv[k] - xmv[k]) * inv2; o[3 * k + 1] = (ypv[k] - ymv[k]) * inv2; o[3 * k + 2] = (cz[k + 2] - cz[k]) * inv2;
        }
        float4* d = reinterpret_cast<float4*>(st + lane * 12);
        d[0] = make_float4(o[0], o[1], o[2], o[3]); d[1] = make_float4(o[4], o[5], o[6], o[7]); d[2] = make_float4(o[8], o[9], o[10], o[11]);
        __builtin_amdgcn_wave_barrier();
        float* dst = g + 3 * (i - 4 * lane);
        for (int k = 0; k < 3; ++k) {
            const int idx = k * 256 + lane * 4;
            *reinterpret_cast<float4*>(dst + idx) = *reinterpret_cast<const float4*>(st + idx);
        }
        __builtin_amdgcn_wave_barrier();
        prev = cur; cur = next;
    }
}

int main() {
    const int64_t n = 512, N = n * n * n;
    float *f, *g;
    CK(hipMalloc(&f, N * 4)); CK(hipMalloc(&g, N * 12));
    std::vector<float> h((size_t)N);
    for (int64_t i = 0; i < N; ++i) h[(size_t)i] = (float)((i * 2654435761u) % 1000) * 0.01f;
    CK(hipMemcpy(f, h.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s %.4f ms  (%.0f GB/s of 16 B/voxel)\n", name, ms / 20, N * 16.0 / (ms / 20) / 1e6);
    };
    const dim3 g0((unsigned)((N / 4 + kBlock - 1) / kBlock));
    time("V0 full (library structure)", [&] { hipLaunchKernelGGL(k_v0<0>, g0, dim3(kBlock), 0, 0, f, g, n, n, n, 50.0f); });
    time("V0 no stores", [&] { hipLaunchKernelGGL(k_v0<1>, g0, dim3(kBlock), 0, 0, f, g, n, n, n, 50.0f); });
    time("V0 centre loads only", [&] { hipLaunchKernelGGL(k_v0<2>, g0, dim3(kBlock), 0, 0, f, g, n, n, n, 50.0f); });
    time("V0 stores only", [&] { hipLaunchKernelGGL(k_v0<3>, g0, dim3(kBlock), 0, 0, f, g, n, n, n, 50.0f); });
    time("V1 march along x, TX=16", [&] { hipLaunchKernelGGL(k_v1<16>, dim3((unsigned)(n / 2), (unsigned)(n / 16)), dim3(kBlock), 0, 0, f, g, n, n, n, 50.0f); });
    time("V1 march along x, TX=64", [&] { hipLaunchKernelGGL(k_v1<64>, dim3((unsigned)(n / 2), (unsigned)(n / 64)), dim3(kBlock), 0, 0, f, g, n, n, n, 50.0f); });
    CK(hipMemset(g, 0, N * 12));
    time("hipMemset 1.5 GiB", [&] { CK(hipMemsetAsync(g, 0, N * 12, 0)); });
    return 0;
}
