// Micro-benchmark (development aid): what a dependent near-empty launch costs in a stream -- plain, cooperative, and a
// chain of three against one.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe/launch_probe.hip -o tools/probe/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)
__global__ void k_work(float* p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
__global__ void k_guard(const unsigned* g, float* p) { if (*g == 0u) return; p[threadIdx.x] = 1.f; }
struct Big { const unsigned* g; float* p; long pad[30]; };
__global__ void k_guard_big(const Big a) { if (*a.g == 0u) return; a.p[threadIdx.x] = (float)a.pad[threadIdx.x & 15]; }
__global__ void k_guard_lds(const unsigned* g, float* p) { extern __shared__ float sm[]; if (*g == 0u) return; sm[threadIdx.x] = 1.f; __syncthreads(); p[threadIdx.x] = sm[255 - threadIdx.x]; }
__global__ __launch_bounds__(256) void k_guard_vgpr(const unsigned* g, float* p) {
    if (*g == 0u) return;
    float v[120];
    for (int i = 0; i < 120; ++i) v[i] = p[threadIdx.x + 256 * i];
    float s = 0; for (int k = 0; k < 8; ++k) for (int i = 0; i < 120; ++i) { v[i] = v[i] * v[(i + 7) % 120] + 1.f; s += v[i]; }
    p[threadIdx.x] = s;
}
__global__ void k_guard_scratch(const unsigned* g, float* p, int n) {
    if (*g == 0u) return;
    volatile float loc[64];
    for (int i = 0; i < 64; ++i) loc[i] = p[i];
    float s = 0; for (int i = 0; i < n; ++i) s += loc[(i * 7) & 63];
    p[threadIdx.x] = s;
}
int main() {
    const size_t n = 64u << 20;
    float* p; unsigned* g;
    CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, 4)); CK(hipMemset(g, 0, 4)); CK(hipMemset(p, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto&& f) {
        for (int i = 0; i < 5; ++i) f();
        CK(hipEventRecord(e0)); for (int i = 0; i < 100; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-60s %.2f us per iteration\n", name, ms * 10.f);
    };
    auto work = [&] { hipLaunchKernelGGL(k_work, dim3((unsigned)(n / 256)), dim3(256), 0, 0, p, n); };
    time("work only", [&] { work(); });
    for (unsigned grid : {1u, 256u, 1024u, 2048u, 8192u}) {
        char nm[128];
        snprintf(nm, sizeof nm, "work + 1 guarded exit, grid %u", grid);
        time(nm, [&] { work(); hipLaunchKernelGGL(k_guard, dim3(grid), dim3(256), 0, 0, g, p); });
        snprintf(nm, sizeof nm, "work + 3 guarded exits, grid %u", grid);
        time(nm, [&] { work(); for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_guard, dim3(grid), dim3(256), 0, 0, g, p); });
    }
    {
        Big b{}; b.g = g; b.p = p;
        time("work + 3 guarded exits, grid 1024, 256-B kernarg", [&] { work(); for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_guard_big, dim3(1024), dim3(256), 0, 0, b); });
        time("work + 3 guarded exits, grid 1024, 8 KB dynamic LDS", [&] { work(); for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_guard_lds, dim3(1024), dim3(256), 8192, 0, g, p); });
        time("work + 3 guarded exits, grid 1024, ~128 VGPRs", [&] { work(); for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_guard_vgpr, dim3(1024), dim3(256), 0, 0, g, p); });
        time("work + 3 guarded exits, grid 1024, scratch", [&] { work(); for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_guard_scratch, dim3(1024), dim3(256), 0, 0, g, p, 100); });
    }
    for (unsigned grid : {256u}) {
        char nm[128];
        snprintf(nm, sizeof nm, "work + 1 COOPERATIVE guarded exit, grid %u", grid);
        const unsigned* gg = g; float* pp = p;
        void* args[] = {(void*)&gg, (void*)&pp};
        time(nm, [&] { work(); CK(hipLaunchCooperativeKernel((const void*)k_guard, dim3(grid), dim3(256), args, 0, 0)); });
    }
    return 0;
}
