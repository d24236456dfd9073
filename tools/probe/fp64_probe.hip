// Micro-benchmark (development aid): issue cost of the fp64 instructions the far-field kernel's finish uses, per wave64
// instruction and SIMD, dependent chain vs 8 independent chains.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe/fp64_probe.hip -o tools/probe/fp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)
constexpr int kIters = 65536;

template <int MODE, int CH>
__global__ __launch_bounds__(256) void k(double* out, double seed, int n) {
    double v[CH];
    for (int c = 0; c < CH; ++c) v[c] = seed + threadIdx.x + c;
    float w[CH];
    for (int c = 0; c < CH; ++c) w[c] = (float)seed + threadIdx.x + c;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (MODE == 0) v[c] = __builtin_fma(v[c], 1.0000001, 0.5);
            if (MODE == 1) v[c] = __builtin_amdgcn_rsq(v[c]) + 3.0;                 // rsq + add
            if (MODE == 2) v[c] = (double)(int)((float)v[c]) + 1.5;                  // cvt f32<-f64, cvt i32<-f32, cvt f64<-i32, add
            if (MODE == 3) w[c] = __builtin_fmaf(w[c], 1.0000001f, 0.5f);
            if (MODE == 4) v[c] = v[c] * 1.0000001;
            if (MODE == 5) v[c] = v[c] + 1.0000001;
        }
    }
    double s = 0; for (int c = 0; c < CH; ++c) s += v[c] + w[c];
    if (s == 12345.678) out[0] = s;
}


__device__ __forceinline__ double sqrt_exact_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y, h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, s0, 0.5);
    const double s1 = __builtin_fma(s0, r0, s0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-s1, s1, x);
    const double s2 = __builtin_fma(d0, h1, s1);
    const double d1 = __builtin_fma(-s2, s2, x);
    return __builtin_fma(d1, h1, s2);
}
// the far-field kernel's finish on synthetic squared distances: MODE 0 small values shared by many lanes, 1 large values
// that differ in every lane, 2 large but wave-uniform
template <int MODE>
__global__ __launch_bounds__(256) void k_finish(float* out, double res, int n) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        uint32_t h = (t + i * 7919u) * 2654435761u;
        int D;
        if (MODE == 0) D = (int)((h >> 27)) + 1;                 // 1..32
        if (MODE == 1) D = (int)(h % 700000u) + 1;
        if (MODE == 2) D = (int)(((blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u + i * 7919u) % 700000u) + 1;
        acc += (float)(sqrt_exact_pos((double)D) * res);
    }
    if (acc == 12345.f) out[0] = acc;
}
template <int MODE>
void run_finish(const char* name) {
    float* d; CK(hipMalloc(&d, 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_finish<MODE><<<256 * 16, 256>>>(d, 0.01, 2);
    CK(hipEventRecord(a));
    k_finish<MODE><<<256 * 16, 256>>>(d, 0.01, 128);            // 134 M finishes
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-40s %.3f ms for 134 M finishes\n", name, ms);
}

template <int MODE, int CH>
void run(const char* name, int instr_per_iter) {
    double* d; CK(hipMalloc(&d, 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * 4;             // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    k<MODE, CH><<<blocks, 256>>>(d, 1.0, 16);
    CK(hipEventRecord(a));
    k<MODE, CH><<<blocks, 256>>>(d, 1.0, kIters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // per SIMD: 4 waves x kIters x CH x instr_per_iter wave-instructions
    const double winstr = 4.0 * kIters * CH * instr_per_iter;
    printf("%-28s CH=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, CH, ms, ms * 1e-3 * 2.4e9 / winstr);
}

int main() {
    run<0, 1>("fma_f64 dependent", 1); run<0, 8>("fma_f64 8 chains", 1);
    run<4, 1>("mul_f64 dependent", 1); run<4, 8>("mul_f64 8 chains", 1);
    run<5, 1>("add_f64 dependent", 1); run<5, 8>("add_f64 8 chains", 1);
    run<1, 1>("rsq_f64+add dependent", 2); run<1, 8>("rsq_f64+add 8 chains", 2);
    run<2, 1>("cvt x3 + add dependent", 4); run<2, 8>("cvt x3 + add 8 chains", 4);
    run<3, 1>("fma_f32 dependent", 1); run<3, 8>("fma_f32 8 chains", 1);
    run_finish<0>("finish, D in 1..32"); run_finish<1>("finish, D < 700000 per lane"); run_finish<2>("finish, D < 700000 wave-uniform");
    return 0;
}
