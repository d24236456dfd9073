// Micro-benchmark (development aid, not part of the library): the MEMORY PATTERN of the far-field x sweep at 512^3 without
// its search -- a workgroup owns a tile of NLN memory-adjacent lines (row segments of NLN x 4 B at a stride of ny*nz
// elements), reads every row of the tile (16-byte loads, 4 lines per lane), keeps it in LDS, and writes it back as fp32
// with the store mapping of the chunk phase (one element per lane and store: NLN lines x 256 / NLN positions).
// Variants: lines per tile 16 / 32 / 64 (64-, 128-, 256-byte row segments), and the store mapping "rows": each store
// instruction writes whole rows (NLN x 4 B contiguous per row, positions consecutive) instead of 8-position chunks.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/tile_copy_probe.hip -o tools/probe/tile_copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

template <int NLN, int NT, int STORE, int MODE = 0>      // MODE 0: load + store, 1: loads only (one value per lane stored at the end... of every 4096th workgroup), 2: stores only.  STORE 0: chunk mapping (lane = line, slot of 8 positions); 1: row mapping; 2: chunk mapping, input TILE-MAJOR (a tile's rows contiguous)
__global__ __launch_bounds__(NT) void k_tile(const int* __restrict__ in, float* __restrict__ out, int L, uint32_t ls, int xcd_order, uint32_t pad_in = 0, uint32_t pad_out = 0) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int pitch = L + 2;
    const int t = threadIdx.x;
    int64_t tile = blockIdx.x;
    if (xcd_order && (gridDim.x & 31u) == 0u) {
        const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
        tile = ((int64_t)(seq >> 2) * 8 + xcd) * 4 + (seq & 3u);
    }
    const int64_t base = tile * NLN;
    const int* ip = STORE == 2 ? in + tile * (int64_t)NLN * L : in + base;
    const uint32_t lsi = STORE == 2 ? (uint32_t)NLN : ls + pad_in;
    const uint32_t lso = ls + pad_out;
    float* op = out + base;
    constexpr int LPR = NLN / 4;            // lanes per row
    constexpr int PP = NT / LPR;            // rows per load round
    const int sub = t % LPR, r = t / LPR;
    constexpr int NB = 8;
    int acc = 0;
    if (MODE != 2)
    for (int pb = 0; pb < L; pb += PP * NB) {
        int4 v[NB];
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int p = pb + PP * it + r;
            v[it] = *reinterpret_cast<const int4*>(ip + (uint32_t)(p < L ? p : L - 1) * lsi + 4u * sub);
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int p = pb + PP * it + r;
            if (p < L) {
                smem[(4 * sub + 0) * pitch + p] = v[it].x; smem[(4 * sub + 1) * pitch + p] = v[it].y;
                smem[(4 * sub + 2) * pitch + p] = v[it].z; smem[(4 * sub + 3) * pitch + p] = v[it].w;
                acc += v[it].x ^ v[it].y ^ v[it].z ^ v[it].w;
            }
        }
    }
    __syncthreads();
    if (MODE == 1) { if (acc == 0x12345678) op[t] = 1.0f; return; }
    if (STORE == 0 || STORE == 2) {
        const int line = t % NLN, slot = t / NLN;
        constexpr int S = NT / NLN;
        for (int i0 = 0; i0 < L / 8; i0 += S) {
            const int p0 = 8 * (i0 + slot);
#pragma unroll
            for (int k = 0; k < 8; ++k) op[(uint32_t)line + (uint32_t)(p0 + k) * lso] = (float)smem[line * pitch + p0 + k] * 0.01f;
        }
    } else {
        const int line = t % NLN, rr = t / NLN;
        constexpr int S = NT / NLN;
        for (int p = rr; p < L; p += S) op[(uint32_t)line + (uint32_t)p * lso] = (float)smem[line * pitch + p] * 0.01f;
    }
}

template <int NLN, int NT, int STORE, int MODE = 0>
float run(const int* in, float* out, int n, int xcd, uint32_t pad_in = 0, uint32_t pad_out = 0) {
    const int L = n;
    const uint32_t ls = (uint32_t)n * n;
    const unsigned ntiles = (unsigned)((int64_t)n * n / NLN);
    const size_t lds = (size_t)NLN * (L + 2) * 4;
    CK(hipFuncSetAttribute((const void*)k_tile<NLN, NT, STORE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_tile<NLN, NT, STORE, MODE>), dim3(ntiles), dim3(NT), lds, 0, in, out, L, ls, xcd, pad_in, pad_out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_tile<NLN, NT, STORE, MODE>), dim3(ntiles), dim3(NT), lds, 0, in, out, L, ls, xcd, pad_in, pad_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 10;
}

__global__ void k_copy(const int4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { const int4 v = in[i]; out[i] = make_float4(v.x * 0.01f, v.y * 0.01f, v.z * 0.01f, v.w * 0.01f); }
}

int main() {
    const int n = 512;
    const size_t N = (size_t)n * n * n;
    int* in; float* out;
    CK(hipMalloc(&in, N * 4 + (64u << 20))); CK(hipMalloc(&out, N * 4 + (64u << 20)));
    CK(hipMemset(in, 1, N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_copy, dim3((unsigned)(N / 4 / 256)), dim3(256), 0, 0, (const int4*)in, (float4*)out, N / 4);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_copy, dim3((unsigned)(N / 4 / 256)), dim3(256), 0, 0, (const int4*)in, (float4*)out, N / 4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("linear copy 4+4 B/voxel: %.4f ms\n", ms / 10);
    // plane stride padded (the probe's question in round 3: do the rows of a tile, 1 MiB apart, camp on one HBM channel?)
    for (uint32_t pad : {0u, 16u, 64u, 1024u, 4096u + 64u}) {
        printf("pad %u elements: in only %.4f ms, in + out %.4f ms (16 lines x 256 lanes, chunk stores)\n", pad, run<16, 256, 0>(in, out, n, 1, pad, 0), run<16, 256, 0>(in, out, n, 1, pad, pad));
    }
    // round 4: the two sides on their own (the x sweep without its search takes 0.35 ms: 0.19 without its stores)
    printf("loads only:  16 lines natural %.4f ms, plane stride + 4160 %.4f ms, tile-major %.4f ms; 32 lines %.4f ms; 64 lines %.4f ms\n",
           run<16, 256, 0, 1>(in, out, n, 1), run<16, 256, 0, 1>(in, out, n, 1, 4160, 0), run<16, 256, 2, 1>(in, out, n, 1), run<32, 512, 0, 1>(in, out, n, 1), run<64, 1024, 0, 1>(in, out, n, 1));
    printf("stores only: 16 lines chunk mapping %.4f ms, plane stride + 4160 %.4f ms, row mapping %.4f ms; 32 lines chunk %.4f ms, row %.4f ms; 64 lines chunk %.4f ms\n",
           run<16, 256, 0, 2>(in, out, n, 1), run<16, 256, 0, 2>(in, out, n, 1, 0, 4160), run<16, 256, 1, 2>(in, out, n, 1), run<32, 512, 0, 2>(in, out, n, 1), run<32, 512, 1, 2>(in, out, n, 1), run<64, 1024, 0, 2>(in, out, n, 1));
    for (int xcd = 1; xcd >= 1; --xcd) {
        printf("xcd_order=%d\n", xcd);
        printf("  16 lines x 256 lanes, chunk stores: %.4f ms\n", run<16, 256, 0>(in, out, n, xcd));
        printf("  16 lines x 256 lanes, row stores:   %.4f ms\n", run<16, 256, 1>(in, out, n, xcd));
        printf("  16 lines x 256 lanes, chunk stores, tile-major input: %.4f ms\n", run<16, 256, 2>(in, out, n, xcd));
        printf("  32 lines x 512 lanes, chunk stores: %.4f ms\n", run<32, 512, 0>(in, out, n, xcd));
        printf("  32 lines x 512 lanes, row stores:   %.4f ms\n", run<32, 512, 1>(in, out, n, xcd));
        printf("  32 lines x 256 lanes, row stores:   %.4f ms\n", run<32, 256, 1>(in, out, n, xcd));
        printf("  64 lines x 512 lanes, row stores:   %.4f ms\n", run<64, 512, 1>(in, out, n, xcd));
        printf("  64 lines x 1024 lanes, chunk stores: %.4f ms\n", run<64, 1024, 0>(in, out, n, xcd));
    }
    return 0;
}
