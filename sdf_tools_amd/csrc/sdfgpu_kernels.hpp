// sdfgpu_kernels.hpp -- hand-written CDNA4 (gfx950) kernels of the SDF build path.
//
// The reference computes two approximate distance fields by bucket-queue
// propagation (sdf_generation.hpp:95-207) and merges them (:245-269).  Here the
// same result is produced by an exact separable squared-EDT carried in ONE
// signed field (for every voxel exactly one of the two reference fields is 0):
//
//   K1  sweep_z : mask / COLLISION_CELL -> int16  s1 = +-(distance along z to the
//                 nearest voxel of the opposite class), +-32767 = none in the row
//   K2  sweep_y : int16 -> int32  s2 = +-(squared distance inside the x-plane)
//   K3  sweep_x : int32 -> fp32   sdf = +-res*sqrt(d^2) (fp64 sqrt/mul, one cast),
//                 virtual-border clamp and integer extrema fused
//
// Sign convention of the intermediates: + for free voxels (distance to filled),
// - for filled voxels (distance to free).  A candidate site u seen from voxel v
// contributes |s(u)| if u has v's class, 0 otherwise.
//
// K2/K3 are the same "marching" kernel: a lane owns V consecutive z of one
// line bundle and walks T positions along the swept axis, keeping the last
// 2H+1 rows in registers.  Every global access is a full coalesced row segment
// (64 lanes x V elements); nothing is transposed and no LDS is needed.  The
// register window decides every voxel whose squared distance is < (H+1)^2;
// farther voxels continue with an outward scan straight from global memory
// (L2-served) that stops as soon as d^2 >= best, so the result is exact for any
// input while dense scenes never leave the window.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// Non-template kernels are DEFINED in these headers.  The one translation unit that includes them only for their device
// helpers (sdfgpu_envelope_tu.hip, sdfgpu_dense6_tu.hip: one kernel's instantiations each) gives them internal linkage, so
// the library holds one definition of each.
#if defined(SDFGPU_ENVELOPE_TU) || defined(SDFGPU_AUX_TU)
#define SDFGPU_KERNEL static __global__
#else
#define SDFGPU_KERNEL __global__
#endif

namespace sdfgpu {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Grid-wide maxima / flags live in single words.  One same-address atomic costs ~12 ns at the L2, so a
// wave per atomic (8-130 k waves per launch) would serialise for 0.2-1.5 ms -- longer than the kernels
// themselves.  Read the word first (L2-coherent relaxed load) and only issue the atomic when it would
// change the value: after the first few hundred waves the maxima are established and nothing is sent.
__device__ __forceinline__ void atomic_max_if_larger(uint32_t* p, uint32_t v) {
    if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
}
__device__ __forceinline__ void atomic_or_if_new(uint32_t* p, uint32_t bits) {
    if (bits & ~__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(p, bits);
}
// A flag whose only non-zero value is 1 needs no read-modify-write at all: every writer stores the same word.
__device__ __forceinline__ void raise_flag(uint32_t* p) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        __hip_atomic_store(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Extrema accumulation.  Every wave of a final-stage kernel ends with up to two atomic max operations.  Sent to
// ONE address they serialise in the L2 atomic unit at ~12 ns each, and the read-before-atomic filter cannot help
// the first generation of waves: all ~8000 resident waves still see the initial 0 and fire -- a backlog of up to
// ~200 us that the kernel has to drain before it completes, and the better the waves are synchronised (i.e. the
// FASTER the code in front of the atomics), the more of them fire.  (Measured on the dense kernel: removing half
// of its VALU work made it 35 us slower; removing its store phase altogether made it 45 us slower.)  So a wave
// updates one of kSlots 128-byte slots chosen by its global wave index (neighbouring waves -> different L2
// channels, ~16 first-generation waves per slot), and the one-block k_fold_slots launched at the end of every
// ABI call folds the slots into the caller's {max free, max filled} words and clears them again.
constexpr int kSlots = 512;
constexpr int kStatusWords = 24;      // words of the status block that the end-of-build fold publishes and clears
constexpr int kSlotWords = 32;
constexpr uint32_t kReportDense = 0x1000FFu;      // status words a dense-tier report publishes to the host
constexpr uint32_t kReportFar = 0x30u;            // ... a far-flag report (words 4, 5)
__device__ __forceinline__ void slot_max2(uint32_t* slots, uint32_t wave, int mxF, int mxQ) {
    uint32_t* p = slots + (size_t)(wave & (kSlots - 1)) * kSlotWords;
    if (mxF) atomic_max_if_larger(p + 0, (uint32_t)mxF);
    if (mxQ) atomic_max_if_larger(p + 1, (uint32_t)mxQ);
}

// End-of-build form (result != nullptr; `maxdsq` is then the context's 8-word status block {maxima, status,
// uncertified, far flags, fix_needed, -}): the final block is written to `result` (device; what get_extrema reads)
// and to `report` (pinned host memory mapped into the device; the policy's asynchronous "what did this build need"),
// and the status block is cleared for the next build -- one kernel instead of fold + copy kernel + fill kernel.
// (the body as a device function: the stand-by x sweep of a build folds for itself -- one launch less per build -- see
//  k_envelope_dc's LOOP form; loads through the L2, because in that kernel other workgroups of the SAME launch may have written)
template <int NT>
__device__ __forceinline__ void fold_slots_device(uint32_t* __restrict__ slots, uint32_t* __restrict__ maxdsq,
                                                  uint32_t* __restrict__ result, uint32_t* __restrict__ report, const uint32_t report_mask,
                                                  const int t) {
    static_assert(NT % 64 == 0 && NT >= 64 && kSlots % NT == 0, "whole waves, whole rounds over the slots");
    __shared__ uint32_t part[2 * (NT / 64)];
    uint32_t f = 0, q = 0;
#pragma unroll
    for (int i = 0; i < kSlots / NT; ++i) {
        uint32_t* p = slots + (size_t)(t + i * NT) * kSlotWords;
        const uint32_t f1 = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t q1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f1) p[0] = 0;
        if (q1) p[1] = 0;
        f = max(f, f1); q = max(q, q1);
    }
    uint32_t st = 0;
    if (result && t < kStatusWords) st = __hip_atomic_load(maxdsq + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        f = max(f, (uint32_t)__shfl_xor((int)f, off));
        q = max(q, (uint32_t)__shfl_xor((int)q, off));
    }
    if ((t & 63) == 0) { part[2 * (t >> 6)] = f; part[2 * (t >> 6) + 1] = q; }
    __syncthreads();
    if (t < kStatusWords) {                                    // lanes 0..23 of wave 0: one status word each
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { f = max(f, part[2 * w]); q = max(q, part[2 * w + 1]); }
        if (!result) {
            if (t == 0 && f) atomic_max_if_larger(maxdsq + 0, f);
            if (t == 1 && q) atomic_max_if_larger(maxdsq + 1, q);
        } else {
            if (t == 0) st = max(st, f);
            if (t == 1) st = max(st, q);
            result[t] = st;
            // (host copy: the words of report_mask -- kReportDense: words 0..7, and KD's own verdict, word 20, as word 8; kReportFar: the
            //  two far flags.  Every word is a separate write across PCIe that the kernel's end waits for: publishing all 24 made
            //  every build 0.04 ms longer)
            if (report && ((report_mask >> t) & 1u))
                __hip_atomic_store(report + (t == 20 ? 8 : t), st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            maxdsq[t] = 0;
        }
    }
}
SDFGPU_KERNEL __launch_bounds__(kSlots) void k_fold_slots(uint32_t* __restrict__ slots, uint32_t* __restrict__ maxdsq,
                                                       uint32_t* __restrict__ result, uint32_t* __restrict__ report, uint32_t report_mask) {
    fold_slots_device<kSlots>(slots, maxdsq, result, report, report_mask, (int)threadIdx.x);
}

constexpr int kInf16 = 32767;        // "no opposite voxel in this z row"
constexpr int kInf32 = 1 << 30;      // "no opposite voxel" for squared distances
constexpr int kFar = 1 << 20;        // position sentinel for the z sweep
constexpr int kBlock = 256;

// ---------------------------------------------------------------------------
// K1: z sweep.  One workgroup owns `rpb` whole z-rows, packs their occupancy
// into an LDS bitmap (one bit per voxel) and answers "nearest opposite bit"
// with clz/ffs on 64-bit words.
// ---------------------------------------------------------------------------

struct MaskLoader {
    const uint8_t* p;
    __device__ __forceinline__ bool filled(int64_t idx) const { return p[idx] != 0; }
};

// collision_map.hpp:680-712 predicate on raw COLLISION_CELL records.
struct CellLoader {
    const char* p;
    int64_t stride;
    int64_t off;
    int unknown_is_filled;
    __device__ __forceinline__ bool filled(int64_t idx) const {
        const float occ = *reinterpret_cast<const float*>(p + idx * stride + off);
        return (occ > 0.5f) || (unknown_is_filled && (occ == 0.5f));
    }
};

// one bit per voxel, linear order: bit (v & 31) of word (v >> 5) = voxel v (the layout K0 writes for nz % 32 == 0; the input of
// the bits-in entry points, sdfgpu_build_bits*)
struct BitsLoader {
    const uint32_t* p;
    __device__ __forceinline__ bool filled(int64_t idx) const { return (p[idx >> 5] >> (idx & 31)) & 1u; }
};

// 4 mask bytes -> 4 bits (byte k nonzero -> bit k)
__device__ __forceinline__ uint32_t nonzero_bits4(uint32_t w) {
    const uint32_t m = (w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    return (((m >> 7) * 0x01020408u) >> 24) & 0xFu;
}

__device__ __forceinline__ uint64_t valid_mask(int w, int nz) {
    const int nvalid = nz - 64 * w;
    return nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
}

// nearest voxel of class `want_filled` strictly left of word w (position or -kFar)
__device__ __forceinline__ int far_left(const uint64_t* row, int w, bool want_filled) {
    for (int ww = w - 1; ww >= 0; --ww) {
        const uint64_t b = want_filled ? row[ww] : ~row[ww];
        if (b) return 64 * ww + 63 - __clzll((long long)b);
    }
    return -kFar;
}
// nearest voxel of class `want_filled` strictly right of word w (position or +kFar)
__device__ __forceinline__ int far_right(const uint64_t* row, int w, int W, int nz, bool want_filled) {
    for (int ww = w + 1; ww < W; ++ww) {
        const uint64_t b = (want_filled ? row[ww] : ~row[ww]) & valid_mask(ww, nz);
        if (b) return 64 * ww + __ffsll((unsigned long long)b) - 1;
    }
    return kFar;
}

// signed z distance of the voxel at bit zb of `word` (word index w)
__device__ __forceinline__ int z_signed_distance(uint64_t word, uint64_t vm, int zb, int z,
                                                 int LF, int LE, int RF, int RE) {
    const bool own = (word >> zb) & 1ull;
    const uint64_t X = own ? (~word & vm) : word;          // opposite-class voxels in this word
    const uint64_t ml = X & ((1ull << zb) - 1ull);
    const int dl = ml ? zb - (63 - __clzll((long long)ml)) : z - (own ? LE : LF);
    const uint64_t mr = (X >> zb) >> 1;
    const int dr = mr ? __ffsll((unsigned long long)mr) : (own ? RE : RF) - z;
    const int d = min(min(dl, dr), kInf16);
    return own ? -d : d;
}

// Fast path: uint8 mask, nz % 16 == 0, 16-byte aligned base.  A lane handles 16
// consecutive voxels: one 16-B load, two 16-B stores.
SDFGPU_KERNEL __launch_bounds__(kBlock) void k_sweep_z_vec16(const uint8_t* __restrict__ mask,
                                                         int16_t* __restrict__ out,
                                                         int64_t nrows, int nz, int rpb,
                                                         const uint32_t* __restrict__ guard) {
    if (guard && *guard == 0u) return;      // dense path certified every voxel: nothing to do
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* bm = reinterpret_cast<uint64_t*>(smem_raw);
    uint16_t* bm16 = reinterpret_cast<uint16_t*>(smem_raw);
    const int W = (nz + 63) >> 6;
    uint32_t* rowcls = reinterpret_cast<uint32_t*>(bm + (size_t)rpb * W);   // [rpb] behind the bitmap
    const int cpr = nz >> 4;                       // 16-voxel chunks per row
    const int spr = W * 4;                         // uint16 slots per row in the bitmap
    // A lane's slot (row in group, 16-voxel chunk) is the same in every row group when a group is one pass of the
    // workgroup (always, unless nz > 4096): the two divisions are hoisted out of the loop, and the NEXT group's 16 bytes are
    // requested before this group's bit searches, so that the HBM round trip runs under them (the kernel sat at 2.2 TB/s
    // with a quarter of the VALU busy: every group waited for its own load, then for its own stores)
    const bool one_pass = rpb * spr <= kBlock;
    const int t = threadIdx.x;
    const int rA = t / spr, cA = t - rA * spr;     // phase A slot
    const int rB = t / cpr, cB = t - rB * cpr;     // phase B slot
    auto fetch = [&](int64_t row0) -> uint4 {
        if (row0 < nrows && cA < cpr && row0 + rA < nrows && rA < rpb)
            return *reinterpret_cast<const uint4*>(mask + (row0 + rA) * nz + 16 * cA);
        return make_uint4(0u, 0u, 0u, 0u);
    };
    const int64_t step = (int64_t)gridDim.x * rpb;
    uint4 vnext = one_pass ? fetch((int64_t)blockIdx.x * rpb) : make_uint4(0u, 0u, 0u, 0u);
    // persistent loop over row groups: a small grid keeps the guard early-exit cheap
    for (int64_t row0 = (int64_t)blockIdx.x * rpb; row0 < nrows; row0 += step) {
    const int nr = (int)min((int64_t)rpb, nrows - row0);
    // phase A: pack; per row: does it hold any filled (bit 0) / any free (bit 1) voxel?
    for (int s = threadIdx.x; s < nr; s += kBlock) rowcls[s] = 0u;
    __syncthreads();
    if (one_pass) {
        if (t < nr * spr) {
            uint32_t bits = 0;
            if (cA < cpr) {
                const uint4 v = vnext;
                bits = nonzero_bits4(v.x) | (nonzero_bits4(v.y) << 4) | (nonzero_bits4(v.z) << 8) | (nonzero_bits4(v.w) << 12);
                const uint32_t cls = (bits != 0u ? 1u : 0u) | (bits != 0xFFFFu ? 2u : 0u);
                atomicOr(&rowcls[rA], cls);   // (no return value: a fire-and-forget ds_or; reading the word first to skip the
                                              //  atomic put an LDS round trip into every lane's pack step: +0.1 ms at 512^3)
            }
            bm16[t] = (uint16_t)bits;
        }
        vnext = fetch(row0 + step);
    } else {
        for (int s = threadIdx.x; s < nr * spr; s += kBlock) {
            const int r = s / spr, c = s - r * spr;
            uint32_t bits = 0;
            if (c < cpr) {
                const uint4 v = *reinterpret_cast<const uint4*>(mask + (row0 + r) * nz + 16 * c);
                bits = nonzero_bits4(v.x) | (nonzero_bits4(v.y) << 4) | (nonzero_bits4(v.z) << 8) |
                       (nonzero_bits4(v.w) << 12);
                const uint32_t cls = (bits != 0u ? 1u : 0u) | (bits != 0xFFFFu ? 2u : 0u);
                atomicOr(&rowcls[r], cls);
            }
            bm16[s] = (uint16_t)bits;
        }
    }
    __syncthreads();
    // phase B: nearest opposite bit for 16 voxels per lane
    for (int s = threadIdx.x; s < nr * cpr; s += kBlock) {
        const int r = one_pass ? rB : s / cpr, c = one_pass ? cB : s - (s / cpr) * cpr;
        const uint64_t* row = bm + r * W;
        const int w = c >> 2, sub = c & 3;
        // a row of one class only (most rows of a scene with a few objects in free space): every voxel is "none"
        const uint32_t rc = rowcls[r];
        if (rc != 3u) {
            const uint32_t v = rc == 2u ? 0x7fff7fffu : 0x80018001u;       // +32767 (all free) / -32767 (all filled)
            uint4* dst = reinterpret_cast<uint4*>(out + (row0 + r) * nz + 16 * c);
            dst[0] = make_uint4(v, v, v, v);
            dst[1] = make_uint4(v, v, v, v);
            continue;
        }
        const uint64_t word = row[w];
        const uint64_t vm = valid_mask(w, nz);
        // Two running scans over the lane's 16 voxels instead of 16 independent 64-bit bit searches (3x fewer
        // instructions): left to right carrying the position of the last filled / last free voxel seen, right to left
        // carrying the next ones.  The carries start from the rest of the word (one clz / ffs per class and side) or,
        // where that is empty, from the neighbouring words (far_left / far_right: position or -+kFar).
        const int zb0 = 16 * sub, z0 = 64 * w + zb0;
        const uint32_t chunk = (uint32_t)(word >> zb0) & 0xFFFFu;
        const uint64_t lowm = (1ull << zb0) - 1ull;                                   // bits left of the chunk (all valid)
        const uint64_t higm = sub == 3 ? 0ull : (~0ull << (zb0 + 16));                // bits right of it
        const uint64_t fl = word & lowm, el = ~word & lowm;
        const uint64_t fr = word & vm & higm, er = ~word & vm & higm;
        int lastF = fl ? 64 * w + 63 - __clzll((long long)fl) : far_left(row, w, true);
        int lastE = el ? 64 * w + 63 - __clzll((long long)el) : far_left(row, w, false);
        int nextF = fr ? 64 * w + __ffsll((unsigned long long)fr) - 1 : far_right(row, w, W, nz, true);
        int nextE = er ? 64 * w + __ffsll((unsigned long long)er) - 1 : far_right(row, w, W, nz, false);
        int dl[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool own = (chunk >> k) & 1u;
            const int z = z0 + k;
            dl[k] = z - (own ? lastE : lastF);
            lastF = own ? z : lastF;
            lastE = own ? lastE : z;
        }
        uint32_t pk[8];
#pragma unroll
        for (int k = 15; k >= 0; --k) {
            const bool own = (chunk >> k) & 1u;
            const int z = z0 + k;
            const int dr = (own ? nextE : nextF) - z;
            nextF = own ? z : nextF;
            nextE = own ? nextE : z;
            const int d = min(min(dl[k], dr), kInf16);
            const uint32_t v = (uint32_t)(own ? -d : d) & 0xffffu;
            if (k & 1) pk[k >> 1] = v << 16; else pk[k >> 1] |= v;
        }
        uint4* dst = reinterpret_cast<uint4*>(out + (row0 + r) * nz + 16 * c);
        dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
    __syncthreads();                               // bitmap is reused by the next row group
    }
}

// packed 16-bit pairs in one 32-bit register (v_pk_*_u16 / _i16)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_min(as_us2(a), as_us2(b))); }
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_max(as_us2(a), as_us2(b))); }
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) { return as_u32(as_us2(a) + as_us2(b)); }
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return as_u32(as_us2(a) - as_us2(b)); }
// (a & m) | (b & ~m) as ONE v_bfi_b32.  Written as asm because the optimiser, seeing that m is a pair of 16-bit all-or-nothing
// halves, rewrites the C form into two 16-bit selects, a shift and a byte permute (4 - 5 instructions instead of 1: the
// packed window kernels ran with 60 % more VALU instructions than their source suggests).
__device__ __forceinline__ uint32_t bit_select(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ss2, a), __builtin_bit_cast(ss2, b)));
}
__device__ __forceinline__ uint32_t pk_neg_i16(uint32_t a) {
    return __builtin_bit_cast(uint32_t, (ss2)(0) - __builtin_bit_cast(ss2, a));
}

// Wave-private form of the fast path for nz = 16 CPR, CPR in {4 .. 64} lanes per row (nz = 64 .. 1024): a wave owns
// 64 / CPR whole rows per step (1024 voxels), keeps their bitmap in its own 128 bytes of LDS and never meets the other
// waves of the workgroup -- no workgroup barrier between "pack" and "search", so one wave's loads and stores run under
// another wave's bit searches (the barrier form ran at memory time PLUS compute time: 0.115 ms at 512^3 for 0.07 ms of
// traffic and 0.045 ms of VALU work).  The row classes come from two ballots instead of LDS atomics.
// BITS (round 6): `mask` is the linear bit field instead of the byte mask -- a lane's 16 voxels are one aligned 16-bit piece of
// it (2 bytes read per lane instead of 16: the z sweep of a bits-in build reads 1/8 B per voxel).
// row_any != nullptr (round 6, builds that go straight to the far-field pair): row_any[row] = the row holds a filled voxel, one byte per
// z row, written for EVERY row (nothing to clear, no atomics: a first form that raised one bit per x-plane with an atomic OR put
// 262 144 same-address accesses into the room scene's z sweep -- 0.09 -> 2.1 ms).  The far-field y sweep ORs the bytes of its
// x-plane and skips planes without a filled voxel; the x sweep then skips their row loads.  Rows without a filled voxel are NOT
// written to the z field then (the y sweep takes "+32767 everywhere" from the byte).
template <int CPR, bool BITS = false>
__global__ __launch_bounds__(kBlock) void k_sweep_z_wave16(const uint8_t* __restrict__ mask, int16_t* __restrict__ out,
                                                          int64_t nrows, const uint32_t* __restrict__ guard,
                                                          uint8_t* __restrict__ row_any = nullptr) {
    if (guard && *guard == 0u) return;
    constexpr int W = CPR / 4;                     // 64-bit words per row
    constexpr int RW = 64 / CPR;                   // rows per wave step
    constexpr int nz = 16 * CPR;
    __shared__ __attribute__((aligned(16))) uint64_t bm_all[(kBlock / 64) * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t* const bm = bm_all + wave * 16;
    uint16_t* const bm16 = reinterpret_cast<uint16_t*>(bm);
    const int r = lane / CPR, c = lane - r * CPR;
    const uint64_t* const row = bm + r * W;
    const int64_t ngroups = (nrows + RW - 1) / RW;
    const int64_t gstep = (int64_t)gridDim.x * (kBlock / 64);
    auto fetch = [&](int64_t g) -> uint4 {
        const int64_t rr = g * RW + r;
        if constexpr (BITS) {
            if (g < ngroups && rr < nrows) return make_uint4(reinterpret_cast<const uint16_t*>(mask)[rr * CPR + c], 0u, 0u, 0u);
        } else {
            if (g < ngroups && rr < nrows) return *reinterpret_cast<const uint4*>(mask + rr * nz + 16 * c);
        }
        return make_uint4(0u, 0u, 0u, 0u);
    };
    int64_t g = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    uint4 vnext = fetch(g);                 // (a second step in flight per wave changes nothing: measured)
    for (; g < ngroups; g += gstep) {
        const uint4 v = vnext;
        vnext = fetch(g + gstep);
        const int64_t rr = g * RW + r;
        const bool valid = rr < nrows;
        const uint32_t bits = BITS ? v.x : (nonzero_bits4(v.x) | (nonzero_bits4(v.y) << 4) | (nonzero_bits4(v.z) << 8) | (nonzero_bits4(v.w) << 12));
        const uint64_t bF = __ballot(bits != 0u), bE = __ballot(bits != 0xFFFFu);
        const uint64_t rmask = CPR == 64 ? ~0ull : (((1ull << (CPR & 63)) - 1ull) << (r * CPR));
        const bool anyF = (bF & rmask) != 0ull, anyE = (bE & rmask) != 0ull;
        bm16[lane] = (uint16_t)bits;
        if (row_any && valid && c == 0) row_any[rr] = anyF ? 1 : 0;            // (one lane per row)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (with row_any the consumer is the far-field y sweep, which takes "+32767 everywhere" from the row's byte: an all-free row --
        //  nearly every row of a sensed scene -- is not written at all)
        if (valid && !(row_any && !anyF)) {
            uint4* dst = reinterpret_cast<uint4*>(out + rr * nz + 16 * c);
            if (!(anyF && anyE)) {
                const uint32_t u = anyE ? 0x7fff7fffu : 0x80018001u;       // +32767 (all free) / -32767 (all filled)
                dst[0] = make_uint4(u, u, u, u);
                dst[1] = make_uint4(u, u, u, u);
            } else {
                const int w = c >> 2, sub = c & 3;
                const uint64_t word = row[w];
                const int zb0 = 16 * sub, z0 = 64 * w + zb0;
                const uint32_t chunk = bits;
                const uint64_t lowm = (1ull << zb0) - 1ull;
                const uint64_t higm = sub == 3 ? 0ull : (~0ull << (zb0 + 16));
                const uint64_t fl = word & lowm, el = ~word & lowm;
                const uint64_t fr = word & higm, er = ~word & higm;
                int lastF = fl ? 64 * w + 63 - __clzll((long long)fl) : far_left(row, w, true);
                int lastE = el ? 64 * w + 63 - __clzll((long long)el) : far_left(row, w, false);
                int nextF = fr ? 64 * w + __ffsll((unsigned long long)fr) - 1 : far_right(row, w, W, nz, true);
                int nextE = er ? 64 * w + __ffsll((unsigned long long)er) - 1 : far_right(row, w, W, nz, false);
                // Two running scans over the lane's 16 voxels with PACKED counters: one register holds {distance to the last
                // filled voxel, distance to the last free voxel} as 16-bit halves.  A voxel clears the half of its own class
                // (mask m), which leaves its distance to the opposite class in the other half and zero in its own; the
                // counters then advance by one (v_pk_add_u16).  Left and right scans meet in a packed min, and the signed
                // result is (low half) - (high half): +d for a free voxel, -d for a filled one.  "None on this side" starts
                // at kNone, which 16 steps cannot carry past 16 bits and which never survives the min: this branch only
                // runs on rows that hold both classes.
                constexpr int kNone = 0x7C00;
                uint32_t cl = (uint32_t)min(z0 - lastF, kNone) | ((uint32_t)min(z0 - lastE, kNone) << 16);
                uint32_t cr = (uint32_t)min(nextF - (z0 + 15), kNone) | ((uint32_t)min(nextE - (z0 + 15), kNone) << 16);
                uint32_t m[16], dl[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    m[k] = 0x0000FFFFu ^ (uint32_t)__builtin_amdgcn_sbfe((int)chunk, k, 1);   // filled: 0xFFFF0000, free: 0x0000FFFF
                    dl[k] = cl & m[k];
                    cl = pk_add_u16(dl[k], 0x00010001u);
                }
                uint32_t pk[8];
#pragma unroll
                for (int k = 15; k >= 1; k -= 2) {
                    const uint32_t r1 = cr & m[k];
                    cr = pk_add_u16(r1, 0x00010001u);
                    const uint32_t r0 = cr & m[k - 1];
                    cr = pk_add_u16(r0, 0x00010001u);
                    const uint32_t b1 = pk_min_u16(dl[k], r1), b0 = pk_min_u16(dl[k - 1], r0);     // voxels k, k - 1
                    const uint32_t lo = __builtin_amdgcn_perm(b1, b0, 0x05040100u);               // {lo(b0), lo(b1)}
                    const uint32_t hi = __builtin_amdgcn_perm(b1, b0, 0x07060302u);               // {hi(b0), hi(b1)}
                    pk[k >> 1] = pk_sub_u16(lo, hi);
                }
                dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                // the bitmap is rewritten by the next step
    }
}

// Generic path: any nz, any loader (mask bytes or COLLISION_CELL records).
// A wave ballots 64 voxels into one bitmap word; a lane then owns one voxel.
template <class Loader>
__global__ __launch_bounds__(kBlock) void k_sweep_z_generic(Loader ld, int16_t* __restrict__ out,
                                                           int64_t nrows, int nz, int rpb,
                                                           const uint32_t* __restrict__ guard) {
    if (guard && *guard == 0u) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* bm = reinterpret_cast<uint64_t*>(smem_raw);
    const int W = (nz + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row0 = (int64_t)blockIdx.x * rpb; row0 < nrows; row0 += (int64_t)gridDim.x * rpb) {
    const int nr = (int)min((int64_t)rpb, nrows - row0);
    for (int s = wave; s < nr * W; s += kBlock / 64) {
        const int r = s / W, ww = s - r * W;
        const int z = 64 * ww + lane;
        const bool f = (z < nz) ? ld.filled((row0 + r) * nz + z) : false;
        const uint64_t word = __ballot(f);
        if (lane == 0) bm[s] = word;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nr * nz; s += kBlock) {
        const int r = s / nz, z = s - r * nz;
        const uint64_t* row = bm + r * W;
        const int w = z >> 6, zb = z & 63;
        const uint64_t word = row[w];
        const bool own = (word >> zb) & 1ull;
        // only the opposite class is needed for a single voxel
        const int L = far_left(row, w, !own), R = far_right(row, w, W, nz, !own);
        const int d = z_signed_distance(word, valid_mask(w, nz), zb, z, L, L, R, R);
        out[(row0 + r) * nz + z] = (int16_t)d;
    }
    __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K2 / K3: marching sweep along a strided axis.
// ---------------------------------------------------------------------------

struct SweepArgs {
    const void* in;
    void* out;
    int64_t ncols;         // V-wide columns
    int64_t cpl;           // columns per outer unit (K2: nz/V per x-plane; K3: all)
    int64_t outer_stride;  // elements between outer units (K2: ny*nz)
    int64_t line_stride;   // elements between successive positions of a line (K2: nz, K3: ny*nz)
    int L;                 // positions available along the line (K3 slab mode: halo included)
    int out_lo, out_hi;    // positions written: [out_lo, out_hi); output row = p - out_lo
    int T;                 // positions marched per thread
    // K3 only
    double resolution;
    int lo_truncated, hi_truncated;   // real rows exist beyond the buffer (slab mode)
    int64_t x_global;                 // grid x of position out_lo
    int64_t nx_global, ny, nz;        // full extents (virtual border)
    int64_t y_off, ny_glob;           // K3 on a y slab: grid y of local row 0 and the full y extent (0 / ny otherwise)
    uint32_t* maxdsq;                 // [0] free, [1] filled
    uint32_t* status;                 // bit 0: unresolved voxel (slab mode)
    // K2 with 16-bit output (plane16 + side table, see sdfgpu_sweep_x16.hpp)
    int out16;
    int32_t* side;
    const uint32_t* guard;            // non-null: run only if *guard != 0 (dense path left voxels uncertified)
    // bounded outward scan (far-field scenes are redone by the envelope kernel, sdfgpu_envelope.hpp)
    int max_scan;                     // 0 = unbounded
    uint32_t* far_flag;               // raised when a voxel was still undecided after max_scan rows
};

template <int STAGE, int V> struct InVecT;
template <> struct InVecT<2, 4> { using type = short4; };
template <> struct InVecT<2, 1> { using type = short; };
template <> struct InVecT<3, 4> { using type = int4; };
template <> struct InVecT<3, 1> { using type = int; };

// raw vector -> signed squared values
template <int STAGE, int V>
__device__ __forceinline__ void unpack_row(const typename InVecT<STAGE, V>::type& raw, int (&s)[V]) {
    if constexpr (STAGE == 2) {
        int g[V];
        if constexpr (V == 4) { g[0] = raw.x; g[1] = raw.y; g[2] = raw.z; g[3] = raw.w; }
        else { g[0] = raw; }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int a = abs(g[k]);
            const int sq = (a >= kInf16) ? kInf32 : a * a;
            s[k] = g[k] < 0 ? -sq : sq;
        }
    } else {
        if constexpr (V == 4) { s[0] = raw.x; s[1] = raw.y; s[2] = raw.z; s[3] = raw.w; }
        else { s[0] = raw; }
    }
}

template <int STAGE, int V>
__device__ __forceinline__ typename InVecT<STAGE, V>::type load_raw(const void* in, int64_t elem) {
    using VT = typename InVecT<STAGE, V>::type;
    if constexpr (STAGE == 2)
        return *reinterpret_cast<const VT*>(reinterpret_cast<const int16_t*>(in) + elem);
    else
        return *reinterpret_cast<const VT*>(reinterpret_cast<const int32_t*>(in) + elem);
}

// candidate offered by site value su at squared offset dd to a voxel whose class mask is m
// (m = 0 free / -1 filled, negm = -m):  same class -> |su|, other class -> 0.
__device__ __forceinline__ int candidate(int su, int m, int negm, int dd) {
    const int t = (su ^ m) + negm;          // m ? -su : su   (v_xad_u32)
    return max(t, 0) + dd;
}

template <int STAGE, int V, int H, bool VB>
__global__ __launch_bounds__(kBlock) void k_sweep_march(const SweepArgs a) {
    constexpr int R = 2 * H + 1;
    using VT = typename InVecT<STAGE, V>::type;
    if (a.guard && *a.guard == 0u) return;
    int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = c < a.ncols;
    if (!valid) c = a.ncols - 1;            // keep the lane alive for wave-wide ops; stores are masked
    const int p0 = a.out_lo + (int)blockIdx.y * a.T;
    const int p1 = min(a.out_hi, p0 + a.T);
    if (p0 >= p1) return;                   // block-uniform

    int64_t base;
    if (a.cpl == a.ncols) base = c * V;
    else { const int64_t o = c / a.cpl; base = o * a.outer_stride + (c - o * a.cpl) * V; }
    const int64_t ls = a.line_stride;
    const int L = a.L;

    int win[R][V];                          // win[slot(row)] ; slot(row) = (row - p0 + H) mod R
    int mxF = 0, mxQ = 0;                   // K3: max d^2 over free / filled voxels
    bool unresolved = false;
    bool far = false;

    // virtual-border coordinates of this lane's V voxels (K3 only)
    int vy[V], vz[V];
    if constexpr (VB && STAGE == 3) {
        const int64_t q0 = c * V;
        int y0 = (int)(q0 / a.nz), z0 = (int)(q0 - (int64_t)y0 * a.nz);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            int zk = z0 + k, yk = y0;
            while (zk >= a.nz) { zk -= (int)a.nz; ++yk; }
            vy[k] = yk; vz[k] = zk;
        }
    }

    auto load_checked = [&](int p, int (&dst)[V]) {
        if (p >= 0 && p < L) {
            const VT raw = load_raw<STAGE, V>(a.in, base + (int64_t)p * ls);
            unpack_row<STAGE, V>(raw, dst);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) dst[k] = 0;   // never used: callers test the range
        }
    };

    // one output position; CHECK = some window rows may lie outside [0, L)
    // (r is a compile-time constant so every window access has a static register index)
    auto step = [&](int p, auto r_tag, auto check_tag) {
        constexpr int r = decltype(r_tag)::value;
        constexpr bool CHECK = decltype(check_tag)::value;
        const int (&cen)[V] = win[(r + H) % R];
        int best[V], m[V], negm[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            m[k] = cen[k] >> 31;
            negm[k] = -m[k];
            best[k] = (cen[k] ^ m[k]) + negm[k];      // |center|
        }
#pragma unroll
        for (int d = 1; d <= H; ++d) {
            const int dd = d * d;
            const int (&lo)[V] = win[(r + H - d) % R];
            const int (&hi)[V] = win[(r + H + d) % R];
            const bool lo_ok = !CHECK || (p - d >= 0);
            const bool hi_ok = !CHECK || (p + d < L);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                if (lo_ok) best[k] = min(best[k], candidate(lo[k], m[k], negm[k], dd));
                if (hi_ok) best[k] = min(best[k], candidate(hi[k], m[k], negm[k], dd));
            }
        }
        // outward scan beyond the register window (exactness for sparse scenes)
        bool need = false;
        int inexact = 0;                        // bit k: voxel k was still undecided when the bounded scan stopped
#pragma unroll
        for (int k = 0; k < V; ++k) need |= best[k] >= (H + 1) * (H + 1);
        if (__any(need)) {
            for (int d = H + 1;; ++d) {
                const int lo = p - d, hi = p + d;
                if (lo < 0 && hi >= L) break;
                const int dd = d * d;
                bool act = false;
#pragma unroll
                for (int k = 0; k < V; ++k) act |= dd < best[k];
                if (!__any(act)) break;
                if (a.max_scan && d > a.max_scan) {                          // leave it to the envelope kernel
                    far |= act;
#pragma unroll
                    for (int k = 0; k < V; ++k) inexact |= (dd < best[k]) ? (1 << k) : 0;
                    break;
                }
                if (act) {
                    int s[V];
                    if (lo >= 0) {
                        unpack_row<STAGE, V>(load_raw<STAGE, V>(a.in, base + (int64_t)lo * ls), s);
#pragma unroll
                        for (int k = 0; k < V; ++k) best[k] = min(best[k], candidate(s[k], m[k], negm[k], dd));
                    }
                    if (hi < L) {
                        unpack_row<STAGE, V>(load_raw<STAGE, V>(a.in, base + (int64_t)hi * ls), s);
#pragma unroll
                        for (int k = 0; k < V; ++k) best[k] = min(best[k], candidate(s[k], m[k], negm[k], dd));
                    }
                }
            }
        }
        if constexpr (STAGE == 3) {
            // slab mode: a row beyond the buffer at distance dd could still offer dd^2
            if (a.lo_truncated) {
                const int dd = p + 1;
#pragma unroll
                for (int k = 0; k < V; ++k) unresolved |= best[k] > dd * dd;
            }
            if (a.hi_truncated) {
                const int dd = L - p;
#pragma unroll
                for (int k = 0; k < V; ++k) unresolved |= best[k] > dd * dd;
            }
        }
        const int64_t oelem = base + (int64_t)(p - a.out_lo) * ls;
        if constexpr (STAGE == 2) {
            int o[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int D = min(best[k], kInf32);
                o[k] = (D ^ m[k]) + negm[k];
            }
            if (valid) {
                if (a.out16) {
                    // plane16 format: saturate at +-32767, exact values of saturated groups to the side table
                    if constexpr (V == 4) {
                        bool sat = false;
                        int s16[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int mag = abs(o[k]);
                            sat |= mag >= 32767;
                            s16[k] = o[k] < 0 ? -min(mag, 32767) : min(mag, 32767);
                        }
                        if (sat) *reinterpret_cast<int4*>(a.side + oelem) = make_int4(o[0], o[1], o[2], o[3]);
                        *reinterpret_cast<uint2*>(reinterpret_cast<int16_t*>(a.out) + oelem) =
                            make_uint2(((uint32_t)s16[0] & 0xffffu) | ((uint32_t)s16[1] << 16),
                                       ((uint32_t)s16[2] & 0xffffu) | ((uint32_t)s16[3] << 16));
                    }
                } else {
                    int32_t* dst = reinterpret_cast<int32_t*>(a.out) + oelem;
                    if constexpr (V == 4) *reinterpret_cast<int4*>(dst) = make_int4(o[0], o[1], o[2], o[3]);
                    else *dst = o[0];
                }
            }
        } else {
            float o[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                int D = min(best[k], kInf32);
                if constexpr (VB) {
                    // net effect of sdf_generation.hpp:287-419: D = min(D, b^2), b = axis
                    // distance to the virtual layer over axes with more than one cell
                    int64_t b = kInf32;
                    const int64_t gx = a.x_global + (p - a.out_lo);
                    if (a.nx_global > 1) b = min(b, min(gx + 1, a.nx_global - gx));
                    if (a.ny_glob > 1) b = min(b, min((int64_t)vy[k] + a.y_off + 1, a.ny_glob - a.y_off - vy[k]));
                    if (a.nz > 1) b = min(b, min((int64_t)vz[k] + 1, a.nz - vz[k]));
                    if (b < 32768) D = min(D, (int)(b * b));
                }
                // (a voxel the bounded scan left undecided holds an upper bound only: the far-field kernel behind this sweep
                //  rewrites it and reports its true value; it must not reach the extrema from here)
                if (!((inexact >> k) & 1)) { if (m[k]) mxQ = max(mxQ, D); else mxF = max(mxF, D); }
                // sdf_generation.hpp:254-265: sqrt and multiply in double, one narrowing cast
                const float f = (D >= kInf32) ? __builtin_inff()
                                              : (float)(sqrt((double)D) * a.resolution);
                o[k] = m[k] ? -f : f;
            }
            if (valid) {
                float* dst = reinterpret_cast<float*>(a.out) + oelem;
                if constexpr (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                else *dst = o[0];
            }
        }
    };

    // prologue: rows p0-H .. p0+H-1 -> slots 0 .. 2H-1
#pragma unroll
    for (int k = 0; k < 2 * H; ++k) load_checked(p0 - H + k, win[k]);

    for (int pb = p0; pb < p1; pb += R) {
        const bool fast = (pb - H >= 0) && (pb + R - 1 + H < L) && (pb + R <= p1);
        if (fast) {
            // issue the whole batch of row loads first (R independent 8/16-B loads per lane)
            VT raw[R];
#pragma unroll
            for (int r = 0; r < R; ++r) raw[r] = load_raw<STAGE, V>(a.in, base + (int64_t)(pb + r + H) * ls);
            static_for<R>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                unpack_row<STAGE, V>(raw[r], win[(r + 2 * H) % R]);
                step(pb + r, rc, std::false_type{});
            });
        } else {
            static_for<R>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int p = pb + r;
                if (p < p1) {
                    load_checked(p + H, win[(r + 2 * H) % R]);
                    step(p, rc, std::true_type{});
                }
            });
        }
    }

    if (a.far_flag) {
        if (__any(far) && (threadIdx.x & 63) == 0) raise_flag(a.far_flag);
    }
    if constexpr (STAGE == 3) {
        // wave-level max, one atomic per wave per class
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mxF = max(mxF, __shfl_xor(mxF, off));
            mxQ = max(mxQ, __shfl_xor(mxQ, off));
        }
        const bool any_unres = __any(unresolved);
        if ((threadIdx.x & 63) == 0) {
            slot_max2(a.maxdsq, blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), mxF, mxQ);   // a.maxdsq = slot array
            if (any_unres && a.status) raise_flag(a.status);
        }
    }
}

// ---------------------------------------------------------------------------
// N4: TaggedObjectCollisionMapGrid predicates (tagged_object_collision_map.hpp:730-856) on raw
// TAGGED_OBJECT_COLLISION_CELL records -> byte mask.  A cell is filled iff its occupancy says so
// (occ > 0.5, or == 0.5 when unknown_is_filled) AND its object id passes the filter:
//   mode 0: any object                      (free_sdf_filled_fn :736-749, and objects_to_use empty :826)
//   mode 1: object_id > 0                   (object_filled_fn :757-775, "named objects")
//   mode 2: object_id in the given id list  (object_use_map :817-827); the list arrives sorted, any length
// ---------------------------------------------------------------------------
SDFGPU_KERNEL __launch_bounds__(kBlock) void k_classify_tagged(const char* __restrict__ cells, int64_t stride,
                                                           int64_t occ_off, int64_t obj_off, int unknown_is_filled,
                                                           int mode, const uint32_t* __restrict__ ids, int n_ids,
                                                           int64_t n, uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float occ = *reinterpret_cast<const float*>(cells + i * stride + occ_off);
    const uint32_t obj = *reinterpret_cast<const uint32_t*>(cells + i * stride + obj_off);
    bool pass = mode == 0 || (mode == 1 && obj > 0u);
    if (mode == 2) {                                   // ids are sorted ascending by the host: binary search
        int lo = 0, hi = n_ids;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ids[mid] < obj) lo = mid + 1; else hi = mid;
        }
        pass = lo < n_ids && ids[lo] == obj;
    }
    const bool occupied = (occ > 0.5f) || (unknown_is_filled && (occ == 0.5f));
    mask[i] = (pass && occupied) ? 1 : 0;
}

// collision_map.hpp:680-712 predicate on raw COLLISION_CELL records -> byte mask (slab pipelines take masks)
SDFGPU_KERNEL __launch_bounds__(kBlock) void k_classify_cells(CellLoader ld, int64_t n, uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) mask[i] = ld.filled(i) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// N2: point cloud -> occupancy (scripts/3d_sdf_demo_rviz.py:22-29): idx = trunc((p - origin) / res),
// vg[ix, iy, iz] = 1.  Points whose index falls outside the grid are dropped.  fp32 points (the
// PointCloud2 convention), index arithmetic in fp64 like numpy's.
// ---------------------------------------------------------------------------
SDFGPU_KERNEL __launch_bounds__(kBlock) void k_voxelize_points(const float* __restrict__ pts, int64_t n_points,
                                                           double ox, double oy, double oz, double res,
                                                           int64_t nx, int64_t ny, int64_t nz,
                                                           uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_points) return;
    const double fx = ((double)pts[3 * i + 0] - ox) / res;
    const double fy = ((double)pts[3 * i + 1] - oy) / res;
    const double fz = ((double)pts[3 * i + 2] - oz) / res;
    if (!(fx > -1.0 && fy > -1.0 && fz > -1.0 && fx < (double)nx && fy < (double)ny && fz < (double)nz)) return;  // also NaN
    const int64_t ix = (int64_t)fx, iy = (int64_t)fy, iz = (int64_t)fz;      // truncation toward zero, like astype(int64)
    mask[(ix * ny + iy) * nz + iz] = 1;                                       // benign race: every writer stores 1
}

// The same scatter into the bit field the bits-in entry points take (round 6): one atomic OR per point, 1/8 B per voxel to clear
// and nothing to pack afterwards -- a streaming frame goes point cloud -> bits -> SDF without a byte mask in between.
SDFGPU_KERNEL __launch_bounds__(kBlock) void k_voxelize_points_bits(const float* __restrict__ pts, int64_t n_points,
                                                                double ox, double oy, double oz, double res,
                                                                int64_t nx, int64_t ny, int64_t nz,
                                                                uint32_t* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_points) return;
    const double fx = ((double)pts[3 * i + 0] - ox) / res;
    const double fy = ((double)pts[3 * i + 1] - oy) / res;
    const double fz = ((double)pts[3 * i + 2] - oz) / res;
    if (!(fx > -1.0 && fy > -1.0 && fz > -1.0 && fx < (double)nx && fy < (double)ny && fz < (double)nz)) return;  // also NaN
    const int64_t ix = (int64_t)fx, iy = (int64_t)fy, iz = (int64_t)fz;      // truncation toward zero, like astype(int64)
    const int64_t v = (ix * ny + iy) * nz + iz;
    atomic_or_if_new(bits + (v >> 5), 1u << (v & 31));
}

// ---------------------------------------------------------------------------
// N1: grid-aligned gradient of the whole field, sdf.hpp:432-526.
// The reciprocals the reference computes per call -- 1 / (2 res) in the interior (:447), 1 / ((hi - lo) res) on the
// boundary shell (:464-512, hi - lo = 1 or 2) -- are computed ONCE on the host with the same double operations and passed
// in: fp64 divisions on the device cost ~30 instructions each, and at nz = 512 every wave holds a z-boundary voxel.
// ---------------------------------------------------------------------------
struct GradScale {
    double inv2;      // 1.0 / (2.0 * res)               interior (:447)
    double inv_w1;    // 1.0 / ((double)1 * res)          shell, clamped interval of 1 cell
    double inv_w2;    // 1.0 / ((double)2 * res)          shell, interval of 2 cells
    float inv2f;      // (float)inv2 when that is exact (F32SCALE kernels)
};

// one voxel, any position: interior -> float subtraction, double scale (:447-458); shell -> clamped indices, double
// subtraction (:464-512); shell without edge gradients -> NaN (the reference returns an empty vector)
__device__ __forceinline__ void gradient_one(const float* __restrict__ f, int64_t i, int64_t x, int64_t y, int64_t z,
                                             int64_t nx, int64_t ny, int64_t nz, const GradScale& sc, int edge, double (&g)[3]) {
    const int64_t sx = ny * nz, sy = nz;
    const bool interior = x > 0 && y > 0 && z > 0 && x < nx - 1 && y < ny - 1 && z < nz - 1;
    if (interior) {
        g[0] = (double)(f[i + sx] - f[i - sx]) * sc.inv2;
        g[1] = (double)(f[i + sy] - f[i - sy]) * sc.inv2;
        g[2] = (double)(f[i + 1] - f[i - 1]) * sc.inv2;
    } else if (edge) {
        const int64_t lx = max((int64_t)0, x - 1), hx = min(nx - 1, x + 1);
        const int64_t ly = max((int64_t)0, y - 1), hy = min(ny - 1, y + 1);
        const int64_t lz = max((int64_t)0, z - 1), hz = min(nz - 1, z + 1);
        const int wx = (int)(hx - lx), wy = (int)(hy - ly), wz = (int)(hz - lz);
        g[0] = g[1] = g[2] = 0.0;
        if (wx > 0) g[0] = ((double)f[i + (hx - x) * sx] - (double)f[i - (x - lx) * sx]) * (wx == 2 ? sc.inv_w2 : sc.inv_w1);
        if (wy > 0) g[1] = ((double)f[i + (hy - y) * sy] - (double)f[i - (y - ly) * sy]) * (wy == 2 ? sc.inv_w2 : sc.inv_w1);
        if (wz > 0) g[2] = ((double)f[i + (hz - z)] - (double)f[i - (z - lz)]) * (wz == 2 ? sc.inv_w2 : sc.inv_w1);
    } else {
        g[0] = g[1] = g[2] = __builtin_nan("");
    }
}

template <typename OutT>
__global__ __launch_bounds__(kBlock) void k_gradient(const float* __restrict__ f, OutT* __restrict__ g,
                                                    int64_t nx, int64_t ny, int64_t nz, const GradScale sc, int edge) {
    const int64_t n = nx * ny * nz;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t z = i % nz, y = (i / nz) % ny, x = i / (nz * ny);
    double v[3];
    gradient_one(f, i, x, y, z, nx, ny, nz, sc, edge, v);
    g[3 * i + 0] = (OutT)v[0];
    g[3 * i + 1] = (OutT)v[1];
    g[3 * i + 2] = (OutT)v[2];
}

// fp32 output, 4 consecutive z per lane (nz % 4 == 0, 16-byte aligned): interior groups take 16-byte loads of the six
// neighbour rows and write 48 contiguous bytes; a group that touches a grid face finishes its voxels one by one with
// gradient_one (at nz = 512 that is one lane in every wave, so that path is kept lean: no divisions, see GradScale).
// F32SCALE: 1 / (2 res) is exactly representable in fp32 (res = 0.01, 0.02, 0.05, 0.1, 0.25, 1 ...): the product of the
// fp32 difference with it has at most 48 significant bits, so the reference's double multiply is exact and its narrowing
// to float is ONE rounding of the exact product -- which is what the fp32 multiply computes.  Same bits, a third of the
// instructions (no conversions, no fp64 multiply).
template <bool F32SCALE>
__global__ __launch_bounds__(kBlock) void k_gradient_f32x4(const float* __restrict__ f, float* __restrict__ g,
                                                          int64_t nx, int64_t ny, int64_t nz, const GradScale sc, int edge,
                                                          int gshift) {
    // A lane's 4 voxels give 12 consecutive floats (48 B).  Written straight from the lane, every store instruction
    // would touch 64 x 16 B at a 48 B stride (24 cache lines instead of 8); the wave's 3 KiB are therefore transposed
    // through LDS so that each of the 3 store instructions writes one contiguous 1 KiB.
    __shared__ __attribute__((aligned(16))) float stage[(kBlock / 64) * 64 * 12];
    // (tried in round 2, no gain at 512^3: an XCD-contiguous workgroup order and non-temporal stores)
    // Grid: blockIdx.y walks the x planes, blockIdx.x the groups of 4 voxels inside a plane -- one 32-bit division (a shift
    // when nz / 4 is a power of two) per lane instead of three 64-bit ones, which were ~250 of the kernel's ~350
    // instructions per lane and kept it off the memory roofline (0.49 -> 0.43 ms at 512^3).
    const uint32_t G = (uint32_t)(nz >> 2);                      // groups per z row
    const uint32_t P = (uint32_t)ny * G;                         // groups per x plane (the launcher checks the range)
    const int lane = threadIdx.x & 63;
    float* st = stage + (threadIdx.x >> 6) * (64 * 12);
    for (int64_t x = blockIdx.y; x < nx; x += gridDim.y) {       // (more than one pass only when nx > 65535)
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;        // group inside the plane
    if (r < P) {
        const uint32_t yy = gshift >= 0 ? r >> gshift : r / G;
        const int64_t y = yy, z = (int64_t)(r - yy * G) * 4;
        const int64_t sx = ny * nz, sy = nz;
        const int64_t i = x * sx + (int64_t)r * 4;
        float o[12];
        if (x > 0 && x < nx - 1 && y > 0 && y < ny - 1) {
            // x and y interior.  A group on a z face (z == 0 or z + 4 == nz: one lane of EVERY wave at nz = 512) still takes
            // this path -- its face voxel is patched below from the rows already in registers; sending the whole group
            // through the per-voxel path put four dependent load chains on every wave's critical path (0.63 ms at 512^3
            // where the same stencil without them runs in 0.43)
            const bool zlo = z == 0, zhi = z + 4 >= nz;
            const float4 c = *reinterpret_cast<const float4*>(f + i);
            const float4 xp = *reinterpret_cast<const float4*>(f + i + sx), xm = *reinterpret_cast<const float4*>(f + i - sx);
            const float4 yp = *reinterpret_cast<const float4*>(f + i + sy), ym = *reinterpret_cast<const float4*>(f + i - sy);
            const float zm = zlo ? c.x : f[i - 1], zp = zhi ? c.w : f[i + 4];
            const float cz[6] = {zm, c.x, c.y, c.z, c.w, zp};
            const float xpv[4] = {xp.x, xp.y, xp.z, xp.w}, xmv[4] = {xm.x, xm.y, xm.z, xm.w};
            const float ypv[4] = {yp.x, yp.y, yp.z, yp.w}, ymv[4] = {ym.x, ym.y, ym.z, ym.w};
            if constexpr (F32SCALE) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[3 * k + 0] = (xpv[k] - xmv[k]) * sc.inv2f;
                    o[3 * k + 1] = (ypv[k] - ymv[k]) * sc.inv2f;
                    o[3 * k + 2] = (cz[k + 2] - cz[k]) * sc.inv2f;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[3 * k + 0] = (float)((double)(xpv[k] - xmv[k]) * sc.inv2);
                    o[3 * k + 1] = (float)((double)(ypv[k] - ymv[k]) * sc.inv2);
                    o[3 * k + 2] = (float)((double)(cz[k + 2] - cz[k]) * sc.inv2);
                }
            }
            if (zlo | zhi) {
                // the face voxel is on the boundary shell (sdf.hpp:464-512): double subtraction, x / y over 2 cells, z over the
                // one cell inside the grid; NaN when edge gradients are off.  (nz >= 8 here: nz % 4 == 0 and a group with both
                // faces, nz == 4, is handled too: voxel 0 and voxel 3 are patched independently.)
                const double nan = __builtin_nan("");
                if (zlo) {
                    o[0] = edge ? (float)(((double)xp.x - (double)xm.x) * sc.inv_w2) : (float)nan;
                    o[1] = edge ? (float)(((double)yp.x - (double)ym.x) * sc.inv_w2) : (float)nan;
                    o[2] = edge ? (float)(((double)c.y - (double)c.x) * sc.inv_w1) : (float)nan;
                }
                if (zhi) {
                    o[9] = edge ? (float)(((double)xp.w - (double)xm.w) * sc.inv_w2) : (float)nan;
                    o[10] = edge ? (float)(((double)yp.w - (double)ym.w) * sc.inv_w2) : (float)nan;
                    o[11] = edge ? (float)(((double)c.w - (double)c.z) * sc.inv_w1) : (float)nan;
                }
            }
        } else if (!edge) {                       // rows on an x or y face: boundary shell, no edge gradients (:464)
            const float nanf = (float)__builtin_nan("");
#pragma unroll
            for (int k = 0; k < 12; ++k) o[k] = nanf;
        } else {
            // rows on an x or y face of the grid: every voxel of the group is on the boundary shell (:464-512) -- clamped
            // neighbour indices, double subtraction, interval of 1 or 2 cells per axis (0: singleton axis, gradient 0).
            // Same five 16-byte row loads as the interior path with the out-of-grid neighbour replaced by the row itself:
            // whole waves take this path (a row is 128 lanes at nz = 512), and the per-voxel form (gradient_one: six scalar
            // loads per voxel behind index arithmetic) made the 0.8 % of waves on the faces cost 15 % of the kernel.
            const int64_t dxl = x > 0 ? sx : 0, dxh = x < nx - 1 ? sx : 0;
            const int64_t dyl = y > 0 ? sy : 0, dyh = y < ny - 1 ? sy : 0;
            const int wx = (x > 0 ? 1 : 0) + (x < nx - 1 ? 1 : 0), wy = (y > 0 ? 1 : 0) + (y < ny - 1 ? 1 : 0);
            const double kx = wx == 2 ? sc.inv_w2 : (wx == 1 ? sc.inv_w1 : 0.0);
            const double ky = wy == 2 ? sc.inv_w2 : (wy == 1 ? sc.inv_w1 : 0.0);
            const float4 c = *reinterpret_cast<const float4*>(f + i);
            const float4 xp = *reinterpret_cast<const float4*>(f + i + dxh), xm = *reinterpret_cast<const float4*>(f + i - dxl);
            const float4 yp = *reinterpret_cast<const float4*>(f + i + dyh), ym = *reinterpret_cast<const float4*>(f + i - dyl);
            const float zm = z > 0 ? f[i - 1] : c.x, zp = z + 4 < nz ? f[i + 4] : c.w;
            const float cz[6] = {zm, c.x, c.y, c.z, c.w, zp};
            const float xpv[4] = {xp.x, xp.y, xp.z, xp.w}, xmv[4] = {xm.x, xm.y, xm.z, xm.w};
            const float ypv[4] = {yp.x, yp.y, yp.z, yp.w}, ymv[4] = {ym.x, ym.y, ym.z, ym.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool lo = z + k > 0, hi = z + k < nz - 1;
                const double kz = (lo && hi) ? sc.inv_w2 : ((lo || hi) ? sc.inv_w1 : 0.0);
                const float zl = lo ? cz[k] : cz[k + 1], zh = hi ? cz[k + 2] : cz[k + 1];
                o[3 * k + 0] = wx ? (float)(((double)xpv[k] - (double)xmv[k]) * kx) : 0.0f;     // (inf - inf on a singleton axis)
                o[3 * k + 1] = wy ? (float)(((double)ypv[k] - (double)ymv[k]) * ky) : 0.0f;
                o[3 * k + 2] = (float)(((double)zh - (double)zl) * kz);
            }
        }
        float4* d = reinterpret_cast<float4*>(st + lane * 12);
        d[0] = make_float4(o[0], o[1], o[2], o[3]);
        d[1] = make_float4(o[4], o[5], o[6], o[7]);
        d[2] = make_float4(o[8], o[9], o[10], o[11]);
    }
    __syncthreads();
    const uint32_t r0 = r - (uint32_t)lane;                          // first group of this wave
    if (r0 < P) {
        const int valid = (int)min(64u, P - r0) * 12;                // floats this wave produced (multiple of 4)
        float* dst = g + 3 * (x * ny * nz + (int64_t)r0 * 4);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = k * 256 + lane * 4;
            if (idx < valid) *reinterpret_cast<float4*>(dst + idx) = *reinterpret_cast<const float4*>(st + idx);
        }
    }
    __syncthreads();                                                 // (the staging area is reused by the next plane)
    }
}

// ---------------------------------------------------------------------------
// N1 (query side): batched SignedDistanceField::EstimateDistance4d (sdf.hpp:947-961) and GetGradient4d
// (:383-430) at arbitrary world-frame points, one lane per point.  The estimate is the reference's trilinear
// inter/extrapolation of the 8 surrounding cell centres (:836-902) whose values are first shrunk by half a
// cell toward the surface (:773-796); the neighbour pair per axis follows :798-833 (shifted inward at a grid
// face, collapsed on a singleton axis).  Double arithmetic throughout, like the reference.
// ---------------------------------------------------------------------------
struct QueryArgs {
    const float* sdf;
    const double* points;      // [n][3]
    double* distance;          // [n] or null
    double* gradient;          // [n][3] or null
    uint8_t* flags;            // [n] or null: bit0 inside, bit1 gradient available
    int64_t n, nx, ny, nz;
    double res, inv_res, oob;
    double w2g[12];            // row-major 3x4 inverse origin transform
    double rot[9];             // row-major 3x3 rotation grid -> world
    int edge;
};

__device__ __forceinline__ void query_axis_pair(int64_t i, int64_t n, double offset, int64_t& lower, int64_t& upper) {
    lower = i; upper = i;
    if (offset >= 0.0) {
        upper = i + 1;
        if (upper >= n) { upper = i; lower = i - 1; if (lower < 0) lower = i; }
    } else {
        lower = i - 1;
        if (lower < 0) { upper = i + 1; lower = i; if (upper >= n) upper = i; }
    }
}

__device__ __forceinline__ double query_bilinear(double l1, double h1, double l2, double h2, double q1, double q2,
                                                 double ll, double lh, double hl, double hh) {
    const double multiplier = 1.0 / ((h1 - l1) * (h2 - l2));
    const double a0 = multiplier * (h1 - q1), a1 = multiplier * (q1 - l1);
    const double r0 = a0 * ll + a1 * hl, r1 = a0 * lh + a1 * hh;
    return r0 * (h2 - q2) + r1 * (q2 - l2);
}

SDFGPU_KERNEL __launch_bounds__(kBlock) void k_query_points(const QueryArgs a) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= a.n) return;
    const double px = a.points[3 * t], py = a.points[3 * t + 1], pz = a.points[3 * t + 2];
    const double gx = a.w2g[0] * px + a.w2g[1] * py + a.w2g[2] * pz + a.w2g[3];
    const double gy = a.w2g[4] * px + a.w2g[5] * py + a.w2g[6] * pz + a.w2g[7];
    const double gz = a.w2g[8] * px + a.w2g[9] * py + a.w2g[10] * pz + a.w2g[11];
    const double fx = floor(gx * a.inv_res), fy = floor(gy * a.inv_res), fz = floor(gz * a.inv_res);
    const bool inside = fx >= 0.0 && fy >= 0.0 && fz >= 0.0 && fx < (double)a.nx && fy < (double)a.ny && fz < (double)a.nz;
    const double nan = __builtin_nan("");
    if (!inside) {
        if (a.distance) a.distance[t] = a.oob;
        if (a.gradient) { a.gradient[3 * t] = nan; a.gradient[3 * t + 1] = nan; a.gradient[3 * t + 2] = nan; }
        if (a.flags) a.flags[t] = 0;
        return;
    }
    const int64_t x = (int64_t)fx, y = (int64_t)fy, z = (int64_t)fz;
    const int64_t sx = a.ny * a.nz, sy = a.nz;
    const float* f = a.sdf;
    if (a.distance) {
        const double half = a.res * 0.5;
        auto D = [&](int64_t xi, int64_t yi, int64_t zi) -> double {
            const double d = (double)f[xi * sx + yi * sy + zi];
            return d >= 0.0 ? d - half : d + half;
        };
        int64_t x0, x1, y0, y1, z0, z1;
        query_axis_pair(x, a.nx, gx - a.res * ((double)x + 0.5), x0, x1);
        query_axis_pair(y, a.ny, gy - a.res * ((double)y + 0.5), y0, y1);
        query_axis_pair(z, a.nz, gz - a.res * ((double)z + 0.5), z0, z1);
        const double lx = a.res * ((double)x0 + 0.5), ly = a.res * ((double)y0 + 0.5), lz = a.res * ((double)z0 + 0.5);
        const double mz = query_bilinear(lx, lx + a.res, ly, ly + a.res, gx, gy, D(x0, y0, z0), D(x0, y1, z0), D(x1, y0, z0), D(x1, y1, z0));
        const double pzv = query_bilinear(lx, lx + a.res, ly, ly + a.res, gx, gy, D(x0, y0, z1), D(x0, y1, z1), D(x1, y0, z1), D(x1, y1, z1));
        const double slope = (pzv - mz) * (1.0 / a.res);
        a.distance[t] = mz + ((gz - lz) * slope);
    }
    bool have_grad = false;
    if (a.gradient || a.flags) {
        const int64_t i = x * sx + y * sy + z;
        double g0 = nan, g1 = nan, g2 = nan;
        const bool interior = x > 0 && y > 0 && z > 0 && x < a.nx - 1 && y < a.ny - 1 && z < a.nz - 1;
        if (interior) {
            const double inv2 = 1.0 / (2.0 * a.res);
            g0 = (double)(f[i + sx] - f[i - sx]) * inv2;
            g1 = (double)(f[i + sy] - f[i - sy]) * inv2;
            g2 = (double)(f[i + 1] - f[i - 1]) * inv2;
            have_grad = true;
        } else if (a.edge) {
            const int64_t lx = max((int64_t)0, x - 1), hx = min(a.nx - 1, x + 1);
            const int64_t ly = max((int64_t)0, y - 1), hy = min(a.ny - 1, y + 1);
            const int64_t lz = max((int64_t)0, z - 1), hz = min(a.nz - 1, z + 1);
            const double ix = (double)(hx - lx) * a.res, iy = (double)(hy - ly) * a.res, iz = (double)(hz - lz) * a.res;
            g0 = g1 = g2 = 0.0;
            if (ix > 0.0) g0 = ((double)f[i + (hx - x) * sx] - (double)f[i - (x - lx) * sx]) * (1.0 / ix);
            if (iy > 0.0) g1 = ((double)f[i + (hy - y) * sy] - (double)f[i - (y - ly) * sy]) * (1.0 / iy);
            if (iz > 0.0) g2 = ((double)f[i + (hz - z)] - (double)f[i - (z - lz)]) * (1.0 / iz);
            have_grad = true;
        }
        if (a.gradient) {
            if (have_grad) {
                a.gradient[3 * t + 0] = a.rot[0] * g0 + a.rot[1] * g1 + a.rot[2] * g2;
                a.gradient[3 * t + 1] = a.rot[3] * g0 + a.rot[4] * g1 + a.rot[5] * g2;
                a.gradient[3 * t + 2] = a.rot[6] * g0 + a.rot[7] * g1 + a.rot[8] * g2;
            } else {
                a.gradient[3 * t] = nan; a.gradient[3 * t + 1] = nan; a.gradient[3 * t + 2] = nan;
            }
        }
    }
    if (a.flags) a.flags[t] = (uint8_t)(1 | (have_grad ? 2 : 0));
}

}  // namespace sdfgpu
