// sdfgpu_sweep_x16.hpp -- K3 over the 16-bit plane field: x sweep + signed merge, 2 B/voxel in, 4 B out.
//
// Plane-field format ("p16 + side table"): the in-plane signed squared distance of every voxel is
// stored as int16, saturated at +-32767; wherever a group of 4 consecutive voxels holds a saturated
// value the exact int32 values of that group are also written to a full-size int32 side table.  Dense
// scenes never touch the side table, so the z/y -> x hop costs 2 + 2 B/voxel instead of 4 + 4.
//
// A lane owns 8 consecutive voxels of the (y,z) plane and marches along x with a register window of
// 2H+1 rows held as PACKED unsigned 16-bit pairs, split by class: P = distance-to-filled function
// (nonzero on free voxels), Q = distance-to-free function (nonzero on filled voxels), both clamped to
// 16383 so that adding d^2 <= 9 cannot overflow.  One candidate row costs a bit-select, a packed add
// and a packed min per PAIR of voxels (v_bfi_b32, v_pk_add_u16, v_pk_min_u16).  The packed result is
// exact whenever it is < (H+1)^2 (clamping only touches values >= 16383, and rows outside the window
// are at least (H+1)^2 away).  Voxels that are not decided by the window take the exact 32-bit
// outward scan (reads p16, and the side table where p16 is saturated).
// The finish uses an LDS table of float(sqrt(double(D)) * resolution) for D < 1024, built per
// workgroup with the same fp64 arithmetic as sdf_generation.hpp:254-265, so results stay bit-identical
// while the per-voxel fp64 sqrt disappears from the hot path.
#pragma once
#include "sdfgpu_kernels.hpp"

namespace sdfgpu {

constexpr int kSat16 = 32767;        // saturated plane-field value: look in the side table
constexpr unsigned kCap16 = 16383;   // in-window clamp (leaves head-room for + d^2)
constexpr int kLutN = 1024;

// two signed int16 plane values -> packed (P, Q) clamped to kCap16
__device__ __forceinline__ void split_pq(uint32_t w, uint32_t& P, uint32_t& Q) {
    const uint32_t cap = kCap16 | (kCap16 << 16);
    P = pk_min_u16(pk_max_i16(w, 0u), cap);
    Q = pk_min_u16(pk_max_i16(pk_neg_i16(w), 0u), cap);
}

struct SweepX16Args {
    const int16_t* in16;     // plane field, [L rows][plane]
    const int32_t* side32;   // exact values for saturated groups, same indexing
    float* out;              // [out_hi - out_lo rows][plane]
    int64_t ncols;           // V-wide columns in a plane
    int64_t plane;           // elements per row (ny*nz)
    int L, out_lo, out_hi, T;
    int side_lo, side_hi;    // rows of the buffer whose side-table entries are valid
    double resolution;
    int lo_truncated, hi_truncated;
    int64_t x_global, nx_global, ny, nz;
    uint32_t* maxdsq;        // [0] free, [1] filled
    uint32_t* status;        // bit 0: a voxel needed data beyond the buffer (slab mode)
    const uint32_t* guard;   // non-null: run only if *guard != 0
    int max_scan;            // 0 = unbounded outward scan
    uint32_t* far_flag;      // raised when a voxel was still undecided after max_scan rows
};

template <int V>
struct ExactScan {
    int D[V];
    int unresolved;
    int inexact;             // bit k: voxel k was still undecided when the bounded scan stopped
};

template <int V> struct Raw16T;
template <> struct Raw16T<4> { using type = uint2; };
template <> struct Raw16T<8> { using type = uint4; };

template <int V>
__device__ __forceinline__ void raw16_words(const typename Raw16T<V>::type& raw, uint32_t (&w)[V / 2]) {
    w[0] = raw.x; w[1] = raw.y;
    if constexpr (V == 8) { w[2] = raw.z; w[3] = raw.w; }
}

// fp64 finish for squared distances beyond the LDS table (rare on dense scenes): out of line so the
// unrolled per-voxel finish stays small.
__device__ __noinline__ float finish_large(int D, double resolution) {
    return (D >= kInf32) ? __builtin_inff() : (float)(sqrt((double)D) * resolution);
}

// Exact 32-bit outward scan along x for the 8 voxels of one lane, starting from the centre row.
// Kept out of line: it is the rare path (sparse scenes) and inlining it into every unrolled step
// of the marching loop costs ~80 VGPRs of occupancy on the dense path.
// On entry D[k] holds the window result for decided voxels and any value >= lim for the others.
template <int V>
__device__ __noinline__ ExactScan<V> x16_exact_scan(const int16_t* __restrict__ in16, const int32_t* __restrict__ side32,
                                                    int64_t base, int64_t ls, int L, int side_lo, int side_hi,
                                                    int p, int lim, int max_scan, ExactScan<V> st) {
    using RawT = typename Raw16T<V>::type;
    auto fetch_exact = [&](int q, int (&s)[V]) {
        const RawT raw = *reinterpret_cast<const RawT*>(in16 + base + (int64_t)q * ls);
        uint32_t w[V / 2];
        raw16_words<V>(raw, w);
#pragma unroll
        for (int j = 0; j < V / 2; ++j) { s[2 * j] = (int)(short)(w[j] & 0xffffu); s[2 * j + 1] = (int)(short)(w[j] >> 16); }
#pragma unroll
        for (int g = 0; g < V / 4; ++g) {
            bool sat = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) sat |= abs(s[4 * g + k]) >= kSat16;
            if (sat) {
                if (q >= side_lo && q < side_hi) {
                    const int4 e = *reinterpret_cast<const int4*>(side32 + base + 4 * g + (int64_t)q * ls);
                    s[4 * g] = e.x; s[4 * g + 1] = e.y; s[4 * g + 2] = e.z; s[4 * g + 3] = e.w;
                } else {
                    st.unresolved = 1;      // halo row without its side table: 32767 is only a lower bound
                }
            }
        }
    };
    int cen[V], m[V];
    fetch_exact(p, cen);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        m[k] = cen[k] >> 31;
        if (st.D[k] >= lim) st.D[k] = (cen[k] ^ m[k]) - m[k];
    }
    for (int d = 1;; ++d) {
        const int lo = p - d, hi = p + d;
        if (lo < 0 && hi >= L) break;
        const int dd = d * d;
        bool act = false;
#pragma unroll
        for (int k = 0; k < V; ++k) act |= dd < st.D[k];
        if (!__any(act)) break;
        if (max_scan && d > max_scan) {           // far field: the envelope kernel redoes this sweep
#pragma unroll
            for (int k = 0; k < V; ++k) st.inexact |= (dd < st.D[k]) ? (1 << k) : 0;
            break;
        }
        if (act) {
            int s[V];
            if (lo >= 0) {
                fetch_exact(lo, s);
#pragma unroll
                for (int k = 0; k < V; ++k) st.D[k] = min(st.D[k], candidate(s[k], m[k], -m[k], dd));
            }
            if (hi < L) {
                fetch_exact(hi, s);
#pragma unroll
                for (int k = 0; k < V; ++k) st.D[k] = min(st.D[k], candidate(s[k], m[k], -m[k], dd));
            }
        }
    }
    return st;
}

template <int V, int H, bool VB, bool SLAB>
__global__ __launch_bounds__(kBlock) void k_sweep_x16(const SweepX16Args a) {
    constexpr int NP = V / 2, R = 2 * H + 1;
    using RawT = typename Raw16T<V>::type;
    constexpr uint32_t kLim = (H + 1) * (H + 1);
    if (a.guard && *a.guard == 0u) return;
    __shared__ float lut[kLutN];
    for (int i = threadIdx.x; i < kLutN; i += kBlock) lut[i] = (float)(sqrt((double)i) * a.resolution);
    __syncthreads();

    int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = c < a.ncols;
    if (!valid) c = a.ncols - 1;
    const int p0 = a.out_lo + (int)blockIdx.y * a.T;
    const int p1 = min(a.out_hi, p0 + a.T);
    if (p0 >= p1) return;
    const int64_t base = c * V;
    const int64_t ls = a.plane;
    const int L = a.L;

    uint32_t WP[R][NP], WQ[R][NP];       // window, packed u16 pairs
    int mxF = 0, mxQ = 0;
    uint32_t mxF16 = 0u, mxQ16 = 0u;     // ... of the rows the window decided alone: packed halves, folded into mxF / mxQ at the end
    bool unresolved = false;
    bool far = false;

    int vy[V], vz[V];
    if constexpr (VB) {
        int y0 = (int)(base / a.nz), z0 = (int)(base - (int64_t)y0 * a.nz);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            int zk = z0 + k, yk = y0;
            while (zk >= a.nz) { zk -= (int)a.nz; ++yk; }
            vy[k] = yk; vz[k] = zk;
        }
    }

    auto load_raw = [&](int p) { return *reinterpret_cast<const RawT*>(a.in16 + base + (int64_t)p * ls); };
    auto unpack = [&](const RawT& raw, uint32_t (&P)[NP], uint32_t (&Q)[NP]) {
        uint32_t w[NP];
        raw16_words<V>(raw, w);
#pragma unroll
        for (int j = 0; j < NP; ++j) split_pq(w[j], P[j], Q[j]);
    };
    auto step = [&](int p, auto r_tag, auto check_tag) {
        constexpr int r = decltype(r_tag)::value;
        constexpr bool CHECK = decltype(check_tag)::value;
        const uint32_t (&cP)[NP] = WP[(r + H) % R];
        const uint32_t (&cQ)[NP] = WQ[(r + H) % R];
        uint32_t best[NP], mfree[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            mfree[j] = pk_sub_u16(pk_min_u16(cQ[j], 0x00010001u), 0x00010001u);   // 0xFFFF where the voxel is free
            best[j] = cP[j] | cQ[j];
        }
#pragma unroll
        for (int d = 1; d <= H; ++d) {
            const uint32_t dd2 = (uint32_t)(d * d) * 0x00010001u;
            const bool lo_ok = !CHECK || (p - d >= 0);
            const bool hi_ok = !CHECK || (p + d < L);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (lo_ok) {
                    const uint32_t sel = bit_select(mfree[j], WP[(r + H - d) % R][j], WQ[(r + H - d) % R][j]);
                    best[j] = pk_min_u16(best[j], pk_add_u16(sel, dd2));
                }
                if (hi_ok) {
                    const uint32_t sel = bit_select(mfree[j], WP[(r + H + d) % R][j], WQ[(r + H + d) % R][j]);
                    best[j] = pk_min_u16(best[j], pk_add_u16(sel, dd2));
                }
            }
        }
        uint32_t worst = pk_max_u16(best[0], best[1]);
        if constexpr (V == 8) worst = pk_max_u16(worst, pk_max_u16(best[2], best[3]));
        const bool need = ((worst & 0xffffu) >= kLim) || ((worst >> 16) >= kLim);
        const bool any_need = __any(need);
        if constexpr (!VB && !SLAB) {
            if (!any_need) {
                // Every voxel of the wave decided by the window (all values < (H+1)^2): finish from the packed values -- maxima
                // tracked as packed halves per class, distance from the table, sign bit from the class mask.  4 VALU
                // instructions and one LDS read per voxel instead of ~15.
                float o[V];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const uint32_t nfree = ~mfree[j];
                    mxF16 = pk_max_u16(mxF16, best[j] & mfree[j]);
                    mxQ16 = pk_max_u16(mxQ16, best[j] & nfree);
                    const float f0 = lut[best[j] & 0xffffu], f1 = lut[best[j] >> 16];
                    o[2 * j] = __uint_as_float(__float_as_uint(f0) ^ (nfree << 31));
                    o[2 * j + 1] = __uint_as_float(__float_as_uint(f1) ^ (nfree & 0x80000000u));
                }
                if (valid) {
                    float4* dst = reinterpret_cast<float4*>(a.out + base + (int64_t)(p - a.out_lo) * ls);
#pragma unroll
                    for (int q = 0; q < V / 4; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                }
                return;
            }
        }
        int D[V];
        bool filled[V];
        int inexact = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            D[k] = (int)((best[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            filled[k] = ((mfree[k >> 1] >> (16 * (k & 1))) & 1u) == 0u;
        }
        if (any_need) {
            // exact 32-bit outward scan for the voxels the window did not decide (out of line)
            ExactScan<V> st;
#pragma unroll
            for (int k = 0; k < V; ++k) st.D[k] = D[k];
            st.unresolved = 0;
            st.inexact = 0;
            st = x16_exact_scan<V>(a.in16, a.side32, base, ls, L, a.side_lo, a.side_hi, p, (int)kLim, a.max_scan, st);
#pragma unroll
            for (int k = 0; k < V; ++k) D[k] = st.D[k];
            unresolved |= st.unresolved != 0;
            inexact = st.inexact;
            far |= inexact != 0;
        }
        if constexpr (SLAB) {
            if (a.lo_truncated) {
                const int dd = p + 1;
#pragma unroll
                for (int k = 0; k < V; ++k) unresolved |= D[k] > dd * dd;
            }
            if (a.hi_truncated) {
                const int dd = L - p;
#pragma unroll
                for (int k = 0; k < V; ++k) unresolved |= D[k] > dd * dd;
            }
        }
        float o[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            int Dk = min(D[k], kInf32);
            if constexpr (VB) {
                int64_t b = kInf32;
                const int64_t gx = a.x_global + (p - a.out_lo);
                if (a.nx_global > 1) b = min(b, min(gx + 1, a.nx_global - gx));
                if (a.ny > 1) b = min(b, min((int64_t)vy[k] + 1, a.ny - vy[k]));
                if (a.nz > 1) b = min(b, min((int64_t)vz[k] + 1, a.nz - vz[k]));
                if (b < 32768) Dk = min(Dk, (int)(b * b));
            }
            if (!((inexact >> k) & 1)) { if (filled[k]) mxQ = max(mxQ, Dk); else mxF = max(mxF, Dk); }
            const float f = (Dk < kLutN) ? lut[Dk] : finish_large(Dk, a.resolution);
            o[k] = filled[k] ? -f : f;
        }
        if (valid) {
            float4* dst = reinterpret_cast<float4*>(a.out + base + (int64_t)(p - a.out_lo) * ls);
#pragma unroll
            for (int q = 0; q < V / 4; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
    };

    // prologue
    static_for<2 * H>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const int p = p0 - H + k;
        if (p >= 0 && p < L) unpack(load_raw(p), WP[k], WQ[k]);
    });

    // Batches of R rows.  Interior batches are double-buffered: the loads of batch b+1 are issued
    // before batch b is processed, so every wave always has R row loads in flight while it computes.
    auto is_fast = [&](int pb) { return (pb - H >= 0) && (pb + R - 1 + H < L) && (pb + R <= p1); };
    auto slow_batch = [&](int pb) {
        static_for<R>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int p = pb + r;
            if (p < p1) {
                if (p + H < L) unpack(load_raw(p + H), WP[(r + 2 * H) % R], WQ[(r + 2 * H) % R]);
                step(p, rc, std::true_type{});
            }
        });
    };
    int pb = p0;
    while (pb < p1 && !is_fast(pb)) { slow_batch(pb); pb += R; }
    if (pb < p1) {
        RawT cur[R];
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = load_raw(pb + r + H);
        for (;;) {
            const int nb = pb + R;
            const bool nf = nb < p1 && is_fast(nb);
            RawT nxt[R];
            if (nf) {
#pragma unroll
                for (int r = 0; r < R; ++r) nxt[r] = load_raw(nb + r + H);
            }
            static_for<R>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                unpack(cur[r], WP[(r + 2 * H) % R], WQ[(r + 2 * H) % R]);
                step(pb + r, rc, std::false_type{});
            });
            pb = nb;
            if (!nf) break;
#pragma unroll
            for (int r = 0; r < R; ++r) cur[r] = nxt[r];
        }
        while (pb < p1) { slow_batch(pb); pb += R; }
    }

    mxF = max(mxF, (int)max(mxF16 & 0xffffu, mxF16 >> 16));
    mxQ = max(mxQ, (int)max(mxQ16 & 0xffffu, mxQ16 >> 16));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mxF = max(mxF, __shfl_xor(mxF, off));
        mxQ = max(mxQ, __shfl_xor(mxQ, off));
    }
    const bool any_unres = __any(unresolved);
    const bool any_far = __any(far);
    if ((threadIdx.x & 63) == 0) {
        if (any_far && a.far_flag) raise_flag(a.far_flag);
        slot_max2(a.maxdsq, blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), mxF, mxQ);      // a.maxdsq = slot array
        if (any_unres && a.status) raise_flag(a.status);
    }
}

// Shared by the producers (fused z+y kernel, K2 with 16-bit output): store one group of 4 voxels.
// sD = signed exact squared distance (+ free / - filled, magnitude <= kInf32).
__device__ __forceinline__ uint2 pack_plane16_group(const int (&sD)[4], int32_t* side_group) {
    int s16[4];
    bool sat = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int mag = abs(sD[k]);
        sat |= mag >= kSat16;
        s16[k] = sD[k] < 0 ? -min(mag, kSat16) : min(mag, kSat16);
    }
    if (sat) *reinterpret_cast<int4*>(side_group) = make_int4(sD[0], sD[1], sD[2], sD[3]);
    return make_uint2(((uint32_t)s16[0] & 0xffffu) | ((uint32_t)s16[1] << 16),
                      ((uint32_t)s16[2] & 0xffffu) | ((uint32_t)s16[3] << 16));
}

}  // namespace sdfgpu
