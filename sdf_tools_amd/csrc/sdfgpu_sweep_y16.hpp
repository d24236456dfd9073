// sdfgpu_sweep_y16.hpp -- K2 over packed 16-bit pairs: the y sweep of the marching tier, int16 z field in (2 B/voxel),
// int16 plane field + side table out (2 B/voxel), radius-3 register window.
//
// Same scheme as K3/16 (sdfgpu_sweep_x16.hpp): a lane owns 4 consecutive z of one (x, z) column bundle and marches
// along y with a window of 2H+1 rows held as PACKED unsigned 16-bit pairs split by class -- P = the "distance to
// filled" function (non-zero on free voxels), Q = the "distance to free" function (non-zero on filled voxels).  The row
// values are the squares of the z distances, squared in 16 bits after clamping |g| to 127 (127^2 = 16129: anything
// clamped is far above the (H+1)^2 = 16 below which the window decides a voxel, so a clamped value can only make a
// voxel undecided, never wrong).  One candidate row costs a bit-select, v_pk_add_u16 and v_pk_min_u16 per PAIR of
// voxels; the 32-bit form k_sweep_march<2, 4, 3> spent 37 VALU instructions per voxel on the same work and ran at 0.37
// of the HBM roofline.  A lane whose 4 voxels are all decided by the window packs its signed
// result without unpacking (2 instructions per pair); undecided voxels take the exact 32-bit outward scan out of line
// (bounded by max_scan: the far-field kernel then redoes the sweep).  Exactness: a window value < (H+1)^2 is exact
// (rows outside the window are at least (H+1)^2 away, clamping only touches values >= 16129).
#pragma once
#include "sdfgpu_kernels.hpp"
#include "sdfgpu_sweep_x16.hpp"

namespace sdfgpu {

__device__ __forceinline__ uint32_t pk_mul_lo_u16(uint32_t a, uint32_t b) { return as_u32(as_us2(a) * as_us2(b)); }

// two signed int16 z distances -> packed squares (P: free voxels, Q: filled voxels), |g| clamped to 127
__device__ __forceinline__ void split_pq_z(uint32_t w, uint32_t& P, uint32_t& Q) {
    const uint32_t cap = 127u | (127u << 16);
    const uint32_t gp = pk_min_u16(pk_max_i16(w, 0u), cap);
    const uint32_t gq = pk_min_u16(pk_max_i16(pk_neg_i16(w), 0u), cap);
    P = pk_mul_lo_u16(gp, gp);
    Q = pk_mul_lo_u16(gq, gq);
}

// Exact 32-bit outward scan along y for the 4 voxels of one lane (rare path, out of line).  On entry D[k] holds the
// window result for decided voxels and any value >= lim for the others.
__device__ __noinline__ ExactScan<4> y16_exact_scan(const int16_t* __restrict__ in16, int64_t base, int64_t ls, int L,
                                                    int p, int lim, int max_scan, ExactScan<4> st) {
    auto fetch = [&](int q, int (&s)[4]) {
        const uint2 raw = *reinterpret_cast<const uint2*>(in16 + base + (int64_t)q * ls);
        const int g[4] = {(int)(short)(raw.x & 0xffffu), (int)raw.x >> 16, (int)(short)(raw.y & 0xffffu), (int)raw.y >> 16};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = abs(g[k]);
            const int sq = (a >= kInf16) ? kInf32 : a * a;
            s[k] = g[k] < 0 ? -sq : sq;
        }
    };
    int cen[4], m[4];
    fetch(p, cen);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] = cen[k] >> 31;
        if (st.D[k] >= lim) st.D[k] = (cen[k] ^ m[k]) - m[k];
    }
    for (int d = 1;; ++d) {
        const int lo = p - d, hi = p + d;
        if (lo < 0 && hi >= L) break;
        const int dd = d * d;
        bool act = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) act |= dd < st.D[k];
        if (!__any(act)) break;
        if (max_scan && d > max_scan) {           // far field: the far-field kernel redoes this sweep
#pragma unroll
            for (int k = 0; k < 4; ++k) st.inexact |= (dd < st.D[k]) ? (1 << k) : 0;
            break;
        }
        if (act) {
            int s[4];
            if (lo >= 0) {
                fetch(lo, s);
#pragma unroll
                for (int k = 0; k < 4; ++k) st.D[k] = min(st.D[k], candidate(s[k], m[k], -m[k], dd));
            }
            if (hi < L) {
                fetch(hi, s);
#pragma unroll
                for (int k = 0; k < 4; ++k) st.D[k] = min(st.D[k], candidate(s[k], m[k], -m[k], dd));
            }
        }
    }
    return st;
}

template <int H>
__global__ __launch_bounds__(kBlock, 4) void k_sweep_y16(const SweepArgs a) {
    constexpr int V = 4, NP = 2, R = 2 * H + 1;
    constexpr uint32_t kLim = (H + 1) * (H + 1);
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
    const int16_t* const in16 = reinterpret_cast<const int16_t*>(a.in);
    int16_t* const out16 = reinterpret_cast<int16_t*>(a.out);

    uint32_t WP[R][NP], WQ[R][NP];          // window, packed u16 pairs
    bool far = false;

    auto load_raw = [&](int p) { return *reinterpret_cast<const uint2*>(in16 + base + (int64_t)p * ls); };
    auto unpack = [&](const uint2& raw, uint32_t (&P)[NP], uint32_t (&Q)[NP]) {
        split_pq_z(raw.x, P[0], Q[0]);
        split_pq_z(raw.y, P[1], Q[1]);
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
        const uint32_t worst = pk_max_u16(best[0], best[1]);
        const bool need = ((worst & 0xffffu) >= kLim) || ((worst >> 16) >= kLim);
        const int64_t oelem = base + (int64_t)(p - a.out_lo) * ls;
        uint2 ow;
        if (!__any(need)) {
            // every voxel of the wave decided by the window: signed 16-bit results straight from the packed values
            // ((x ^ 0xFFFF) - 0xFFFF = -x in 16 bits for the filled voxels)
            const uint32_t n0 = ~mfree[0], n1 = ~mfree[1];
            ow.x = pk_sub_u16(best[0] ^ n0, n0);
            ow.y = pk_sub_u16(best[1] ^ n1, n1);
        } else {
            ExactScan<4> st;
            bool filled[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                st.D[k] = (int)((best[k >> 1] >> (16 * (k & 1))) & 0xffffu);
                filled[k] = ((mfree[k >> 1] >> (16 * (k & 1))) & 1u) == 0u;
            }
            st.unresolved = 0;
            st.inexact = 0;
            st = y16_exact_scan(in16, base, ls, L, p, (int)kLim, a.max_scan, st);
            far |= st.inexact != 0;
            int sD[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int D = min(st.D[k], kInf32); sD[k] = filled[k] ? -D : D; }
            ow = pack_plane16_group(sD, a.side + oelem);      // (a lane past the last column repeats that column: same values)
        }
        if (valid) *reinterpret_cast<uint2*>(out16 + oelem) = ow;
    };

    // prologue
    static_for<2 * H>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const int p = p0 - H + k;
        if (p >= 0 && p < L) unpack(load_raw(p), WP[k], WQ[k]);
    });

    // Batches of R rows.  Interior batches are double-buffered: the loads of batch b+1 are issued before batch b is
    // processed, so every wave always has R row loads in flight while it computes.
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
        uint2 cur[R];
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = load_raw(pb + r + H);
        for (;;) {
            const int nb = pb + R;
            const bool nf = nb < p1 && is_fast(nb);
            uint2 nxt[R];
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
    if (a.far_flag && __any(far) && (threadIdx.x & 63) == 0) raise_flag(a.far_flag);
}

}  // namespace sdfgpu
