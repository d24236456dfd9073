// sdfgpu_envelope_dc.hpp -- KE2 / KE3, second generation: the far-field y / x sweep as a PARALLEL exact
// min-plus search instead of a per-lane parabola stack.
//
// Along one line the sweep computes D(p) = min_q F(q) + (p - q)^2 (F = squared distance inside the rows already
// swept, "no site" = +inf).  The cost c(p, q) = F(q) + (p - q)^2 is a Monge array: c(p,q) + c(p',q') <=
// c(p,q') + c(p',q) for p < p', q < q' (the F terms cancel, the rest is 2 (p'-p)(q-q') <= 0).  Hence for any
// argmin a(p1) at p1 < p2 there is an argmin a(p2) >= a(p1) (and symmetrically from the right): once the argmin of
// a middle position is known, the positions left of it only need the candidates up to it and the positions right
// of it only the candidates from it on.  Recursing on halves gives ~ (log2 L + 1) x L candidate evaluations per
// line (9.8 per position at L = 512, fewer when the sites cover only part of the line) -- about the instruction
// count of the stack algorithm -- but every evaluation is independent of every other one in its level: no
// push / pop chain, no per-line stack in memory (the first-generation kernel k_envelope spilled 8 B/voxel of stack
// to a global scratch array and ran one lane per line, 16 waves per CU, at ~10 % of the HBM roofline).
// tools/envelope_dc_model.py is an executable model of exactly this schedule, checked against brute force.
//
// Work decomposition.  A workgroup (256 lanes) owns a tile of 16 neighbouring lines (z-adjacent, so every global
// row access is a contiguous 32 / 64-byte segment) and stages the whole tile in LDS as 32-bit KEYS
//     key[q] = ((F(q) + q^2) << B) | q          B = bits of a position, "no site" uses F = finf
// so that for a position p
//     key[q] + (p^2 << B) - ((2p << B) * q)  =  ((F(q) + (p - q)^2) << B) | q       (mod 2^32, exact: < 2^32)
// One unsigned min over such values yields the distance AND an argmin; walking q upward the linear term is
// updated by one add, so a candidate costs ~2 VALU instructions + half an LDS read (keys are read in pairs).
//   upper levels  positions p = 0, 8, 16, ... (coarse grid) by binary subdivision; 16 lanes per line: the first
//                 levels split one position's candidate range over 16 / 8 / 4 / 2 lanes and min-reduce, later levels
//                 give every lane its own positions.  Argmins go to a small LDS array.
//   chunk phase   lane = (line, chunk of 8 positions): all 8 positions against every candidate between the argmins of the
//                 chunk's two coarse neighbours, results converted and stored straight from the lane (16 lines x 2 / 4 B
//                 contiguous per position).
// Both classes of the signed field take one pass each (sites of "distance to filled" evaluated on free voxels,
// then the reverse); a tile without filled voxels skips the second pass, and in it free voxels are sites only
// next to a filled voxel of their line (only the ends of a run can be nearest to anything outside it).
// Exactness: integer arithmetic throughout; the finish is the reference's (sqrt and multiply in fp64, one cast,
// sdf_generation.hpp:254-265).  Shapes whose keys do not fit 32 bits (finf + L^2 >= 2^(32-B)), L > 1024 or
// line counts that are not multiples of 16 fall back to k_envelope (sdfgpu_envelope.hpp).
#pragma once
#include "sdfgpu_kernels.hpp"
#include "sdfgpu_sweep_x16.hpp"

namespace sdfgpu {

constexpr int kDcLines = 16;          // lines per tile
constexpr int kDcChunk = 8;           // positions per lane in the chunk phase
constexpr int kDcBatch = 2;           // staging: row loads in flight per lane (register budget: 128 VGPRs at 4 waves per SIMD)
constexpr int kDcLocalMax = 64;       // pass 0 finishes filled voxels whose in-row squared distance is at most this ...
constexpr int kDcLocalFilled = 448;   // ... when the tile holds at most this many filled voxels (of 16 x L; a full second
                                      // pass costs as much as the first: at 224 a 3 %-occupied scene -- 245 per tile --
                                      // ran it everywhere, 1.9 instead of 0.57 ms; one packed word per entry keeps 448
                                      // of them inside the LDS budget of 4 workgroups per CU)

struct EnvDcArgs {
    const int16_t* in16;      // STAGE 2: z field (+-g, 32767 = none); STAGE 3: plane field p16
    const int32_t* side_in;   // STAGE 3: exact values where p16 is saturated (valid for the whole group of 4)
    void* out;                // STAGE 2: int16 plane field; STAGE 3: float sdf
    int32_t* side_out;        // STAGE 2: exact plane values for groups of 4 that hold a saturated value
    const int32_t* in_i32;    // STAGE 3, slab pipelines: the input is an int32 plane field (in16 / side_in unused)
    int32_t* out_i32;         // STAGE 2, slab pipelines: write an int32 plane field (+-2^30 = none) instead of p16 + side
    int64_t y_off, ny_glob;   // STAGE 3 on a y slab: grid y of local row 0 and the full y extent (virtual border)
    int64_t tiles_per_outer;  // STAGE 2: nz / 16 tiles per x-plane; STAGE 3: all tiles
    int64_t outer_stride;     // elements between outer units (STAGE 2: ny*nz; STAGE 3: 0)
    int64_t line_stride;      // elements between successive positions of a line
    int L;                    // positions per line
    int B;                    // bits of the position field of a key
    uint32_t finf;            // "no site": larger than every real squared distance of this grid
    int pitch;                // LDS words per line of keys (>= L + 2, == 1 mod 32)
    int M, Kp;                // coarse positions ceil(L / 8) and levels (bit length of M)
    double resolution;        // STAGE 3
    int vb;
    int64_t nx, ny, nz;       // full extents (virtual border)
    uint32_t* maxdsq;         // slot array (STAGE 3)
    const uint32_t* guard;    // nullptr: always run; else run iff (*guard != 0) != guard_invert
    int guard_invert;
    // probe mode (device-side tier selection): only every probe_stride-th tile is processed, nothing is stored; the first
    // pass counts the voxels it produced ([1]) and those whose squared distance is >= probe_thr ([0]) into probe_out
    int probe_stride;
    int probe_thr;
    int probe_thr2;           // second, lower threshold (y probe: radius-8 vs radius-3 marching window), counted into probe_out[-1]
    uint32_t* probe_out;
    // both axes far-field: when *i32_flag != 0 the y sweep hands its result to the x sweep as an exact int32 plane field
    // (out_i32 / in_i32, the side-table buffer used whole) instead of p16 + side table; nullptr = the pointers alone decide
    const uint32_t* i32_flag;
    int h;                    // third generation: centre of the key coordinates, ceil(L / 2)
    int64_t group_lines;      // third generation: lines per outer unit (tiles that stick out replicate the last line)
    unsigned long long* clocks;   // SDFGPU_PHASE_CLOCKS builds only: per-phase shader-clock sums over waves ([stage - 2][8])
    int dbg;                  // profiling aid (wrong results!): bit0 skip upper levels, bit1 skip chunk search, bit2 skip stores,
                              // bit3 skip the second class, bit4 skip key conversion
};

// Correctly rounded fp64 square root of a positive normal number: exactly the Goldschmidt / Newton sequence the compiler
// emits for sqrt(double) (rsq seed, one coupled iteration, two residual corrections), without its input scaling for
// arguments below 2^-767 and its zero / infinity selects -- the argument here is an integer in [1, 2^30).  Same
// operations in the same order, so the same bits; 10 instructions instead of ~20 per voxel.
__device__ __forceinline__ double sqrt_exact_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y;
    const double h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, s0, 0.5);
    const double s1 = __builtin_fma(s0, r0, s0);
    const double h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-s1, s1, x);
    const double s2 = __builtin_fma(d0, h1, s1);
    const double d1 = __builtin_fma(-s2, s2, x);
    return __builtin_fma(d1, h1, s2);
}

inline size_t envelope_dc_lds_bytes(int L, int pitch, int lines = kDcLines) {
    const int SW = (L + 31) / 32, M = (L + kDcChunk - 1) / kDcChunk;
    return ((size_t)lines * pitch + (size_t)lines * SW + 2 * (size_t)lines + 32 + kDcLocalFilled) * 4 + (size_t)lines * (M + 2) * 2;
}

template <int STAGE, int NL>          // NL lines per tile (16 or 8), 16 lanes per line
__global__ __launch_bounds__(16 * NL, 4) void k_envelope_dc(const EnvDcArgs a) {
    constexpr int NT = 16 * NL;
    constexpr int LSUB = NL / 4;        // staging: lanes per position (4 lines each)
    constexpr int PS = NT / LSUB;       // positions staged per pass of the workgroup (= 64)
    constexpr int NS = 16;              // (line, slot) mapping: slots per line
    extern __shared__ __attribute__((aligned(16))) uint32_t dc_smem[];
    if (a.guard) {
        const uint32_t gv = *a.guard;
        if ((gv != 0u) == (a.guard_invert != 0)) return;
    }
    const bool i32 = !a.i32_flag || *a.i32_flag != 0u;        // (block-uniform)
    if (a.probe_stride > 0 && a.i32_flag && i32) return;       // the x tier is already decided: no probe
    const int L = a.L, B = a.B, pitch = a.pitch, M = a.M, Kp = a.Kp;
    const int SW = (L + 31) >> 5, AP = M + 2;
    const uint32_t mask = (1u << B) - 1u, finf = a.finf;
    uint32_t* keys = dc_smem;                                   // [16][pitch]
    uint32_t* sgn = keys + NL * pitch;                    // [16][SW]   bit p: voxel p of the line is filled
    uint32_t* span = sgn + NL * SW;                       // [16][2]    first / last site of the line
    uint32_t* flg = span + 2 * NL;                                  // [16] line holds a filled voxel, [16] = tile does
    uint32_t* flist = flg + 32;                                 // [kDcLocalFilled] filled voxels of pass 0: line << 24 | p << 8 | min(S, 255)
    uint16_t* args = reinterpret_cast<uint16_t*>(flist + kDcLocalFilled);   // [16][AP]   argmin of coarse position i' (1-based)
    const int t = threadIdx.x;

    // tile of this workgroup.  Four consecutive tiles share cache lines (16 lines x 2 B = 32 B of a 128-B line), so an XCD
    // (workgroups are dealt to the 8 XCDs round-robin, each with its own L2) gets runs of 4 consecutive tiles; the runs
    // themselves are dealt round-robin, so that every XCD sees every part of the grid (work is not uniform in space:
    // a tile far from every object is trivial)
    int64_t tile = blockIdx.x;
    const bool probe = a.probe_stride > 0;
    if (probe) {                                                // a sample spread over the whole grid
        tile = (int64_t)blockIdx.x * a.probe_stride + (blockIdx.x * 7u) % (uint32_t)a.probe_stride;
    } else if ((gridDim.x & 31u) == 0u) {
        const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
        tile = ((int64_t)(seq >> 2) * 8 + xcd) * 4 + (seq & 3u);
    }
    const int64_t o = tile / a.tiles_per_outer;
    const int64_t c0 = (tile - o * a.tiles_per_outer) * NL;
    const int64_t base = o * a.outer_stride + c0;               // element index of (line 0, position 0)
    const uint32_t ls = (uint32_t)a.line_stride;                // (the launcher guarantees nx*ny*nz < 2^31: 32-bit element offsets)
    const int16_t* const in16 = a.in16 + base;
    const int32_t* const side_in = STAGE == 3 ? a.side_in + base : nullptr;
    const int32_t* const in32 = (STAGE == 3 && a.in_i32 && i32) ? a.in_i32 + base : nullptr;

    // one candidate range for one position: lanes u = 0 .. G-1 of a group take the pairs lo + 2u, lo + 2u + 2G, ...
    // (the second key of the last pair may be candidate hi + 1: it can tie but never beat the range's minimum, and on a
    // tie the smaller position wins the unsigned min, so the argmin stays inside [lo, hi])
    auto scan = [&](const uint32_t* kl, uint32_t p, int lo, int hi, int u, int G) -> uint32_t {
        const uint32_t c = (2u * p) << B;                       // < 2^22: 24-bit multiplies are exact mod 2^32
        int q = lo + 2 * u;
        uint32_t R = (__umul24(p, p) << B) - __umul24(c, (uint32_t)q);
        const uint32_t dR = __umul24(c, (uint32_t)(2 * G));
        uint32_t best = 0xFFFFFFFFu;
        const int step = 2 * G;
        for (; q + step <= hi; q += 2 * step) {                 // two pairs per trip: half the loop overhead
            const uint32_t k0 = kl[q], k1 = kl[q + 1], k2 = kl[q + step], k3 = kl[q + step + 1];
            const uint32_t R2 = R - dR;
            best = min(best, min(k0 + R, k1 + R - c));
            best = min(best, min(k2 + R2, k3 + R2 - c));
            R = R2 - dR;
        }
        if (q <= hi) {
            const uint32_t k0 = kl[q], k1 = kl[q + 1];
            best = min(best, min(k0 + R, k1 + R - c));
        }
        return best;
    };

    int mxF = 0, mxQ = 0;
    // virtual-border distance of a line to the padded layer over y and z (STAGE 3: line = (y, z))
    auto byz_of = [&](int line) -> int {
        int64_t b = kInf32;
        if constexpr (STAGE == 3) {
            if (a.vb) {
                const int64_t c = c0 + line;
                const int64_t vyl = c / a.nz, vz = c - vyl * a.nz;
                const int64_t vy = vyl + a.y_off;
                if (a.ny_glob > 1) b = min(b, min(vy + 1, a.ny_glob - vy));
                if (a.nz > 1) b = min(b, min(vz + 1, a.nz - vz));
            }
        }
        return (int)b;
    };
    const int byz = byz_of(t & (NL - 1));             // the lane's own line in the chunk phase

    // exact signed value of voxel q of `line`, re-read from global memory (rare paths only)
    auto raw_signed = [&](int line, int q) -> int {
        const uint32_t idx = (uint32_t)line + (uint32_t)q * ls;
        if (STAGE == 3 && in32) return in32[idx];
        int v = in16[idx];
        if constexpr (STAGE == 2) {
            const int g = abs(v);
            const int sq = g >= kInf16 ? kInf32 : g * g;
            return v < 0 ? -sq : sq;
        } else {
            if (abs(v) >= kSat16) v = side_in[idx];
            return v;
        }
    };
    // finish one voxel: STAGE 2 plane field (+ side table on request), STAGE 3 the reference's merge arithmetic
    int probe_far = 0, probe_tot = 0, probe_mid = 0;
    auto emit = [&](int line, int p, int D, bool filled, bool side, int b_yz) {
        if (probe) { probe_tot += 1; probe_far += D >= a.probe_thr ? 1 : 0; probe_mid += D >= a.probe_thr2 ? 1 : 0; return; }
        const uint32_t oi = (uint32_t)line + (uint32_t)p * ls;
        if constexpr (STAGE == 2) {
            if (a.out_i32 && i32) { (a.out_i32 + base)[oi] = filled ? -D : D; return; }
            (reinterpret_cast<int16_t*>(a.out) + base)[oi] = (int16_t)(filled ? -min(D, kSat16) : min(D, kSat16));
            if (side) (a.side_out + base)[oi] = filled ? -D : D;
        } else {
            if (a.vb) {
                int b = b_yz;
                if (a.nx > 1) b = min(b, (int)min((int64_t)p + 1, a.nx - p));
                if (b < 32768) D = min(D, b * b);
            }
            if (filled) mxQ = max(mxQ, D); else mxF = max(mxF, D);
            const float f = (D >= kInf32) ? __builtin_inff() : (float)(sqrt_exact_pos((double)D) * a.resolution);
            (reinterpret_cast<float*>(a.out) + base)[oi] = filled ? -f : f;
        }
    };

#pragma unroll 1
    for (int cls = 0; cls < 2; ++cls) {         // 0: sites of "distance to filled" (for free voxels); 1: the reverse
        if (cls == 0 && t == 0) { flg[16] = 0u; flg[17] = 0u; flg[18] = 0u; flg[19] = 0u; flg[20] = 0u; }   // (ordered before their users by the barriers of pass 0)
        // ---- stage the tile: rows -> keys, one pass ----------------------------------------------------------------------
        // A lane reads 4 lines x 1 position (8 B; STAGE 3 adds the 16-B side-table group where the 16-bit value is
        // saturated, or reads 16 B of an int32 plane field), kDcBatch independent row loads in flight, and writes the four
        // keys straight to LDS.  Filled voxels are rare in the scenes this kernel serves: their sign bits go to the bit
        // array with an LDS atomic each.  First / last site of a line: a bit per (line, iteration) in registers, two LDS
        // atomics per lane at the end.
        if (cls == 1 && ((a.dbg & 8) || probe)) break;
        // The second pass (distance to free, for filled voxels) runs only when pass 0 asked for it: a filled voxel whose
        // in-row squared distance S is small is finished by pass 0 itself with a local search (candidates at offset d can
        // only matter while d^2 < S), which covers the thin surfaces of sensed scenes; pass 0 raises flg[17] for the rest.
        if (cls == 1 && flg[17] == 0u) break;   // (block-uniform)
        if (cls == 0) for (int i = t; i < NL * SW; i += NT) sgn[i] = 0u;
        if (t < NL) { span[2 * t] = 0xFFFFFFFFu; span[2 * t + 1] = 0u; }
        __syncthreads();
        {
            const int sub = t & (LSUB - 1), r = t / LSUB;
            uint32_t seen[4] = {0u, 0u, 0u, 0u};            // bit it: iteration `it` of this lane found a site on line 4 sub + k
            uint32_t* const kbase = keys + (4 * sub) * pitch + r;
            for (int pb = 0, itb = 0; pb < L; pb += PS * kDcBatch, itb += kDcBatch) {
                int sv[kDcBatch][4];
                if (STAGE == 3 && in32) {
#pragma unroll
                    for (int it = 0; it < kDcBatch; ++it) {
                        const int p = min(pb + PS * it + r, L - 1);
                        const int4 e = *reinterpret_cast<const int4*>(in32 + ((uint32_t)p * ls + 4u * sub));
                        sv[it][0] = e.x; sv[it][1] = e.y; sv[it][2] = e.z; sv[it][3] = e.w;
                    }
                } else {
                    uint2 raw[kDcBatch];
#pragma unroll
                    for (int it = 0; it < kDcBatch; ++it) {
                        const int p = min(pb + PS * it + r, L - 1);
                        raw[it] = *reinterpret_cast<const uint2*>(in16 + ((uint32_t)p * ls + 4u * sub));
                    }
#pragma unroll
                    for (int it = 0; it < kDcBatch; ++it) {
                        sv[it][0] = (int)(short)(raw[it].x & 0xffffu); sv[it][1] = (int)raw[it].x >> 16;
                        sv[it][2] = (int)(short)(raw[it].y & 0xffffu); sv[it][3] = (int)raw[it].y >> 16;
                    }
                    if constexpr (STAGE == 3) {
#pragma unroll
                        for (int it = 0; it < kDcBatch; ++it) {
                            // a 16-bit lane is saturated iff it holds +-32767 (-32768 is never stored): |v| + 1 has bit 15 set
                            // per half: v >= 0 -> v, v < 0 -> ~v = |v| - 1; adding 1 (+ 1 more for the negative ones) gives |v| + 1
                            const uint32_t n0 = (raw[it].x >> 15) & 0x00010001u, n1 = (raw[it].y >> 15) & 0x00010001u;
                            const uint32_t m0 = (raw[it].x ^ (n0 * 0xFFFFu)) + 0x00010001u + n0;
                            const uint32_t m1 = (raw[it].y ^ (n1 * 0xFFFFu)) + 0x00010001u + n1;
                            const bool sat = ((m0 | m1) & 0x80008000u) != 0u;
                            if (sat) {
                                const int p = min(pb + PS * it + r, L - 1);
                                const int4 e = *reinterpret_cast<const int4*>(side_in + ((uint32_t)p * ls + 4u * sub));
                                sv[it][0] = e.x; sv[it][1] = e.y; sv[it][2] = e.z; sv[it][3] = e.w;
                            }
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < kDcBatch; ++it) {
                    const int p = pb + PS * it + r;
                    if (p < L) {
                        const uint32_t pp = __umul24((uint32_t)p, (uint32_t)p);
                        const uint32_t seen_bit = 1u << (itb + it);
                        if (cls == 0) {
                            // Pass 0, branch-free: a free voxel is a site with its own value, a filled one a site with 0, "no
                            // filled voxel in the rows swept so far" becomes finf by the clamp (finf > every real distance, and
                            // the "none" codes 32767^2 / kInf32 are above it).  The bookkeeping of the (rare) filled voxels is
                            // kept out of the way behind one test per group of 4.
                            int neg = 0;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int s1 = sv[it][k];
                                uint32_t F;
                                if constexpr (STAGE == 2) {
                                    const uint32_t m = (uint32_t)max(s1, 0);
                                    F = min(__umul24(m, m), finf);
                                } else {
                                    F = (uint32_t)min(max(s1, 0), (int)finf);
                                }
                                kbase[k * pitch + (pb + PS * it)] = ((F + pp) << B) | (uint32_t)p;
                                seen[k] |= F != finf ? seen_bit : 0u;
                                neg |= s1;
                            }
                            if (neg < 0) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const int s1 = sv[it][k];
                                    if (s1 < 0) {
                                        uint32_t S = (uint32_t)(-s1);        // squared distance to the nearest free voxel so far
                                        if constexpr (STAGE == 2) S = S >= (uint32_t)kInf16 ? (uint32_t)kInf32 : __umul24(S, S);
                                        atomicOr(&sgn[(4 * sub + k) * SW + (p >> 5)], 1u << (p & 31));
                                        const uint32_t e = atomicAdd(&flg[16], 1u);          // few per tile: list them for the local search
                                        if (e < (uint32_t)kDcLocalFilled) flist[e] = ((uint32_t)(4 * sub + k) << 24) | ((uint32_t)p << 8) | min(S, 255u);
                                    }
                                }
                            }
                            continue;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {           // pass 1: sites of "distance to free"
                            int s1 = sv[it][k];
                            if constexpr (STAGE == 2) {
                                const int gz = abs(s1);
                                const int sq = gz >= kInf16 ? kInf32 : (int)__umul24((uint32_t)gz, (uint32_t)gz);
                                s1 = s1 < 0 ? -sq : sq;
                            }
                            uint32_t F;
                            bool none;
                            if (s1 < 0) {
                                F = (uint32_t)(-s1);
                                none = -s1 >= kInf32;
                            } else {            // free voxel: a zero-valued site only next to a filled voxel of its line
                                F = 0u;
                                const uint32_t* sg = sgn + (4 * sub + k) * SW;
                                bool nb = false;
                                if (p > 0) nb |= (sg[(p - 1) >> 5] >> ((p - 1) & 31)) & 1u;
                                if (p + 1 < L) nb |= (sg[(p + 1) >> 5] >> ((p + 1) & 31)) & 1u;
                                none = !nb;
                            }
                            kbase[k * pitch + (pb + PS * it)] = (((none ? finf : F) + pp) << B) | (uint32_t)p;
                            if (!none) seen[k] |= seen_bit;
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (seen[k]) {                  // iteration it of this lane is position r + 64 it
                    atomicMin(&span[2 * (4 * sub + k)], (uint32_t)(r + PS * (__ffs((int)seen[k]) - 1)));
                    atomicMax(&span[2 * (4 * sub + k) + 1], (uint32_t)(r + PS * (31 - __clz((int)seen[k]))));
                }
            }
            if (t < NL) keys[t * pitch + L] = ((finf + (uint32_t)L * (uint32_t)L) << B) | ((uint32_t)L & mask);   // sentinel
        }
        __syncthreads();

        // ---- upper levels: coarse positions p = 8 (i' - 1), i' = 1 .. M ---------------------------------------------------
        // Levels with few positions split each position's candidate range over several lanes of the line's 16-lane row
        // (min-reduced with shuffles); levels with many positions switch to lane = (line, slot): the 16 lanes of a row then
        // work on the SAME position of 16 neighbouring lines, whose ranges are alike (the scene is coherent across
        // lines), so the per-lane loops of a row have nearly the same trip count.
        const int lineT = t & (NL - 1), slotT = t / NL;
        const uint32_t qmnT = span[2 * lineT], qmxT = span[2 * lineT + 1];
        const bool actT = qmnT <= qmxT;
        {
            const int lineU = t >> 4, g = t & 15;
            const uint32_t* klU = keys + lineU * pitch;
            uint16_t* aU = args + lineU * AP;
            const uint32_t qmn = span[2 * lineU], qmx = span[2 * lineU + 1];
            const bool actU = qmn <= qmx;
            const uint32_t* klT = keys + lineT * pitch;
            uint16_t* aT = args + lineT * AP;
            for (int l = 0; l < ((a.dbg & 1) ? 0 : Kp); ++l) {
                const int h = 1 << (Kp - 1 - l);
                const int n = (M / h + 1) >> 1;                 // positions of this level: i' = h (2 j + 1) <= M
                if (n <= 8) {
                    int sh = 4;                                 // log2 of the lanes per position
                    while ((16 >> sh) < n) --sh;
                    const int G = 1 << sh;
                    const int j = g >> sh, u = g & (G - 1);
                    const int ip = h * (2 * j + 1);
                    const bool valid = actU && ip <= M;
                    int lo = 1, hi = 0;
                    if (valid) {
                        lo = (ip - h == 0) ? (int)qmn : (int)aU[ip - h];
                        hi = (ip + h > M) ? (int)qmx : (int)aU[ip + h];
                        // The first levels scan (nearly) the whole site span per position.  Any candidate's cost is an upper
                        // bound v of the optimum, and a candidate farther than sqrt(v) from p costs more than v: clip the
                        // range to p +- (sqrt(v) + 1).  On lines that pass near objects (every line of a uniformly sparse
                        // scene, every position outside the span) this shrinks the scan from the span to a few candidates.
                        const int pp = 8 * (ip - 1);
                        const int pc = min(max(pp, lo), hi);
                        const uint32_t cc = (2u * (uint32_t)pp) << B, p2b = __umul24((uint32_t)pp, (uint32_t)pp) << B;
                        const uint32_t v = min(klU[pc] + p2b - __umul24(cc, (uint32_t)pc),
                                               min(klU[lo] + p2b - __umul24(cc, (uint32_t)lo), klU[hi] + p2b - __umul24(cc, (uint32_t)hi))) >> B;
                        if (v < finf) {
                            const int w = (int)__builtin_sqrtf((float)v) + 2;
                            lo = max(lo, pp - w);
                            hi = min(hi, pp + w);
                        }
                    }
                    uint32_t best = scan(klU, 8u * (uint32_t)(ip - 1), lo, hi, u, G);
                    for (int off = 1; off < G; off <<= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off));
                    if (valid && u == 0) aU[ip] = (uint16_t)(best & mask);
                } else {
                    for (int j = slotT; j < n; j += NS) {
                        const int ip = h * (2 * j + 1);
                        if (actT) {
                            const int lo = (ip - h == 0) ? (int)qmnT : (int)aT[ip - h];
                            const int hi = (ip + h > M) ? (int)qmxT : (int)aT[ip + h];
                            const uint32_t best = scan(klT, 8u * (uint32_t)(ip - 1), lo, hi, 0, 1);
                            aT[ip] = (uint16_t)(best & mask);
                        }
                    }
                }
                __syncthreads();
            }
        }

        // ---- chunk phase: lane = (line, chunk of 8 positions); finish and store -------------------------------------------
        {
            const int line = lineT, slot = slotT;
            const uint32_t* kl = keys + line * pitch;
            const uint16_t* al = args + line * AP;
            const uint32_t qmx = qmxT;
            const bool act = actT;
            for (int i0 = 0; i0 < M; i0 += NS) {
                const int i = i0 + slot;
                const bool live = i < M;
                const int p0 = kDcChunk * i;
                int D[kDcChunk];
#pragma unroll
                for (int k = 0; k < kDcChunk; ++k) D[k] = kInf32;
                const int a0 = (live && act && !(a.dbg & 2)) ? (int)al[i + 1] : 0;
                const int a8 = (live && act && !(a.dbg & 2)) ? ((i + 2 <= M) ? (int)al[i + 2] : (int)qmx) : -1;
                if (live && act) {
                    // all 8 positions against every candidate of [a0, a8] (usually a few: the argmin moves ~ half a site per
                    // position), branch-free, two candidates per step; 2 VALU per evaluation, no per-position set-up.  Long ranges
                    // take the same loop: a per-lane divide-and-conquer over the chunk's positions (3 R instead of 8 R evaluations
                    // for a range of R) was kept for ranges >= 24 at first -- it ran the whole wave through both code paths and lost
                    // everywhere (KE3 0.82 -> 0.75 ms on the streaming scene, KE2 0.80 -> 0.63 ms at Bernoulli p = 0.003).
                    // With p_k = p0 + k: R_k(q) = (p_k^2 << B) - ((2 p_k) << B) q, R_k(a0) - R_{k-1}(a0) = ((2 (p0 - a0) + 2k - 1) << B).
                    uint32_t R[kDcChunk], nc[kDcChunk], best[kDcChunk];
                    const uint32_t W = (uint32_t)(2 * (p0 - a0)) << B;
                    R[0] = (__umul24((uint32_t)p0, (uint32_t)p0) << B) - __umul24(((uint32_t)(2 * p0)) << B, (uint32_t)a0);
                    nc[0] = 0u - ((uint32_t)(2 * p0) << B);
                    best[0] = 0xFFFFFFFFu;
#pragma unroll
                    for (int k = 1; k < kDcChunk; ++k) {
                        R[k] = R[k - 1] + W + ((uint32_t)(2 * k - 1) << B);
                        nc[k] = nc[k - 1] - (2u << B);
                        best[k] = 0xFFFFFFFFu;
                    }
                    for (int q = a0; q <= a8; q += 2) {
                        const uint32_t k0 = kl[q], k1 = kl[q + 1];
#pragma unroll
                        for (int k = 0; k < kDcChunk; ++k) {
                            best[k] = min(best[k], min(k0 + R[k], k1 + R[k] + nc[k]));
                            R[k] += 2u * nc[k];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < kDcChunk; ++k) {
                        const uint32_t d = best[k] >> B;
                        D[k] = d >= finf ? kInf32 : (int)d;
                    }
                }
                uint32_t cm = 0u;                               // bit k: voxel p0 + k of this line is filled
                if (live) cm = (sgn[line * SW + (p0 >> 5)] >> (p0 & 31)) & 0xFFu;
#pragma unroll
                for (int k = 0; k < kDcChunk; ++k) {
                    const int p = p0 + k;
                    const bool inl = live && p < L;
                    const bool filled = (cm >> k) & 1u;
                    const bool mine = inl && (filled == (cls == 1)) && !(a.dbg & 4);
                    if constexpr (STAGE == 2) {
                        // side-table convention (sdfgpu_sweep_x16.hpp): if ANY voxel of a group of 4 is saturated, the
                        // exact values of the whole group must be in the side table.  A pass only knows its own class, so a
                        // voxel also writes its exact value when its group holds a voxel of the other class (filled voxels
                        // always write theirs).
                        const int need = mine ? (D[k] >= kSat16 ? 1 : 0) : (inl ? 1 : 0);
                        // OR over the 4 lanes of the group (= 4 neighbouring lines): two DPP quad permutes, no LDS traffic
                        int any = need | __builtin_amdgcn_mov_dpp(need, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
                        any |= __builtin_amdgcn_mov_dpp(any, 0x4E, 0xF, 0xF, true);                  // quad_perm [2,3,0,1]
                        if (mine) emit(line, p, D[k], filled, cls == 1 || any, byz);
                    } else {
                        if (mine) emit(line, p, D[k], filled, false, byz);
                    }
                }
            }
            // Pass 0 finishes the tile's filled voxels itself when they are few and shallow (thin surfaces): one lane per
            // listed voxel, exact local search along its line -- a candidate at offset d can only win while d^2 < the best so
            // far, and S = 1 (a free voxel next door in the rows already swept) needs no search at all.  Deep or numerous
            // filled voxels raise flg[17] and the second pass does the class properly.
            if (cls == 0 && !(a.dbg & 8) && !probe) {
                const uint32_t nf = flg[16];
                if (nf > (uint32_t)kDcLocalFilled) {
                    if (t == 0) flg[17] = 1u;
                } else {
                    for (uint32_t e = (uint32_t)t; e < nf; e += (uint32_t)NT) {
                        const uint32_t ent = flist[e];
                        const int fl = (int)(ent >> 24), p = (int)((ent >> 8) & 0xffffu);
                        int D1 = (int)(ent & 0xffu);
                        if (D1 > kDcLocalMax) { flg[17] = 1u; continue; }
                        for (int d = 1; (int)__umul24(d, d) < D1; ++d) {
                            if (p - d >= 0) D1 = min(D1, (int)__umul24(d, d) + max(-raw_signed(fl, p - d), 0));
                            if (p + d < L) D1 = min(D1, (int)__umul24(d, d) + max(-raw_signed(fl, p + d), 0));
                        }
                        emit(fl, p, D1, true, true, byz_of(fl));
                    }
                }
            }
        }
        __syncthreads();                        // keys / args are rebuilt by the next class
    }

    if (probe) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            probe_far += __shfl_xor(probe_far, off);
            probe_tot += __shfl_xor(probe_tot, off);
            probe_mid += __shfl_xor(probe_mid, off);
        }
        // one pair of global atomics per workgroup (same-address atomics serialise at ~12 ns each)
        if ((t & 63) == 0) { atomicAdd(&flg[18], (uint32_t)probe_far); atomicAdd(&flg[19], (uint32_t)probe_tot); atomicAdd(&flg[20], (uint32_t)probe_mid); }
        __syncthreads();
        if (t == 0) { atomicAdd(a.probe_out, flg[18]); atomicAdd(a.probe_out + 1, flg[19]); if (a.probe_thr2 > 0) atomicAdd(a.probe_out - 1, flg[20]); }
        return;
    }
    if constexpr (STAGE == 3) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mxF = max(mxF, __shfl_xor(mxF, off));
            mxQ = max(mxQ, __shfl_xor(mxQ, off));
        }
        if ((t & 63) == 0) slot_max2(a.maxdsq, blockIdx.x * (NT / 64) + (t >> 6), mxF, mxQ);
    }
}

// Device-side tier selection for one axis (stage 0 = y, 1 = x), one thread.  small = the context's status block:
// [3] uncertified (dense tier could not decide the scene), [4] / [5] "y / x sweep belongs to the envelope kernel" (also
// raised by a marching sweep that hits its scan bound), [8 + 2 stage] guard word of the marching sweep, [12] / [13] the
// probe's counters.  The marching sweep runs iff the general pipeline is needed at all and the probe found the axis
// near-field; otherwise the envelope flag is raised and the (flag-guarded) envelope kernel does the sweep.
__global__ void k_decide_tier(uint32_t* __restrict__ small, int stage, int dense_tried, int force, int num, int den, int handoff,
                              int mid_den) {
    const bool active = dense_tried ? small[3] != 0u : true;
    const uint32_t far_n = small[12], tot = small[13], mid_n = small[11];
    // far when more than num / den of the sampled voxels need a long scan (64-bit: counts are < 2^24, factors small)
    bool far = force >= 0 ? force != 0 : (uint64_t)far_n * (uint64_t)den > (uint64_t)tot * (uint64_t)num;
    if (stage == 1 && small[7] != 0u) far = true;               // the y probe chose the far-field pair with int32 hand-off
    // y sweep, near-field: radius-8 register windows when more than 1 / mid_den of the voxels are beyond the radius-3 one
    const bool wide = stage == 0 && mid_den > 0 && (uint64_t)mid_n * (uint64_t)mid_den > (uint64_t)tot;
    small[8 + 2 * stage] = (active && !far && !wide) ? 1u : 0u;
    if (stage == 0) small[9] = (active && !far && wide) ? 1u : 0u;
    if (active && far) small[4 + stage] = 1u;
    if (stage == 0 && active && far && handoff) small[7] = 1u;
    small[11] = 0u;
    small[12] = 0u;
    small[13] = 0u;
    small[14 + stage] = tot ? (far_n * 1000u) / tot : 0u;         // per-mille of far voxels in the sample (diagnostics)
}

}  // namespace sdfgpu
