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
//   chunk phase   lane = (line, chunk of 8 positions): the 7 interior positions by the same subdivision with the
//                 bounds held in registers, results converted and stored straight from the lane (16 lines x 2 / 4 B
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

struct EnvDcArgs {
    const int16_t* in16;      // STAGE 2: z field (+-g, 32767 = none); STAGE 3: plane field p16
    const int32_t* side_in;   // STAGE 3: exact values where p16 is saturated (valid for the whole group of 4)
    void* out;                // STAGE 2: int16 plane field; STAGE 3: float sdf
    int32_t* side_out;        // STAGE 2: exact plane values for groups of 4 that hold a saturated value
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
};

inline size_t envelope_dc_lds_bytes(int L, int pitch) {
    const int SW = (L + 31) / 32, M = (L + kDcChunk - 1) / kDcChunk;
    return ((size_t)kDcLines * pitch + (size_t)kDcLines * SW + 64) * 4 + (size_t)kDcLines * (M + 2) * 2;
}

template <int STAGE>
__global__ __launch_bounds__(256) void k_envelope_dc(const EnvDcArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dc_smem[];
    if (a.guard) {
        const uint32_t gv = *a.guard;
        if ((gv != 0u) == (a.guard_invert != 0)) return;
    }
    const int L = a.L, B = a.B, pitch = a.pitch, M = a.M, Kp = a.Kp;
    const int SW = (L + 31) >> 5, AP = M + 2;
    const uint32_t mask = (1u << B) - 1u, finf = a.finf;
    uint32_t* keys = dc_smem;                                   // [16][pitch]
    uint32_t* sgn = keys + kDcLines * pitch;                    // [16][SW]   bit p: voxel p of the line is filled
    uint32_t* span = sgn + kDcLines * SW;                       // [16][2]    first / last site of the line
    uint32_t* flg = span + 32;                                  // [16] line holds a filled voxel, [16] = tile does
    uint16_t* args = reinterpret_cast<uint16_t*>(flg + 32);     // [16][AP]   argmin of coarse position i' (1-based)
    const int t = threadIdx.x;

    // tile of this workgroup; consecutive tiles share cache lines (16 lines x 2 B = 32 B of a 128-B line), so the
    // workgroups that an XCD receives round-robin are mapped to one contiguous range of tiles
    int64_t tile = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) tile = (int64_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t o = tile / a.tiles_per_outer;
    const int64_t c0 = (tile - o * a.tiles_per_outer) * kDcLines;
    const int64_t base = o * a.outer_stride + c0;               // element index of (line 0, position 0)
    const int64_t ls = a.line_stride;

    // one candidate range for one position: lanes u = 0 .. G-1 of a group take the pairs lo + 2u, lo + 2u + 2G, ...
    // (the second key of the last pair may be candidate hi + 1: it can tie but never beat the range's minimum, and on a
    // tie the smaller position wins the unsigned min, so the argmin stays inside [lo, hi])
    auto scan = [&](const uint32_t* kl, uint32_t p, int lo, int hi, int u, int G) -> uint32_t {
        const uint32_t c = (2u * p) << B;
        int q = lo + 2 * u;
        uint32_t R = ((p * p) << B) - c * (uint32_t)q;
        const uint32_t dR = c * (uint32_t)(2 * G);
        uint32_t best = 0xFFFFFFFFu;
        for (; q <= hi; q += 2 * G) {
            const uint32_t k0 = kl[q], k1 = kl[q + 1];
            best = min(best, min(k0 + R, k1 + R - c));
            R -= dR;
        }
        return best;
    };

    int mxF = 0, mxQ = 0;
    // virtual-border distance of this lane's line in the chunk phase (STAGE 3: line = (y, z))
    int byz = kInf32;
    if constexpr (STAGE == 3) {
        if (a.vb) {
            const int64_t c = c0 + (t & 15);
            const int64_t vy = c / a.nz, vz = c - vy * a.nz;
            int64_t b = kInf32;
            if (a.ny > 1) b = min(b, min(vy + 1, a.ny - vy));
            if (a.nz > 1) b = min(b, min(vz + 1, a.nz - vz));
            byz = (int)b;
        }
    }

#pragma unroll 1
    for (int cls = 0; cls < 2; ++cls) {         // 0: sites of "distance to filled" (for free voxels); 1: the reverse
        // ---- stage the tile: rows -> keys ----------------------------------------------------------------------------
        if (cls == 0) {
            for (int i = t; i < kDcLines * SW; i += 256) sgn[i] = 0u;
            if (t < 32) flg[t] = 0u;
        }
        if (t < 16) { span[2 * t] = 0xFFFFFFFFu; span[2 * t + 1] = 0u; }
        __syncthreads();
        if (cls == 1 && flg[16] == 0u) break;   // no filled voxel in the tile: nothing to produce (block-uniform)
        {
            const int sub = t & 3, r = t >> 2;
            uint32_t mn[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[4] = {0u, 0u, 0u, 0u};
            bool anyf[4] = {false, false, false, false};
            for (int p = r; p < L; p += 64) {
                const int64_t idx = base + (int64_t)p * ls + 4 * sub;
                const uint2 raw = *reinterpret_cast<const uint2*>(a.in16 + idx);
                int s[4] = {(int)(short)(raw.x & 0xffffu), (int)(short)(raw.x >> 16), (int)(short)(raw.y & 0xffffu),
                            (int)(short)(raw.y >> 16)};
                if constexpr (STAGE == 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int g = abs(s[k]);
                        const int sq = g >= kInf16 ? kInf32 : g * g;
                        s[k] = s[k] < 0 ? -sq : sq;
                    }
                } else {
                    bool sat = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sat |= abs(s[k]) >= kSat16;
                    if (sat) {
                        const int4 e = *reinterpret_cast<const int4*>(a.side_in + idx);
                        s[0] = e.x; s[1] = e.y; s[2] = e.z; s[3] = e.w;
                    }
                }
                const uint32_t pp = (uint32_t)p * (uint32_t)p;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int line = 4 * sub + k;
                    uint32_t F;
                    bool none;
                    if (cls == 0) {
                        if (s[k] < 0) { atomicOr(&sgn[line * SW + (p >> 5)], 1u << (p & 31)); anyf[k] = true; }
                        F = s[k] > 0 ? (uint32_t)s[k] : 0u;
                        none = s[k] >= kInf32;
                    } else if (s[k] < 0) {
                        F = (uint32_t)(-s[k]);
                        none = -s[k] >= kInf32;
                    } else {                    // free voxel: a zero-valued site only next to a filled voxel of its line
                        F = 0u;
                        bool nb = false;
                        if (p > 0) nb |= (sgn[line * SW + ((p - 1) >> 5)] >> ((p - 1) & 31)) & 1u;
                        if (p + 1 < L) nb |= (sgn[line * SW + ((p + 1) >> 5)] >> ((p + 1) & 31)) & 1u;
                        none = !nb;
                    }
                    keys[line * pitch + p] = (((none ? finf : F) + pp) << B) | (uint32_t)p;
                    if (!none) { mn[k] = min(mn[k], (uint32_t)p); mx[k] = max(mx[k], (uint32_t)p); }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int line = 4 * sub + k;
                if (mn[k] != 0xFFFFFFFFu) { atomicMin(&span[2 * line], mn[k]); atomicMax(&span[2 * line + 1], mx[k]); }
                if (anyf[k]) { flg[line] = 1u; flg[16] = 1u; }
            }
            if (t < 16) keys[t * pitch + L] = ((finf + (uint32_t)L * (uint32_t)L) << B) | ((uint32_t)L & mask);   // sentinel
        }
        __syncthreads();

        // ---- upper levels: coarse positions p = 8 (i' - 1), i' = 1 .. M, 16 lanes per line -------------------------------
        {
            const int lineU = t >> 4, g = t & 15;
            const uint32_t* klU = keys + lineU * pitch;
            uint16_t* aU = args + lineU * AP;
            const uint32_t qmn = span[2 * lineU], qmx = span[2 * lineU + 1];
            const bool actU = qmn <= qmx;
            for (int l = 0; l < Kp; ++l) {
                const int h = 1 << (Kp - 1 - l);
                const int sh = l < 4 ? 4 - l : 0;               // log2 of the lanes per position
                const int G = 1 << sh;
                const int npl = l > 4 ? 1 << (l - 4) : 1;       // positions per lane
                for (int k = 0; k < npl; ++k) {
                    const int j = (g >> sh) + 16 * k;
                    const int u = g & (G - 1);
                    const int ip = h * (2 * j + 1);
                    const bool valid = actU && ip <= M;
                    int lo = 1, hi = 0;
                    if (valid) {
                        lo = (ip - h == 0) ? (int)qmn : (int)aU[ip - h];
                        hi = (ip + h > M) ? (int)qmx : (int)aU[ip + h];
                    }
                    uint32_t best = scan(klU, 8u * (uint32_t)(ip - 1), lo, hi, u, G);
                    for (int off = 1; off < G; off <<= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off));
                    if (valid && u == 0) aU[ip] = (uint16_t)(best & mask);
                }
                __syncthreads();
            }
        }

        // ---- chunk phase: lane = (line, chunk of 8 positions); finish and store -------------------------------------------
        {
            const int line = t & 15, slot = t >> 4;
            const uint32_t* kl = keys + line * pitch;
            const uint16_t* al = args + line * AP;
            const uint32_t qmn = span[2 * line], qmx = span[2 * line + 1];
            const bool act = qmn <= qmx;
            for (int i0 = 0; i0 < M; i0 += 16) {
                const int i = i0 + slot;
                const bool live = i < M;
                const int p0 = kDcChunk * i;
                int D[kDcChunk];
#pragma unroll
                for (int k = 0; k < kDcChunk; ++k) D[k] = kInf32;
                if (live && act) {
                    const int a0 = al[i + 1];
                    const int a8 = (i + 2 <= M) ? (int)al[i + 2] : (int)qmx;
                    // position p0 + k over [lo, hi]: distance to D[k], argmin returned (positions beyond the line end
                    // only pass the upper bound on)
                    auto pos = [&](auto kc, int lo, int hi) -> int {
                        constexpr int k = decltype(kc)::value;
                        if (p0 + k >= L) return hi;
                        const uint32_t best = scan(kl, (uint32_t)(p0 + k), lo, hi, 0, 1);
                        const uint32_t d = best >> B;
                        D[k] = d >= finf ? kInf32 : (int)d;
                        return (int)(best & mask);
                    };
                    pos(std::integral_constant<int, 0>{}, a0, a0);
                    const int a4 = pos(std::integral_constant<int, 4>{}, a0, a8);
                    const int a2 = pos(std::integral_constant<int, 2>{}, a0, a4);
                    const int a6 = pos(std::integral_constant<int, 6>{}, a4, a8);
                    pos(std::integral_constant<int, 1>{}, a0, a2);
                    pos(std::integral_constant<int, 3>{}, a2, a4);
                    pos(std::integral_constant<int, 5>{}, a4, a6);
                    pos(std::integral_constant<int, 7>{}, a6, a8);
                }
                uint32_t cm = 0u;                               // bit k: voxel p0 + k of this line is filled
                if (live) cm = (sgn[line * SW + (p0 >> 5)] >> (p0 & 31)) & 0xFFu;
#pragma unroll
                for (int k = 0; k < kDcChunk; ++k) {
                    const int p = p0 + k;
                    const bool inl = live && p < L;
                    const bool filled = (cm >> k) & 1u;
                    const bool mine = inl && (filled == (cls == 1));
                    const int64_t oi = base + line + (int64_t)p * ls;
                    if constexpr (STAGE == 2) {
                        // side-table convention (sdfgpu_sweep_x16.hpp): if ANY voxel of a group of 4 is saturated, the
                        // exact values of the whole group must be in the side table.  This pass only knows its own
                        // class, so a voxel also writes its exact value when its group holds a voxel of the other class
                        // (those write theirs in their own pass: always).
                        const int need = mine ? (D[k] >= kSat16 ? 1 : 0) : (inl ? 1 : 0);
                        int any = need | __shfl_xor(need, 1);
                        any |= __shfl_xor(any, 2);
                        if (mine) {
                            const int sD = filled ? -D[k] : D[k];
                            reinterpret_cast<int16_t*>(a.out)[oi] = (int16_t)(filled ? -min(D[k], kSat16) : min(D[k], kSat16));
                            if (cls == 1 || any) a.side_out[oi] = sD;
                        }
                    } else {
                        if (mine) {
                            int Dk = D[k];
                            if (a.vb) {
                                int b = byz;
                                if (a.nx > 1) b = min(b, (int)min((int64_t)p + 1, a.nx - p));
                                if (b < 32768) Dk = min(Dk, b * b);
                            }
                            if (filled) mxQ = max(mxQ, Dk); else mxF = max(mxF, Dk);
                            const float f = (Dk >= kInf32) ? __builtin_inff() : (float)(sqrt((double)Dk) * a.resolution);
                            reinterpret_cast<float*>(a.out)[oi] = filled ? -f : f;
                        }
                    }
                }
            }
        }
        __syncthreads();                        // keys / args are rebuilt by the next class
    }

    if constexpr (STAGE == 3) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mxF = max(mxF, __shfl_xor(mxF, off));
            mxQ = max(mxQ, __shfl_xor(mxQ, off));
        }
        if ((t & 63) == 0) slot_max2(a.maxdsq, blockIdx.x * 4 + (t >> 6), mxF, mxQ);
    }
}

}  // namespace sdfgpu
