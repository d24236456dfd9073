// sdfgpu_dense3.hpp -- KD3: the dense ball kernel with |offset| <= 3 and the 13 complete levels d^2 in {1..6, 8..14}.
//
// The second dense stage for the density band between "almost dense" and far-field (Bernoulli p = 0.02 .. 0.045): KD with
// d^2 <= 8 leaves 0.96^92 = 2.3 % of the voxels undecided at p = 0.04 and 6 % at p = 0.03, more than the fix-up kernel can
// take (exact up to d^2 = 64, but a 16-lane row per voxel); d^2 <= 14 leaves 0.96^250 = 4e-5 and 0.97^250 = 5e-4 .. 2e-3.
// Evaluated level by level with the wave-uniform early stop it costs what KD costs on the scenes KD decides.
// Same structure as KD (sdfgpu_dense.hpp: staged bit tile with halo, levels in increasing d^2, cumulative, wave-uniform
// early stop; level index per voxel as bit-planes; signed pair table; 4 voxels per lane and store), with a halo of 3,
// 4 level bit-planes, a 1024-entry pair table, no virtual border (b = 3 would bind inside the ball: such scenes keep
// the other tiers) and one more way to give up: a wave with more than kBall3MaxUndecided undecided voxels raises
// `uncertified` at once (the fix-up kernel's per-tile cap would do the same one launch later).
// Where it runs (sdfgpu.hip): in KD's place, with KF behind it, whenever the host policy enqueues the fix-up stage; and behind
// KD in the same build, guarded on KD's verdict (DenseArgs::guard), when a build does not expect KD to decide the scene.
#pragma once
#include "sdfgpu_dense.hpp"

namespace sdfgpu {

constexpr int kBall3R = 3;
constexpr int kBall3Levels = 13;
constexpr int kBall3MaxUndecided = kFixCap;                   // per wave (2048 voxels): a wave with more undecided voxels than the fix-up kernel
                                                              // takes per TILE (4 waves) says so at once.  (Undecided voxels come in clusters --
                                                              // cavities: at p = 0.03, 0.17 % of the voxels, 3.5 per wave on average, 44 in the
                                                              // worst wave of a 64 x 64 x 128 grid -- so a tighter per-wave bound rejects scenes
                                                              // whose total is small.)
__host__ __device__ constexpr int ball3_level(int d2) {      // d^2 -> level index, -1 = not a complete level of the cube
    return (d2 >= 1 && d2 <= 6) ? d2 - 1 : (d2 >= 8 && d2 <= 14) ? d2 - 2 : -1;
}
__device__ constexpr int kBall3D2[kBall3Levels] = {1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14};

template <int LV>
__device__ __forceinline__ uint32_t ball3_level_pass(const uint32_t* c0, int hy, int rw, uint32_t O) {
    uint32_t acc = 0;
#pragma unroll
    for (int dx = -kBall3R; dx <= kBall3R; ++dx) {
#pragma unroll
        for (int dy = -kBall3R; dy <= kBall3R; ++dy) {
            bool any = false;
#pragma unroll
            for (int dz = -kBall3R; dz <= kBall3R; ++dz) any |= ball3_level(dx * dx + dy * dy + dz * dz) == LV;
            if (!any) continue;                              // (compile time: rows without an offset of this level are never loaded)
            const uint32_t* p = c0 + (dx * hy + dy) * rw;
            const uint32_t prev = p[-1], cur = p[0], next = p[1];
#pragma unroll
            for (int dz = -kBall3R; dz <= kBall3R; ++dz) {
                if (ball3_level(dx * dx + dy * dy + dz * dz) != LV) continue;
                const uint32_t S = dz == 0 ? cur
                                 : dz > 0 ? __builtin_amdgcn_alignbit(next, cur, dz)
                                          : __builtin_amdgcn_alignbit(cur, prev, 32 + dz);
                acc = __builtin_amdgcn_bitop3_b32(acc, O, S, 0xF6);          // acc | (O ^ S)
            }
        }
    }
    return acc;
}

// ZINV = nz <= BD * 4 (every expansion pass covers whole z-rows)
// NZW, TY: 0 = words per row and tile rows along y from the arguments; otherwise compile-time (the launcher picks the instance
// that matches: nz = 512 with 4 x 4-row tiles).  The level passes visit ~150 (dx, dy) rows of the staged tile; with the row pitch
// and the halo's width known, their LDS addresses are immediates of the ds_read instead of a scalar multiply and a vector add
// per row -- a sixth of the kernel's VALU instructions, and this kernel is VALU-bound wherever it is the dense stage (round 5).
template <int BD, bool ZINV, int NZW = 0, int TY = 0>
__global__ __launch_bounds__(BD) void k_ball_dense3(const DenseArgs a) {
    static_assert((NZW == 0) == (TY == 0) && (NZW & (NZW - 1)) == 0 && (TY & (TY - 1)) == 0, "both or neither; powers of two");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (a.guard && __hip_atomic_load(a.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;   // KD decided the scene
    // (block-uniform; an atomic load: a plain one may be served from a scalar / L1 cache line read before the flag went up)
    if (a.early_out && __hip_atomic_load(a.uncertified, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const int nzw = NZW ? NZW : a.nzw, lg = NZW ? __builtin_ctz(NZW ? NZW : 1) : a.log2_nzw;
    const int a_ty = TY ? TY : a.ty, a_log2_ty = TY ? __builtin_ctz(TY ? TY : 1) : a.log2_ty;
    const int a_tx = TY ? BD / (NZW ? NZW : 1) / TY : a.tx;
    const int a_inv_hy = TY ? (65536 + (TY + 2 * kBall3R) - 1) / (TY + 2 * kBall3R) : a.inv_hy;
    const int rwu = nzw + 2;                                  // words used per staged row (edge words replicated)
    // row pitch: a wave reads 64/nzw tile rows at once; pitch = nzw (mod 32) puts them on disjoint banks
    const int rw = nzw < 32 ? nzw + 32 : nzw + 2;
    const int hx = a_tx + 2 * kBall3R, hy = a_ty + 2 * kBall3R;
    // LDS: [signed pair table 8 KiB][magnitudes][level planes BD x 16 B][class words BD x 4 B][tile]
    float2* lut2 = reinterpret_cast<float2*>(smem_raw);                   // [1024] signed pair table
    float* magl = reinterpret_cast<float*>(smem_raw + 1024 * 8);          // [16] level magnitudes (64 B slot)
    uint32_t* uni = reinterpret_cast<uint32_t*>(smem_raw + 1024 * 8 + 64);     // [2 x waves] "all zero" / "all one" per staging wave
    uint32_t* planes = reinterpret_cast<uint32_t*>(smem_raw + 1024 * 8 + 128); // [BD][4]: level bits b0, b1, b2, b3
    uint32_t* cls = planes + BD * 4;                                      // [BD] class word
    uint32_t* tile = cls + BD;                                            // [hx][hy][rw]
    const int t = threadIdx.x;

    // signed pair table: bits {0,1} = class of voxels a,b (1 = filled -> negative); {2,3} = level bit 0; {4,5} = bit 1;
    // {6,7} = bit 2; {8,9} = bit 3.  Level 13 = "not found" -> +-0 (the fix-up kernel or the general pipeline rewrites it).
    // The 13 magnitudes come from the host as scalar kernel arguments, picked by lanes 0..15 with a select chain (see KD).
    if (t < 16) {
        float m = a.mag3[0];
        m = t == 1 ? a.mag3[1] : m; m = t == 2 ? a.mag3[2] : m; m = t == 3 ? a.mag3[3] : m; m = t == 4 ? a.mag3[4] : m;
        m = t == 5 ? a.mag3[5] : m; m = t == 6 ? a.mag3[6] : m; m = t == 7 ? a.mag3[7] : m; m = t == 8 ? a.mag3[8] : m;
        m = t == 9 ? a.mag3[9] : m; m = t == 10 ? a.mag3[10] : m; m = t == 11 ? a.mag3[11] : m; m = t == 12 ? a.mag3[12] : m;
        m = t >= 13 ? 0.0f : m;
        magl[t] = m;
    }

    const int x0 = a.out_lo + (int)blockIdx.y * a_tx;         // first tile plane (buffer coordinates)
    const int y0 = (int)blockIdx.x * a_ty;
    // stage the bit-rows of the tile + halo; rows outside the buffer / grid replicate the nearest row
    if (nzw >= 4) {
        // one 16-byte load per lane and staged quarter-row: (hx*hy rows) x (nzw/4 quads); all loads of a
        // workgroup are independent, so staging costs a single L2 round trip
        const int lq = lg - 2;                                // log2(quads per row)
        const int total = (hx * hy) << lq;
        uint32_t any1 = 0u, all1 = ~0u;                       // OR / AND of everything this lane stages
        for (int i0 = 0; i0 < total; i0 += 2 * BD) {
            uint4 v[2];
            int rowi[2], quad[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = min(i0 + u * BD + t, total - 1);
                rowi[u] = i >> lq; quad[u] = i & ((1 << lq) - 1);
                const int jx = (rowi[u] * a_inv_hy) >> 16, jy = rowi[u] - jx * hy;
                const int gx = min(max(x0 + jx - kBall3R, 0), a.rows_x - 1);
                const int gy = min(max(y0 + jy - kBall3R, 0), a.ny - 1);
                v[u] = *reinterpret_cast<const uint4*>(a.bits + ((int64_t)gx * a.ny + gy) * nzw + 4 * quad[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (i0 + u * BD + t < total) {
                    uint32_t* dst = tile + rowi[u] * rw + 1 + 4 * quad[u];
                    dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
                    any1 |= v[u].x | v[u].y | v[u].z | v[u].w;
                    all1 &= v[u].x & v[u].y & v[u].z & v[u].w;
                    if (quad[u] == 0) dst[-1] = (v[u].x & 1u) ? ~0u : 0u;                    // replicate the first voxel
                    if (quad[u] == (1 << lq) - 1) dst[4] = (v[u].w >> 31) ? ~0u : 0u;        // ... and the last one
                }
            }
        }
        // tile + halo of one class only (empty or solid space): no voxel of this tile can be decided -- say so before the
        // level passes (a far-field scene's first, staged build then costs a staging round per resident workgroup, not 13
        // level passes and an expansion: first streaming frame 1.74 -> 1.6 ms)
        const bool z = !__any(any1 != 0u), o = !__any(all1 != ~0u);
        if ((t & 63) == 0) { uni[2 * (t >> 6)] = z ? 1u : 0u; uni[2 * (t >> 6) + 1] = o ? 1u : 0u; }
    } else {
        if ((t & 63) == 0) { uni[2 * (t >> 6)] = 0u; uni[2 * (t >> 6) + 1] = 0u; }
        // narrow rows (nz = 32 or 64): word-wise staging, lanes laid out as (row-in-pass, word)
        const int lgp = max(lg + 1, 2);                       // 2^lgp >= nzw + 2 lanes per staged row
        const int lw = t & ((1 << lgp) - 1), lr = t >> lgp;   // word slot, row-in-pass
        const int rpp = BD >> lgp;                            // rows staged per pass
        if (lw < rwu) {
            for (int jy = lr; jy < hy; jy += rpp) {
                const int gy = min(max(y0 + jy - kBall3R, 0), a.ny - 1);
                for (int jx = 0; jx < hx; ++jx) {
                    const int gx = min(max(x0 + jx - kBall3R, 0), a.rows_x - 1);
                    const uint32_t* row = a.bits + ((int64_t)gx * a.ny + gy) * nzw;
                    uint32_t x;
                    if (lw == 0) x = (row[0] & 1u) ? ~0u : 0u;
                    else if (lw == rwu - 1) x = (row[nzw - 1] >> 31) ? ~0u : 0u;
                    else x = row[lw - 1];
                    tile[(jx * hy + jy) * rw + lw] = x;
                }
            }
        }
    }
    __syncthreads();

    if (a.early_out) {                                        // (block-uniform)
        bool z = true, o = true;
#pragma unroll
        for (int k = 0; k < BD / 64; ++k) { z = z && uni[2 * k] != 0u; o = o && uni[2 * k + 1] != 0u; }
        if (z || o) {
            if (t == 0) { raise_flag(a.uncertified); note_reason(a.reason, kGiveUpOneClassTile); }
            return;
        }
    }
    const int r = t >> lg, w = t & (nzw - 1);                 // tile row, word in row
    const int ty_ = r & (a_ty - 1), tx_ = r >> a_log2_ty;
    const uint32_t* c0 = tile + ((tx_ + kBall3R) * hy + (ty_ + kBall3R)) * rw + (w + 1);
    const uint32_t O = c0[0];
    // Levels in increasing d^2, cumulative; stop as soon as every voxel of the wave has been decided.
    // On Bernoulli(0.5) occupancy 98.4 % of the voxels have a face neighbour of the other class and all
    // but ~2^-18 are decided by d^2 <= 2, so a wave normally evaluates 18 of the 92 offsets.
    uint32_t acc[kBall3Levels];
    {
        uint32_t cum = 0;
        bool done = false;
        static_for<kBall3Levels>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if (!done) {
                cum |= ball3_level_pass<l>(c0, hy, rw, O);
                done = __all(cum == ~0u);
            }
            acc[l] = cum;
        });
    }

    // extrema (max d^2 per class) and certification, per word
    int mxF = 0, mxQ = 0;
    {
        uint32_t prevc = 0;
        static_for<kBall3Levels>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            {
                const uint32_t first = acc[l] & ~prevc;
                if (first & ~O) mxF = kBall3D2[l];
                if (first & O) mxQ = kBall3D2[l];
                prevc = acc[l];
            }
        });
    }
    const bool row_in_grid = (x0 + tx_ < a.out_hi) && (y0 + ty_ < a.ny);
    const bool uncert = row_in_grid && (~acc[kBall3Levels - 1] != 0u);
    if (!row_in_grid) { mxF = 0; mxQ = 0; }

    // level index per voxel = number of levels it was NOT found at (0..12, 13 = not found): the U_l are nested, so bit k of
    // the count is the parity of the U_l with l = 2^k - 1 (mod 2^(k+1))
    {
        uint32_t U[kBall3Levels];
#pragma unroll
        for (int l = 0; l < kBall3Levels; ++l) U[l] = ~acc[l];
        uint4 pl;
        pl.x = U[0] ^ U[1] ^ U[2] ^ U[3] ^ U[4] ^ U[5] ^ U[6] ^ U[7] ^ U[8] ^ U[9] ^ U[10] ^ U[11] ^ U[12];
        pl.y = U[1] ^ U[3] ^ U[5] ^ U[7] ^ U[9] ^ U[11];
        pl.z = U[3] ^ U[7] ^ U[11];
        pl.w = U[7];
        reinterpret_cast<uint4*>(planes)[t] = pl;
        cls[t] = O;
    }
    for (int i = t; i < 1024; i += BD) {
        const int la = ((i >> 2) & 1) | ((i >> 3) & 2) | ((i >> 4) & 4) | ((i >> 5) & 8);
        const int lb = ((i >> 3) & 1) | ((i >> 4) & 2) | ((i >> 5) & 4) | ((i >> 6) & 8);
        lut2[i] = make_float2(__uint_as_float(__float_as_uint(magl[la]) | ((uint32_t)(i & 1) << 31)),
                              __uint_as_float(__float_as_uint(magl[lb]) | ((uint32_t)(i & 2) << 30)));
    }
    __syncthreads();

    // expansion: a lane finishes 4 consecutive voxels per pass -> every store instruction writes one
    // fully contiguous 1 KiB segment per wave (8 voxels per lane halves the instruction count but
    // makes each store half-strided: measured 20 % slower).  Two signed pair-table lookups give the 4
    // finished floats; with ZINV (a pass of BD*4 voxels is a whole number of z-rows, nz <= BD*4) the lane's
    // z, its bit position and the plane address are loop-invariant, and the destination is a wave-uniform
    // tile pointer plus a 32-bit lane offset.
    const int nz = nzw << 5;
    const int lgz = lg + 5;
    char* const tile_out = reinterpret_cast<char*>(a.out + ((int64_t)(x0 - a.out_lo) * a.ny + y0) * nz);
    const uint4* planes4 = reinterpret_cast<const uint4*>(planes);
    typedef float f4v __attribute__((ext_vector_type(4)));
    auto expand = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;        // whole tile inside the output range: no bounds checks
        const int v0 = t << 2;
        const int zi = v0 & (nz - 1), r0 = v0 >> lgz;         // ZINV: this lane's z and first row
        const int rs = (BD * 4) >> lgz;                       //       rows per pass
        const uint4* pbase = planes4 + (r0 << lg) + (zi >> 5);
        const uint32_t* cbase = cls + (r0 << lg) + (zi >> 5);
        // ZINV: row r0 + j*rs splits into (tx, ty) without carries between the lane part r0 (< rs) and the
        // wave-uniform part j*rs, so the byte offset is a per-lane constant plus a scalar per pass
        const int ty0 = r0 & (a_ty - 1), tx0 = r0 >> a_log2_ty;
        const uint32_t lane_off = (uint32_t)(((((int)__umul24(tx0, a.ny) + ty0) << lgz) + zi) << 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {                         // fully unrolled: 8 independent LDS->LUT->store chains
            int rr, z;
            uint4 pl;
            uint32_t cw;
            if constexpr (ZINV) {
                rr = r0 + j * rs; z = zi;
                pl = pbase[j * (BD / 8)];                     // (rs << lg) == BD / 8 plane entries per pass
                cw = cbase[j * (BD / 8)];
            } else {
                const int v = j * (BD * 4) + v0;              // voxel index inside the tile (row-major)
                rr = v >> lgz; z = v & (nz - 1);
                pl = planes4[(rr << lg) + (z >> 5)];
                cw = cls[(rr << lg) + (z >> 5)];
            }
            const uint32_t sh = (uint32_t)z & 31u;
            const uint32_t ia = __builtin_amdgcn_ubfe(cw, sh, 2u) | (__builtin_amdgcn_ubfe(pl.x, sh, 2u) << 2) |
                                (__builtin_amdgcn_ubfe(pl.y, sh, 2u) << 4) | (__builtin_amdgcn_ubfe(pl.z, sh, 2u) << 6) |
                                (__builtin_amdgcn_ubfe(pl.w, sh, 2u) << 8);
            const uint32_t ib = __builtin_amdgcn_ubfe(cw, sh + 2u, 2u) | (__builtin_amdgcn_ubfe(pl.x, sh + 2u, 2u) << 2) |
                                (__builtin_amdgcn_ubfe(pl.y, sh + 2u, 2u) << 4) | (__builtin_amdgcn_ubfe(pl.z, sh + 2u, 2u) << 6) |
                                (__builtin_amdgcn_ubfe(pl.w, sh + 2u, 2u) << 8);
            const float2 fa = lut2[ia], fb = lut2[ib];
            const int tyy = rr & (a_ty - 1), txx = rr >> a_log2_ty;
            if (FULL || (x0 + txx < a.out_hi && y0 + tyy < a.ny)) {
                f4v ov;
                ov.x = fa.x; ov.y = fa.y; ov.z = fb.x; ov.w = fb.y;
                f4v* dst;
                if constexpr (ZINV) {
                    const int rj = j * rs;                                                    // wave-uniform
                    const int64_t uoff = (((int64_t)(rj >> a_log2_ty) * a.ny + (rj & (a_ty - 1))) << lgz) << 2;
                    dst = reinterpret_cast<f4v*>(tile_out + uoff + lane_off);
                } else {
                    dst = reinterpret_cast<f4v*>(tile_out + (uint32_t)((((int)__umul24(txx, a.ny) + tyy) << lgz) + z) * 4u);
                }
                if (a.nt_store) __builtin_nontemporal_store(ov, dst);
                else *dst = ov;
            }
        }
    };
    if (!(a.checked & 1) && (x0 + a_tx <= a.out_hi) && (y0 + a_ty <= a.ny)) expand(std::true_type{});
    else expand(std::false_type{});

#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mxF = max(mxF, __shfl_xor(mxF, off));
        mxQ = max(mxQ, __shfl_xor(mxQ, off));
    }
    // more than kBall3MaxUndecided undecided voxels in this wave's 2048: the scene is too sparse for this tier (the fix-up
    // kernel behind would take longer than the sweeps) -- say so now, the later workgroups return at once
    int nund = row_in_grid ? __popc(~acc[kBall3Levels - 1]) : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nund += __shfl_xor(nund, off);
    const bool hopeless = nund > a.max_undecided;
    const bool any_uncert = __any(uncert);
    const uint32_t open_words = (uint32_t)__popcll(__ballot(uncert));    // words of this wave that hold undecided voxels
    const bool all_undecided = __all(row_in_grid && acc[kBall3Levels - 1] == 0u);
    if (any_uncert && a.unc) {
        if (row_in_grid)
            a.unc[((int64_t)(x0 + tx_ - a.out_lo) * a.ny + (y0 + ty_)) * nzw + w] = ~acc[kBall3Levels - 1];
    }
    if ((t & 63) == 0) {
        const uint32_t tile_id = (uint32_t)blockIdx.y * gridDim.x + blockIdx.x;
        // The early-out test at the top is per wave: if the flag rises between the loads of two waves of one workgroup, some
        // waves leave and the others run on over a partly staged tile -- what they find is garbage.  The field is rewritten
        // by the stage behind (the flag is up), but maxima are max-folded: they must not leave this wave.  A wave that can
        // have been affected sees the flag set HERE (it only ever rises), and then the stage behind recomputes every
        // maximum anyway.
        const bool void_maxima = a.early_out && __hip_atomic_load(a.uncertified, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        if (!void_maxima) slot_max2(a.slots, tile_id * (BD / 64) + ((uint32_t)t >> 6), mxF, mxQ);
        // (every 16th wave adds its undecided voxels to one of the 512 extrema slots' third word: a 1 / 16 sample of the scene's total
        //  for k_shell_budget.  ONE status word for all of them -- a thousand same-address atomics -- cost this kernel 76 - 190 us.)
        if (a.und_sample && nund) {
            const uint32_t wg = tile_id * (BD / 64) + ((uint32_t)t >> 6);
            if ((wg & 15u) == 0u) atomicAdd(a.und_sample + (size_t)((wg >> 4) & (kSlots - 1)) * kSlotWords + 2, (uint32_t)nund);
        }
        if (any_uncert) {
            if (a.unc) {
                // one word per tile: no same-address pile-up.  Bits 0 .. 15: the wave's flag; bits 16 ..: the number of WORDS this wave
                // leaves undecided voxels in, summed over the tile's waves -- the shell pass reads four of these words and knows at
                // once whether the group is worth staging (round 5).  (Every wave adds its own bit exactly once: add == or.)
                atomicAdd(a.tileflag + tile_id, (1u << (t >> 6)) + (open_words << 16));
                raise_flag(a.fix_needed);
                // a wave without a single decided voxel sits in empty (or solid) space: nothing for the fix-up kernel
                if ((a.early_out && all_undecided) || hopeless) {
                    raise_flag(a.uncertified);
                    note_reason(a.reason, hopeless ? kGiveUpWaveTooMany : kGiveUpWaveAllUndecided);
                }
            } else {
                raise_flag(a.uncertified);
                note_reason(a.reason, kGiveUpBeyondBall);
            }
        }
    }
}


}  // namespace sdfgpu
