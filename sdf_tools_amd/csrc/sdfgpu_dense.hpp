// sdfgpu_dense.hpp -- K0 + KD: the dense-scene path, bit-parallel and exact-or-flagged.
//
//   K0  k_pack_bits   occupancy bytes / COLLISION_CELL records -> 1 bit per voxel (32 voxels per word,
//                     z fastest), 1 B read + 1/8 B written per voxel
//   KD  k_ball_dense  bit field -> fp32 SDF for every voxel whose nearest opposite-class voxel lies
//                     within squared distance 8, and a flag if any voxel is farther than that
//
// Why: on dense scenes (the benchmark's Bernoulli p = 0.5 grids have D <= ~6) the three separable
// sweeps spend their time expanding bits into integers and back.  KD never leaves the bit domain
// until the final store: a lane owns one 32-voxel word; for each of the 92 lattice offsets o with
// |o|^2 <= 8 it XORs its word with the (funnel-shifted) word at that offset -- a set bit means "the
// voxel at this offset has the other class" -- and ORs the result into the plane of level |o|^2
// (7 levels: 1,2,3,4,5,6,8).  The first level with a set bit is the voxel's exact squared distance,
// for both classes at once (the XOR is symmetric), which is exactly the signed merge of
// sdf_generation.hpp:245-269.  ~9 integer ops per voxel instead of ~95, and the only HBM traffic is
// the bit field (1/8 B/voxel, re-read ~4x from L2 for the halo) and the 4 B/voxel output.
// Neighbour rows come from an LDS tile of bit-rows with a 2-row halo; rows / bits outside the grid
// replicate the nearest in-grid voxel, which cannot create a false hit (the replicated voxel is a
// real voxel at a smaller-or-equal offset that is itself enumerated).
//
// Exactness: a voxel with no hit has D >= 9; it raises *uncertified and the caller's general
// pipeline (K12/K1+K2, K3 -- launched right behind, guarded by the same flag) recomputes the whole
// grid.  If the flag stays 0 those kernels exit immediately.  So results are exact for any input;
// only the speed depends on the scene.
#pragma once
#include "sdfgpu_kernels.hpp"

namespace sdfgpu {

// ---------------------------------------------------------------------------------------------
// K0: pack.  bits[(row) * nzw + w] bit i = voxel z = 32 w + i of that row is filled.
// ---------------------------------------------------------------------------------------------

// uint8 mask, nz % 32 == 0, 16-byte aligned: a lane folds 16 bytes to 16 bits, lane pairs form one 32-bit
// word (DPP quad permute, no LDS).  CHUNKS independent 16-byte loads per lane are in flight at once,
// taken a whole grid apart so every load instruction stays a contiguous 1 KiB per wave.  The mask is read
// exactly once, so the loads are non-temporal: measured 53 -> 31 us at 512^3, and the ball kernel behind
// it gets faster too because the 134 MB mask no longer evicts the 17 MB bit field from L2 / MALL.
template <int CHUNKS, bool NT>
__global__ __launch_bounds__(kBlock) void k_pack_bits_mask(const uint8_t* __restrict__ mask,
                                                          uint32_t* __restrict__ bits, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // 16-voxel chunk index
    uint4 v[CHUNKS];
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
        const int64_t i = i0 + k * stride;
        if (i < n16) {
            const uint4* src = reinterpret_cast<const uint4*>(mask + 16 * i);
            if constexpr (NT) {
                v[k].x = __builtin_nontemporal_load(&src->x); v[k].y = __builtin_nontemporal_load(&src->y);
                v[k].z = __builtin_nontemporal_load(&src->z); v[k].w = __builtin_nontemporal_load(&src->w);
            } else {
                v[k] = *src;
            }
        } else {
            v[k] = make_uint4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
        const int64_t i = i0 + k * stride;
        const uint32_t b = nonzero_bits4(v[k].x) | (nonzero_bits4(v[k].y) << 4) | (nonzero_bits4(v[k].z) << 8) |
                           (nonzero_bits4(v[k].w) << 12);
        const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        if (i < n16 && (threadIdx.x & 1) == 0) bits[i >> 1] = b | (other << 16);
    }
}

// any loader (COLLISION_CELL records, unaligned masks): one ballot = 64 voxels; needs nz % 32 == 0
template <class Loader>
__global__ __launch_bounds__(kBlock) void k_pack_bits_generic(Loader ld, uint32_t* __restrict__ bits, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;       // voxel index
    const bool f = (i < n) ? ld.filled(i) : false;
    const uint64_t word = __ballot(f);
    const int lane = threadIdx.x & 63;
    const int64_t w0 = (i - lane) >> 5;
    if (lane == 0 && i < n) {
        bits[w0] = (uint32_t)word;
        if (i + 32 < n) bits[w0 + 1] = (uint32_t)(word >> 32);
    }
}

// The reverse of K0 for the host-buffer entry points (round 5): the host's thread team classifies the caller's cells / mask
// into ONE BIT per voxel while it fills the pinned staging chunk (1/8 B per voxel over PCIe instead of 1 B of mask or 8 B of
// COLLISION_CELL records: 16 MiB instead of 1 GiB at 512^3), and this kernel spreads the bits back into the 0 / 1 byte mask
// every tier's first kernel reads.  Linear bit order: bit (v & 31) of word (v >> 5) = voxel v.  A lane writes 16 bytes, a
// wave one contiguous 1 KiB per store instruction; 1/8 B read + 1 B written per voxel (~30 us at 512^3).
SDFGPU_KERNEL __launch_bounds__(kBlock) void k_unpack_bits_mask(const uint32_t* __restrict__ bits, uint8_t* __restrict__ mask, int64_t n) {
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;       // 16-voxel chunk index
    const int64_t v0 = 16 * c;
    if (v0 >= n) return;
    const uint32_t b = (bits[c >> 1] >> ((c & 1) * 16)) & 0xffffu;
    auto spread4 = [](uint32_t x) -> uint32_t {                         // bits 0..3 -> bytes 0..3 (0 / 1)
        return ((x & 1u) | ((x & 2u) << 7) | ((x & 4u) << 14) | ((x & 8u) << 21));
    };
    const uint4 o = make_uint4(spread4(b), spread4(b >> 4), spread4(b >> 8), spread4(b >> 12));
    if (v0 + 16 <= n && (reinterpret_cast<uintptr_t>(mask) & 15) == 0) {
        *reinterpret_cast<uint4*>(mask + v0) = o;
    } else {
        for (int k = 0; k < 16 && v0 + k < n; ++k) mask[v0 + k] = (uint8_t)((b >> k) & 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// KD: dense ball kernel.
// ---------------------------------------------------------------------------------------------

struct DenseArgs {
    const uint32_t* bits;   // [rows_x][ny][nzw]; rows_x includes halo planes in slab mode
    float* out;             // [out rows][ny][nz]
    int nzw;                // words per z-row (nz / 32), power of two, <= 64
    int log2_nzw;
    int ny;
    int rows_x;             // x-planes present in `bits`
    int out_lo, out_hi;     // x-planes (buffer coordinates) whose voxels are written
    int tx, ty;             // tile rows per workgroup along x / y (powers of two); tx * ty * nzw == block size
    int log2_ty;
    int inv_hy;             // ceil(65536 / (ty + 4)): row / hy = (row * inv_hy) >> 16 for row < 4096
    float mag[8];           // float(sqrt(double(level d^2)) * resolution) per level; [7] = 0 (not found)
    float mag3[16];         // ... of the 13 levels of KD3 (sdfgpu_dense3.hpp); [13..15] = 0
    uint32_t* slots;        // [kSlots][kSlotWords]: per-slot {max d^2 free, max d^2 filled}, see slot_max2 / k_fold_slots
    uint32_t* uncertified;  // set to 1 if some voxel has no opposite-class voxel within d^2 <= 8
    uint32_t* reason;       // nullptr, or a word that collects WHY the tier gave up (kGiveUp* bits; diagnostics, whole builds only)
    // fix-up mode (k_ball_fixup runs behind this launch): instead of raising `uncertified`, a wave that holds
    // undecided voxels writes its 64 "undecided" words, sets its bit in the tile's flag word and raises fix_needed
    uint32_t* unc;          // [out rows][ny][nzw] undecided bits (only written by waves that have some)
    uint32_t* tileflag;     // [tiles] bit w = wave w of the tile wrote its unc words (cleared again by k_ball_fixup)
    uint32_t* fix_needed;
    // early out (whole-build launches only): once `uncertified` is up the general sweeps will redo the whole grid, so
    // workgroups that start later return at once -- a far-field scene then costs one wave of workgroups, not the kernel
    int early_out;
    // virtual border (sdf_generation.hpp:287-419, net effect D <- min(D, b^2), b = axis distance to the padded layer over the
    // axes with more than one cell): inside the ball only b = 1 and b = 2 can bind, folded into levels 0 and 3
    int vb;
    int nx_glob;            // x extent of the whole grid (buffer plane 0 = grid plane 0 in this mode)
    int checked;            // debugging: take the bounds-checked expansion even for interior tiles
    int nt_store;           // write the output with non-temporal stores (it is never re-read here)
    const uint32_t* guard;  // KD3 only: non-null = run iff *guard != 0 (the staged fix-up stage behind KD in the same build)
    uint32_t* und_sample;   // KD3 only: nullptr, or the slot array: every 16th wave adds its undecided voxels to word 2 of a slot -- a 1 / 16
                            // sample of the scene's total, which k_shell_budget folds and holds against the shell pass's budget
    int max_undecided;      // KD3 only: a wave (2048 voxels) with more undecided voxels than this gives the tier up at once
                            // (kBall3MaxUndecided = KF's per-tile cap; with the shell pass KD6 behind KD3 the bound is KD6's reach)
};

// why the dense tier handed a scene on (status word 21 of whole builds; sdfgpu_last_dense_certified reports them from bit 8 up)
constexpr uint32_t kGiveUpOneClassTile = 1u, kGiveUpWaveAllUndecided = 2u, kGiveUpWaveTooMany = 4u, kGiveUpTileOverCap = 8u,
                   kGiveUpBeyondReach = 16u, kGiveUpBeyondBall = 32u, kGiveUpTooSparse = 64u;
__device__ __forceinline__ void note_reason(uint32_t* reason, uint32_t bit) {
    if (reason && !(__hip_atomic_load(reason, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(reason, bit);
}


constexpr int kBallR = 2;                                    // |dx|,|dy|,|dz| <= 2
__host__ __device__ constexpr int ball_level(int d2) {       // d^2 -> level index, -1 = not in the ball
    return d2 == 1 ? 0 : d2 == 2 ? 1 : d2 == 3 ? 2 : d2 == 4 ? 3 : d2 == 5 ? 4 : d2 == 6 ? 5 : d2 == 8 ? 6 : -1;
}
__device__ constexpr int kLevelD2[7] = {1, 2, 3, 4, 5, 6, 8};

// Hits of one level (all offsets with ball_level(|o|^2) == LV) for the 32 voxels of word O.  The triple
// loop is resolved at compile time; rows that hold no offset of this level are never loaded.
template <int LV>
__device__ __forceinline__ uint32_t ball_level_pass(const uint32_t* c0, int hy, int rw, uint32_t O) {
    uint32_t acc = 0;
#pragma unroll
    for (int dx = -kBallR; dx <= kBallR; ++dx) {
#pragma unroll
        for (int dy = -kBallR; dy <= kBallR; ++dy) {
            const uint32_t* p = c0 + (dx * hy + dy) * rw;
            const uint32_t prev = p[-1], cur = p[0], next = p[1];
#pragma unroll
            for (int dz = -kBallR; dz <= kBallR; ++dz) {
                if (ball_level(dx * dx + dy * dy + dz * dz) != LV) continue;
                // S bit i = voxel (x+dx, y+dy, z+dz) for this word's voxel i
                const uint32_t S = dz == 0 ? cur
                                 : dz > 0 ? __builtin_amdgcn_alignbit(next, cur, dz)
                                          : __builtin_amdgcn_alignbit(cur, prev, 32 + dz);
                acc = __builtin_amdgcn_bitop3_b32(acc, O, S, 0xF6);          // acc | (O ^ S) in one v_bitop3_b32
            }
        }
    }
    return acc;
}

// BD = workgroup size: a larger tile amortises the 2-row halo (4x -> 3x -> 2.25x rows staged)
// ZINV = nz <= BD * 4 (every expansion pass covers whole z-rows)
template <int BD, bool ZINV>
__global__ __launch_bounds__(BD) void k_ball_dense(const DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // (block-uniform; an atomic load: a plain one may be served from a scalar / L1 cache line read before the flag went up)
    if (a.early_out && __hip_atomic_load(a.uncertified, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const int nzw = a.nzw, lg = a.log2_nzw;
    const int rwu = nzw + 2;                                  // words used per staged row (edge words replicated)
    // row pitch: a wave reads 64/nzw tile rows at once; pitch = nzw (mod 32) puts them on disjoint banks
    // (pitch nzw + 2 cost a 2-way conflict on every one of the 75 tile reads: 8.8 M conflict cycles at 512^3)
    const int rw = nzw < 32 ? nzw + 32 : nzw + 2;
    const int hx = a.tx + 2 * kBallR, hy = a.ty + 2 * kBallR;
    // LDS: [signed pair table 2 KiB][planes BD x 16 B][tile] -- the two fixed-size parts first, so their
    // addresses are compile-time constants (folded into the ds_read offsets)
    float2* lut2 = reinterpret_cast<float2*>(smem_raw);                   // [256] signed pair table
    float* magl = reinterpret_cast<float*>(smem_raw + 256 * 8);           // [8] level magnitudes (64 B slot)
    uint32_t* planes = reinterpret_cast<uint32_t*>(smem_raw + 256 * 8 + 64);   // [BD][4]: b0, b1, b2, class
    uint32_t* tile = planes + BD * 4;                                     // [hx][hy][rw]
    const int t = threadIdx.x;

    // signed pair table: bits {0,1} = class of voxels a,b (1 = filled -> negative); {2,3} = level bit 0;
    // {4,5} = level bit 1; {6,7} = level bit 2.  Level 7 = "not found" -> +-0 (the general pipeline rewrites
    // it).  Class in the LOW bits: dense scenes use levels 0..1, i.e. entries 0..15 = 32 distinct banks (with
    // the class on top the 4 class combinations would fall on the same banks).
    // The 7 magnitudes float(sqrt(double d^2) * resolution) come from the host as scalar kernel arguments (no
    // f64 work per workgroup).  Lanes 0..7 pick theirs with a select chain -- indexing the argument array with
    // a lane-varying index makes the compiler issue VECTOR loads from the kernarg segment, measured +15 us
    // per launch -- and park them in LDS; the table itself is built behind the first barrier.
    if (t < 8) {
        float m = a.mag[0];
        m = t == 1 ? a.mag[1] : m; m = t == 2 ? a.mag[2] : m; m = t == 3 ? a.mag[3] : m;
        m = t == 4 ? a.mag[4] : m; m = t == 5 ? a.mag[5] : m; m = t == 6 ? a.mag[6] : m;
        m = t == 7 ? 0.0f : m;
        magl[t] = m;
    }

    const int x0 = a.out_lo + (int)blockIdx.y * a.tx;         // first tile plane (buffer coordinates)
    const int y0 = (int)blockIdx.x * a.ty;
    // stage the bit-rows of the tile + halo; rows outside the buffer / grid replicate the nearest row
    if (nzw >= 4) {
        // one 16-byte load per lane and staged quarter-row: (hx*hy rows) x (nzw/4 quads); all loads of a
        // workgroup are independent, so staging costs a single L2 round trip
        const int lq = lg - 2;                                // log2(quads per row)
        const int total = (hx * hy) << lq;
        for (int i0 = 0; i0 < total; i0 += 2 * BD) {
            uint4 v[2];
            int rowi[2], quad[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = min(i0 + u * BD + t, total - 1);
                rowi[u] = i >> lq; quad[u] = i & ((1 << lq) - 1);
                const int jx = (rowi[u] * a.inv_hy) >> 16, jy = rowi[u] - jx * hy;
                const int gx = min(max(x0 + jx - kBallR, 0), a.rows_x - 1);
                const int gy = min(max(y0 + jy - kBallR, 0), a.ny - 1);
                v[u] = *reinterpret_cast<const uint4*>(a.bits + ((int64_t)gx * a.ny + gy) * nzw + 4 * quad[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (i0 + u * BD + t < total) {
                    uint32_t* dst = tile + rowi[u] * rw + 1 + 4 * quad[u];
                    dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
                    if (quad[u] == 0) dst[-1] = (v[u].x & 1u) ? ~0u : 0u;                    // replicate the first voxel
                    if (quad[u] == (1 << lq) - 1) dst[4] = (v[u].w >> 31) ? ~0u : 0u;        // ... and the last one
                }
            }
        }
    } else {
        // narrow rows (nz = 32 or 64): word-wise staging, lanes laid out as (row-in-pass, word)
        const int lgp = max(lg + 1, 2);                       // 2^lgp >= nzw + 2 lanes per staged row
        const int lw = t & ((1 << lgp) - 1), lr = t >> lgp;   // word slot, row-in-pass
        const int rpp = BD >> lgp;                            // rows staged per pass
        if (lw < rwu) {
            for (int jy = lr; jy < hy; jy += rpp) {
                const int gy = min(max(y0 + jy - kBallR, 0), a.ny - 1);
                for (int jx = 0; jx < hx; ++jx) {
                    const int gx = min(max(x0 + jx - kBallR, 0), a.rows_x - 1);
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

    const int r = t >> lg, w = t & (nzw - 1);                 // tile row, word in row
    const int ty_ = r & (a.ty - 1), tx_ = r >> a.log2_ty;
    const uint32_t* c0 = tile + ((tx_ + kBallR) * hy + (ty_ + kBallR)) * rw + (w + 1);
    const uint32_t O = c0[0];
    // Levels in increasing d^2, cumulative; stop as soon as every voxel of the wave has been decided.
    // On Bernoulli(0.5) occupancy 98.4 % of the voxels have a face neighbour of the other class and all
    // but ~2^-18 are decided by d^2 <= 2, so a wave normally evaluates 18 of the 92 offsets.
    uint32_t acc[7];
    {
        uint32_t cum = 0;
        bool done = false;
        static_for<7>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if (!done) {
                cum |= ball_level_pass<l>(c0, hy, rw, O);
                done = __all(cum == ~0u);
            }
            acc[l] = cum;
        });
    }

    if (a.vb) {                                               // (block-uniform)
        int bxy = 1 << 20;
        const int gx = x0 + tx_, gy = y0 + ty_;
        if (a.nx_glob > 1) bxy = min(bxy, min(gx + 1, a.nx_glob - gx));
        if (a.ny > 1) bxy = min(bxy, min(gy + 1, a.ny - gy));
        uint32_t b1 = bxy == 1 ? ~0u : 0u, b2 = bxy <= 2 ? ~0u : 0u;
        // (nz = 32 nzw >= 32: the first / last voxel of the row are bit 0 of word 0 / bit 31 of the last word)
        if (w == 0) { b1 |= 1u; b2 |= 3u; }
        if (w == nzw - 1) { b1 |= 0x80000000u; b2 |= 0xC0000000u; }
        acc[0] |= b1; acc[1] |= b1; acc[2] |= b1;
        acc[3] |= b2; acc[4] |= b2; acc[5] |= b2; acc[6] |= b2;
    }
    // extrema (max d^2 per class) and certification, per word
    int mxF = 0, mxQ = 0;
    {
        uint32_t prevc = 0;
        static_for<7>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            {
                const uint32_t first = acc[l] & ~prevc;
                if (first & ~O) mxF = kLevelD2[l];
                if (first & O) mxQ = kLevelD2[l];
                prevc = acc[l];
            }
        });
    }
    const bool row_in_grid = (x0 + tx_ < a.out_hi) && (y0 + ty_ < a.ny);
    const bool uncert = row_in_grid && (~acc[6] != 0u);
    if (!row_in_grid) { mxF = 0; mxQ = 0; }

    // level index per voxel = number of levels it was NOT found at (0..6, 7 = not found) as 3 bit-planes
    {
        const uint32_t U0 = ~acc[0], U1 = ~acc[1], U2 = ~acc[2], U3 = ~acc[3], U4 = ~acc[4], U5 = ~acc[5], U6 = ~acc[6];
        uint4 pl;
        pl.x = (U0 & ~U1) | (U2 & ~U3) | (U4 & ~U5) | U6;
        pl.y = (U1 & ~U3) | U5;
        pl.z = U3;
        pl.w = O;
        reinterpret_cast<uint4*>(planes)[t] = pl;
    }
    if (t < 256) {
        const float lut_a = magl[((t >> 2) & 1) | ((t >> 3) & 2) | ((t >> 4) & 4)];
        const float lut_b = magl[((t >> 3) & 1) | ((t >> 4) & 2) | ((t >> 5) & 4)];
        lut2[t] = make_float2(__uint_as_float(__float_as_uint(lut_a) | ((uint32_t)(t & 1) << 31)),
                              __uint_as_float(__float_as_uint(lut_b) | ((uint32_t)(t & 2) << 30)));
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
        // ZINV: row r0 + j*rs splits into (tx, ty) without carries between the lane part r0 (< rs) and the
        // wave-uniform part j*rs, so the byte offset is a per-lane constant plus a scalar per pass
        const int ty0 = r0 & (a.ty - 1), tx0 = r0 >> a.log2_ty;
        const uint32_t lane_off = (uint32_t)(((((int)__umul24(tx0, a.ny) + ty0) << lgz) + zi) << 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {                         // fully unrolled: 8 independent LDS->LUT->store chains
            int rr, z;
            uint4 pl;
            if constexpr (ZINV) {
                rr = r0 + j * rs; z = zi;
                pl = pbase[j * (BD / 8)];                     // (rs << lg) == BD / 8 plane entries per pass
            } else {
                const int v = j * (BD * 4) + v0;              // voxel index inside the tile (row-major)
                rr = v >> lgz; z = v & (nz - 1);
                pl = planes4[(rr << lg) + (z >> 5)];
            }
            const uint32_t sh = (uint32_t)z & 31u;
            const uint32_t ia = __builtin_amdgcn_ubfe(pl.w, sh, 2u) | (__builtin_amdgcn_ubfe(pl.x, sh, 2u) << 2) |
                                (__builtin_amdgcn_ubfe(pl.y, sh, 2u) << 4) | (__builtin_amdgcn_ubfe(pl.z, sh, 2u) << 6);
            const uint32_t ib = __builtin_amdgcn_ubfe(pl.w, sh + 2u, 2u) | (__builtin_amdgcn_ubfe(pl.x, sh + 2u, 2u) << 2) |
                                (__builtin_amdgcn_ubfe(pl.y, sh + 2u, 2u) << 4) | (__builtin_amdgcn_ubfe(pl.z, sh + 2u, 2u) << 6);
            const float2 fa = lut2[ia], fb = lut2[ib];
            const int tyy = rr & (a.ty - 1), txx = rr >> a.log2_ty;
            if (FULL || (x0 + txx < a.out_hi && y0 + tyy < a.ny)) {
                f4v ov;
                ov.x = fa.x; ov.y = fa.y; ov.z = fb.x; ov.w = fb.y;
                f4v* dst;
                if constexpr (ZINV) {
                    const int rj = j * rs;                                                    // wave-uniform
                    const int64_t uoff = (((int64_t)(rj >> a.log2_ty) * a.ny + (rj & (a.ty - 1))) << lgz) << 2;
                    dst = reinterpret_cast<f4v*>(tile_out + uoff + lane_off);
                } else {
                    dst = reinterpret_cast<f4v*>(tile_out + (uint32_t)((((int)__umul24(txx, a.ny) + tyy) << lgz) + z) * 4u);
                }
                if (a.nt_store) __builtin_nontemporal_store(ov, dst);
                else *dst = ov;
            }
        }
    };
    if (!(a.checked & 1) && (x0 + a.tx <= a.out_hi) && (y0 + a.ty <= a.ny)) expand(std::true_type{});
    else expand(std::false_type{});

#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mxF = max(mxF, __shfl_xor(mxF, off));
        mxQ = max(mxQ, __shfl_xor(mxQ, off));
    }
    const bool any_uncert = __any(uncert);
    const bool all_undecided = __all(row_in_grid && acc[6] == 0u);
    if (any_uncert && a.unc) {
        if (row_in_grid)
            a.unc[((int64_t)(x0 + tx_ - a.out_lo) * a.ny + (y0 + ty_)) * nzw + w] = ~acc[6];
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
        if (any_uncert) {
            if (a.unc) {
                atomicOr(a.tileflag + tile_id, 1u << (t >> 6));      // one word per tile: no same-address pile-up
                raise_flag(a.fix_needed);
                // a wave without a single decided voxel sits in empty (or solid) space: nothing for the fix-up kernel
                if (a.early_out && all_undecided) { raise_flag(a.uncertified); note_reason(a.reason, kGiveUpWaveAllUndecided); }
            } else {
                raise_flag(a.uncertified);
                note_reason(a.reason, kGiveUpBeyondBall);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// KF: fix-up behind KD for "almost dense" scenes.
//
// At Bernoulli p = 0.1 the ball of d^2 <= 8 leaves 0.9^92 = 6e-5 of the free voxels undecided -- 7 000 of 134 M
// -- and recomputing the whole grid with the general sweeps for them costs 5x the dense kernels.  KF visits only
// those voxels: the tiles whose flag word KD set stage their bit rows again with a halo of kFixR planes / rows, the
// undecided voxels of the tile are compacted into an LDS list, and one wave per voxel scans the (2 kFixR + 1)^2 (dx, dy) rows
// (3 per lane), taking the nearest opposite bit within |dz| <= kFixR of each row with two bit scans, and
// min-reduces the candidates.  A candidate <= kFixR^2 is the exact squared
// distance (every offset that could beat it lies inside the scanned cube; rows / bits beyond the grid replicate
// the nearest in-grid voxel exactly as in KD); anything else -- or a tile with more than kFixCap undecided
// voxels (p = 0.03 leaves 6 % undecided: KF would take 2.5 ms where the sweeps take 1.1) -- raises `uncertified` and the guarded general sweeps redo the grid, so results are exact for any input.
// ---------------------------------------------------------------------------------------------
constexpr int kFixR = 16;                                     // (round 3: 6 -> 8.  At 512^3 the LARGEST distance of a Bernoulli p = 0.03 / 0.04
                                                              //  scene is d^2 = 40 .. 41 on two seeds of three: one voxel beyond 36 sent the whole
                                                              //  grid to the sweeps.  The rows go past sorted by dx^2 + dy^2 with an early exit, so
                                                              //  the voxels that never needed the outer rows do not pay for them.  Round 5: 8 -> 16,
                                                              //  d^2 <= 256.  With the shell pass KD6 in front KF sees only what lies beyond d^2 = 36
                                                              //  -- at p = 0.015 a few hundred voxels, most of them at the grid's faces and corners,
                                                              //  where a voxel sees a fraction of the ball and the largest distances of a noise
                                                              //  scene sit: d^2 = 51 .. 77 at 512^3 -- and ONE voxel beyond its reach sends the
                                                              //  whole grid to the far-field pair ("beyond_fixup_reach" ended the tier at p = 0.015))
constexpr int kFixCap = 512;                                  // undecided voxels a tile may hand to KF (round 3: 160 -> 512 with the
                                                              // early exit below: Bernoulli p = 0.05 leaves 0.9 % undecided -- ~280 per
                                                              // tile -- and took the marching sweeps at 0.84 ms instead; p = 0.03 leaves
                                                              // 6 %, ~2000 per tile: there the sweeps are cheaper and the cap sends it on)
constexpr int kFixDirect = 24;                                // tiles with at most this many undecided voxels read the bit field directly
constexpr int kFixTileR = 6;                                  // halo of the staged LDS tile; the rows beyond it (dx^2 + dy^2 >= 49: the last
                                                              // two batches, reached by the few voxels still open then) come from the L2-resident
                                                              // bit field -- a halo of 8 made the staging of every tile 1.56x larger (p = 0.05:
                                                              // 0.46 -> 0.58 ms per build)
constexpr int kFixRows = (2 * kFixR + 1) * (2 * kFixR + 1);   // 1089 (dx, dy) rows
constexpr int kFixRA = 8;                                     // phase A of a voxel's scan: the rows with dx^2 + dy^2 <= 64, bit windows of +-8 -- round 4's
                                                              // reach, and all that nearly every voxel ever needs; phase B (rows up to 256, windows of
                                                              // +-16) only for a voxel that is still open behind it
__host__ __device__ constexpr int fix_rows_within(int r2max) {
    int n = 0;
    for (int dx = -kFixR; dx <= kFixR; ++dx)
        for (int dy = -kFixR; dy <= kFixR; ++dy) n += (dx * dx + dy * dy <= r2max) ? 1 : 0;
    return n;
}
constexpr int kFixRowsA = fix_rows_within(kFixRA * kFixRA);  // 197: a prefix of the table (sorted by dx^2 + dy^2)
constexpr int kFixRowsB = fix_rows_within(kFixR * kFixR);    // 797
constexpr int kFixNearRows = 320;                             // rows whose table entries are copied into LDS (sorted by dx^2 + dy^2: everything inside
                                                              // the staged halo, dx^2 + dy^2 <= 72, and what round 4's reach of 8 covered); the rows
                                                              // behind them are the rare far voxel's and come from the table in global memory
constexpr int kFixOrderPad = (kFixNearRows + 7) & ~7;         // LDS words reserved for the row table
constexpr int kFixOutside = 0x40000000;                       // rowofs[] marker: the row lies beyond the staged halo

struct FixArgs {
    const uint32_t* bits;   // [rows_x][ny][nzw]
    float* out;
    const uint32_t* unc;
    uint32_t* tileflag;
    const uint32_t* fix_needed;   // guard
    const uint32_t* order;  // [kFixRows] (dx + R) | (dy + R) << 8 | (dx^2 + dy^2) << 16, sorted by the last
    int nzw, log2_nzw, ny, rows_x, out_lo, out_hi, tx, ty, log2_ty;
    double resolution;
    uint32_t* slots;
    uint32_t* uncertified;
    uint32_t* reason;       // see DenseArgs
};

template <int BD>
__global__ __launch_bounds__(BD) void k_ball_fixup(const FixArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fix_smem[];
    if (*a.fix_needed == 0u) return;
    const uint32_t tile_id = (uint32_t)blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t flags = a.tileflag[tile_id] & 0xFFFFu;     // wave-uniform (bits 16 ..: KD3's count of open words, for the shell pass)
    if (flags == 0u) return;
    const int t = threadIdx.x;
    const int nzw = a.nzw, lg = a.log2_nzw;
    const int rw = nzw + 2;                                   // one replicated edge word on each side
    const int hx = a.tx + 2 * kFixTileR, hy = a.ty + 2 * kFixTileR;
    uint32_t* order = reinterpret_cast<uint32_t*>(fix_smem);  // [kFixRows] (+ pad)
    int* rowofs = reinterpret_cast<int*>(order + kFixOrderPad);   // [kFixRows] (+ pad): LDS word offset of row (dx, dy) from the voxel's own
                                                              // row, or kFixOutside for rows beyond the staged halo (round 4: the scan loop
                                                              // decoded dx, dy and multiplied them out for every row of every voxel)
    uint32_t* list = reinterpret_cast<uint32_t*>(rowofs + kFixOrderPad);   // [kFixCap] (tile row << 16) | z
    uint32_t* count = list + kFixCap;                         // [1] (+ pad to 4 words)
    uint32_t* tile = count + 4;                               // [hx][hy][rw]
    // "the general sweeps will redo the grid anyway": ONE lane reads the flag for the whole workgroup (count[1]) -- read per
    // wave, two waves could see it on either side of its rise, wave 0 would leave without clearing the list counter and the
    // others would walk a garbage list.  (The flag words stay zero between builds: cleared either way.)
    if (t == 0) {
        *count = 0u;
        count[1] = __hip_atomic_load(a.uncertified, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = t; i < kFixNearRows; i += BD) {
        const uint32_t o = a.order[i];
        order[i] = o;
        const int dx = (int)(o & 0xffu) - kFixR, dy = (int)((o >> 8) & 0xffu) - kFixR;
        rowofs[i] = (dx >= -kFixTileR && dx <= kFixTileR && dy >= -kFixTileR && dy <= kFixTileR) ? (dx * hy + dy) * rw : kFixOutside;
    }
    const int x0 = a.out_lo + (int)blockIdx.y * a.tx, y0 = (int)blockIdx.x * a.ty;
    __syncthreads();
    if (t == 0) a.tileflag[tile_id] = 0u;                     // (behind the barrier: every wave has read its copy of the word)
    if (count[1] != 0u) return;                               // (block-uniform)
    // this lane's word (same mapping as KD) -> list of its undecided voxels
    {
        const int r = t >> lg, w = t & (nzw - 1);
        const int ty_ = r & (a.ty - 1), tx_ = r >> a.log2_ty;
        const bool row_in_grid = (x0 + tx_ < a.out_hi) && (y0 + ty_ < a.ny);
        uint32_t u = 0;
        if (row_in_grid && ((flags >> (t >> 6)) & 1u))
            u = a.unc[((int64_t)(x0 + tx_ - a.out_lo) * a.ny + (y0 + ty_)) * nzw + w];
        while (u) {
            const int b = __builtin_ctz(u);
            u &= u - 1;
            const uint32_t slot = atomicAdd(count, 1u);
            if (slot < (uint32_t)kFixCap) list[slot] = ((uint32_t)r << 16) | (uint32_t)(w * 32 + b);
        }
    }
    __syncthreads();
    const uint32_t n = *count;
    if (n > (uint32_t)kFixCap) {                              // too many: let the general sweeps do the whole grid
        if (t == 0) { raise_flag(a.uncertified); note_reason(a.reason, kGiveUpTileOverCap); }
        return;
    }
    // A tile with a handful of undecided voxels (Bernoulli p = 0.1: two per tile, in nearly every tile) reads the rows it
    // needs straight from the L2-resident bit field -- 64 .. 289 rows of 3 words per voxel -- instead of staging the whole
    // halo tile in LDS ((tx + 12) x (ty + 12) rows for two voxels was most of this kernel's time there).
    const bool direct = n <= (uint32_t)kFixDirect;            // (block-uniform)
    if (!direct) {
        // rows / edge words replicate the nearest in-grid voxel.  Round 4: a lane per WORD of the bit row (nzw lanes per row,
        // BD / nzw rows per pass, every lane busy; the two replicated edge words are derived by the lanes that hold the row's
        // first / last word) and the row -> (jx, jy) split by a multiply-shift -- the first form spent 14 of every 32 lanes on
        // nothing and two integer divisions per element: 40 % of this kernel's instructions at Bernoulli p = 0.02.
        const int j = t & (nzw - 1), rp = t >> lg, rpp = BD >> lg;
        const uint32_t inv_hy = ((1u << 20) + (uint32_t)hy - 1u) / (uint32_t)hy;      // (exact for rows < 2^20 / hy)
        const int nrows = hx * hy;
        for (int row0 = rp; row0 < nrows; row0 += 4 * rpp) {                          // 4 independent loads in flight per lane
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = min(row0 + k * rpp, nrows - 1);
                const int jx = (int)(((uint32_t)row * inv_hy) >> 20), jy = row - jx * hy;
                const int gx = min(max(x0 + jx - kFixTileR, 0), a.rows_x - 1);
                const int gy = min(max(y0 + jy - kFixTileR, 0), a.ny - 1);
                v[k] = a.bits[((int64_t)gx * a.ny + gy) * nzw + j];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = row0 + k * rpp;
                if (row < nrows) {
                    uint32_t* dst = tile + row * rw;
                    dst[j + 1] = v[k];
                    if (j == 0) dst[0] = (v[k] & 1u) ? ~0u : 0u;
                    if (j == nzw - 1) dst[rw - 1] = (v[k] >> 31) ? ~0u : 0u;
                }
            }
        }
        __syncthreads();
    }
    int mxF = 0, mxQ = 0;
    bool failed = false;
    const int nz = nzw << 5;
    // one 16-lane ROW per voxel (4 voxels per wave): the 289 (dx, dy) rows, sorted by dx^2 + dy^2, go past in batches of
    // 64 (4 per lane, no dependence between them), the candidates are min-reduced over the DPP row, and a voxel stops as
    // soon as its best candidate is no larger than the smallest in-plane offset still to come -- the usual case after the
    // first batch (it covers dx^2 + dy^2 <= 18; at p = 0.05 the nearest opposite voxel lies at d^2 ~ 9 .. 14).
    // (Round 2 ran one WAVE per voxel without the early exit: right for the handful of voxels per tile of p = 0.1, 0.4 us
    // per voxel at the ~280 per tile of p = 0.05.)
    auto rowmin = [](int v) -> int {
        v = min(v, __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true));      // row_ror:8
        v = min(v, __builtin_amdgcn_mov_dpp(v, 0x124, 0xF, 0xF, true));      // row_ror:4
        v = min(v, __builtin_amdgcn_mov_dpp(v, 0x122, 0xF, 0xF, true));      // row_ror:2
        v = min(v, __builtin_amdgcn_mov_dpp(v, 0x121, 0xF, 0xF, true));      // row_ror:1
        return v;
    };
    const int gl = t & 15, grp = t >> 4;
    for (uint32_t i0 = 0; i0 < n; i0 += BD / 16) {             // (uniform trip count: every lane takes part in the reductions)
        const uint32_t i = i0 + (uint32_t)grp;
        const bool live = i < n;
        const uint32_t e = list[live ? i : 0u];
        const int r = (int)(e >> 16), z = (int)(e & 0xffffu);
        const int ty_ = r & (a.ty - 1), tx_ = r >> a.log2_ty;
        const int w = z >> 5, b = z & 31;
        const uint32_t* c0 = tile + ((tx_ + kFixTileR) * hy + (ty_ + kFixTileR)) * rw + (w + 1);
        // the three words around bit b of row (dx, dy); rows / edge words beyond the grid replicate the nearest in-grid voxel
        auto words = [&](int dx, int dy, uint32_t& prev, uint32_t& cur, uint32_t& next) {
            if (!direct && dx >= -kFixTileR && dx <= kFixTileR && dy >= -kFixTileR && dy <= kFixTileR) {
                const uint32_t* p = c0 + (dx * hy + dy) * rw;
                prev = p[-1]; cur = p[0]; next = p[1];
            } else {
                const int gx = min(max(x0 + tx_ + dx, 0), a.rows_x - 1), gy = min(max(y0 + ty_ + dy, 0), a.ny - 1);
                const uint32_t* src = a.bits + ((int64_t)gx * a.ny + gy) * nzw;
                cur = src[w];
                prev = w > 0 ? src[w - 1] : ((src[0] & 1u) ? ~0u : 0u);
                next = w + 1 < nzw ? src[w + 1] : ((src[nzw - 1] >> 31) ? ~0u : 0u);
            }
        };
        uint32_t cw0, cw1, cw2;
        words(0, 0, cw0, cw1, cw2);
        const uint32_t cls = (cw1 >> b) & 1u;
        const uint32_t flip = cls ? ~0u : 0u;                 // after the XOR a set bit = voxel of the OTHER class
        // One phase of the scan: the first `nrows` rows of the table (a prefix: everything with dx^2 + dy^2 <= R^2), the bits around z
        // of a row as two funnel shifts (v_alignbit_b32) whose shift counts depend on the voxel only:
        //   up: bit dz of (next : cur) >> b            = voxel z + dz,      dz = 0 .. R
        //   dn: bit i  of (cur : prev) >> (32 + b - R) = voxel z - R + i,   i = 0 .. R - 1
        // A candidate <= R^2 is then the exact squared distance: every offset that could beat it has dx^2 + dy^2 <= R^2 and |dz| <= R.
        auto scan = [&](auto rc, int nrows, bool want) -> int {
            constexpr int R = decltype(rc)::value;
            const bool dn_in_cur = b >= R;                        // ... all inside `cur`: (0 : cur) >> (b - R)
            const bool up_past_cur = b + R > 31;                  // the upward window reaches into `next`
            const uint32_t dn_sh = (uint32_t)(b - R) & 31u;
            int best = 1 << 20;
            bool done = !want;
#pragma unroll 1
            for (int k0 = 0; k0 < nrows; k0 += 64) {
                if (k0 > 0) {
                    done = done || rowmin(best) <= (int)((k0 < kFixNearRows ? order[k0] : a.order[k0]) >> 16);
                    if (__all(done)) break;
                }
                if (!done) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int k = k0 + gl + 16 * m;
                        if (k < nrows) {
                            const uint32_t o = k < kFixNearRows ? order[k] : a.order[k];
                            const int d2 = (int)(o >> 16), ofs = k < kFixNearRows ? rowofs[k] : kFixOutside;
                            uint32_t prev = 0u, cur, next = 0u;
                            if (!direct && ofs != kFixOutside) {
                                // (the neighbour words only where this voxel's bit window leaves `cur`: half of the voxels need neither,
                                //  and every LDS access spared is a bank conflict spared -- the rows of a group's 16 lanes are scattered)
                                const uint32_t* p = c0 + ofs;
                                cur = p[0];
                                if (!dn_in_cur) prev = p[-1];
                                if (up_past_cur) next = p[1];
                            } else {
                                words((int)(o & 0xffu) - kFixR, (int)((o >> 8) & 0xffu) - kFixR, prev, cur, next);
                            }
                            prev ^= flip; cur ^= flip; next ^= flip;
                            // (no hit: the forced top bit / clz(0) = 32 give dz = 31 / R + 1, i.e. a candidate above R^2 that the
                            //  final test rejects; it cannot end the scan early before every row that could hold a real hit has gone past)
                            const uint32_t up = (__builtin_amdgcn_alignbit(next, cur, (uint32_t)b) & ((2u << R) - 1u)) | 0x80000000u;
                            const int dzu = __builtin_ctz(up);
                            best = min(best, d2 + dzu * dzu);
                            const uint32_t dn = __builtin_amdgcn_alignbit(dn_in_cur ? 0u : cur, dn_in_cur ? cur : prev, dn_sh) & ((1u << R) - 1u);
                            const int dzd = R - 31 + __clz((int)dn);
                            best = min(best, d2 + dzd * dzd);
                        }
                    }
                }
            }
            return rowmin(best);
        };
        int best = scan(std::integral_constant<int, kFixRA>{}, kFixRowsA, live);
        // phase B, for the rare voxel beyond d^2 = 64 (the faces and corners of a sparse noise grid): the whole table, wide windows
        // (its result REPLACES phase A's: a phase's value above R^2 may be the "no hit" sentinel (R + 1)^2 of a bit window, not a candidate)
        if (__any(live && best > kFixRA * kFixRA)) {
            const bool open_ = live && best > kFixRA * kFixRA;
            const int wide = scan(std::integral_constant<int, kFixR>{}, kFixRowsB, open_);
            if (open_) best = wide;
        }
        if (live) {
            if (best <= kFixR * kFixR) {
                if (gl == 0) {
                    const float f = (float)(sqrt((double)best) * a.resolution);
                    const int gx = x0 + tx_, gy = y0 + ty_;
                    a.out[((int64_t)(gx - a.out_lo) * a.ny + gy) * nz + z] = cls ? -f : f;
                }
                if (cls) mxQ = max(mxQ, best); else mxF = max(mxF, best);
            } else {
                failed = true;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mxF = max(mxF, __shfl_xor(mxF, off));
        mxQ = max(mxQ, __shfl_xor(mxQ, off));
    }
    const bool any_failed = __any(failed);
    if ((t & 63) == 0) {
        slot_max2(a.slots, tile_id * (BD / 64) + ((uint32_t)t >> 6), mxF, mxQ);
        if (any_failed) { raise_flag(a.uncertified); note_reason(a.reason, kGiveUpBeyondReach); }
    }
}


// ---------------------------------------------------------------------------------------------
// Generic forms of K0 / KD for the shapes the tuned kernels above do not take (round 2): any nz (rows are padded
// to whole 32-bit words: nzw = ceil(nz / 32), the bits past the row end replicate the row's last voxel) and
// add_virtual_border.  Same algorithm, same exactness contract (a voxel without a hit raises *uncertified), no
// power-of-two index arithmetic and no LDS tile: a lane owns one word and reads the 25 x 3 neighbour words of the
// ball straight from the (L2-resident) bit field.  These are the reference's own grid shapes -- 100 x 100 x 50
// (src/compute_convex_segments_test.cpp:13-41), 40^3 (src/sdf_tools_tutorial.cpp:23-59), 25 x 20 x 15
// (scripts/3d_sdf_demo_rviz.py:107-111) -- and every add_virtual_border = true call (sdf_generation.hpp:287-419).
//
// Virtual border inside the ball: the padded layer is at axis distance b = min over axes with n > 1 of
// min(i + 1, n - i); D <- min(D, b^2) only binds for b^2 in {1, 4} (9 > 8), i.e. level 0 for the outermost layer
// of voxels and level 3 for the second one; a voxel with b <= 2 is therefore always certified.
// ---------------------------------------------------------------------------------------------
template <class Loader>
__global__ __launch_bounds__(kBlock) void k_pack_bits_rows(Loader ld, uint32_t* __restrict__ bits, int64_t nrows, int nz, int nzw) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;       // word index
    if (i >= nrows * nzw) return;
    const int64_t row = i / nzw;
    const int w = (int)(i - row * nzw);
    const int z0 = 32 * w, nv = min(32, nz - z0);
    uint32_t word = 0;
    bool last = false;
    for (int k = 0; k < nv; ++k) {
        last = ld.filled(row * nz + z0 + k);
        word |= (last ? 1u : 0u) << k;
    }
    if (nv < 32 && last) word |= ~0u << nv;                             // replicate the row's last voxel
    bits[i] = word;
}

struct DenseGenArgs {
    const uint32_t* bits;   // [nx][ny][nzw]
    float* out;             // [nx][ny][nz]
    int nx, ny, nz, nzw;
    int vb;
    float mag[8];           // level magnitudes as in DenseArgs
    uint32_t* slots;
    uint32_t* uncertified;
};

SDFGPU_KERNEL __launch_bounds__(kBlock) void k_ball_dense_generic(const DenseGenArgs a) {
    __shared__ float lut[16];                                           // [class << 3 | level] -> signed magnitude
    if (threadIdx.x < 16) {
        const int l = threadIdx.x & 7;
        float m = a.mag[0];
        m = l == 1 ? a.mag[1] : m; m = l == 2 ? a.mag[2] : m; m = l == 3 ? a.mag[3] : m;
        m = l == 4 ? a.mag[4] : m; m = l == 5 ? a.mag[5] : m; m = l == 6 ? a.mag[6] : m;
        m = l == 7 ? 0.0f : m;
        lut[threadIdx.x] = (threadIdx.x & 8) ? -m : m;
    }
    __syncthreads();
    const int64_t nwords = (int64_t)a.nx * a.ny * a.nzw;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < nwords;
    const int64_t ii = live ? i : nwords - 1;
    const int64_t row = ii / a.nzw;
    const int w = (int)(ii - row * a.nzw);
    const int x = (int)(row / a.ny), y = (int)(row - (int64_t)x * a.ny);
    const uint32_t* rowp = a.bits + row * a.nzw;
    const uint32_t O = rowp[w];
    uint32_t acc[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int dx = -kBallR; dx <= kBallR; ++dx) {
#pragma unroll
        for (int dy = -kBallR; dy <= kBallR; ++dy) {
            const int gx = min(max(x + dx, 0), a.nx - 1), gy = min(max(y + dy, 0), a.ny - 1);
            const uint32_t* p = a.bits + ((int64_t)gx * a.ny + gy) * a.nzw;
            const uint32_t cur = p[w];
            const uint32_t prev = w > 0 ? p[w - 1] : ((cur & 1u) ? ~0u : 0u);                 // replicate the first voxel
            const uint32_t next = w + 1 < a.nzw ? p[w + 1] : ((cur >> 31) ? ~0u : 0u);       // ... and the (replicated) last bit
#pragma unroll
            for (int dz = -kBallR; dz <= kBallR; ++dz) {
                const int d2 = dx * dx + dy * dy + dz * dz;
                const int lv = ball_level(d2);
                if (lv < 0) continue;
                const uint32_t S = dz == 0 ? cur
                                 : dz > 0 ? __builtin_amdgcn_alignbit(next, cur, dz)
                                          : __builtin_amdgcn_alignbit(cur, prev, 32 + dz);
                acc[lv] |= O ^ S;
            }
        }
    }
#pragma unroll
    for (int l = 1; l < 7; ++l) acc[l] |= acc[l - 1];                   // cumulative: found at level <= l
    const int z0 = 32 * w, nv = min(32, a.nz - z0);
    const uint32_t valid = nv >= 32 ? ~0u : ((1u << nv) - 1u);
    if (a.vb) {
        // voxels with b == 1 are found at level 0, with b == 2 at level 3 (D <- min(D, b^2))
        int bxy = 1 << 20;
        if (a.nx > 1) bxy = min(bxy, min(x + 1, a.nx - x));
        if (a.ny > 1) bxy = min(bxy, min(y + 1, a.ny - y));
        uint32_t b1 = bxy == 1 ? ~0u : 0u, b2 = bxy <= 2 ? ~0u : 0u;
        if (a.nz > 1) {
            // bits of this word whose z distance to the padded layer is 1 / <= 2
            uint32_t z1 = 0u, z2 = 0u;
            for (int k = 0; k < nv; ++k) {
                const int z = z0 + k, bz = min(z + 1, a.nz - z);
                z1 |= (bz == 1 ? 1u : 0u) << k;
                z2 |= (bz <= 2 ? 1u : 0u) << k;
            }
            b1 |= z1; b2 |= z2;
        }
        acc[0] |= b1; acc[1] |= b1; acc[2] |= b1;
        acc[3] |= b2 | b1; acc[4] |= b2 | b1; acc[5] |= b2 | b1; acc[6] |= b2 | b1;
    }
    int mxF = 0, mxQ = 0;
    {
        uint32_t prevc = 0u;
#pragma unroll
        for (int l = 0; l < 7; ++l) {
            const uint32_t first = acc[l] & ~prevc & valid;
            if (first & ~O) mxF = kLevelD2[l];
            if (first & O) mxQ = kLevelD2[l];
            prevc = acc[l];
        }
    }
    const bool uncert = live && ((~acc[6] & valid) != 0u);
    if (!live) { mxF = 0; mxQ = 0; }
    if (live) {
        // level index = number of levels a voxel was NOT found at (7 = not found: +-0, rewritten by the general pipeline)
        float* dst = a.out + row * a.nz + z0;
        for (int k = 0; k < nv; ++k) {
            int lvl = 0;
#pragma unroll
            for (int l = 0; l < 7; ++l) lvl += ((acc[l] >> k) & 1u) ? 0 : 1;
            dst[k] = lut[(((O >> k) & 1u) << 3) | (uint32_t)lvl];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mxF = max(mxF, __shfl_xor(mxF, off));
        mxQ = max(mxQ, __shfl_xor(mxQ, off));
    }
    const bool any_uncert = __any(uncert);
    if ((threadIdx.x & 63) == 0) {
        slot_max2(a.slots, blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), mxF, mxQ);
        if (any_uncert) raise_flag(a.uncertified);
    }
}

}  // namespace sdfgpu
