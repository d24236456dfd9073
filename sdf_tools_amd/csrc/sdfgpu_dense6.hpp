// sdfgpu_dense6.hpp -- KD6: the bit-parallel SHELL pass of the dense tier's fix-up stage (round 5).
//
// KD3 (sdfgpu_dense3.hpp) decides every voxel with d^2 <= 14; what it leaves -- Bernoulli p = 0.02: 0.98^250 = 0.6 % of the voxels,
// p = 0.015: 2.3 %, p = 0.01: 8 % -- went to KF (sdfgpu_dense.hpp), which takes ONE voxel per 16-lane row and scans (dx, dy)
// rows around it: ~4 M voxels per ms, the whole cost of the p = 0.02 build (0.23 of 0.45 ms) and the reason the tier ended at
// p ~ 0.018 (below: the far-field pair, 0.96 ms -- a 2.1x step in the density sweep).  KD6 sits between the two and stays in
// the bit domain: a lane takes one 32-voxel word that still holds undecided voxels and evaluates the SHELL of complete levels
// behind KD3's cube -- every lattice offset with 16 <= d^2 <= 36 (15 is not a sum of three squares; all components <= 6),
// 668 offsets in 18 levels -- exactly like KD / KD3 do for theirs: one funnel shift (v_alignbit_b32) and one
// acc[level] |= own ^ shifted (v_bitop3_b32) per offset and 32 voxels, for both classes at once.  The first level with a hit is
// the exact squared distance (every smaller offset has been checked: d^2 <= 14 by KD3, the levels in between here).  Words without
// undecided voxels are compacted away first (p = 0.02: 18 % of the words are active), the shell is evaluated in two halves
// (d^2 <= 24, then <= 36) with a wave-uniform stop in between, and what is still open after d^2 = 36 -- 0.98^925 = 8e-9 of the
// voxels at p = 0.02, 9e-5 at p = 0.01 -- stays in the undecided words for KF (exact up to d^2 = 64) exactly as before.
//
// Rows / bits beyond the grid replicate the nearest in-grid voxel, as in KD: a replicated voxel is a real voxel at a
// component-wise smaller-or-equal offset, which is itself enumerated at a level that is not later -- it cannot create a false
// hit.  Exactness never depends on this kernel: a voxel it cannot decide goes on to KF, and what KF cannot decide raises
// `uncertified` for the guarded general pipeline.
#pragma once
#include "sdfgpu_dense3.hpp"

namespace sdfgpu {

constexpr int kShellR = 6;                                    // |dx|, |dy|, |dz| <= 6
constexpr int kShellLevels = 18;
__host__ __device__ constexpr int shell_level(int d2) {      // d^2 -> level index 0 .. 17, -1 = not a level of the shell
    constexpr int tab[37] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                             0, 1, 2, 3, 4, 5, 6, -1, 7, 8, 9, 10, -1, 11, 12, -1, 13, 14, 15, 16, 17};
    return (d2 >= 0 && d2 <= 36) ? tab[d2] : -1;
}
__device__ constexpr int kShellD2[kShellLevels] = {16, 17, 18, 19, 20, 21, 22, 24, 25, 26, 27, 29, 30, 32, 33, 34, 35, 36};
constexpr int kShellSplit = 8;                                // levels 0 .. 7 (d^2 <= 24) first, 8 .. 17 only if somebody is still open

struct ShellArgs {
    const uint32_t* bits;   // [rows_x][ny][nzw]
    float* out;
    uint32_t* unc;          // undecided words written by KD3 (only by waves that set their bit in the tile's flag word): updated in place
    const uint32_t* tileflag;
    const uint32_t* fix_needed;   // guard
    const uint32_t* uncertified;  // set: the general pipeline redoes the grid anyway (k_shell_budget raises it for scenes beyond the budget)
    int nzw, log2_nzw, ny, rows_x, out_lo, out_hi, tx, ty, log2_ty;
    double resolution;      // a level's magnitude is float(sqrt(double(d^2)) * resolution), the reference's arithmetic (sdf_generation.hpp:254-265);
                            // d^2 is a compile-time constant per level, so its correctly rounded square root is folded by the compiler
    uint32_t* slots;
    int min_words;          // active words of a group below which it is left to KF
};

// The (dx, dy) rows of the shell, grouped by r2 = dx^2 + dy^2: which dz of a row belong to the shell, and to which level, depends
// on r2 alone, so the rows of a group share ONE piece of code (a run-time loop over the group's rows around a compile-time set
// of funnel shifts) and the level accumulators keep compile-time indices.  (The first version unrolled all 668 offsets: ~36 KB
// of straight-line code that every wave streamed through the instruction cache once, with a dozen waves per CU at different
// places in it -- 58 M VALU wave instructions took 1.26 ms at Bernoulli p = 0.02 where KD3 issues 89 M in 0.21 ms.)
constexpr int kShellR2Count = 19;
__host__ __device__ constexpr int shell_r2(int g) {
    constexpr int tab[kShellR2Count] = {0, 1, 2, 4, 5, 8, 9, 10, 13, 16, 17, 18, 20, 25, 26, 29, 32, 34, 36};
    return tab[g];
}
__host__ __device__ constexpr int shell_r2_group(int r2) {   // -1: no row with this dx^2 + dy^2 inside the shell's square
    for (int g = 0; g < kShellR2Count; ++g) if (shell_r2(g) == r2) return g;
    return -1;
}
__host__ __device__ constexpr int shell_group_rows(int g) {  // rows (dx, dy), |dx|, |dy| <= 6, of group g
    int n = 0;
    for (int dx = -kShellR; dx <= kShellR; ++dx)
        for (int dy = -kShellR; dy <= kShellR; ++dy) n += (dx * dx + dy * dy == shell_r2(g)) ? 1 : 0;
    return n;
}
__host__ __device__ constexpr int shell_group_start(int g) { // first slot of group g in the row table
    int n = 0;
    for (int k = 0; k < g; ++k) n += shell_group_rows(k);
    return n;
}
constexpr int kShellRows = shell_group_start(kShellR2Count);  // 113
static_assert(kShellRows == 113, "rows of the shell");
// slot of row (dx, dy) in the table (rows of a group in enumeration order), -1 = outside the shell
__host__ __device__ constexpr int shell_row_slot(int dx, int dy) {
    const int g = shell_r2_group(dx * dx + dy * dy);
    if (g < 0) return -1;
    int rank = 0;
    for (int ex = -kShellR; ex <= kShellR; ++ex)
        for (int ey = -kShellR; ey <= kShellR; ++ey) {
            if (ex == dx && ey == dy) return shell_group_start(g) + rank;
            if (ex * ex + ey * ey == shell_r2(g)) ++rank;
        }
    return -1;
}

// acc[l] |= (hits of every offset of level l, LO <= l < HI) for the 32 voxels of word O; rowofs = LDS word offsets of the rows
template <int LO, int HI>
__device__ __forceinline__ void shell_pass(const uint32_t* c0, const int* rowofs, uint32_t O, uint32_t (&acc)[kShellLevels]) {
    static_for<kShellR2Count>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int r2 = shell_r2(g);
        constexpr bool pos = [] { for (int dz = 1; dz <= kShellR; ++dz) { const int l = shell_level(r2 + dz * dz); if (l >= LO && l < HI) return true; } return false; }();
        constexpr bool zero = shell_level(r2) >= LO && shell_level(r2) < HI;
        if constexpr (pos || zero) {
            constexpr int k0 = shell_group_start(g), k1 = k0 + shell_group_rows(g);
#pragma unroll 1
            for (int k = k0; k < k1; ++k) {
                const uint32_t* p = c0 + rowofs[k];
                const uint32_t cur = p[0];
                uint32_t prev = 0u, next = 0u;
                if constexpr (pos) { prev = p[-1]; next = p[1]; }          // (dz and -dz share a level)
                static_for<2 * kShellR + 1>([&](auto zc) {
                    constexpr int dz = decltype(zc)::value - kShellR;
                    constexpr int l = shell_level(r2 + dz * dz);
                    if constexpr (l >= LO && l < HI) {
                        const uint32_t S = dz == 0 ? cur
                                         : dz > 0 ? __builtin_amdgcn_alignbit(next, cur, dz)
                                                  : __builtin_amdgcn_alignbit(cur, prev, 32 + dz);
                        acc[l] = __builtin_amdgcn_bitop3_b32(acc[l], O, S, 0xF6);            // acc | (O ^ S)
                    }
                });
            }
        }
    });
}

// The budget of the stage behind KD3 (one workgroup, between KD3 and KD6): KD3's sampled waves left their undecided voxels in word 2
// of the extrema slots (a 1 / 16 sample of the scene).  Beyond the budget -- a fifth of the voxels undecided behind KD3, Bernoulli
// p ~ 0.006: KD6 then turns every word around and KF still gets 0.4 % of the voxels -- the far-field pair is the cheaper tool:
// `uncertified` goes up, KD6 and KF return on it and the guarded general pipeline redoes the grid.  Clears the sample words.
SDFGPU_KERNEL __launch_bounds__(kSlots) void k_shell_budget(uint32_t* __restrict__ slots, const uint32_t* __restrict__ fix_needed,
                                                         uint32_t* __restrict__ uncertified, uint32_t* __restrict__ reason, uint32_t budget) {
    __shared__ uint32_t part[kSlots / 64];
    uint32_t* p = slots + (size_t)threadIdx.x * kSlotWords + 2;
    uint32_t v = *p;
    if (v) *p = 0u;
    if (*fix_needed == 0u) return;                          // (block-uniform; nothing undecided anywhere: the words were zero)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < kSlots / 64; ++k) tot += part[k];
        if (tot > budget) { raise_flag(uncertified); note_reason(reason, kGiveUpTooSparse); }
    }
}

// One workgroup takes kShellGroup consecutive tiles of KD3 (along y): the halo of 6 rows is then staged once for the group
// ((tx + 12) x (4 ty + 12) rows instead of 4 x (tx + 12) x (ty + 12): 2.3x fewer for 4 x 4-row tiles), and the group's active
// words fill whole waves even where a single tile holds a dozen (p = 0.03: 13 per tile).  The first version -- one tile per
// workgroup -- ran ONE wave per tile behind a 256-row staging round with three workgroups per CU: 0.93 ms at p = 0.03, every LDS
// latency of its 339 row reads exposed (63 M VALU wave instructions at a sixth of KD3's rate).
constexpr int kShellGroup = 2;
constexpr int kShellMinWords = 128;                           // active words (of kShellGroup x 256) below which a group is left to KF

template <int BD>
__global__ __launch_bounds__(BD) void k_ball_shell(const ShellArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char shell_smem[];
    if (*a.fix_needed == 0u) return;
    const int t = threadIdx.x;
    const int nzw = a.nzw, lg = a.log2_nzw;
    const int tiles_y = (a.ny + a.ty - 1) / a.ty;             // KD3's grid: blockIdx.x = y tile, blockIdx.y = x tile
    const int ty0 = (int)blockIdx.x * kShellGroup;            // first tile of the group
    const int ntile = min(kShellGroup, tiles_y - ty0);
    uint32_t flags[kShellGroup];
    int n = 0;                                                // words of the group that hold undecided voxels (KD3 counted them)
#pragma unroll
    for (int g = 0; g < kShellGroup; ++g) {
        const uint32_t f = g < ntile ? a.tileflag[(uint32_t)blockIdx.y * (uint32_t)tiles_y + (uint32_t)(ty0 + g)] : 0u;   // (block-uniform)
        flags[g] = f & 0xFFFFu;
        n += (int)(f >> 16);
    }
    // A group with few active words is KF's case (one 16-lane row per VOXEL, rows read straight from the L2-resident bit field):
    // staging 448 halo rows for them costs more than KF spends on their voxels (Bernoulli p = 0.03: 16 active words per group,
    // p = 0.02: 184 of 1024 -- KF 0.23 ms, this pass 0.20 + KF 0.05; p = 0.015: 535).  Their undecided words stay as they are.
    if (n < a.min_words) return;
    const int rw = nzw + 2;                                   // one replicated edge word on each side
    const int gty = kShellGroup * a.ty;                       // rows of the group along y
    const int hx = a.tx + 2 * kShellR, hy = gty + 2 * kShellR;
    uint32_t* list = reinterpret_cast<uint32_t*>(shell_smem); // [kShellGroup * BD] (tile in group << 16) | lane index of an active word
    uint32_t* ulist = list + kShellGroup * BD;                // [kShellGroup * BD] its undecided bits
    uint32_t* count = ulist + kShellGroup * BD;               // [0] active words, [1] the `uncertified` flag as one lane read it
    int* rowofs = reinterpret_cast<int*>(count + 4);          // [kShellRows] (+ pad) LDS word offset of row (dx, dy) from the word's own row
    uint32_t* tile = count + 4 + 128;                         // [hx][hy][rw]
    if (t == 0) {
        count[0] = 0u;
        count[1] = __hip_atomic_load(a.uncertified, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int x0 = a.out_lo + (int)blockIdx.y * a.tx, y0 = ty0 * a.ty;
    __syncthreads();
    if (count[1] != 0u) return;                               // (block-uniform: the general sweeps will redo the grid)
    // this lane's word in each tile of the group (KD3's mapping) -> compacted list of the words that still hold undecided voxels,
    // so that they fill whole waves (p = 0.015: half of the words are active).  (Leaving every lane its own words when nearly
    // all are active -- p = 0.01: 93 %, rows at the staged pitch, no bank conflicts -- was measured too: 0.84 ms per build
    // against 0.78 with the list.)
    {
        const int r = t >> lg, w = t & (nzw - 1);
        const int ty_ = r & (a.ty - 1), tx_ = r >> a.log2_ty;
#pragma unroll
        for (int g = 0; g < kShellGroup; ++g) {
            const int gy = y0 + g * a.ty + ty_;
            uint32_t u = 0;
            if ((x0 + tx_ < a.out_hi) && (gy < a.ny) && ((flags[g] >> (t >> 6)) & 1u))
                u = a.unc[((int64_t)(x0 + tx_ - a.out_lo) * a.ny + gy) * nzw + w];
            // (ordered within the wave -- ballot + prefix, one atomic per wave and tile: neighbouring entries are neighbouring words
            //  of a row, so a wave's LDS reads in the shell pass spread over the banks; one atomic per WORD left them in arrival order)
            const uint64_t bal = __ballot(u != 0u);
            if (bal) {
                uint32_t base = 0u;
                if ((t & 63) == 0) base = atomicAdd(count, (uint32_t)__popcll(bal));
                base = (uint32_t)__shfl((int)base, 0);
                if (u) {
                    const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << (t & 63)) - 1ull));
                    list[slot] = ((uint32_t)g << 16) | (uint32_t)t;
                    ulist[slot] = u;
                }
            }
        }
    }
    // row table: LDS word offset of row (dx, dy), grouped by dx^2 + dy^2 (the slot of a row is a compile-time function of (dx, dy),
    // evaluated here per lane through a small search)
    if (t < (2 * kShellR + 1) * (2 * kShellR + 1)) {
        const int dx = t / (2 * kShellR + 1) - kShellR, dy = t % (2 * kShellR + 1) - kShellR;
        const int r2v = dx * dx + dy * dy;
        if (r2v <= kShellR * kShellR) {
            int base = 0;
            static_for<kShellR2Count>([&](auto gc) {
                constexpr int gg = decltype(gc)::value;
                if (shell_r2(gg) == r2v) base = shell_group_start(gg);
            });
            int rank = 0;
            for (int ex = -kShellR; ex <= kShellR; ++ex)
                for (int ey = -kShellR; ey <= kShellR; ++ey)
                    if (ex * ex + ey * ey == r2v && (ex < dx || (ex == dx && ey < dy))) ++rank;
            rowofs[base + rank] = (dx * hy + dy) * rw;
        }
    }
    // stage the bit rows of the group + halo (a lane per word of a row, BD / nzw rows per pass, 4 independent loads in flight;
    // rows / edge words beyond the grid replicate the nearest in-grid voxel)
    {
        const int j = t & (nzw - 1), rp = t >> lg, rpp = BD >> lg;
        const uint32_t inv_hy = ((1u << 20) + (uint32_t)hy - 1u) / (uint32_t)hy;      // (exact for rows < 2^20 / hy)
        const int nrows = hx * hy;
        for (int row0 = rp; row0 < nrows; row0 += 4 * rpp) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = min(row0 + k * rpp, nrows - 1);
                const int jx = (int)(((uint32_t)row * inv_hy) >> 20), jy = row - jx * hy;
                const int gx = min(max(x0 + jx - kShellR, 0), a.rows_x - 1);
                const int gy = min(max(y0 + jy - kShellR, 0), a.ny - 1);
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
    }
    __syncthreads();
    const int nz = nzw << 5;
    int mxF = 0, mxQ = 0;
    // one word: the shell in two halves, levels in increasing d^2 -- a voxel takes the first level that shows a voxel of the other class
    const bool merged = 4 * n >= 3 * kShellGroup * BD;        // three quarters of the group's words are open (Bernoulli p <= ~0.012)
    auto do_word = [&](int g, int tt, uint32_t U, bool live) {
        const int r = tt >> lg, w = tt & (nzw - 1);
        const int ty_ = g * a.ty + (r & (a.ty - 1)), tx_ = r >> a.log2_ty;
        const uint32_t* c0 = tile + ((tx_ + kShellR) * hy + (ty_ + kShellR)) * rw + (w + 1);
        const uint32_t O = c0[0];
        const int64_t rowi = (int64_t)(x0 + tx_ - a.out_lo) * a.ny + (y0 + ty_);
        float* const orow = a.out + rowi * nz + w * 32;
        uint32_t acc[kShellLevels];
#pragma unroll
        for (int l = 0; l < kShellLevels; ++l) acc[l] = 0u;
        auto resolve = [&](auto lo_c, auto hi_c) {
            static_for<decltype(hi_c)::value - decltype(lo_c)::value>([&](auto lc) {
                constexpr int l = decltype(lo_c)::value + decltype(lc)::value;
                uint32_t m = acc[l] & U;
                U &= ~acc[l];
                if (m) {
                    if (m & ~O) mxF = max(mxF, kShellD2[l]);         // (max: a later word of this lane may stop at a lower level)
                    if (m & O) mxQ = max(mxQ, kShellD2[l]);
                    const float f = (float)(__builtin_sqrt((double)kShellD2[l]) * a.resolution);
                    while (m) {
                        const int b = __builtin_ctz(m);
                        m &= m - 1u;
                        orow[b] = ((O >> b) & 1u) ? -f : f;
                    }
                }
            });
        };
        if (merged) {                                         // (block-uniform) sparse scene: nearly every wave needs the outer half too --
            shell_pass<0, kShellLevels>(c0, rowofs, O, acc);  // one walk over the 113 rows instead of 81 + 113
            resolve(std::integral_constant<int, 0>{}, std::integral_constant<int, kShellLevels>{});
        } else {
            shell_pass<0, kShellSplit>(c0, rowofs, O, acc);
            resolve(std::integral_constant<int, 0>{}, std::integral_constant<int, kShellSplit>{});
            if (__any(U != 0u)) {                             // (wave-uniform)
                shell_pass<kShellSplit, kShellLevels>(c0, rowofs, O, acc);
                resolve(std::integral_constant<int, kShellSplit>{}, std::integral_constant<int, kShellLevels>{});
            }
        }
        // what is still open (d^2 > 36) stays in the undecided word for KF; decided words become 0 there
        if (live) a.unc[rowi * nzw + w] = U;
    };
    const int nl = (int)count[0];                             // (== n unless a tile sticks out of the grid)
    for (int i0 = t - (t & 63); i0 < nl; i0 += BD) {          // (whole waves: a wave without an entry leaves)
        const int i = i0 + (t & 63);
        const bool live = i < nl;
        const uint32_t e = list[live ? i : 0];
        do_word((int)(e >> 16), (int)(e & 0xffffu), live ? ulist[i] : 0u, live);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mxF = max(mxF, __shfl_xor(mxF, off));
        mxQ = max(mxQ, __shfl_xor(mxQ, off));
    }
    if ((t & 63) == 0 && (mxF | mxQ)) slot_max2(a.slots, ((uint32_t)blockIdx.y * gridDim.x + blockIdx.x) * (BD / 64) + ((uint32_t)t >> 6), mxF, mxQ);
}

inline size_t shell_lds_bytes(int bd, int tx, int ty, int nzw) {
    return (size_t)(2 * kShellGroup * bd + 4 + 128) * 4 + (size_t)(tx + 2 * kShellR) * (kShellGroup * ty + 2 * kShellR) * (nzw + 2) * 4;
}

// The instantiations the launcher uses, compiled in their own translation unit (sdfgpu_dense6_tu.hip).
#define SDFGPU_SHELL_INSTANCES(X) X(256)       // (KD3, the stage in front, exists for 256-lane tiles only)
#ifndef SDFGPU_DENSE6_TU
#define SDFGPU_SHELL_DECLARE(BD) extern template __global__ void k_ball_shell<BD>(const ShellArgs);
SDFGPU_SHELL_INSTANCES(SDFGPU_SHELL_DECLARE)
#undef SDFGPU_SHELL_DECLARE
#endif

}  // namespace sdfgpu
