// sdfgpu_envelope_dc.hpp -- KE2 / KE3 (third generation of the far-field kernel): the far-field y / x sweep as a three-level, 8-ary exact
// min-plus search built from ONE inner loop.
//
// Along one line the sweep computes D(p) = min_q F(q) + (p - q)^2 (F = squared distance inside the rows already
// swept, "no site" = +inf).  c(p, q) = F(q) + (p - q)^2 is a Monge array, so for any argmin a(p1) at p1 < p2 there is
// an argmin a(p2) >= a(p1): once the argmins of two positions are known, every position between them only needs the
// candidates between those argmins (the F terms cancel in the quadrangle inequality c(p,q) + c(p',q') <= c(p,q') + c(p',q) for p < p', q < q': what is left is 2 (p'-p)(q-q') <= 0).
//
// What changed against the second generation (binary subdivision, one position per lane per level, ~10 barriers per
// tile, 2.5 wave-level VALU instructions per voxel of which the levels above the chunk phase were 40 %; the kernel
// is VALU-bound -- 90 % of the SIMD issue slots -- so instructions are what counts):
//   * ONE primitive does the searching: a lane holds 8 positions with a common stride and streams a candidate range
//     past them, two candidates per LDS read (ds_read_b64), 3 VALU per position and candidate pair:
//         v0 = key[q] - c_k * q',  v1 = key[q+1] - c_k * (q'+1)      (v_mad_i32_i24 x 2)
//         best_k = min(best_k, v0, v1)                               (v_min3_u32)
//     Level B runs it with stride 8 inside each interval of level A, level C with stride 1 inside each interval of
//     level B (the "chunk phase").  Level A (positions 64 i) scans one position per lane over a range clipped by a
//     distance bound (below); 5 barriers per tile and pass, no per-position set-up code in the loops.
//   * Centred coordinates.  With h = ceil(L / 2), q' = q - h, p' = p - h and
//         key[q] = ((F(q) + q'^2 + h^2) << B) | q
//     the candidate value key[q] - ((2 p') << B) * q' equals ((F(q) + (p - q)^2 + (h^2 - p'^2)) << B) | q, which is
//     non-negative and below 2^32 whenever finf + (L + 2)^2 < 2^(32 - B): the multiply-add form needs no more key
//     bits than the running-sum form of the second generation did, and no per-position term inside the loop.
//   * Distance-bound clipping with the tile's smallest site value m: any candidate's value v bounds the optimum of a
//     position p, and a candidate farther than sqrt(v - m) from p costs more than v.  The scenes this kernel serves
//     have lines far from every object, where F is large but nearly flat: v - m is then small although v is not.
//   * Staging writes each key with one clamp + one shift-add; which voxels are filled is not kept anywhere: in the
//     pass that serves free voxels a voxel is filled iff its result is 0 (only a filled voxel is a site with F = 0),
//     and the other way round in the second pass.
//   * The site span (first / last candidate) is kept per TILE (wave ballots in the staging loop); the tile's 16 lines
//     see nearly the same sites (in the first pass exactly the same: a row / plane either holds a filled voxel or not).
//   * Any line count: tiles that stick out of their line group replicate the last line on load and mask the stores;
//     shapes without 4-element alignment use scalar loads (template parameter VEC).
// Measured and not kept (512^3, two-box scene and Bernoulli 1 % .. 0.01 %; tools/ff_check.sh, profiling builds with
// -DSDFGPU_DEBUG_HOOKS / -DSDFGPU_PHASE_CLOCKS):
//   * 512 lanes per tile inside 64 VGPRs (8 waves per SIMD): +-5 % either way; 8-line tiles (20 KB of LDS, 7 - 8 workgroups
//     per CU) with 128 or 256 lanes: 12 - 25 % slower; padding the LDS so that only 2 workgroups fit a CU: 1.55x slower --
//     4 workgroups of 4 waves per CU is a plateau;
//   * 32-line tiles (128-byte row segments, 512 lanes): +-3 %, and twice the filled voxels per tile push Bernoulli 3 % into the
//     second pass; the bare memory pattern (tools/probe/tile_copy_probe.hip) is 0.27 ms with 16-line and 0.22 ms with 32-line
//     tiles against 0.17 ms for a linear copy of the same 8 B/voxel;
//   * a persistent grid whose workgroups request the rows of their next tile before searching the current one: 10 - 25 %
//     SLOWER on every scene (120 VGPRs, static tile assignment);
//   * an LDS table of the finished values of small squared distances instead of the fp64 finish: no change -- removing the
//     fp64 finish altogether does not change the time either (it overlaps with the integer pipe);
//   * finishing 4 chunks per lane first and storing their 32 values back to back (store bursts like the bare memory pattern's):
//     no change;
//   * ablation of the x sweep on the two-box scene: no search 0.37 ms (of 0.55), no search and no stores 0.19 ms.
// Exactness: integer arithmetic throughout; the finish is the reference's (sqrt and multiply in fp64, one cast,
// sdf_generation.hpp:254-265).
#pragma once
#include "sdfgpu_kernels.hpp"
#include "sdfgpu_sweep_x16.hpp"
#include "sdfgpu_finish.hpp"

namespace sdfgpu {

// Rows the marching kernels scan outward before handing the sweep to the far-field kernel.  Long enough that mid-sparse
// scenes (p = 0.01: distances up to ~30) never pay for a far-field pass they do not need.
constexpr int kScanExpectNear = 40;

constexpr int kDcLocalSat = 1023;     // pass 0 finishes a filled voxel itself when its result is below this: its own in-row squared distance or a free
                                      // voxel within 31 positions along the line (round 4: the bound used to be on the in-row distance alone, 64 --
                                      // a wall or a floor a few voxels thick, whose in-row distance is "none", sent every tile to the second pass;
                                      // 31 because a tile lists at most 448 / 16 = 28 filled voxels per line) ...
constexpr int kDcLocalFilled = 448;   // ... when the tile holds at most this many filled voxels (of 16 x L; a full second pass costs as much
                                      // as the first: a 3 %-occupied 512^3 scene has 245 per tile and must stay below)

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
    int pitch;                // LDS words per line of keys (>= L + 2, == 2 mod 64)
    int M, Kp;                // coarse positions ceil(L / 8); Kp unused
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
    int probe_thr3;           // y probe: the x sweep's far threshold, counted into probe_out[5] (status word 17): the x result is
                              // never larger than the y result, so a y probe that finds few such voxels settles the x tier too
    uint32_t* probe_out;
    // probe launches also turn their counters into the tier decision (the last workgroup to finish does what k_decide_tier
    // does): decide_small = the status block, nullptr = no decision here
    uint32_t* decide_small;
    int decide_stage, decide_dense_tried, decide_force, decide_den, decide_handoff, decide_mid_den, decide_xden;
    // both axes far-field: when *i32_flag != 0 the y sweep hands its result to the x sweep as an exact int32 plane field
    // (out_i32 / in_i32, the side-table buffer used whole) instead of p16 + side table; nullptr = the pointers alone decide
    const uint32_t* i32_flag;
    int h;                    // centre of the key coordinates, ceil(L / 2)
    int64_t group_lines;      // lines per outer unit (tiles that stick out replicate the last line)
    unsigned long long* clocks;   // SDFGPU_PHASE_CLOCKS builds only: per-phase shader-clock sums over waves ([stage - 2][8])
    int dbg;                  // SDFGPU_DEBUG_HOOKS builds only (wrong results): bit0 no search, bit1 no fp64 finish, bit2 no stores, bit3 no level-C scans, bit4 no level-B scans (with bit3)
                              // bit5 the y sweep's stores to one block per tile, bit6 no local-search rounds; two-valued tiles: bit7 path off (results stay right),
                              // bit8 no distance chains, bit9 no nearest-site scans, bit10 no classification
    int64_t ntiles;           // LOOP form: tiles of the whole launch (a workgroup takes tiles blockIdx.x, + gridDim.x, ...)
    const uint32_t* bits;     // STAGE 2, scalar (VEC = false) form only: the dense tier's bit field ([x][y][nzw] words, bit i of word w =
    int nzw;                  // voxel z = 32 w + i is filled) INSTEAD of the z field: the stand-by behind a trusted dense tier computes
                              // the z distances itself, so that it needs no z sweep launched in front of it (see zdist_from_bits)
    uint32_t* ran_flag;       // nullptr, or a status word that a launch which does work raises (the stand-by pair reports itself)
    // LOOP form, STAGE 3 -- the last launch of a stand-by build -- also does the end-of-build fold (k_fold_slots's job: maxima,
    // status block -> result / report, clear): fold_status = the status block (nullptr: no fold here), fold_ticket = a word of it
    // that counts the workgroups that have finished.  A launch that leaves on its guard has written nothing: workgroup 0 folds
    // straight away; one that works folds in the workgroup that finishes last.
    uint32_t* fold_status;
    uint32_t* fold_result;
    uint32_t* fold_report;
    uint32_t* fold_ticket;
    uint32_t fold_report_mask;
    // round 6, plane sparsity (builds that go straight to the far-field pair; nullptr = no such knowledge).  The z sweep in front writes
    // one byte per z row -- "holds a filled voxel" -- and does NOT write the z field of rows without one; k_pack_row_flags packs the bytes
    // to one bit per row, one byte per x-plane and a "some plane is empty" status word.  STAGE 2 skips the tiles of planes without a filled
    // voxel (their result, "free, no filled voxel in this plane" everywhere, is not written) and takes "+32767" for rows without one
    // instead of loading them -- the plane's mask comes through SCALAR loads (16 words at ny = 512): a first form with per-lane byte
    // loads put a vector-memory round trip in front of every tile's row loads (+6 % on a scene that has no such row).  STAGE 3 takes
    // "no site" for the rows of skipped planes instead of loading them, and looks nothing up when no plane was skipped.
    const uint32_t* row_bits; // (packed by k_pack_row_flags: [x][row_words] words, bit y & 31 of word y >> 5 = z row (x, y) holds a filled voxel)
    int row_words;
    const uint8_t* plane_any; // [x]: 0 = x-plane x holds no filled voxel, 1 = some of its z rows do, 2 = every z row does
    uint32_t* some_empty;     // status word STAGE 2 raises when it skips a plane: a scene without one (walls) costs STAGE 3 no look-ups at all
    // round 6, flat-positive tiles (STAGE 2, pass 0, planes whose every z row holds a filled voxel -- a floor).  When every entry of a line
    // that is not a zero site (F = 0: a filled voxel) carries the SAME value c -- the height above the floor, for every y line of the open
    // volume -- then min_q F(q) + (p - q)^2 = min(c, d0(p)^2), d0 = distance along the line to the nearest zero site: a candidate is a zero
    // site (cost (p - q)^2) or a c site (cost >= c, = c at q = p).  A tile whose 16 lines are all of that kind skips the three search
    // levels and computes d0 from the zero sites of its chunks (tools/flat_tile_model.py: 51 % of the room's y tiles).  With TWO values
    // mn < mx outside the zero sites -- a line that passes over or under a table top -- the answer is min(mx, d0^2, mn + d1^2), d1 = distance
    // to the nearest mn site: that covers every y tile of the room.  0 = never.
    int flat_on;              // 1: as long as it pays (flat_score); 2: in every such tile
    // The habit lives on the device: flat_score[0] is a (signed) word of the handle that no build clears.  A tile that tries and does not
    // qualify has paid ~15 % for nothing (a floor under noise: every tile would), one that qualifies saves ~40 %: every 16th tile that
    // tried votes +1 / -3 (atomic adds, nobody waits for them); k_pack_row_flags, in front of every such y sweep, clamps the sum to
    // [0, kDcFlatCap] and publishes the gate flat_score[1] = "below kDcFlatOff", which the y sweep reads with its other status words --
    // a fresh load of the score itself sat on every tile's critical path: +4 %.  With the gate down every 64th tile still tries, and
    // votes: the habit is back one build after the scene changes.  Results never depend on it.
    uint32_t* flat_score;
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


// Test hook (sdfgpu_debug_finish_table): the finish of squared distances 0 .. n - 1 in the fp32 form of sdfgpu_finish.hpp -- fast path,
// wave-wide ballot, the x sweep's own fp64 sequence (sqrt_exact_pos) for the rounds that ask for it -- or, fin.ok = 0, in the fp64
// sequence alone: tests compare EVERY D with the reference's arithmetic on the device's own instructions; slow_count = lanes that
// raised the flag.  (The fp32 form is not used by the x sweep: measured 2.4 % slower, see sdfgpu_finish.hpp.)
SDFGPU_KERNEL __launch_bounds__(256) void k_finish_table(float* __restrict__ out, int64_t n, double resolution, FinishFast fin,
                                                       uint32_t* __restrict__ slow_count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int D = (int)(i < n ? i : 1);
    bool slow;
    float f = finish_fast(D, fin, slow);
    slow = (slow || !fin.ok) && D != 0;
    if (__ballot(slow) != 0ull) {
        const float g = (float)(sqrt_exact_pos((double)D) * resolution);
        f = slow ? g : f;
    }
    if (D == 0) f = 0.0f;
    if (i < n) {
        out[i] = f;
        if (slow && fin.ok) atomicAdd(slow_count, 1u);
    }
}

// Signed z distance of voxel z of one bit row, as the z sweep (K1) defines it: + distance to the nearest FILLED voxel of the row
// for a free voxel, - distance to the nearest FREE one for a filled voxel, +-kInf16 when the row holds no voxel of the other
// class.  Word-at-a-time scans in both directions: slow next to K1 (up to 2 nzw loads per voxel on an empty row), which is fine
// where it is used -- the stand-by pair behind a trusted dense tier runs once, in the build in which the scene left that tier;
// what matters there is that the stand-by is ONE launch shorter (every launch costs the dense-certified steady state ~2.6 us:
// 0.1405 -> 0.1431 ms per 512^3 step measured with a separate guarded K1 in front).  Pad bits of a ragged last word replicate
// the row's last voxel (k_pack_bits_rows): a hit there is never nearer than that voxel itself, and is dropped.
__device__ __forceinline__ int zdist_from_bits(const uint32_t* __restrict__ row, int nzw, int nz, int z) {
    const int w0 = z >> 5, b = z & 31;
    const uint32_t cw = row[w0];
    const bool own = (cw >> b) & 1u;
    const uint32_t flip = own ? 0xFFFFFFFFu : 0u;              // word ^ flip: set bits = voxels of the other class
    int best = kInf16;
    {
        uint32_t x = (cw ^ flip) & (0xFFFFFFFEu << b);          // above z (b = 31: nothing left in this word)
        int w = w0;
        while (x == 0u && ++w < nzw) x = row[w] ^ flip;
        if (x != 0u) { const int zz = w * 32 + (__ffs((int)x) - 1); if (zz < nz) best = zz - z; }
    }
    {
        uint32_t x = (cw ^ flip) & ((1u << b) - 1u);            // below z
        int w = w0;
        while (x == 0u && --w >= 0) x = row[w] ^ flip;
        if (x != 0u) { const int zz = w * 32 + (31 - __clz((int)x)); best = best < z - zz ? best : z - zz; }
    }
    return own ? -best : best;
}

// The same distances for the staging loop of the stand-by y sweep, where the 4 lanes of a DPP quad hold the 4 x 4 voxels of one
// tile row (16 consecutive z = half a word of the bit row): the quad reads the row's words ONCE, interleaved (lane j takes
// words 4 i + j: independent loads, 16 bytes per quad and step), every lane reduces its words to four positions -- first set /
// first clear bit in the words above the tile's word w0, last set / last clear bit in the words below it -- the quad combines
// them with two quad-permute steps, and each lane finishes its 4 voxels from w0's word and those four positions.  (One
// zdist_from_bits call per voxel walked up to 2 nzw dependent loads on an empty row: the transition build of the two-box scene
// took 2.6 ms, 2.9x the steady-state far-field build; this form: see DESIGN.)  All lanes of the wave must be active.
struct ZRowBits { uint32_t W; int fa1, fa0, lb1, lb0; };
__device__ __forceinline__ ZRowBits zrow_scan_quad(const uint32_t* __restrict__ row, int nzw, int w0, int j) {
    int a1 = 1 << 20, a0 = 1 << 20, b1 = -(1 << 20), b0 = -(1 << 20);
    uint32_t W = 0u;
    for (int i = 0; 4 * i < nzw; ++i) {
        const int w = 4 * i + j;
        if (w < nzw) {
            const uint32_t x = row[w], nx = ~x;
            if (w == w0) W = x;
            if (w > w0) {
                if (x) a1 = min(a1, 32 * w + __ffs((int)x) - 1);
                if (nx) a0 = min(a0, 32 * w + __ffs((int)nx) - 1);
            }
            if (w < w0) {
                if (x) b1 = max(b1, 32 * w + 31 - __clz((int)x));
                if (nx) b0 = max(b0, 32 * w + 31 - __clz((int)nx));
            }
        }
    }
    auto quad = [](int v, auto op) {
        v = op(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
        return op(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    };
    ZRowBits r;
    r.W = (uint32_t)quad((int)W, [](int x, int y) { return x | y; });
    r.fa1 = quad(a1, [](int x, int y) { return x < y ? x : y; });
    r.fa0 = quad(a0, [](int x, int y) { return x < y ? x : y; });
    r.lb1 = quad(b1, [](int x, int y) { return x > y ? x : y; });
    r.lb0 = quad(b0, [](int x, int y) { return x > y ? x : y; });
    return r;
}
__device__ __forceinline__ int zdist_from_row(const ZRowBits& r, int w0, int nz, int z) {
    const int b = z & 31;
    const bool own = (r.W >> b) & 1u;
    const uint32_t X = own ? ~r.W : r.W;                        // set bits = voxels of the other class
    const uint32_t up = X & (0xFFFFFFFEu << b), dn = X & ((1u << b) - 1u);
    const int zu = up ? 32 * w0 + __ffs((int)up) - 1 : (own ? r.fa0 : r.fa1);
    const int zd = dn ? 32 * w0 + 31 - __clz((int)dn) : (own ? r.lb0 : r.lb1);
    int best = kInf16;
    if (zu < nz) best = zu - z;
    if (zd >= 0 && z - zd < best) best = z - zd;
    return own ? -best : best;
}

// Plane sparsity, between the z sweep and the far-field y sweep: packs every x-plane's row bytes (row_any[x * ny + y])
// into ceil(ny / 32) words, sums them up in plane_any[x] (0 / 1 / 2, below) and raises *some_empty for a plane without a filled voxel.
constexpr uint32_t kDcFlatOff = 16u, kDcFlatCap = 256u;        // (EnvDcArgs::flat_score)
SDFGPU_KERNEL __launch_bounds__(1024) void k_pack_row_flags(const uint8_t* __restrict__ row_any, int nx, int ny, int row_words, uint32_t* __restrict__ row_bits,
                                                         uint8_t* __restrict__ plane_any, uint32_t* __restrict__ some_empty, uint32_t* __restrict__ flat_score) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && flat_score) {      // the habit of the two-valued tiles: last build's votes -> this build's gate
        int v = (int)flat_score[0];
        v = v < 0 ? 0 : v > (int)kDcFlatCap ? (int)kDcFlatCap : v;
        flat_score[0] = (uint32_t)v;
        flat_score[1] = v < (int)kDcFlatOff ? 1u : 0u;
    }
    // a WAVE per x-plane at a time, 16 waves per workgroup, at most 16 workgroups: the two status words see one store per WORKGROUP.
    // (Device-scope stores to one address cost ~28 ns each: forms with a store per plane or per wave -- 512 to 1024 of them -- took
    //  20 - 29 us at 512^3, whatever else the kernel did.)
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 16 + (threadIdx.x >> 6), nwaves = gridDim.x * 16;
    bool saw_empty = false, saw_hole = false;
    for (int x = wave; x < nx; x += nwaves) {
        bool any = false, hole = false;
        // (every byte of the plane requested before the first ballot: one load at a time made this kernel 20 us of load latency)
        int fv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int y = 64 * k + lane;
            fv[k] = (y < ny) ? (int)row_any[(int64_t)x * ny + y] : 0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int y0 = 64 * k;
            if (y0 >= 32 * row_words) break;
            const int y = y0 + lane;
            const int f = fv[k] != 0;
            const uint64_t b = __ballot(f);
            const uint64_t valid = __ballot(y < ny);
            any |= b != 0ull;
            hole |= (valid & ~b) != 0ull;
            if (lane == 0) {
                const int w = y0 >> 5;
                row_bits[(int64_t)x * row_words + w] = (uint32_t)b;
                if (w + 1 < row_words) row_bits[(int64_t)x * row_words + w + 1] = (uint32_t)(b >> 32);
            }
        }
        // 0: no filled voxel in the plane; 1: some of its rows hold one; 2: every row does (a floor, a wall: the row mask has nothing to say)
        if (lane == 0) plane_any[x] = !any ? 0 : hole ? 1 : 2;
        saw_empty |= !any;
        saw_hole |= hole;
    }
    // some_empty[0]: a plane without a filled voxel exists (the x sweep then looks planes up); some_empty[-3] (status word 19): a row
    // without one exists (the y sweep then looks at plane_any / the row masks at all: a scene of floors and walls never does)
    const int wg_empty = __syncthreads_or(saw_empty ? 1 : 0), wg_hole = __syncthreads_or(saw_hole ? 1 : 0);
    if (threadIdx.x == 0 && wg_empty) __hip_atomic_store(some_empty, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0 && wg_hole) __hip_atomic_store(some_empty - 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kDcLines = 16;          // lines per tile (the kernel is a template over 8 / 16 lines and 128 / 256 / 512 lanes: 16 x 256 is the measured optimum)
constexpr int NB = 8;          // staging: row loads in flight per lane (a 512-line is staged from ONE round of loads)

constexpr int kDcMisc = 80;           // words of per-tile bookkeeping in LDS (k_envelope_dc: misc)
inline int envelope_dc_pitch(int L) { return ((L + 63) / 64) * 64 + 2; }        // >= L + 2, == 2 mod 64, even (8-byte pairs)
inline size_t envelope_dc_lds_bytes(int L, int lines = kDcLines) {
    const int M = (L + 7) / 8;
    return ((size_t)lines * envelope_dc_pitch(L) + (size_t)(M + 2) * lines + kDcMisc + kDcLocalFilled) * 4;
}

// a * b + c on the low 24 bits of a and b (signed), low 32 bits of the result: one full-rate instruction.  (Written as
// __mul24(a, b) + c the compiler keeps explicit sign extensions of the operands inside the loop.)
__device__ __forceinline__ uint32_t mad_i24(int a, int b, uint32_t c) {
    uint32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (HIP's overloaded min() picks the double version for some unsigned argument pairs: v_cvt_f64_u32 + v_min_f64 + v_cvt_u32_f64)
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// Profiling builds (-DSDFGPU_PHASE_CLOCKS, never the shipped library): shader-clock time per phase, summed over waves.
// With -DSDFGPU_TRIP_COUNTS on top, the same eight slots count trips of the scan loops instead: [2 l] = trips issued by the
// waves (the slowest lane's), [2 l + 1] = trips the lanes needed, summed, for level l = 0 (A), 1 (B), 2 (C).
#if defined(SDFGPU_PHASE_CLOCKS) && defined(SDFGPU_TRIP_COUNTS)
#define DC_TRIP ++dbgt
#define DC_STAMP(k) do { int mx_ = dbgt, sm_ = dbgt; \
        for (int off_ = 32; off_ >= 1; off_ >>= 1) { mx_ = max(mx_, __shfl_xor(mx_, off_)); sm_ += __shfl_xor(sm_, off_); } \
        if ((k) >= 1 && (k) <= 3) { clk[2 * ((k) - 1)] += (unsigned long long)mx_; clk[2 * ((k) - 1) + 1] += (unsigned long long)sm_; } \
        dbgt = 0; (void)tprev; } while (0)
#elif defined(SDFGPU_PHASE_CLOCKS)
#define DC_TRIP do {} while (0)
#define DC_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); clk[k] += now_ - tprev; tprev = now_; } while (0)
#else
#define DC_TRIP do {} while (0)
#define DC_STAMP(k) do {} while (0)
#endif

// Device-side tier selection for one axis (stage 0 = y, 1 = x), one thread.  small = the context's status block:
// [3] uncertified (dense tier could not decide the scene), [4] / [5] "y / x sweep belongs to the far-field kernel" (also
// raised by a marching sweep that hits its scan bound), [8 + 2 stage] guard word of the marching sweep, [12] / [13] the
// probe's counters.  The marching sweep runs iff the general pipeline is needed at all and the probe found the axis
// near-field; otherwise the far-field flag is raised and the (flag-guarded) far-field kernel does the sweep.
__device__ __forceinline__ void decide_tier(uint32_t* __restrict__ small, int stage, int dense_tried, int force, int num, int den,
                                            int handoff, int mid_den, int xden = 0) {
    auto ld = [&](int i) -> uint32_t { return __hip_atomic_load(small + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const bool active = dense_tried ? ld(3) != 0u : true;
    const uint32_t far_n = ld(12), tot = ld(13), mid_n = ld(11);
    // far when more than num / den of the sampled voxels need a long scan (64-bit: counts are < 2^24, factors small)
    bool far = force >= 0 ? force != 0 : (uint64_t)far_n * (uint64_t)den > (uint64_t)tot * (uint64_t)num;
    if (stage == 1 && ld(7) != 0u) far = true;                  // the y probe chose the far-field pair with int32 hand-off
    if (stage == 1 && force < 0 && ld(18) != 0u) far = false;   // the y probe already settled it: near-field (no x probe ran)
    // (only when the y sweep itself stays near-field: a far-field y sweep may hand the x sweep int32 values, which only the
    //  far-field x kernel reads)
    if (stage == 0) small[18] = (force < 0 && !far && tot != 0u && (uint64_t)ld(17) * (uint64_t)xden <= (uint64_t)tot) ? 1u : 0u;
    // y sweep, near-field: radius-8 register windows when more than 1 / mid_den of the voxels are beyond the radius-3 one
    const bool wide = stage == 0 && mid_den > 0 && (uint64_t)mid_n * (uint64_t)mid_den > (uint64_t)tot;
    small[8 + 2 * stage] = (active && !far && !wide) ? 1u : 0u;
    if (stage == 0) small[9] = (active && !far && wide) ? 1u : 0u;
    if (active && far) small[4 + stage] = 1u;
    if (stage == 0 && active && far && handoff) small[7] = 1u;
    small[11] = 0u;
    small[12] = 0u;
    small[13] = 0u;
    small[17] = 0u;
    if (stage == 1) small[18] = 0u;
    small[14 + stage] = tot ? (far_n * 1000u) / tot : 0u;         // per-mille of far voxels in the sample (diagnostics)
}
SDFGPU_KERNEL void k_decide_tier(uint32_t* __restrict__ small, int stage, int dense_tried, int force, int num, int den, int handoff,
                              int mid_den) {
    decide_tier(small, stage, dense_tried, force, num, den, handoff, mid_den, 0);
}
// ... called by the last workgroup of a probe launch (every workgroup passes here exactly once, early exits included)
__device__ __forceinline__ void probe_done(const EnvDcArgs& a) {
    if (!a.decide_small || threadIdx.x != 0) return;
    __threadfence();
    if (atomicAdd(a.decide_small + 16, 1u) == gridDim.x - 1u) {
        __threadfence();
        decide_tier(a.decide_small, a.decide_stage, a.decide_dense_tried, a.decide_force, 1, a.decide_den, a.decide_handoff, a.decide_mid_den, a.decide_xden);
        a.decide_small[16] = 0u;
    }
}

// The probe as a window statistic (round 3).  What the tier decision needs is the share of free voxels whose distance
// after this sweep is still >= thr (16 for y, 9 for x): small thresholds, and "d^2 >= thr" is decided exactly by the
// candidates within |offset| < sqrt(thr) along the axis -- 2 W + 1 values of the sweep's own input per sampled voxel, no
// tile staging, no search.  One lane per sample, ~32 k samples spread over the grid, one set of atomics per workgroup, the
// last workgroup turns the counters into the decision (decide_tier): 7 us instead of the 25 us of level A on 256 sampled
// tiles.  (k_envelope_dc keeps its probe mode for thresholds above 81.)
struct ProbeArgs {
    const int16_t* in16;      // z field (stage 2) / 16-bit plane field (stage 3), or nullptr
    const int32_t* in32;      // int32 plane field (stage 3 of shapes without the 16-bit one), or nullptr
    int64_t n, ls, step;      // voxels, line stride, sample stride
    int L, W;                 // positions along the axis, window radius
    int thr, thr2, thr3;
    uint32_t nsamples;
    uint32_t* probe_out;      // status word 12 (see decide_tier)
    const uint32_t* guard;    // nullptr: always run; else run iff *guard != 0
    const uint32_t* i32_flag; // stage 3: non-zero = the y probe chose the int32 hand-off, the x tier is decided
    uint32_t* decide_small;
    int decide_stage, decide_dense_tried, decide_force, decide_den, decide_handoff, decide_mid_den, decide_xden;
};
// WMAX: compile-time bound of the window radius (3 or 8): the 2 WMAX + 1 loads of a sample are issued together, at clamped
// addresses, and masked afterwards (a loop over the run-time radius waited for every load in turn: 18 us instead of 7).
template <int STAGE, int WMAX>
__global__ __launch_bounds__(256) void k_probe_window(const ProbeArgs a) {
    __shared__ uint32_t cnt[4];
    const int t = threadIdx.x;
    bool run = true;
    if (a.guard && *a.guard == 0u) run = false;
    if (STAGE == 3 && a.i32_flag && *a.i32_flag != 0u) run = false;
    if (STAGE == 3 && __hip_atomic_load(a.decide_small + 18, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) run = false;   // settled by the y probe
    if (run) {                                                  // (block-uniform)
        if (t < 4) cnt[t] = 0u;
        __syncthreads();
        const uint32_t sidx = blockIdx.x * 256u + (uint32_t)t;
        int far = 0, tot = 0, mid = 0, x9 = 0;
        if (sidx < a.nsamples) {
            // one sample in every stretch of `step` voxels, at a hashed offset inside it (a plain stride is a multiple of the row
            // length on power-of-two grids: every sample on the z = 0 face)
            const int64_t idx = (int64_t)sidx * a.step + (int64_t)(((sidx * 0x9E3779B1u) >> 8) % (uint32_t)a.step);
            const int p = (int)((idx / a.ls) % a.L);
            auto value = [&](int64_t i) -> int {
                if (a.in32) return a.in32[i];
                return (int)a.in16[i];
            };
            const int c = value(idx);
            if (c > 0) {                                        // a free voxel: distance to the nearest filled one
                int D = kInf32;
                int v[2 * WMAX + 1];
#pragma unroll
                for (int k = 0; k <= 2 * WMAX; ++k) {
                    const int q = imin(imax(p + k - WMAX, 0), a.L - 1);
                    v[k] = value(idx + (int64_t)(q - p) * a.ls);
                }
#pragma unroll
                for (int k = 0; k <= 2 * WMAX; ++k) {
                    const int d = k - WMAX, q = p + d;
                    int f;
                    if (v[k] <= 0) f = 0;
                    else if (STAGE == 2) f = v[k] >= kInf16 ? kInf32 : v[k] * v[k];          // z distances
                    else f = (!a.in32 && v[k] >= 32767) ? kInf32 : imin(v[k], kInf32);     // squared in-plane distances (16-bit: saturated)
                    const bool use = d >= -a.W && d <= a.W && q >= 0 && q < a.L;
                    D = imin(D, (use && f < kInf32) ? f + d * d : kInf32);
                }
                tot = 1;
                far = D >= a.thr ? 1 : 0;
                mid = (a.thr2 > 0 && D >= a.thr2) ? 1 : 0;
                x9 = (a.thr3 > 0 && D >= a.thr3) ? 1 : 0;
            }
        }
        const uint32_t bf = (uint32_t)__popcll(__ballot(far)), bt = (uint32_t)__popcll(__ballot(tot));
        const uint32_t bm = (uint32_t)__popcll(__ballot(mid)), bx = (uint32_t)__popcll(__ballot(x9));
        if ((t & 63) == 0) { atomicAdd(&cnt[0], bf); atomicAdd(&cnt[1], bt); atomicAdd(&cnt[2], bm); atomicAdd(&cnt[3], bx); }
        __syncthreads();
        if (t == 0) {
            atomicAdd(a.probe_out, cnt[0]); atomicAdd(a.probe_out + 1, cnt[1]);
            if (a.thr2 > 0) atomicAdd(a.probe_out - 1, cnt[2]);
            if (a.thr3 > 0) atomicAdd(a.probe_out + 5, cnt[3]);
        }
    }
    if (t == 0) {                                               // every workgroup passes here once: the last one decides
        __threadfence();
        if (atomicAdd(a.decide_small + 16, 1u) == gridDim.x - 1u) {
            __threadfence();
            decide_tier(a.decide_small, a.decide_stage, a.decide_dense_tried, a.decide_force, 1, a.decide_den, a.decide_handoff,
                        a.decide_mid_den, a.decide_xden);
            a.decide_small[16] = 0u;
        }
    }
}

// NL lines per tile, NT lanes: S = NT / NL lanes ("slots") per line.  The kernel is sensitive to how many WORKGROUPS a CU
// holds (their phases interleave; 2 instead of 4 per CU is 1.55x slower), and that number is set by the LDS footprint:
// 16 lines x 512 positions x 4 B of keys = 39 KB -> 4 per CU; 8 lines -> 20 KB -> 7 - 8 per CU.
//
// LOOP (round 4): the stand-by form behind a trusted dense tier.  Such a launch nearly always exits on its guard, and what a
// guarded exit costs grows with the grid (1.7 us up to 2048 workgroups, 2.9 us at 8192, tools/probe/launch_probe.hip): the
// LOOP form is launched with a small grid (option "standby_grid", 1024 workgroups = 4 per CU) whose workgroups take tiles
// blockIdx.x, + gridDim.x, ... -- static assignment, a few % slower than one tile per workgroup when it does run (the build in
// which a scene leaves the dense tier).
// WPS = waves per SIMD the register budget is set for: 4 workgroups of 256 lanes per CU for lines up to 512 (LDS: 37 KB each), 2 workgroups
// of 512 lanes for longer lines (76 KB each at 1024) -- round 4: lines above 512 used to run 2 x 256 lanes per CU, 2 waves per SIMD.
// LC: 0 = the line's geometry (length, key bits, LDS pitch, chunk count, centre) from the arguments; otherwise the compile-time line length
// (the launcher picks the instance that matches: 512-voxel lines): loop bounds, the level structure and the LDS addressing are then constants.
__host__ __device__ constexpr int dc_clog2(int L) { int b = 1; while ((1 << b) < L) ++b; return b; }
template <int STAGE, bool VEC, int NT = 256, int NL = kDcLines, bool LOOP = false, int WPS = (NL == 8 ? 8 : 4) * (NT / 64) / 4, int LC = 0>
__global__ __launch_bounds__(NT, WPS) void k_envelope_dc(const EnvDcArgs a) {
    constexpr int S = NT / NL;              // lanes per line
    constexpr int LPR = NL / 4;             // staging: lanes per row (4 lines each)
    constexpr int LPR_SH = NL == 16 ? 2 : 1;
    constexpr int PP = NT / LPR;            // rows staged per load round of the workgroup
    constexpr int NB = 512 / PP > 0 ? 512 / PP : 1;         // loads in flight per lane (a 512-line is staged from ONE round)
    constexpr int NW = NT / 64;             // waves
    static_assert(NL == 8 || NL == 16, "tile of 8 or 16 lines");
    extern __shared__ __attribute__((aligned(16))) uint32_t dc_smem[];
    if (a.guard) {
        const uint32_t gv = *a.guard;
        if ((gv != 0u) == (a.guard_invert != 0)) {
            if constexpr (LOOP && STAGE == 3) {
                if (a.fold_status && blockIdx.x == 0)           // (block-uniform; nothing of this launch writes the slots or the status block)
                    fold_slots_device<NT>(a.maxdsq, a.fold_status, a.fold_result, a.fold_report, a.fold_report_mask, (int)threadIdx.x);
            }
            probe_done(a);
            return;
        }
    }
    const bool i32 = !a.i32_flag || *a.i32_flag != 0u;        // (block-uniform)
    // plane sparsity: the y sweep in front skipped at least one x-plane (launch-uniform; its launch is complete)
    const bool planes = STAGE == 3 && a.plane_any && __hip_atomic_load(a.some_empty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    // ... STAGE 2: the z sweep in front left at least one z row unwritten (k_pack_row_flags raises status word 19)
    // ... none: every z row of the grid holds a filled voxel; flat_gate: the habit of the two-valued tiles says "try" (EnvDcArgs::flat_score).
    // (The two words are requested TOGETHER: with the gate's load behind the test of word 19 every tile waited a second round trip, +3.6 %.)
    uint32_t w19 = 0u, wgate = 0u;
    if (STAGE == 2 && a.row_bits) {
        w19 = __hip_atomic_load(a.some_empty - 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wgate = __hip_atomic_load(a.flat_score + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool rows_known = STAGE == 2 && a.row_bits && w19 != 0u;
    // (lines longer than 512: any scene may try -- boxes in open space have two-valued y lines too, c over the box and "no site" beside it:
    //  y sweep 1.73 -> 1.48 ms at 1024^3, shells 1.12 -> 1.00; at 512^3 the path costs what the search costs there: shells +8 %.  The habit
    //  closes the gate on scenes that do not qualify either way.)
    const bool floorlike = STAGE == 2 && a.row_bits && a.flat_on && (w19 == 0u || (LC ? LC : a.L) > 512);
    const bool flat_gate = floorlike && (a.flat_on == 2 || wgate != 0u);
    if (a.probe_stride > 0 && a.i32_flag && i32) { probe_done(a); return; }       // the x tier is already decided: no probe
    if (a.probe_stride > 0 && STAGE == 3 && a.decide_small &&
        __hip_atomic_load(a.decide_small + 18, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { probe_done(a); return; }   // ... settled by the y probe
    const int L = LC ? LC : a.L, B = LC ? dc_clog2(LC ? LC : 2) : a.B, pitch = LC ? ((LC + 63) / 64) * 64 + 2 : a.pitch, M = LC ? (LC + 7) / 8 : a.M,
              h = LC ? (LC + 1) / 2 : a.h;
    const int MA = (L + 63) >> 6;
    const uint32_t mask = (1u << B) - 1u, finf = a.finf;
    // position multipliers -(2 p') << B (|.| < 2^23: 24-bit multiplies), as shifts of h - p
    auto ncof = [&](int p) -> int { return (int)((uint32_t)(h - p) << (B + 1)); };
    uint32_t* const keys = dc_smem;                             // [16][pitch]
    uint32_t* const args = keys + NL * pitch;                   // [M + 2][16]  best value (distance << B | argmin) of coarse position 8 i
    uint32_t* const misc = args + (M + 2) * NL;                 // per wave: [0..7] span lo, [8..15] span hi, [16..23] smallest site value; [24] filled voxels listed, [25] second pass wanted, [26..28] probe
    uint32_t* const fl_mn = misc + 48;                          // [16] per line: smallest non-zero entry MINUS ONE (0xFFFFFFFF: none), ...
    uint32_t* const fl_mx = misc + 64;                          // [16] ... largest entry
    uint32_t* const flist = misc + kDcMisc;                         // [kDcLocalFilled] filled voxels of pass 0: line << 28 | p << 12 | min(S, kDcLocalSat)
    const int t = threadIdx.x;
#ifdef SDFGPU_PHASE_CLOCKS
    unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
    int dbgt = 0;
    (void)dbgt;
#endif
    if (a.ran_flag && blockIdx.x == 0 && t == 0) *a.ran_flag = 1u;
    int mxF = 0, mxQ = 0;
    const uint32_t nblk = LOOP ? (uint32_t)a.ntiles : gridDim.x;          // "virtual" workgroups = tiles
    uint32_t vblk = blockIdx.x;
    do {

    // tile of this workgroup.  Four consecutive tiles share cache lines (16 lines x 2 B = 32 B of a 128-B line), so an XCD
    // (workgroups are dealt to the 8 XCDs round-robin, each with its own L2) gets runs of 4 consecutive tiles; the runs
    // themselves are dealt round-robin, so that every XCD sees every part of the grid (work is not uniform in space)
    int64_t tile = vblk;
    const bool probe = !LOOP && a.probe_stride > 0;
    if (probe) {
        tile = (int64_t)blockIdx.x * a.probe_stride + (blockIdx.x * 7u) % (uint32_t)a.probe_stride;
    } else if ((nblk & 31u) == 0u) {
        const uint32_t xcd = vblk & 7u, seq = vblk >> 3;
        // (round 6 measured the runs in a scattered -- multiplicatively permuted -- order, so that the ~1000 tiles in flight spread over
        //  the whole plane instead of one 64 KB window: x sweep +3 % at 512^3, -1 % at 1024^3, profiles/r06_tile_order_ab.txt.  Not kept.)
        tile = ((int64_t)(seq >> 2) * 8 + xcd) * 4 + (seq & 3u);
    }
    const int64_t o = tile / a.tiles_per_outer;
    const int64_t c0 = (tile - o * a.tiles_per_outer) * NL;
    uint32_t rwm = 0xFFFFFFFFu;             // STAGE 2, plane sparsity: bit j = the j-th row this lane stages holds a filled voxel
    bool masked = false;                    // ... and some bit of the tile is clear (block-uniform); STAGE 3: some x-plane was skipped
    if constexpr (STAGE == 2) {
        // an x-plane without a single filled voxel: every voxel of it is free with no site in its plane, and the x sweep will know
        if (rows_known) {
            const int pstate = a.plane_any[o];                  // (block-uniform, a scalar load)
            if (pstate == 0) { if constexpr (LOOP) continue; else return; }
            masked = pstate == 1;                               // (2: every row holds a filled voxel -- nothing to look up, nothing to replace)
            // the plane's row mask through scalar loads (uniform address), then one bit per row this lane will stage: row PP * j + t / LPR
            constexpr int kLPR = NL / 4, kPP = NT / kLPR, kWPJ = kPP / 32;       // words of the mask per staging step (2 at 256 lanes, 4 at 512)
            static_assert(kPP % 32 == 0, "a staging step covers whole mask words");
            const uint32_t* __restrict__ pm = a.row_bits + o * a.row_words;
            const int r0 = (int)threadIdx.x / kLPR, q0 = r0 >> 5, b0 = r0 & 31;
            if (masked) rwm = 0u;
#pragma unroll
            for (int j = 0; j < 32 / kWPJ; ++j) {                              // (up to 32 mask words = lines of up to 1024 voxels)
                if (masked && j * kPP < a.L) {
                    uint32_t word = 0u;
#pragma unroll
                    for (int q = 0; q < kWPJ; ++q) {
                        const int wi = kWPJ * j + q;                            // (a compile-time index: the word sits in a scalar register)
                        const uint32_t m = wi < a.row_words ? pm[wi] : 0u;
                        word = q0 == q ? m : word;
                    }
                    // (rows past the end of the line re-read the last row when they are staged: give them its bit)
                    const int last = a.L - 1;
                    const uint32_t lastbit = (pm[last >> 5] >> (last & 31)) & 1u;
                    const uint32_t bit = (kPP * j + r0 < a.L) ? ((word >> b0) & 1u) : lastbit;
                    rwm |= bit << j;
                }
            }
        }
    }
    // two-valued tiles (see EnvDcArgs::flat_on): tried when every z row of the grid holds a filled voxel -- a floor -- or the lines are long (block-uniform)
    bool flat_try = false;
    uint32_t flat_hash = 0u;
    if constexpr (STAGE == 2 && NL == 16 && VEC && !LOOP) {
        // (samples by a hash of the tile index: tile & 63 == 0 is the same 16 lines of z in every second plane -- the floor's own tiles)
        flat_hash = (uint32_t)tile * 2654435761u;
        flat_try = floorlike && M <= 4 * S && L <= 2040 && (flat_gate || (flat_hash >> 26) == 0u);
    }
    // (STAGE 3 was tried too -- enabled when no z row of the grid is all free -- with one value per line: 27 % of the room's x tiles
    //  qualify, and the x sweep got 2 % SLOWER: its search is the smaller part of its time and the test is paid by every tile.  Not kept.)
    const int nvalid = (int)min((int64_t)NL, a.group_lines - c0);     // lines of this tile that exist
    const int64_t base = o * a.outer_stride + c0;               // element index of (line 0, position 0)
    const uint32_t ls = (uint32_t)a.line_stride;                // (the launcher guarantees nx*ny*nz < 2^31: 32-bit element offsets)
    const int16_t* const in16 = a.in16 + base;
    const int32_t* const side_in = STAGE == 3 ? a.side_in + base : nullptr;
    const int32_t* const in32 = (STAGE == 3 && a.in_i32 && i32) ? a.in_i32 + base : nullptr;
    const bool out32 = STAGE == 2 && a.out_i32 && i32;

    auto byz_of = [&](int line) -> int {        // virtual-border distance of a line to the padded layer over y and z
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
    const int lineT = t & (NL - 1), slotT = t / NL;             // (line, slot) mapping of levels B and C
    const bool lineT_ok = lineT < nvalid;
    const int byz = byz_of(lineT);

    // exact signed value of voxel q of `line`, re-read from global memory (rare paths only)
    auto raw_signed = [&](int line, int q) -> int {
        const uint32_t idx = (uint32_t)line + (uint32_t)q * ls;
        if (STAGE == 3 && planes && a.plane_any[q] == 0) return kInf32;     // (a plane the y sweep skipped)
        if (STAGE == 3 && in32) return in32[idx];
        int v;
        if (STAGE == 2 && !VEC && a.bits) v = zdist_from_bits(a.bits + (o * a.ny + q) * a.nzw, a.nzw, (int)a.nz, (int)c0 + line);
        else if (STAGE == 2 && rows_known && ((a.row_bits[o * a.row_words + (q >> 5)] >> (q & 31)) & 1u) == 0u) v = kInf16;   // (a row the z sweep did not write: all free)
        else v = in16[idx];
        if constexpr (STAGE == 2) {
            const int g = abs(v);
            const int sq = g >= kInf16 ? kInf32 : g * g;
            return v < 0 ? -sq : sq;
        } else {
            if (abs(v) >= kSat16) v = side_in[idx];
            return v;
        }
    };
    // finish one FILLED voxel of pass 0's local search (rare path; the chunk phase has its own, vectorised finish)
    auto emit_filled = [&](uint32_t oi, int p, int D, int b_yz) {
        if constexpr (STAGE == 2) {
            if (out32) { (a.out_i32 + base)[oi] = -D; return; }
            (reinterpret_cast<int16_t*>(a.out) + base)[oi] = (int16_t)(-imin(D, kSat16));
            (a.side_out + base)[oi] = -D;
        } else {
            if (a.vb) {
                int b = b_yz;
                if (a.nx > 1) b = imin(b, (int)min((int64_t)p + 1, a.nx - p));
                if (b < 32768) D = imin(D, (int)__umul24((uint32_t)b, (uint32_t)b));
            }
            mxQ = imax(mxQ, D);
            const float f = (D >= kInf32) ? __builtin_inff() : (float)(sqrt_exact_pos((double)D) * a.resolution);
            (reinterpret_cast<float*>(a.out) + base)[oi] = -f;
        }
    };
    int probe_far = 0, probe_tot = 0, probe_mid = 0, probe_x9 = 0;

    // THE inner loop: 8 positions (multipliers nc[k] = -((2 p'_k) << B)) against the candidates qs, qs + step, ... <= qe
    // taken in aligned pairs (qs even).  Candidates outside the caller's range that ride along in a pair are harmless:
    // any candidate's value is an upper bound of the optimum, and whatever wins is a true argmin of the position.
    auto scan8 = [&](const uint32_t* kl, int qs, int qe, int step, const int (&nc)[8], uint32_t (&best)[8]) {
        int qc = qs - h;
        const uint32_t* kp = kl + qs;
        for (int q = qs; q <= qe; q += step) {
            const uint2 kk = *reinterpret_cast<const uint2*>(kp);
            const int qc1 = qc + 1;
#pragma unroll
            for (int k = 0; k < 8; ++k) best[k] = umin(best[k], umin(mad_i24(qc, nc[k], kk.x), mad_i24(qc1, nc[k], kk.y)));
            kp += step;
            qc += step;
            DC_TRIP;
        }
    };

    // The same scan under WAVE-UNIFORM control (round 4).  A wave issues a trip of the loop for all 64 lanes while any lane
    // has candidates left, so the few long ranges of a tile -- the intervals in which the argmin jumps from one object to
    // another: 1.4 % of the two-box scene's chunks hold a third of the useful trips -- used to cost every lane of their wave
    // their full length.  The lanes of a wave are 4 rows of 16 (the tile's lines x 4 slots), and its units of work are groups
    // of `urows` rows (a chunk of level C = one row; an interval of level B = Hs rows).  scan8_calm runs trips in the lanes,
    // in blocks of 4, 8, 16, ..., until after a block at most `kmax` units are still active and one of them has 6 trips or
    // more to go, and coop8 spreads the rest of each such unit over all 4 rows (every row takes every 4th pair of the unit's
    // line), reduces the partial minima across the rows (16 ds_bpermute) and hands them to the unit's first row.  Exact for
    // the same reason as any other split of a scan: the candidate set of a unit is unchanged (tools/envelope_dc_model.py,
    // scan_unit, restates the hand-over on the CPU).  Both are only entered by a wave that holds a long range at all (one
    // ballot; noise-like scenes, whose ranges are all alike, take the plain scan8).
    auto scan8_calm = [&](const uint32_t* kl, int& q, int qe, int step, const int (&nc)[8], uint32_t (&best)[8], auto urows_tag, int kmax) {
        constexpr int urows = decltype(urows_tag)::value;
        int qc = q - h, blk = 4 * step;
        const uint32_t* kp = kl + q;
        for (;;) {
            const int qcap = imin(qe, q + blk - step);          // a block of trips (4, 8, 16, ...) in the lanes: the plain loop
            for (; q <= qcap; q += step) {
                const uint2 kk = *reinterpret_cast<const uint2*>(kp);
                const int qc1 = qc + 1;
#pragma unroll
                for (int k = 0; k < 8; ++k) best[k] = umin(best[k], umin(mad_i24(qc, nc[k], kk.x), mad_i24(qc1, nc[k], kk.y)));
                kp += step;
                qc += step;
                DC_TRIP;
            }
            const uint64_t act = __ballot(q <= qe);             // (wave-uniform from here)
            if (act == 0ull) return;
            if (__ballot(q + 5 * step <= qe) != 0ull) {         // somebody has 6 trips or more to go
                const uint32_t alo = (uint32_t)act, ahi = (uint32_t)(act >> 32);
                int units;
                if constexpr (urows == 1) units = ((alo & 0xFFFFu) != 0) + ((alo >> 16) != 0) + ((ahi & 0xFFFFu) != 0) + ((ahi >> 16) != 0);
                else units = (alo != 0) + (ahi != 0);
                if (units <= kmax) return;
            }
            blk *= 2;
        }
    };
    // pos_step: positions between the first positions of consecutive units (their multipliers differ by pos_step << (B + 1))
    auto coop8 = [&](const uint32_t* kl, int& q, int qe, const int (&nc)[8], uint32_t (&best)[8], auto urows_tag, int pos_step) {
        constexpr int urows = decltype(urows_tag)::value;
        const int lane = t & 63, row = lane >> 4;
        uint64_t act = __ballot(q <= qe);
        while (act != 0ull) {                                   // (wave-uniform)
            const int r0 = ((__ffsll((unsigned long long)act) - 1) >> 4) & ~(urows - 1);     // first row of the first active unit
            const int src = (lane & 15) + 16 * r0;              // its lane of my line: the share that is furthest behind
            const int bq = __shfl(q, src), bqe = __shfl(qe, src);
            const int d = ((r0 - (row & ~(urows - 1))) / urows) * pos_step * (1 << (B + 1));
            int ncb[8];
            uint32_t hb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { ncb[k] = nc[k] - d; hb[k] = 0xFFFFFFFFu; }
            scan8(kl, bq + 2 * row, bqe, 8, ncb, hb);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hb[k] = umin(hb[k], (uint32_t)__shfl_xor((int)hb[k], 16));
                hb[k] = umin(hb[k], (uint32_t)__shfl_xor((int)hb[k], 32));
            }
            if (row == r0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) best[k] = umin(best[k], hb[k]);
            }
            if ((row & ~(urows - 1)) == r0) q = qe + 2;         // (this unit is done)
            act &= ~((urows == 1 ? 0xFFFFull : 0xFFFFFFFFull) << (16 * r0));
        }
    };

    // one pass over the tile.  CLS 0: sites of "distance to filled" (results for free voxels); 1: the reverse, on the
    // negated field.
    auto run_pass = [&](auto cls_tag) {
        constexpr int cls = decltype(cls_tag)::value;
        for (int i = t; i < (M + 2) * NL; i += NT) args[i] = 0xFFFFFFFFu;
        if (cls == 0 && t < kDcMisc) misc[t] = (t >= 48 && t < 64) ? 0xFFFFFFFFu : 0u;

        // ---- stage the tile: rows -> keys ------------------------------------------------------------------------------------
        // A lane reads 4 lines x 1 position per load (VEC: one 8 / 16-byte load), NB loads in flight, and writes the
        // four keys.
        const int sub = t & (LPR - 1), r = t / LPR;
        const bool sparse = STAGE == 3 ? planes : masked;       // plane sparsity has something to say about this tile's rows (block-uniform)
        uint32_t* const kb = keys + (4 * sub) * pitch + r;
        int lsel[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lsel[k] = imin(4 * sub + k, nvalid - 1);
        const uint32_t off_r = (uint32_t)r * ls + 4u * sub, off_last = (uint32_t)(L - 1) * ls + 4u * sub;
        int lo_w = 0x7fffffff, hi_w = -1;                       // span seen by this wave (uniform)
        uint32_t mt = 0xFFFFFFFFu;                              // smallest site value seen by this lane
        uint32_t fmn[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, fmx[4] = {0u, 0u, 0u, 0u};   // flat-positive test: per line of this lane
        for (int pb = 0; pb < L; pb += PP * NB) {
            int sv[NB][4];
            // plane sparsity: the verdicts of this lane's rows -- STAGE 3: the row's x-plane holds a filled voxel (the y sweep skipped the
            // others), STAGE 2: the z row does (the z sweep did not write the others) -- requested before the row loads (all "present"
            // without that knowledge)
            uint32_t pw[NB];
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                pw[it] = 1u;
                const int prow = imin(pb + PP * it + r, L - 1);
                if constexpr (STAGE == 3) {
                    if (planes) pw[it] = a.plane_any[prow] != 0;
                } else {
                    pw[it] = rwm >> ((pb / PP) + it);                       // STAGE 2: z rows without a filled voxel were not written
                    (void)prow;
                }
            }
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                // (rows past the end of the line re-read the last row; their keys are not written)
                const uint32_t off = (pb + PP * it + r < L) ? off_r + (uint32_t)(pb + PP * it) * ls : off_last;
                if constexpr (VEC) {
                    // (plane sparsity: a lane whose row is known -- an x-plane / z row without a filled voxel -- does not branch around its
                    //  load, which would put a wait behind every one of the NB loads: it loads the tile's first row instead, one cached
                    //  line for all such lanes, and the value is replaced below)
                    const uint32_t offl = (!sparse || (pw[it] & 1u)) ? off : 4u * (uint32_t)sub;
                    if (STAGE == 3 && in32) {
                        const int4 e = *reinterpret_cast<const int4*>(in32 + offl);
                        sv[it][0] = e.x; sv[it][1] = e.y; sv[it][2] = e.z; sv[it][3] = e.w;
                    } else {
                        const uint2 raw = *reinterpret_cast<const uint2*>(in16 + offl);
                        sv[it][0] = (int)raw.x; sv[it][1] = (int)raw.y;         // (unpacked below, once every row is requested)
                    }
                } else if (STAGE == 2 && a.bits) {
                    // stand-by: z distances straight from the dense tier's bit field.  The LPR = 4 lanes of a tile row are one
                    // DPP quad (sub = t & 3) and the tile's 16 lines lie in ONE word of the bit row (c0 is a multiple of 16)
                    static_assert(LPR == 4 && NL == 16, "the lanes of a tile row must be one DPP quad");
                    static_assert(NT % 64 == 0, "zrow_scan_quad's quad permutes need whole waves");
#ifdef SDFGPU_DEBUG_HOOKS
                    if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();     // (a partial wave would read zeros from its inactive lanes)
#endif
                    const int prow = (pb + PP * it + r < L) ? pb + PP * it + r : L - 1;
                    const int w0 = (int)(c0 >> 5);
                    const ZRowBits zr = zrow_scan_quad(a.bits + (o * a.ny + prow) * a.nzw, a.nzw, w0, sub);
#pragma unroll
                    for (int k = 0; k < 4; ++k) sv[it][k] = zdist_from_row(zr, w0, (int)a.nz, (int)c0 + lsel[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t idx = off - 4u * sub + (uint32_t)lsel[k];
                        int v;
                        if (STAGE == 3 && in32) v = (pw[it] & 1u) ? in32[idx] : kInf32;
                        else if (STAGE == 2 && !(pw[it] & 1u)) v = kInf16;
                        else {
                            v = in16[idx];
                            if constexpr (STAGE == 3) if (abs(v) >= kSat16) v = side_in[idx];
                        }
                        sv[it][k] = v;
                    }
                }
            }
            if (pb == 0) __syncthreads();                       // args / misc initialised (the loads above are in flight meanwhile)
            if constexpr (VEC) {
                if (STAGE == 3 && in32) {
                    if (sparse) {                               // (block-uniform)
#pragma unroll
                        for (int it = 0; it < NB; ++it)
                            if (!(pw[it] & 1u)) { sv[it][0] = kInf32; sv[it][1] = kInf32; sv[it][2] = kInf32; sv[it][3] = kInf32; }
                    }
                } else {
                    if (STAGE == 2 && sparse) {                 // (block-uniform)
#pragma unroll
                        for (int it = 0; it < NB; ++it)
                            if (!(pw[it] & 1u)) { sv[it][0] = (int)0x7fff7fffu; sv[it][1] = (int)0x7fff7fffu; }     // +32767 x 4
                    }
#pragma unroll
                    for (int it = 0; it < NB; ++it) {
                        const uint32_t rx = (uint32_t)sv[it][0], ry = (uint32_t)sv[it][1];
                        sv[it][0] = (int)(short)(rx & 0xffffu); sv[it][1] = (int)rx >> 16;
                        sv[it][2] = (int)(short)(ry & 0xffffu); sv[it][3] = (int)ry >> 16;
                        if constexpr (STAGE == 3) {
                            // a 16-bit lane is saturated iff it holds +-32767 (-32768 is never stored): |v| + 1 has bit 15 set
                            const uint32_t n0 = (rx >> 15) & 0x00010001u, n1 = (ry >> 15) & 0x00010001u;
                            const uint32_t m0 = (rx ^ (n0 * 0xFFFFu)) + 0x00010001u + n0;
                            const uint32_t m1 = (ry ^ (n1 * 0xFFFFu)) + 0x00010001u + n1;
                            if (((m0 | m1) & 0x80008000u) != 0u) {
                                const uint32_t off = (pb + PP * it + r < L) ? off_r + (uint32_t)(pb + PP * it) * ls : off_last;
                                const int4 e = *reinterpret_cast<const int4*>(side_in + off);
                                sv[it][0] = e.x; sv[it][1] = e.y; sv[it][2] = e.z; sv[it][3] = e.w;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                const int p = pb + PP * it + r;
                const bool inl = p < L;
                const int pc = p - h;
                const uint32_t cpos = (mad_i24(pc, pc, (uint32_t)(h * h)) << B) | (uint32_t)p;
                uint32_t F[4];
                int neg = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int s1 = cls == 0 ? sv[it][k] : -sv[it][k];
                    if constexpr (STAGE == 2) {
                        const uint32_t m = (uint32_t)imax(s1, 0);
                        F[k] = umin(__umul24(m, m), finf);
                    } else {
                        F[k] = (uint32_t)imin(imax(s1, 0), (int)finf);
                    }
                    neg |= s1;
                }
                if (inl) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) kb[k * pitch + (pb + PP * it)] = (F[k] << B) + cpos;
                }
                if (cls == 0 && flat_try) {                     // (block-uniform) smallest non-zero and largest entry of each line
#pragma unroll
                    for (int k = 0; k < 4; ++k) {               // (rows past the end of the line re-read its last row: harmless here)
                        fmx[k] = umax(fmx[k], F[k]);
                        fmn[k] = umin(fmn[k], F[k] - 1u);       // F - 1: a zero entry wraps to the identity of the minimum
                    }
                }
                const uint32_t fmin4 = inl ? umin(umin(F[0], F[1]), umin(F[2], F[3])) : finf;
                mt = umin(mt, fmin4);
                const uint64_t bal = __ballot(fmin4 < finf);
                if (bal) {                                      // (wave-uniform: scalar code)
                    const int pw = pb + PP * it + (t >> 6) * (64 / LPR);
                    lo_w = imin(lo_w, pw + ((__ffsll((unsigned long long)bal) - 1) >> LPR_SH));
                    hi_w = imax(hi_w, pw + ((63 - __clzll((long long)bal)) >> LPR_SH));
                }
                if (cls == 0 && neg < 0 && inl) {               // filled voxels (rare in the scenes this kernel serves): list them
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int s1 = sv[it][k];
                        if (s1 < 0 && 4 * sub + k < nvalid) {
                            uint32_t S = (uint32_t)(-s1);       // squared distance to the nearest free voxel so far
                            if constexpr (STAGE == 2) S = S >= (uint32_t)kInf16 ? (uint32_t)kInf32 : __umul24(S, S);
                            const uint32_t e = atomicAdd(&misc[24], 1u);
                            if (e < (uint32_t)kDcLocalFilled) flist[e] = ((uint32_t)(4 * sub + k) << 28) | ((uint32_t)p << 12) | umin(S, (uint32_t)kDcLocalSat);
                        }
                    }
                }
            }
        }
        if (cls == 0 && flat_try) {                             // lanes sub, sub + LPR, ... of a wave hold the same 4 lines
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    fmn[k] = umin(fmn[k], (uint32_t)__shfl_xor((int)fmn[k], off));
                    fmx[k] = umax(fmx[k], (uint32_t)__shfl_xor((int)fmx[k], off));
                }
            }
            if ((t & 63) < LPR) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { atomicMin(&fl_mn[4 * sub + k], fmn[k]); atomicMax(&fl_mx[4 * sub + k], fmx[k]); }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mt = umin(mt, (uint32_t)__shfl_xor((int)mt, off));
        if ((t & 63) == 0) { misc[t >> 6] = (uint32_t)lo_w; misc[8 + (t >> 6)] = (uint32_t)hi_w; misc[16 + (t >> 6)] = mt; }
        if (t < 2 * NL) {                                       // two sentinels behind every line (pairs are read 8-byte aligned)
            const int line = t >> 1, q = L + (t & 1), qc = q - h;
            keys[line * pitch + q] = ((finf + (uint32_t)(qc * qc) + (uint32_t)(h * h)) << B) | ((uint32_t)q & mask);
        }
        __syncthreads();
        DC_STAMP(0);
        int lo_t = 0x7fffffff, hi_t = -1;
        uint32_t mt_t = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < NW; ++w) { lo_t = imin(lo_t, (int)misc[w]); hi_t = imax(hi_t, (int)misc[8 + w]); mt_t = umin(mt_t, misc[16 + w]); }
#ifdef SDFGPU_DEBUG_HOOKS
        const bool act = lo_t <= hi_t && !(a.dbg & 1);          // profiling builds: dbg bit 0 = no search (wrong results), bit 1 = no fp64 finish, bit 2 = no stores
#else
        const bool act = lo_t <= hi_t;                          // the tile holds a site (block-uniform)
#endif
        // Two-valued tiles.  Every lane classifies the positions of its chunks against its line's smallest non-zero entry mn and
        // largest entry mx: zero site, mn site, mx site -- or something else, and then the tile is searched as usual (the keys
        // are untouched until the verdict; tracking + classification cost a searched tile ~15 %: hence the habit, EnvDcArgs::flat_score).
        bool flat = false, two = false;
        constexpr bool kAllIn = LC != 0 && LC % 8 == 0;          // (every position of every chunk lies inside the line)
        uint32_t zm = 0u, lm = 0u;                              // zero sites / mn sites of this lane's chunks, 8 bits per round of level C
        if (cls == 0 && flat_try && act) {
            const uint32_t* kl = keys + lineT * pitch;
            const uint32_t mn1 = fl_mn[lineT], mx = fl_mx[lineT];
            const uint32_t mn = mn1 == 0xFFFFFFFFu ? mn1 : mn1 + 1u;
            // does any line hold two values at all?  (block-uniform: every lane reads the same 32 words.)  A tile of one-valued lines --
            // half of the room's -- does without the mn sites: mn = mx, and min(mx, d0^2) is the whole answer
#pragma unroll
            for (int l = 0; l < NL; ++l) { const uint32_t a1 = fl_mn[l] + 1u, b1 = fl_mx[l]; two = two || (b1 != 0u && a1 != b1); }
            bool other = false;
            int n = 0;
#ifdef SDFGPU_DEBUG_HOOKS
            if (a.dbg & 1024) {} else                   // profiling builds: bit 10 = no classification (wrong results)
#endif
            if (!two) {
                for (int i0 = 0; i0 < M; i0 += S, ++n) {
                    const int i = i0 + slotT;
                    if (i >= M) continue;
                    const int p0 = 8 * i, pc0 = p0 - h;
                    uint32_t cz = mad_i24(pc0, pc0, (uint32_t)(h * h));
                    int inc = 2 * pc0 + 1;
                    uint32_t z8 = 0u;
                    if (kAllIn || p0 + 8 <= L) {
                        // whole chunk inside the line: arithmetic instead of compares -- min(e, 1) is the "not a zero site" bit, and
                        // min(e, e ^ mx) is non-zero iff the entry is neither 0 nor mx
                        uint32_t nz = 0u, bad = 0u;
#pragma unroll
                        for (int k = 0; k < 8; k += 2) {
                            const uint2 kk = *reinterpret_cast<const uint2*>(kl + p0 + k);
                            const uint32_t f0 = (kk.x >> B) - cz;
                            cz += (uint32_t)inc; inc += 2;
                            const uint32_t f1 = (kk.y >> B) - cz;
                            cz += (uint32_t)inc; inc += 2;
                            nz |= (umin(f0, 1u) << k) | (umin(f1, 1u) << (k + 1));
                            bad |= umin(f0, f0 ^ mx) | umin(f1, f1 ^ mx);
                        }
                        z8 = ~nz & 0xFFu;
                        other = other || bad != 0u;
                    } else {
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        const uint2 kk = *reinterpret_cast<const uint2*>(kl + p0 + k);
                        const uint32_t f0 = (kk.x >> B) - cz;
                        cz += (uint32_t)inc; inc += 2;
                        const uint32_t f1 = (kk.y >> B) - cz;
                        cz += (uint32_t)inc; inc += 2;
                        const bool in0 = p0 + k < L, in1 = p0 + k + 1 < L;
                        z8 |= ((f0 == 0u && in0) ? 1u : 0u) << k;
                        z8 |= ((f1 == 0u && in1) ? 1u : 0u) << (k + 1);
                        other = other || (in0 && f0 != 0u && f0 != mx) || (in1 && f1 != 0u && f1 != mx);
                    }
                    }
                    zm |= z8 << (8 * n);
                }
            } else
            for (int i0 = 0; i0 < M; i0 += S, ++n) {
                const int i = i0 + slotT;
                if (i >= M) continue;
                const int p0 = 8 * i, pc0 = p0 - h;
                uint32_t cz = mad_i24(pc0, pc0, (uint32_t)(h * h));        // the position's own term (p - h)^2 + h^2: F = (key >> B) - cz
                int inc = 2 * pc0 + 1;
                uint32_t z8 = 0u, l8 = 0u;
                if (kAllIn || p0 + 8 <= L) {
                    uint32_t nz = 0u, nl = 0u, bad = 0u;
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        const uint2 kk = *reinterpret_cast<const uint2*>(kl + p0 + k);
                        const uint32_t f0 = (kk.x >> B) - cz;
                        cz += (uint32_t)inc; inc += 2;
                        const uint32_t f1 = (kk.y >> B) - cz;
                        cz += (uint32_t)inc; inc += 2;
                        const uint32_t x0 = f0 ^ mn, x1 = f1 ^ mn;
                        nz |= (umin(f0, 1u) << k) | (umin(f1, 1u) << (k + 1));
                        nl |= (umin(x0, 1u) << k) | (umin(x1, 1u) << (k + 1));
                        bad |= umin(umin(f0, x0), f0 ^ mx) | umin(umin(f1, x1), f1 ^ mx);
                    }
                    z8 = ~nz & 0xFFu;
                    l8 = ~nl & 0xFFu;
                    other = other || bad != 0u;
                } else {
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const uint2 kk = *reinterpret_cast<const uint2*>(kl + p0 + k);
                    const uint32_t f0 = (kk.x >> B) - cz;
                    cz += (uint32_t)inc; inc += 2;
                    const uint32_t f1 = (kk.y >> B) - cz;
                    cz += (uint32_t)inc; inc += 2;
                    const bool in0 = p0 + k < L, in1 = p0 + k + 1 < L;
                    z8 |= ((f0 == 0u && in0) ? 1u : 0u) << k;
                    z8 |= ((f1 == 0u && in1) ? 1u : 0u) << (k + 1);
                    l8 |= ((f0 == mn && in0) ? 1u : 0u) << k;
                    l8 |= ((f1 == mn && in1) ? 1u : 0u) << (k + 1);
                    other = other || (in0 && f0 != 0u && f0 != mn && f0 != mx) || (in1 && f1 != 0u && f1 != mn && f1 != mx);
                }
                }
                zm |= z8 << (8 * n);
                lm |= l8 << (8 * n);
            }
            flat = __syncthreads_and(other ? 0 : 1) != 0;      // (block-uniform; everybody has read its keys)
            if (a.flat_on == 1 && (flat_hash >> 28) == 0u && t == 0)     // this tile's vote (every 16th, and every one of the 64th that try with the gate down)
                (void)__hip_atomic_fetch_add(reinterpret_cast<int*>(a.flat_score), flat ? -3 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            DC_STAMP(1);
#ifdef SDFGPU_DEBUG_HOOKS
            if (a.dbg & 128) flat = false;
#endif
        }

        if (flat) {
            constexpr int kFar = 1 << 20;
            // the key area is free now: six arrays [16][M + 2] -- first / last zero site and mn site of every chunk, then the nearest
            // ones before / behind every chunk
            const int MP = M + 2;
            uint32_t* const cz0 = keys, * const cl0 = keys + NL * MP;
            uint32_t* const nzl = keys + 2 * NL * MP, * const nzr = keys + 3 * NL * MP, * const nll = keys + 4 * NL * MP, * const nlr = keys + 5 * NL * MP;
            {
                int n = 0;
                for (int i0 = 0; i0 < M; i0 += S, ++n) {
                    const int i = i0 + slotT;
                    if (i >= M) continue;
                    const int p0 = 8 * i;
                    const uint32_t z8 = (zm >> (8 * n)) & 0xFFu, l8 = (lm >> (8 * n)) & 0xFFu;
                    cz0[lineT * MP + i] = z8 ? ((uint32_t)(p0 + 31 - __clz((int)z8)) << 16) | (uint32_t)(p0 + __ffs((int)z8) - 1) : 0xFFFFFFFFu;
                    if (two) cl0[lineT * MP + i] = l8 ? ((uint32_t)(p0 + 31 - __clz((int)l8)) << 16) | (uint32_t)(p0 + __ffs((int)l8) - 1) : 0xFFFFFFFFu;
                }
            }
            __syncthreads();
            // per line, 16 lanes, each with a run of chunks: exclusive prefix maximum of "last" / suffix minimum of "first" over the chunks
#ifdef SDFGPU_DEBUG_HOOKS
            if (a.dbg & 512) {} else                    // profiling builds: bit 9 = no nearest-site scans (wrong results)
#endif
            if (t < 16 * NL) {
                const int line2 = t >> 4, j = t & 15;
                const int CP = (M + 15) >> 4, cb = j * CP, ce = imin(cb + CP, M);
                auto nearest = [&](const uint32_t* info, uint32_t* before, uint32_t* behind) {
                    int pre = -kFar, suf = kFar;
                    for (int c = cb; c < ce; ++c) {
                        const uint32_t w = info[line2 * MP + c];
                        if (w != 0xFFFFFFFFu) { pre = imax(pre, (int)(w >> 16)); suf = imin(suf, (int)(w & 0xFFFFu)); }
                    }
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) {
                        const int v = __shfl_up(pre, d, 16), u = __shfl_down(suf, d, 16);
                        if (j >= d) pre = imax(pre, v);
                        if (j + d < 16) suf = imin(suf, u);
                    }
                    int run = __shfl_up(pre, 1, 16), nxt = __shfl_down(suf, 1, 16);
                    if (j == 0) run = -kFar;
                    if (j == 15) nxt = kFar;
                    for (int c = cb; c < ce; ++c) {
                        before[line2 * MP + c] = (uint32_t)run;
                        const uint32_t w = info[line2 * MP + c];
                        if (w != 0xFFFFFFFFu) run = (int)(w >> 16);
                    }
                    for (int c = ce - 1; c >= cb; --c) {
                        behind[line2 * MP + c] = (uint32_t)nxt;
                        const uint32_t w = info[line2 * MP + c];
                        if (w != 0xFFFFFFFFu) nxt = (int)(w & 0xFFFFu);
                    }
                };
                nearest(cz0, nzl, nzr);
                if (two) nearest(cl0, nll, nlr);
            }
            __syncthreads();
            DC_STAMP(2);
        } else if (act) {
            // ---- level A: positions 64 i ---------------------------------------------------------------------------------------
            // Two forms, chosen per wave (= 64 / S whole lines): (a) one position per lane group over a range clipped by the
            // distance bound -- a few candidates per position wherever the line runs near sites or far from ALL of them;
            // (b) when the clipped ranges stay long (objects at different distances: the bound v - m is then large), the 16
            // lanes split the span and every lane takes its share of the candidates for 8 positions at once.
            {
                const int lineA = t / S, u = t % S;
                const uint32_t* klA = keys + lineA * pitch;
                const int span_pairs = (hi_t - (lo_t & ~1)) / 2 + 1;
                int G = 1;                                      // form (a): lanes per position
                while (2 * G * MA <= S) G *= 2;
                const int v = u & (G - 1), i1 = u / G;
                int lo = lo_t, hi = hi_t, nc1 = 0;
                bool clip = false;
                if (MA <= S) {
                    int prs = 0;
                    if (i1 < MA) {
                        const int p = 64 * i1;
                        nc1 = ncof(p);
                        // any candidate bounds the optimum: try the candidate next to p and the two ends of the span
                        const int pcl = imin(imax(p, lo_t), hi_t);
                        const uint32_t ub = umin(mad_i24(pcl - h, nc1, klA[pcl]), umin(mad_i24(lo_t - h, nc1, klA[lo_t]), mad_i24(hi_t - h, nc1, klA[hi_t])));
                        const uint32_t dub = (ub >> B) - __umul24((uint32_t)p, (uint32_t)(2 * h - p));
                        if (dub < finf) {
                            const int w = (int)__builtin_sqrtf((float)(dub - umin(mt_t, dub))) + 2;
                            lo = imax(lo, p - w);
                            hi = imin(hi, p + w);
                        }
                        prs = (hi - (lo & ~1)) / 2 + 1;
                    }
                    // cost per lane: (a) ceil(prs / G) trips of ~8 instructions, (b) ceil(span_pairs / S) trips of ~29 per 8 positions;
                    // decided per wave (whole lines: 64 / S of them)
                    clip = __all(((prs + G - 1) / G) * 8 <= ((span_pairs + S - 1) / S) * 29 * ((MA + 7) / 8));
                }
                if (clip) {
                    if (i1 < MA) {
                        uint32_t best = 0xFFFFFFFFu;
                        int q = (lo & ~1) + 2 * v, qc = q - h;
                        const uint32_t* kp = klA + q;
                        for (; q <= hi; q += 2 * G) {
                            const uint2 kk = *reinterpret_cast<const uint2*>(kp);
                            best = umin(best, umin(mad_i24(qc, nc1, kk.x), mad_i24(qc + 1, nc1, kk.y)));
                            kp += 2 * G;
                            qc += 2 * G;
                            DC_TRIP;
                        }
                        atomicMin(&args[(8 * i1) * NL + lineA], best);
                    }
                } else {
                    for (int g = 0; g < MA; g += 8) {
                        int nc[8];
                        uint32_t best[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            nc[k] = ncof(64 * imin(g + k, MA - 1));
                            best[k] = 0xFFFFFFFFu;
                        }
                        scan8(klA, (lo_t & ~1) + 2 * u, hi_t, 2 * S, nc, best);
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (g + k < MA) atomicMin(&args[(8 * (g + k)) * NL + lineA], best[k]);
                    }
                }
            }
            __syncthreads();
            DC_STAMP(1);
            if (probe) {
                // the probe's statistic from the coarse positions alone (64 i of every line: an unbiased sample of the voxels)
                if (slotT < imin(MA, S)) {
                    const int i = slotT, p = 64 * i;
                    const uint32_t d = (args[(8 * i) * NL + lineT] >> B) - __umul24((uint32_t)p, (uint32_t)(2 * h - p));
                    const int D = d >= finf ? kInf32 : (int)d;
                    if (lineT < nvalid) {
                        probe_tot += D != 0 ? 1 : 0;
                        probe_far += D >= a.probe_thr ? 1 : 0;
                        probe_mid += D >= a.probe_thr2 ? 1 : 0;
                        probe_x9 += D >= a.probe_thr3 ? 1 : 0;
                    }
                }
                return;                                     // (from the pass; the counters are summed below)
            }
            // ---- level B: positions 64 i + 8 k inside interval i; lane = (line, interval[, share of the candidates]) ------------
            {
                int Hs = 1;                                     // lanes per interval (S slots per line)
                while (2 * Hs * MA <= S) Hs *= 2;
                const uint32_t* kl = keys + lineT * pitch;
                const bool coop = NL == 16 && Hs <= 2;          // (block-uniform: 2 or 4 intervals per wave)
                for (int ib = 0; ib < MA; ib += S / Hs) {
                    const int i = ib + slotT / Hs;
                    if (!coop && i >= MA) continue;
                    const int ic = imin(i, MA - 1);             // (coop: lanes past the last interval stay in the wave with an empty range)
                    const int u = slotT % Hs;
                    const int lo = (int)(args[(8 * ic) * NL + lineT] & mask);
                    const int hi = i >= MA ? -1 : (i + 1 < MA) ? (int)(args[(8 * (i + 1)) * NL + lineT] & mask) : hi_t;
                    int nc[8];
                    uint32_t best[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        nc[k] = ncof(64 * i + 8 * k);
                        best[k] = 0xFFFFFFFFu;
                    }
#ifdef SDFGPU_DEBUG_HOOKS
                    if (a.dbg & 16) continue;
#endif
                    if (coop && __ballot(hi - lo >= 108) != 0ull) {             // (an interval's range without a jump: <= 64 + noise)
                        int q = (lo & ~1) + 2 * u;
                        if (Hs == 1) {                          // (block-uniform)
                            scan8_calm(kl, q, hi, 2, nc, best, std::integral_constant<int, 1>{}, 2);
                            coop8(kl, q, hi, nc, best, std::integral_constant<int, 1>{}, 64);
                        } else {
                            scan8_calm(kl, q, hi, 4, nc, best, std::integral_constant<int, 2>{}, 1);
                            coop8(kl, q, hi, nc, best, std::integral_constant<int, 2>{}, 64);
                        }
                        if (i >= MA) continue;
                    } else {
                        scan8(kl, (lo & ~1) + 2 * u, hi, 2 * Hs, nc, best);
                    }
#pragma unroll
                    for (int k = 1; k < 8; ++k)
                        if (8 * i + k < M) atomicMin(&args[(8 * i + k) * NL + lineT], best[k]);
                }
            }
            __syncthreads();
            DC_STAMP(2);
        }

        if (probe) {                            // (a tile without sites: every sampled voxel is at "infinity")
            if (slotT < imin(MA, S) && lineT < nvalid) { probe_tot += 1; probe_far += 1; probe_mid += 1; probe_x9 += 1; }
            return;
        }
        // ---- level C: lane = (line, chunk of 8 positions); finish and store ---------------------------------------------------
        {
            const uint32_t* kl = keys + lineT * pitch;
            for (int i0 = 0, n = 0; i0 < M; i0 += S, ++n) {
                const int i = i0 + slotT;
                const bool mine = i < M && lineT_ok;
                if (NL != 16 && !mine) continue;                // (16-line tiles: the lane stays in its wave with an empty range)
                const int p0 = 8 * i;
                int D[8];
                if (flat) {
                    // D = min(mx, d0^2, mn + d1^2): d0 / d1 = distance along the line to the nearest zero site / mn site (a zero site
                    // costs (p - q)^2, an mn site mn + (p - q)^2, an mx site at least mx -- which the position itself attains unless it
                    // is a zero or mn site, and then one of the other two terms is already smaller)
                    const int ic = imin(i, M - 1), MP = M + 2;
                    const uint32_t z8 = (zm >> (8 * n)) & 0xFFu, l8 = (lm >> (8 * n)) & 0xFFu;
                    const uint32_t mn1 = fl_mn[lineT], mx = fl_mx[lineT];
                    const uint32_t mn = mn1 == 0xFFFFFFFFu ? mn1 : mn1 + 1u;
                    int lz = (int)keys[(2 * NL + lineT) * MP + ic], rz = (int)keys[(3 * NL + lineT) * MP + ic];
                    int ll = 0, rl = 0;
                    if (two) { ll = (int)keys[(4 * NL + lineT) * MP + ic]; rl = (int)keys[(5 * NL + lineT) * MP + ic]; }
                    // (a chunk-level shortcut -- no site inside, the nearest ones too far to matter: one value for all eight positions,
                    //  decided per wave -- was measured: y sweep +2 ... 3 %; the wall at low y reaches half of the room's chunks)
#ifdef SDFGPU_DEBUG_HOOKS
                    if (a.dbg & 256) {                          // profiling builds: bit 8 = no distance chains (wrong results)
#pragma unroll
                        for (int k = 0; k < 8; ++k) D[k] = mx >= finf ? kInf32 : (int)mx;
                    } else
#endif
                    if (!two) {                                 // (block-uniform) one value per line
                        int dz[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            if ((z8 >> k) & 1u) lz = p0 + k;
                            dz[k] = p0 + k - lz;
                        }
#pragma unroll
                        for (int k = 7; k >= 0; --k) {
                            if ((z8 >> k) & 1u) rz = p0 + k;
                            const int d0 = imin(dz[k], rz - (p0 + k));
                            const uint32_t t0 = d0 > 2047 ? 0xFFFFFFFFu : __umul24((uint32_t)d0, (uint32_t)d0);
                            const uint32_t d = umin(mx, t0);
                            D[k] = d >= finf ? kInf32 : (int)d;
                        }
                    } else {
                    int dz[8], dl[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if ((z8 >> k) & 1u) lz = p0 + k;
                        if ((l8 >> k) & 1u) ll = p0 + k;
                        dz[k] = p0 + k - lz;
                        dl[k] = p0 + k - ll;
                    }
#pragma unroll
                    for (int k = 7; k >= 0; --k) {
                        if ((z8 >> k) & 1u) rz = p0 + k;
                        if ((l8 >> k) & 1u) rl = p0 + k;
                        const int d0 = imin(dz[k], rz - (p0 + k)), d1 = imin(dl[k], rl - (p0 + k));
                        const uint32_t t0 = d0 > 2047 ? 0xFFFFFFFFu : __umul24((uint32_t)d0, (uint32_t)d0);
                        const uint32_t t1 = d1 > 2047 ? 0xFFFFFFFFu : mn + __umul24((uint32_t)d1, (uint32_t)d1);
                        const uint32_t d = umin(mx, umin(t0, t1));
                        D[k] = d >= finf ? kInf32 : (int)d;
                    }
                    }
                    DC_STAMP(3);
                } else if (act) {
                    const int ic = imin(i, M - 1);
                    const int a0 = (int)(args[ic * NL + lineT] & mask);
                    const int a8 = !mine ? -1 : (i + 1 < M) ? (int)(args[(i + 1) * NL + lineT] & mask) : hi_t;
                    int nc[8];
                    uint32_t best[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        nc[k] = ncof(p0 + k);
                        best[k] = 0xFFFFFFFFu;
                    }
                    DC_STAMP(4);
#ifdef SDFGPU_DEBUG_HOOKS
                    if (a.dbg & 8) {}                           // profiling builds: bit 3 = no level-C scan, bit 4 = no level-B scan (use with bit 3)
                    else
#endif
                    if (NL == 16 && __ballot(a8 - a0 >= 34) != 0ull) {          // (a chunk's range without a jump: <= 8 + noise)
                        int q = a0 & ~1;
                        scan8_calm(kl, q, a8, 2, nc, best, std::integral_constant<int, 1>{}, 2);
                        coop8(kl, q, a8, nc, best, std::integral_constant<int, 1>{}, 8);
                    } else {
                        scan8(kl, a0 & ~1, a8, 2, nc, best);
                    }
                    DC_STAMP(3);
                    // D = (best >> B) - (h^2 - p'^2), h^2 - p'^2 = p (2 h - p): a running value, + (2 h - 2 p - 1) per position
                    uint32_t hp = __umul24((uint32_t)p0, (uint32_t)(2 * h - p0));
                    const uint32_t c1 = (uint32_t)(2 * h - 2 * p0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t d = (best[k] >> B) - hp;
                        D[k] = d >= finf ? kInf32 : (int)d;
                        hp += c1 - (uint32_t)(2 * k + 1);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) D[k] = kInf32;
                }
                if (!mine) continue;
                // positions past the end of the line count as "not mine" (D = 0).  Pass 0: a voxel is filled iff its distance
                // to the nearest filled voxel is 0; pass 1: free iff its distance to the nearest free voxel is 0 -- so in
                // both passes the voxels this pass must write are exactly those with D != 0.
                if (p0 + 8 > L) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (p0 + k >= L) D[k] = 0;
                }
                const uint32_t ob = (uint32_t)lineT + (uint32_t)p0 * ls;      // element offset of the chunk's first voxel
                if (probe) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        probe_tot += D[k] != 0 ? 1 : 0;
                        probe_far += D[k] >= a.probe_thr ? 1 : 0;
                        probe_mid += D[k] >= a.probe_thr2 ? 1 : 0;
                    }
                    continue;
                }
                if constexpr (STAGE == 2) {
                    if (out32) {
                        char* op = reinterpret_cast<char*>(a.out_i32 + base);      // (uniform base + 32-bit byte offset)
                        uint32_t bo = ob * 4u, bstep = 4u * ls;
#ifdef SDFGPU_DEBUG_HOOKS
                        if (a.dbg & 32) {                       // profiling builds: bit 5 = the tile's stores to ONE contiguous 16 x L block (wrong results)
                            op = reinterpret_cast<char*>(a.out_i32 + ((((c0 >> 4) * a.nx + o) * (int64_t)L) << 4));
                            bo = ((uint32_t)lineT + (uint32_t)p0 * 16u) * 4u;
                            bstep = 64u;
                        }
#endif
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
#ifdef SDFGPU_DEBUG_HOOKS
                            if ((a.dbg & 4) && (k || slotT)) { bo += bstep; continue; }
#endif
                            if (D[k] != 0) *reinterpret_cast<int32_t*>(op + bo) = cls == 1 ? -D[k] : D[k];
                            bo += bstep;
                        }
                    } else {
                        // side-table convention (sdfgpu_sweep_x16.hpp): if ANY voxel of a group of 4 is saturated, the exact
                        // values of the whole group must be in the side table.  A pass only knows its own class, so a voxel
                        // also writes its exact value when its group holds a voxel of the other class.
                        int16_t* const op = reinterpret_cast<int16_t*>(a.out) + base;
                        int32_t* const sp = a.side_out + base;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const bool inl = p0 + k < L, mine = D[k] != 0;
                            const int need = mine ? (D[k] >= kSat16 ? 1 : 0) : (inl ? 1 : 0);
                            int any = need | __builtin_amdgcn_mov_dpp(need, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
                            any |= __builtin_amdgcn_mov_dpp(any, 0x4E, 0xF, 0xF, true);                  // quad_perm [2,3,0,1]
                            if (mine) {
                                const uint32_t oi = ob + (uint32_t)k * ls;
                                const int Ds = imin(D[k], kSat16);
                                op[oi] = (int16_t)(cls == 1 ? -Ds : Ds);
                                if (cls == 1 || any) sp[oi] = cls == 1 ? -D[k] : D[k];
                            }
                        }
                    }
                } else {
                    if (a.vb) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            int b = byz;
                            if (a.nx > 1) b = imin(b, (int)min((int64_t)(p0 + k) + 1, a.nx - (p0 + k)));
                            // (b >= 1 inside the line: a voxel of the other class stays 0.  Positions past the end of the line -- the last
                            //  chunk of a line whose length is not a multiple of 8 -- have b <= 0, and from the second one on b^2 as a 24-bit
                            //  product is NEGATIVE: without the test on p0 + k such a position stopped being "not mine" (D = 0) and its
                            //  value was stored one or more x planes past the end of the field.  Round 5's fuzz, shape 9 x 777 x 64 with a
                            //  virtual border: the first time the field behind was the allocation's last page.)
                            if (b < 32768 && p0 + k < L) D[k] = imin(D[k], (int)__umul24((uint32_t)b, (uint32_t)b));
                        }
                    }
                    char* const op = reinterpret_cast<char*>(reinterpret_cast<float*>(a.out) + base);     // (uniform base + 32-bit byte offset)
                    uint32_t bo = ob * 4u;
                    int mx = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        mx = imax(mx, D[k]);
                        // (round 6 built an fp64-free form of this finish -- sdfgpu_finish.hpp: exact, and 2.4 % SLOWER here: on this chip
                        //  v_fma_f64 issues at the fp32 rate, and the fp32 form with its exactness test is the longer sequence)
#ifdef SDFGPU_DEBUG_HOOKS
                        float f = (a.dbg & 2) ? (float)D[k] : (float)(sqrt_exact_pos((double)D[k]) * a.resolution);
                        if ((a.dbg & 4) && (k || slotT)) { bo += 4u * ls; continue; }
#else
                        float f = (float)(sqrt_exact_pos((double)D[k]) * a.resolution);                 // (D = 0: not stored)
#endif
                        f = D[k] >= kInf32 ? __builtin_inff() : f;
                        if (D[k] != 0) *reinterpret_cast<float*>(op + bo) = cls == 1 ? -f : f;
                        bo += 4u * ls;
                    }
                    if (cls == 1) mxQ = imax(mxQ, mx); else mxF = imax(mxF, mx);
                }
            }
            DC_STAMP(4);
            // Pass 0 finishes the tile's filled voxels itself when they are few and shallow (thin surfaces): one lane per
            // listed voxel, exact local search along its line.  Deep or numerous filled voxels raise misc[25] and the second
            // pass does the class properly.
            if (cls == 0 && !probe) {
                const uint32_t nf = misc[24];
                if (nf > (uint32_t)kDcLocalFilled) {
                    if (t == 0) misc[25] = 1u;
                } else {
                    for (uint32_t e = (uint32_t)t; e < nf; e += (uint32_t)NT) {
                        const uint32_t ent = flist[e];
                        const int fl = (int)(ent >> 28), p = (int)((ent >> 12) & 0xffffu);
                        int D1 = (int)(ent & 0xfffu);           // (saturated at kDcLocalSat: then only a candidate found below can finish the voxel)
                        // offsets in rounds of 4 (8 loads in flight; D1 <= 1023: at most 8 rounds).  (Round 6 measured rounds of 8 in the y sweep: slower
                        // on every scene -- room y sweep +3 %, two-box +5 %.)  A candidate beyond the bound that
                        // rides along in a round is still a candidate: harmless.
#ifdef SDFGPU_DEBUG_HOOKS
                        if (a.dbg & 64) D1 = 1;                 // profiling builds: bit 6 = no local-search rounds (wrong results)
#endif
                        for (int d0 = 1; (int)__umul24(d0, d0) < D1; d0 += 4) {
                            int v[8];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int d = d0 + j;
                                v[2 * j] = p - d >= 0 ? imax(-raw_signed(fl, p - d), 0) : (1 << 28);
                                v[2 * j + 1] = p + d < L ? imax(-raw_signed(fl, p + d), 0) : (1 << 28);
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) D1 = imin(D1, (int)__umul24(d0 + j, d0 + j) + imin(v[2 * j], v[2 * j + 1]));
                        }
                        if (D1 >= kDcLocalSat) { misc[25] = 1u; continue; }      // deep inside a solid: the second pass
                        emit_filled((uint32_t)fl + (uint32_t)p * ls, p, D1, byz_of(fl));
                    }
                }
            }
        }
        DC_STAMP(5);
        __syncthreads();                        // keys / args are rebuilt by the next class; misc[25] is complete
        DC_STAMP(6);
    };

    run_pass(std::integral_constant<int, 0>{});
    // The second pass (distance to free, for filled voxels) runs only when pass 0 asked for it: a filled voxel whose in-row
    // squared distance S is small is finished by pass 0 itself with a local search (candidates at offset d can only matter
    // while d^2 < S), which covers the thin surfaces of sensed scenes; pass 0 raises misc[25] for the rest.
    const bool second = !probe && misc[25] != 0u;                                  // (block-uniform)
    if constexpr (LOOP) __syncthreads();            // (the next tile's first pass clears misc: everybody has read it)
    if (second) run_pass(std::integral_constant<int, 1>{});

#ifdef SDFGPU_PHASE_CLOCKS
#ifdef SDFGPU_TRIP_COUNTS
    if (a.clocks && (t & 63) == 0 && !probe) {
#else
    if (a.clocks && (t & 63) == 0 && !probe && ((blockIdx.x * 2654435761u) >> 27) == 5u) {     // a sample (hashed: consecutive workgroups are consecutive tiles of a row): same-address atomics serialise
#endif
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(a.clocks + (STAGE - 2) * 8 + k, clk[k]);
    }
#endif
    if (probe) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            probe_far += __shfl_xor(probe_far, off);
            probe_tot += __shfl_xor(probe_tot, off);
            probe_mid += __shfl_xor(probe_mid, off);
            probe_x9 += __shfl_xor(probe_x9, off);
        }
        if ((t & 63) == 0) { atomicAdd(&misc[26], (uint32_t)probe_far); atomicAdd(&misc[27], (uint32_t)probe_tot); atomicAdd(&misc[28], (uint32_t)probe_mid); atomicAdd(&misc[29], (uint32_t)probe_x9); }
        __syncthreads();
        if (t == 0) {
            atomicAdd(a.probe_out, misc[26]); atomicAdd(a.probe_out + 1, misc[27]);
            if (a.probe_thr2 > 0) atomicAdd(a.probe_out - 1, misc[28]);
            if (a.probe_thr3 > 0) atomicAdd(a.probe_out + 5, misc[29]);        // (probe_out = status word 12: + 5 = word 17)
        }
        probe_done(a);
        return;
    }
    } while (LOOP && (vblk += gridDim.x) < nblk);
    if constexpr (STAGE == 3) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mxF = imax(mxF, __shfl_xor(mxF, off));
            mxQ = imax(mxQ, __shfl_xor(mxQ, off));
        }
        if ((t & 63) == 0) slot_max2(a.maxdsq, blockIdx.x * (NT / 64) + (t >> 6), mxF, mxQ);
        if constexpr (LOOP) {
            if (a.fold_status) {                                // (launch-uniform)
                __shared__ uint32_t fold_last;
                __threadfence();                                // this workgroup's slot maxima (and workgroup 0's ran_flag) before its ticket
                __syncthreads();
                if (t == 0) fold_last = (atomicAdd(a.fold_ticket, 1u) == gridDim.x - 1u) ? 1u : 0u;
                __syncthreads();
                if (fold_last) {                                // (block-uniform) every other workgroup has finished
                    __threadfence();
                    fold_slots_device<NT>(a.maxdsq, a.fold_status, a.fold_result, a.fold_report, a.fold_report_mask, t);     // (clears the ticket with the status block)
                }
            }
        }
    }
}

// The instantiations the launcher (launch_envelope, sdfgpu.hip) uses, compiled in their own translation unit
// (sdfgpu_envelope_tu.hip: the far-field kernel is a third of the library's device code and most of its compile time, and it
// is the kernel that changes most often); everywhere else they are only declared.
#define SDFGPU_ENVELOPE_INSTANCES(X) \
    X(2, false, 256, 16) X(3, false, 256, 16) X(2, true, 256, 16) X(3, true, 256, 16) \
    X(2, false, 256, 16, true) X(3, false, 256, 16, true) X(2, true, 256, 16, true) X(3, true, 256, 16, true) \
    X(2, false, 512, 16, false, 4) X(3, false, 512, 16, false, 4) X(2, true, 512, 16, false, 4) X(3, true, 512, 16, false, 4) \
    X(2, true, 256, 16, false, 4, 512) X(3, true, 256, 16, false, 4, 512) X(2, true, 512, 16, false, 4, 1024) X(3, true, 512, 16, false, 4, 1024)
#ifndef SDFGPU_ENVELOPE_TU
#define SDFGPU_ENVELOPE_DECLARE(...) extern template __global__ void k_envelope_dc<__VA_ARGS__>(const EnvDcArgs);
SDFGPU_ENVELOPE_INSTANCES(SDFGPU_ENVELOPE_DECLARE)
#undef SDFGPU_ENVELOPE_DECLARE
#endif

}  // namespace sdfgpu
