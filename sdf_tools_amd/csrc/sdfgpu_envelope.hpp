// sdfgpu_envelope.hpp -- KE2 / KE3: exact lower-envelope sweeps for far-field scenes.
//
// The marching kernels (K2, K3/16) decide every voxel whose nearest opposite-class site is close and
// continue with an outward scan otherwise.  That scan costs O(distance) per voxel, which is fine for
// moderately sparse scenes but not for the scenes robots actually see (a few objects in a large free
// volume: distances of hundreds of voxels -- 512^3 from a 200 k point cloud took ~100 ms).  So the scan
// is bounded (kMaxScan rows); a voxel that is still undecided raises a "far" flag and the sweep is redone
// for the whole grid by the kernel below, which is O(line length) per line whatever the distances.
//
// One lane owns one line and runs the classic lower-envelope-of-parabolas algorithm (Felzenszwalb &
// Huttenlocher / Meijster) in exact integer arithmetic: with A_i = f(i) + i^2, site q dominates the top
// of the stack v_k over its whole interval iff (A_q - A_k)(v_k - v_{k-1}) <= (A_k - A_{k-1})(q - v_k)
// (64-bit cross-multiplication, no divisions, no stored intersection points); the fill pass advances to
// the next parabola when A_{k+1} - A_k <= 2 p (v_{k+1} - v_k).  Each line is processed twice, once per
// class (sites of the "distance to filled" function, evaluated on free voxels, and vice versa) -- the
// same signed-field convention as everywhere else.  Consecutive lanes own consecutive z, so every row
// read and write is a coalesced segment; the per-line stacks live in a global scratch array laid out
// [depth][line], so lanes at equal depth (the common case) access it coalesced as well and the two
// hottest entries of each stack stay in registers.
#pragma once
#include "sdfgpu_kernels.hpp"
#include "sdfgpu_sweep_x16.hpp"

namespace sdfgpu {

// Rows the marching kernels scan outward before handing the sweep to the envelope kernel.  Long enough
// that mid-sparse scenes (p = 0.01: distances up to ~30) never pay for an envelope pass they do not need.
constexpr int kScanExpectNear = 40;

struct EnvArgs {
    const int16_t* in16;      // STAGE 2: z field (+-g, 32767 = none); STAGE 3: plane field p16
    const int32_t* side_in;   // STAGE 3: exact values where p16 is saturated
    void* out;                // STAGE 2: int16 plane field; STAGE 3: float sdf
    int32_t* side_out;        // STAGE 2: exact plane values (written for every voxel)
    int2* scratch;            // [L][nlines] stack entries (v, A)
    int64_t nlines;           // lines = lanes
    int64_t cpl;              // lines per outer unit (STAGE 2: nz per x-plane; STAGE 3: all)
    int64_t outer_stride;     // elements between outer units (STAGE 2: ny*nz)
    int64_t line_stride;      // elements between successive positions of a line
    int L;
    double resolution;        // STAGE 3
    int vb;
    int64_t nx, ny, nz;       // full extents (virtual border)
    uint32_t* maxdsq;
    const uint32_t* guard;    // run only if *guard != 0
};

// vmcnt counts loads and stores alike and the compiler cannot count the stores of a data-dependent loop, so ANY
// global load whose result is used inside the push / pop loop costs an s_waitcnt vmcnt(0) there -- which also
// waits for the scratch store of the site pushed just before: one full memory round trip per site, strictly
// serial.  Therefore the hot loops below never consume a global load:
//   forward   the top kEnvRing entries of a lane's stack are mirrored in an LDS ring (slot = depth mod kEnvRing)
//             and pops read only LDS; if a pop sequence runs below the ring, a refill copies the next
//             kEnvRing / 2 older entries from the scratch into the ring;
//   backward  the next kEnvWin parabolas are fetched with independent loads once per batch of positions into a
//             register window that is shifted on every advance; running out of it mid-batch is the rare case.
// The rare paths load through inline assembly that carries its own wait, so the compiler does not see a pending
// load and puts no conservative wait into the hot loops.
constexpr int kEnvRing = 16;
#ifndef SDFGPU_ENV_WIN
#define SDFGPU_ENV_WIN 4
#endif
#ifndef SDFGPU_ENV_CH
#define SDFGPU_ENV_CH 8
#endif
constexpr int kEnvWin = SDFGPU_ENV_WIN;

__device__ __forceinline__ int2 env_load_waited(const int2* p) {
    int2 e;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(p) : "memory");
    return e;
}
__device__ __forceinline__ void env_ring_refill(int2* ring_lane, const int2* scratch_lane, int64_t nl, int i) {
    for (int c = 0; c < kEnvRing / 2; ++c) {
        const int d = i - c;
        if (d >= 0) ring_lane[(d & (kEnvRing - 1)) * kBlock] = env_load_waited(scratch_lane + (int64_t)d * nl);
    }
}

template <int STAGE>
__global__ __launch_bounds__(kBlock) void k_envelope(const EnvArgs a) {
    __shared__ int2 ring[kEnvRing * kBlock];
    if (a.guard && *a.guard == 0u) return;
    int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = t < a.nlines;
    if (!valid) t = a.nlines - 1;
    const int64_t o = t / a.cpl;
    const int64_t base = o * a.outer_stride + (t - o * a.cpl);
    const int64_t ls = a.line_stride, nl = a.nlines;
    const int L = a.L;
    int mxF = 0, mxQ = 0;

    // exact signed value at position p: + free / - filled, magnitude kInf32 = no site
    auto load_signed = [&](int p) -> int {
        const int v = a.in16[base + (int64_t)p * ls];
        if constexpr (STAGE == 2) {
            const int g = abs(v);
            const int sq = g >= kInf16 ? kInf32 : g * g;
            return v < 0 ? -sq : sq;
        } else {
            if (abs(v) >= kSat16) return a.side_in[base + (int64_t)p * ls];
            return v;
        }
    };

    // virtual-border coordinates of this line (STAGE 3: the line runs along x at fixed (y, z))
    int vy = 0, vz = 0;
    if constexpr (STAGE == 3) { vy = (int)(t / a.nz); vz = (int)(t - (int64_t)vy * a.nz); }

    constexpr int CH = SDFGPU_ENV_CH;          // rows fetched per batch: CH independent loads in flight per lane
    bool has_filled = false;                   // learned during pass 0: does this line hold any filled voxel?
    if (valid) {                               // (lanes past the last line must not touch another line's stack)
#pragma unroll 1
    for (int cls = 0; cls < 2; ++cls) {        // 0: sites of "distance to filled" (for free voxels), 1: the other
        // pass 1 only produces values for filled voxels: a line without any (most lines of a scene with a few
        // objects in free space) skips it entirely
        if (cls == 1 && !has_filled) break;
        // ---- forward: build the envelope -------------------------------------------------------------
        int k = -1, lo = 0;                     // stack top; lowest depth whose ring slot is current
        int vt = 0, At = 0, vs = 0, As = 0;     // top and second entry (copies of scratch[k], scratch[k-1])
        int2* const ring_lane = ring + threadIdx.x;
        const int2* const scratch_lane = a.scratch + t;
        int sv[CH], sn[CH];                     // current batch and the one in flight behind it
        // Site pruning.  The voxels that ARE the sought class have value 0, and along a line only the two ends
        // of a run of them can ever be nearest to a voxel outside the run: (p - end)^2 < (p - interior)^2.
        // So such a voxel is pushed only if one of its line neighbours has the other class.  For pass 1 on a
        // scene with a few objects this turns "every free voxel of the line is a site" (a 512-deep stack and an
        // advance per step of the backward pass) into two sites per object crossing.
        const int none = cls == 0 ? -1 : 1;     // stands for a neighbour beyond the line ends: "same class as the run"
        int sprev = none;
#pragma unroll
        for (int u = 0; u < CH; ++u) sn[u] = (u < L) ? load_signed(u) : (cls == 0 ? kInf32 : -kInf32);
        for (int q0 = 0; q0 < L; q0 += CH) {
#pragma unroll
            for (int u = 0; u < CH; ++u) sv[u] = sn[u];
            if (q0 + CH < L) {                  // issue the next batch's loads before working on this one
#pragma unroll
                for (int u = 0; u < CH; ++u) sn[u] = (q0 + CH + u < L) ? load_signed(q0 + CH + u) : (cls == 0 ? kInf32 : -kInf32);
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int q = q0 + u;
                const int s = sv[u];
                if (q < L) has_filled |= s < 0;
                const int left = u > 0 ? sv[u - 1] : sprev;
                const int right = (q + 1 >= L) ? none : (u < CH - 1 ? sv[u + 1] : sn[0]);
                // value of this class's function at q: |s| on voxels of the class that looks for the other one,
                // 0 on voxels that ARE the sought class (kept only at the ends of their runs)
                int val;
                if (cls == 0) val = s > 0 ? s : ((left > 0 || right > 0) ? 0 : kInf32);
                else val = s < 0 ? -s : ((left < 0 || right < 0) ? 0 : kInf32);
                if (val < kInf32) {             // else: no (useful) site of the sought class at this position
                    const int Aq = val + q * q;
                    while (k >= 1 && (int64_t)(Aq - At) * (vt - vs) <= (int64_t)(At - As) * (q - vt)) {
                        --k; vt = vs; At = As;
                        if (k >= 1) {
                            const int i = k - 1;
                            if (i < lo) {                       // rare: popped below the ring
                                env_ring_refill(ring_lane, scratch_lane, nl, i);
                                lo = max(0, i - kEnvRing / 2 + 1);
                            }
                            const int2 e = ring_lane[(i & (kEnvRing - 1)) * kBlock];
                            vs = e.x; As = e.y;
                        }
                    }
                    ++k;
                    a.scratch[(int64_t)k * nl + t] = make_int2(q, Aq);
                    ring_lane[(k & (kEnvRing - 1)) * kBlock] = make_int2(q, Aq);
                    lo = max(lo, k - kEnvRing + 1);
                    vs = vt; As = At; vt = q; At = Aq;
                }
            }
            sprev = sv[CH - 1];
        }
        // ---- backward: evaluate on the voxels that need this class --------------------------------------
        int j = 0, v0 = 0, A0 = 0;               // current parabola = entry j
        int wv[kEnvWin], wA[kEnvWin];            // entries j+1 .. j+kEnvWin; navail of them are loaded
        int navail = 0;
        if (k >= 0) { const int2 e = a.scratch[t]; v0 = e.x; A0 = e.y; }
#pragma unroll
        for (int i = 0; i < kEnvWin; ++i) { wv[i] = 0; wA[i] = 0; }
        int16_t rawv[CH], rawn[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) rawn[u] = a.in16[base + (int64_t)min(u, L - 1) * ls];
        for (int p0 = 0; p0 < L; p0 += CH) {
#pragma unroll
            for (int u = 0; u < CH; ++u) rawv[u] = rawn[u];
#pragma unroll
            for (int u = 0; u < CH; ++u) rawn[u] = a.in16[base + (int64_t)min(p0 + CH + u, L - 1) * ls];
            // top the window up: independent, unconditional loads (clamped depth), waited for once per batch
            {
                int2 e[kEnvWin];
#pragma unroll
                for (int i = 0; i < kEnvWin; ++i) e[i] = a.scratch[(int64_t)max(0, min(j + 1 + i, k)) * nl + t];
#pragma unroll
                for (int i = 0; i < kEnvWin; ++i) {
                    if (i >= navail) { wv[i] = e[i].x; wA[i] = e[i].y; }
                }
            }
            navail = max(0, min(kEnvWin, k - j));
            // step 1: every advance and distance of the batch, kept in registers; step 2: all stores together
            int Dv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int p = p0 + u;
                Dv[u] = -1;
                if (p < L) {
                while (j < k && (int64_t)(wA[0] - A0) <= (int64_t)2 * p * (wv[0] - v0)) {
                        ++j; v0 = wv[0]; A0 = wA[0];
#pragma unroll
                        for (int i = 0; i + 1 < kEnvWin; ++i) { wv[i] = wv[i + 1]; wA[i] = wA[i + 1]; }
                        --navail;
                        if (navail == 0 && j < k) {               // rare: window exhausted inside a batch
                            const int2 e = env_load_waited(scratch_lane + (int64_t)(j + 1) * nl);
                            wv[0] = e.x; wA[0] = e.y;
                            navail = 1;
                        }
                    }
                    const bool filled = rawv[u] < 0;
                    if ((cls == 0) != filled) {
                        int D = kInf32;
                        if (k >= 0) {
                            const int64_t d = (int64_t)A0 + (int64_t)p * p - (int64_t)2 * p * v0;
                            D = (int)min(d, (int64_t)kInf32);
                        }
                        Dv[u] = D;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int p = p0 + u;
                int D = Dv[u];
                if (D < 0) continue;
                const bool filled = rawv[u] < 0;
                const int64_t oi = base + (int64_t)p * ls;
                if constexpr (STAGE == 2) {
                    a.side_out[oi] = filled ? -D : D;
                    reinterpret_cast<int16_t*>(a.out)[oi] = (int16_t)(filled ? -min(D, kSat16) : min(D, kSat16));
                } else {
                    if (a.vb) {
                        int64_t b = kInf32;
                        if (a.nx > 1) b = min(b, min((int64_t)p + 1, a.nx - p));
                        if (a.ny > 1) b = min(b, min((int64_t)vy + 1, a.ny - vy));
                        if (a.nz > 1) b = min(b, min((int64_t)vz + 1, a.nz - vz));
                        if (b < 32768) D = min(D, (int)(b * b));
                    }
                    if (filled) mxQ = max(mxQ, D); else mxF = max(mxF, D);
                    const float f = (D >= kInf32) ? __builtin_inff() : (float)(sqrt((double)D) * a.resolution);
                    reinterpret_cast<float*>(a.out)[oi] = filled ? -f : f;
                }
            }
        }
    }
    }
    if constexpr (STAGE == 3) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mxF = max(mxF, __shfl_xor(mxF, off));
            mxQ = max(mxQ, __shfl_xor(mxQ, off));
        }
        if ((threadIdx.x & 63) == 0)
            slot_max2(a.maxdsq, blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), mxF, mxQ);      // a.maxdsq = slot array
    }
}

}  // namespace sdfgpu
