// sdfgpu_finish.hpp -- the reference's finishing arithmetic float(sqrt((double)D) * resolution) (sdf_generation.hpp:254-265; D = the
// integer squared distance) WITHOUT fp64 on the fast path (round 6, VERDICT r5 "next round" 1c).
//
// VERDICT r5 asked for it on the premise that fp64 issues at half rate.  Built, exact -- and NOT used by the kernels: on MI355X
// v_fma_f64 issues at the rate of the (unpacked) fp32 FMA, the x sweep's fp64 sequence (sqrt_exact_pos: 10 instructions + convert,
// multiply, convert) is SHORTER than this form with its exactness test, and the x sweep built on this form ran 2.4 % slower than
// round 5's library in one process on one box (two-box scene 0.467 vs 0.456 ms, room 0.714 vs 0.698; profiles/r06_fast_finish_ab.txt
// -- whose first, same-binary A/B had shown -7 % only because its "off" arm computed both forms).  Kept as a tested building block
// (sdfgpu_debug_finish_table, tests/test_gpu_finish.py, tools/probe/finish_fast_check.c) for hardware where fp64 is the slow pipe.
// finish_fast computes the same float in fp32:
//     rs = v_rsq_f32(D), s = D * rs, e = (D - s^2) rs / 2   sqrt(D) = s + e up to 2^-44 relative (one Newton residual; rs: 1 ulp)
//     p = s * rh, pe = fma(s, rh, -p)                  resolution = rh + rl (two floats: 2^-49 relative), p + pe = s * rh exactly
//     c = pe + s * rl + e * rh                         T = sqrt(D) * resolution = p + c up to 2^-20 ulp(p)
//     y = RN(p + (c - thr)), yh = RN(p + (c + thr))    thr = p * 2^-38 = 2^-15 .. 2^-14 ulp(p)
// Rounding is monotone: when y == yh every value between the two sums rounds to y -- T does, and so does the reference's
// double-rounded T (which differs from T by 2^-52 relative) -- across binade boundaries too.  When they differ (a share of
// ~9e-5 of all D, no small D among them for the usual resolutions) the lane raises `slow` and takes the fp64 sequence; a
// caller branches on a wave-wide ballot, so a wave pays for it once in ~200 voxel rounds.
// Exactness: tools/probe/finish_fast_check.c restates this with correctly rounded host arithmetic and perturbs the
// approximate instruction by -2 .. +2 ulp: every D <= 3 * 1024^2 x 15 resolutions x 9 perturbations either raises `slow` or
// returns the reference's float; on the device tests/test_gpu_finish.py compares the kernel's own table of every D.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>

namespace sdfgpu {

struct FinishFast {
    float rh, rl, hrh;      // resolution = rh + rl, hrh = rh / 2
    int ok;                 // 0: every lane takes the fp64 sequence (resolution outside the safe range, squared distances beyond 2^24, option off)
};

// host: max_d = an upper bound of the finite squared distances of this build (beyond it only the "infinite" sentinel occurs)
inline FinishFast make_finish_fast(double resolution, uint64_t max_d, bool enabled = true) {
    FinishFast k{0.0f, 0.0f, 0.0f, 0};
    // s * rh, the residuals and thr must stay normal floats: sqrt(D) in [1, 2^12], so 2^-60 <= resolution <= 2^60 is ample
    if (!enabled || !(resolution >= 0x1p-60) || !(resolution <= 0x1p60) || max_d >= (1ull << 24)) return k;
    k.rh = (float)resolution;
    k.rl = (float)(resolution - (double)k.rh);
    k.hrh = 0.5f * k.rh;
    k.ok = 1;
    return k;
}

// D >= 1 and exactly representable as a float (the caller masks D = 0 and the sentinel out of `slow`)
__device__ __forceinline__ float finish_fast(int D, const FinishFast& k, bool& slow) {
    const float x = (float)D;
    const float rs = __builtin_amdgcn_rsqf(x);                  // the one transcendental (quarter rate); s within ~2 ulp of sqrt(x)
    const float s = x * rs;
    const float r = __builtin_fmaf(-s, s, x);
    const float t1 = r * rs;                                    // e = t1 / 2 (the half sits in hrh)
    const float p = s * k.rh;
    const float pe = __builtin_fmaf(s, k.rh, -p);
    float c = __builtin_fmaf(s, k.rl, pe);
    c = __builtin_fmaf(t1, k.hrh, c);
    const float y = p + __builtin_fmaf(p, -0x1p-38f, c), yh = p + __builtin_fmaf(p, 0x1p-38f, c);
    slow = __builtin_bit_cast(uint32_t, y) != __builtin_bit_cast(uint32_t, yh);
    return y;
}

}  // namespace sdfgpu
