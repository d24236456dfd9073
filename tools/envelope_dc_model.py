#!/usr/bin/env python3
"""Executable model of the far-field sweep of sdfgpu_envelope_dc.hpp (k_envelope_dc, third generation): the same centred
32-bit keys, the three levels (A: positions 64 i, clipped by the distance bound or over the whole span; B: positions 8 j
inside the intervals of A; C: chunks of 8 positions inside the intervals of B), scans in ALIGNED PAIRS -- which read up to
one candidate before and one behind the range they are given -- and a tile-level site span / smallest site value that may
be looser than the line's own.  Run line by line on the CPU and compared with a brute-force min-plus evaluation.
Development aid and a CPU test (tests/test_host_cpu.py); the GPU tests compare the kernel itself with the exact oracle.
Run: python tools/envelope_dc_model.py"""
import math
import random
import struct
import sys

INF = 1 << 30
M32 = 0xFFFFFFFF


def f32(x):
    return struct.unpack("f", struct.pack("f", float(x)))[0]


def dc_line(F, FINF, lo_t=None, hi_t=None, mt_t=None, force_clip=None, coop_rng=None, coop_scale=1, flat=False, stats=None):
    """F: site values (INF = no site).  lo_t / hi_t: span handed to the line (a superset of its sites), mt_t: a lower bound
    of the site values (the kernel keeps both per tile of 16 lines).  Returns (D, candidate evaluations)."""
    L = len(F)
    B = max(1, (L - 1).bit_length())
    mask = (1 << B) - 1
    h = (L + 1) // 2
    assert FINF + (L + 2) ** 2 < (1 << (32 - B)), "keys do not fit 32 bits"
    Fc = [min(v, FINF) for v in F]
    key = [(((Fc[q] + (q - h) ** 2 + h * h) << B) | q) & M32 for q in range(L)]
    for q in (L, L + 1):                                            # two sentinels behind the line
        key.append((((FINF + (q - h) ** 2 + h * h) << B) | (q & mask)) & M32)
    sites = [q for q in range(L) if Fc[q] < FINF]
    if lo_t is None:
        lo_t, hi_t = (sites[0], sites[-1]) if sites else (0, -1)
    if mt_t is None:
        mt_t = min([Fc[q] for q in sites] + [M32])
    if lo_t > hi_t:
        return [INF] * L, 0
    M = (L + 7) // 8
    MA = (L + 63) // 64
    args = [M32] * (M + 2)
    evals = [0]

    def nc(p):                                                      # -(2 p') << B
        return ((h - p) << (B + 1))

    def val(p, q):                                                  # key[q] - ((2 p') << B) * q'   (mod 2^32)
        v = (key[q] + nc(p) * (q - h)) & M32
        # (positions past the end of the line ride along in the last interval / chunk; their values may wrap and are dropped)
        assert p >= L or v == ((((key[q] >> B) - (q - h) ** 2 - h * h + (p - q) ** 2 + h * h - (p - h) ** 2) << B) | (key[q] & mask)), "value left the 32-bit range"
        return v

    def scan(positions, lo, hi):                                    # aligned pairs: candidates (lo & ~1) .. hi | 1
        assert lo <= hi, "empty range: the monotonicity of the argmins is broken"
        best = [M32] * len(positions)
        q = lo & ~1
        while q <= hi:
            for k, p in enumerate(positions):
                best[k] = min(best[k], val(p, q), val(p, q + 1))
                evals[0] += 2
            q += 2
        return best

    def scan_unit(positions, lo, hi, nshares, rng):
        """Levels B and C of round 4 (scan8_calm + coop8) for one unit of one line: its `nshares` lanes (interleaved shares, step
        2 * nshares) run trips in blocks of 4, 8, 16, ... in lock step -- the control is the wave's --, and after any block the
        wave may decide (here: at random) to spread what is left over its 4 rows: row r takes the pairs bq + 2 r + 8 j, bq = the
        FIRST share's next candidate.  Must see exactly the candidates of the plain scan."""
        best = [M32] * len(positions)
        step = 2 * nshares
        qs = [(lo & ~1) + 2 * u for u in range(nshares)]

        def pair(q):
            for k, p in enumerate(positions):
                best[k] = min(best[k], val(p, q), val(p, q + 1))
                evals[0] += 2
        blk = 4
        while any(q <= hi for q in qs):
            for _ in range(blk):
                for u in range(nshares):
                    if qs[u] <= hi:
                        pair(qs[u])
                        qs[u] += step
            blk *= 2
            if any(q <= hi for q in qs) and rng.random() < 0.5:
                assert qs[0] <= hi and all(q >= qs[0] for q in qs), "the first share is the one furthest behind"
                for row in range(4):
                    q = qs[0] + 2 * row
                    while q <= hi:
                        pair(q)
                        q += 8
                break
        return best

    def dist(best, p):
        return (best >> B) - p * (2 * h - p)

    # Round 5, a VARIANT that was built into the kernel, measured and taken out again (`flat` = True): the
    # flat-stretch shortcut.  If every link q -> q + 1 of a candidate range is flat (|dF| <= 1), every position of the range
    # is its own argmin and D(p) = F(p): levels B and C need no scan over floors, walls and box faces.  Exact (checked here
    # against the scans it replaces and against brute force; 16 000 fuzz scenes on the GPU), and it removed 27 - 50 % of the
    # scan trips of the room scene -- without making a build any faster: between two workgroup barriers a tile waits for its
    # SLOWEST wave, and the lines of a tile all meet the wall in the same interval, i.e. in the same wave.  The map, from the
    # keys as the kernel built it: the link is flat iff u = key[q + 1] - key[q] - 1 - ((2 q') << B) = (dF + 1) << B is at
    # most 2 << B (unsigned, mod 2^32); range test over at most 3 words of 32 links; links beyond the line do not exist.
    FW = (L + 31) // 32
    flatw = [0] * FW
    if flat:
        for q in range(L - 1):
            u = (key[q + 1] - key[q] - 1 - ((2 * (q - h)) << B)) & M32
            if u <= (2 << B):
                assert abs(Fc[q + 1] - Fc[q]) <= 1, "a link that is not flat passed the key test"
                flatw[q >> 5] |= 1 << (q & 31)
            else:
                assert abs(Fc[q + 1] - Fc[q]) > 1, "a flat link failed the key test"

    def flat_range(a, b):
        if b <= a:
            return True
        w0, w1 = a >> 5, (b - 1) >> 5
        if w1 - w0 > 2:
            return False
        return all(flatw[q >> 5] >> (q & 31) & 1 for q in range(a, b))
    any_flat = any(flatw)

    # level A
    for i in range(MA):
        p = 64 * i
        lo, hi = lo_t, hi_t
        clip = force_clip if force_clip is not None else (i % 2 == 0)
        if clip:
            pcl = min(max(p, lo_t), hi_t)
            ub = min(val(p, pcl), val(p, lo_t), val(p, hi_t))
            dub = dist(ub, p)
            if dub < FINF:
                w = int(math.sqrt(f32(dub - min(mt_t, dub)))) + 2
                lo, hi = max(lo, p - w), min(hi, p + w)
        args[8 * i] = scan([p], lo, hi)[0]
    # level B
    for i in range(MA):
        lo = args[8 * i] & mask
        hi = (args[8 * (i + 1)] & mask) if i + 1 < MA else hi_t
        pos = [64 * i + 8 * k for k in range(8)]
        pb0 = 64 * i
        if any_flat and lo <= pb0 + 8 and hi >= pb0 + 56 and pb0 + 57 < L and flat_range(lo, hi):
            # flat stretch: every position is its own argmin, or its left neighbour at equal value; the pair partners ride along
            # and the packed minimum is the leftmost argmin, like a scan's
            short = [M32] + [min(val(p, p - 2), val(p, p - 1), val(p, p), val(p, p + 1)) for p in pos[1:]]
            evals[0] += 28
            keep = evals[0]
            assert short[1:] == scan(pos, lo, hi)[1:], "level B: the flat-stretch shortcut differs from the scan it replaces"
            evals[0] = keep                                         # (the checking scan is not part of the schedule)
            if stats is not None:
                stats["flat_B"] = stats.get("flat_B", 0) + 1
            for k in range(1, 8):
                if 8 * i + k < M:
                    args[8 * i + k] = min(args[8 * i + k], short[k])
            continue
        best = scan(pos, lo, hi)
        if coop_rng is not None and hi - lo >= 108 // coop_scale:   # a wave that holds a long range: wave-uniform control, hand-over
            for nshares in (1, 2):                                  # (one lane per interval for lines above 512, two below)
                assert scan_unit(pos, lo, hi, nshares, coop_rng) == best, "the split scans of level B saw a different candidate set"
        for k in range(1, 8):
            if 8 * i + k < M:
                args[8 * i + k] = min(args[8 * i + k], best[k])
    # level C
    D = [None] * L
    for i in range(M):
        a0 = args[i] & mask
        a8 = (args[i + 1] & mask) if i + 1 < M else hi_t
        p0 = 8 * i
        if any_flat and a0 <= p0 and a8 >= p0 + 7 and flat_range(a0, a8):
            # flat stretch: D(p) = F(p), the value of the position's own key
            assert a0 <= a8
            short = [val(p0 + k, p0 + k) for k in range(8)]
            evals[0] += 8
            keep = evals[0]
            ref = scan([p0 + k for k in range(8)], a0, a8)
            evals[0] = keep
            assert [v >> B for v in short] == [v >> B for v in ref], "level C: the flat-stretch shortcut differs from the scan it replaces"
            if stats is not None:
                stats["flat_C"] = stats.get("flat_C", 0) + 1
            for k in range(8):
                p = p0 + k
                if p < L:
                    d = dist(short[k], p)
                    D[p] = INF if d >= FINF else d
            continue
        best = scan([8 * i + k for k in range(8)], a0, a8)
        if coop_rng is not None and a8 - a0 >= 34 // coop_scale:
            assert scan_unit([8 * i + k for k in range(8)], a0, a8, 1, coop_rng) == best, "the split scan of level C saw a different candidate set"
        for k in range(8):
            p = 8 * i + k
            if p < L:
                d = dist(best[k], p)
                D[p] = INF if d >= FINF else d
    return D, evals[0]


def i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def vb_chunk_finish(D, p0, L, nx, byz, guarded=True):
    """The x sweep's finish of one chunk of 8 positions under a virtual border (k_envelope_dc, STAGE 3): positions past the end of the
    line were set to D = 0 ("not mine": nothing is stored for them) BEFORE this step, which lowers every D to the squared distance to
    the padded layer -- b = min(byz, p + 1, nx - p), b^2 as the low 32 bits of a 24 x 24-bit product (v_mul_u32_u24), compared as a
    signed int.  Past the end b <= 0, and from b = -1 on the product is 0xFE000001 = negative: without the guard the position is no
    longer 0 and its value is stored x planes behind the field (the bug round 5's fuzz found at nx = 9).  Returns the chunk's D."""
    out = list(D)
    for k in range(8):
        p = p0 + k
        b = byz
        if nx > 1:
            b = min(b, min(p + 1, nx - p))
        if b < 32768 and (p < L or not guarded):
            sq = i32((b & 0xFFFFFF) * (b & 0xFFFFFF))
            out[k] = min(out[k], sq)
    return out


def brute(F):
    L = len(F)
    out = []
    for p in range(L):
        b = INF
        for q in range(L):
            if F[q] < INF:
                b = min(b, F[q] + (p - q) ** 2)
        out.append(b)
    return out


def random_line(rng, L):
    dens = rng.choice([0.0, 0.01, 0.05, 0.3, 1.0])
    vmax = rng.choice([1, 4, 50, 1000, 200000])
    kind = rng.choice(["rand", "smooth", "ties", "runs", "flat", "plateaus", "stairs"])
    c = rng.randrange(-50, L + 50)
    hh = rng.randrange(0, 300)
    level = INF
    F = []
    for q in range(L):
        if kind == "smooth":
            F.append(hh * hh + (q - c) ** 2 if rng.random() < max(dens, 0.3) else INF)
        elif kind == "ties":
            F.append(rng.choice([0, 1, 4]) if rng.random() < dens else INF)
        elif kind == "runs":
            F.append(0 if (q // 7) % 3 == 0 and dens > 0 else INF)
        elif kind == "flat":
            F.append(hh * hh + rng.randrange(0, 3) if rng.random() < max(dens, 0.2) else INF)
        elif kind == "plateaus":            # piecewise constant with a few jumps and holes: floors, walls, box faces (the shortcut's regime)
            if q == 0 or rng.random() < 0.02:
                level = rng.choice([INF, INF, 0, 1, hh * hh, hh * hh + 1, rng.randrange(0, vmax + 1)])
            F.append(level)
        elif kind == "stairs":              # unit steps up and down: every link flat, ties between a position and its left neighbour
            level = hh if q == 0 else max(0, level + rng.choice([-1, 0, 0, 1]))
            F.append(level)
        else:
            F.append(rng.randrange(0, vmax + 1) if rng.random() < dens else INF)
    return F


def check_line(rng, F):
    L = len(F)
    FINF = max([v for v in F if v < INF] + [0]) + (L - 1) ** 2 + 1      # above every real result
    sites = [q for q in range(L) if F[q] < INF]
    kw = {}
    if sites and rng.random() < 0.5:                                    # a looser (tile-level) span and bound
        kw = dict(lo_t=rng.randrange(0, sites[0] + 1), hi_t=rng.randrange(sites[-1], L),
                  mt_t=rng.randrange(0, min(F[q] for q in sites) + 1))
    kw["force_clip"] = rng.choice([None, True, False])
    if rng.random() < 0.5:                                              # round 4: long ranges under wave-uniform control (thresholds scaled
        kw["coop_rng"] = random.Random(rng.randrange(1 << 30))          # down so that the short test lines reach the hand-over too)
        kw["coop_scale"] = rng.choice([1, 4, 16])
    kw["flat"] = rng.random() < 0.5                                     # (the round-5 variant, see dc_line)
    got, ev = dc_line(F, FINF, stats=STATS, **kw)
    assert got == brute(F), (L, F, kw)
    return ev


STATS = {}


def main():
    rng = random.Random(1)
    for trial in range(3000):
        L = rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 40, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 512])
        check_line(rng, random_line(rng, L))
    F = [150 * 150 + (q - 700) ** 2 for q in range(512)]
    got, ev = dc_line(F, 700 ** 2 + 150 ** 2 + 511 ** 2 + 1)
    assert got == brute(F)
    print("ok; smooth 512-line: %d candidate evaluations (%.1f per position)" % (ev, ev / 512.0))
    F = [INF] * 100 + [150 * 150 + (q - 200) ** 2 for q in range(100, 300)] + [INF] * 212
    got, ev = dc_line(F, 3 * 511 ** 2 + 1)
    assert got == brute(F)
    print("ok; sites in [100, 300) of 512: %d candidate evaluations (%.1f per position)" % (ev, ev / 512.0))
    # the shortcut's regime: a floor under the whole line, a box face over part of it -- and how many scans it replaced
    for name, F in (("floor", [90 * 90] * 512), ("floor + box", [90 * 90] * 200 + [30 * 30] * 150 + [90 * 90] * 162),
                    ("wall slab in free space", [INF] * 180 + [12 * 12] * 40 + [INF] * 292)):
        st = {}
        got, ev = dc_line(F, 3 * 511 ** 2 + 1, stats=st, flat=True)
        _, ev0 = dc_line(F, 3 * 511 ** 2 + 1, flat=False)
        assert got == brute(F)
        print("ok; %s: %d candidate evaluations with the flat-stretch shortcut of round 5 (%d of 8 level-B and %d of 64 level-C scans replaced), %d without"
              % (name, ev, st.get("flat_B", 0), st.get("flat_C", 0), ev0))
    assert STATS.get("flat_B", 0) > 20 and STATS.get("flat_C", 0) > 200, STATS       # (the random lines reach both shortcuts)


if __name__ == "__main__":
    sys.exit(main())
