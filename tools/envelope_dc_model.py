#!/usr/bin/env python3
"""Executable model of the divide-and-conquer lower-envelope sweep of sdfgpu_envelope_dc.hpp (k_envelope_dc):
the same key encoding, level structure, pair-wise scans (including the harmless candidate hi + 1) and chunk phase,
run line by line on the CPU and compared with a brute-force min-plus evaluation.  Development aid only (the GPU
tests compare the kernel itself with the exact oracle); run: python tools/envelope_dc_model.py"""
import random
import sys

INF = 1 << 30


def dc_line(F, FINF):
    L = len(F)
    B = max(1, (L - 1).bit_length())
    mask = (1 << B) - 1
    assert FINF + L * L < (1 << (32 - B)), "keys do not fit 32 bits"
    key = [(((F[q] if F[q] < INF else FINF) + q * q) << B) | q for q in range(L)]
    key.append(((FINF + L * L) << B) | (L & mask))                 # sentinel behind the line
    sites = [q for q in range(L) if F[q] < INF]
    if not sites:
        return [INF] * L, 0
    qmin, qmax = sites[0], sites[-1]
    M = (L + 7) // 8
    a = [0] * (M + 2)
    Kp = M.bit_length()
    evals = [0]

    def scan(p, lo, hi, u=0, G=1):
        P2B = (p * p) << B
        c = (2 * p) << B
        q = lo + 2 * u
        best = 0xFFFFFFFF
        while q <= hi:
            R = (P2B - c * q) & 0xFFFFFFFF
            t0 = (key[q] + R) & 0xFFFFFFFF
            t1 = (key[q + 1] + R - c) & 0xFFFFFFFF
            assert t0 == ((((key[q] >> B) - q * q + (p - q) ** 2) << B) | (key[q] & mask))
            best = min(best, t0, t1)
            evals[0] += 2
            q += 2 * G
        return best

    for l in range(Kp):
        h = 1 << (Kp - 1 - l)
        j = 0
        while h * (2 * j + 1) <= M:
            ip = h * (2 * j + 1)
            lo = qmin if ip - h == 0 else a[ip - h]
            hi = qmax if ip + h > M else a[ip + h]
            assert lo <= hi
            p = 8 * (ip - 1)
            n_pos = (M // h + 1) >> 1
            if n_pos <= 8:      # distance-bound clipping of the first levels (any candidate's cost bounds the optimum)
                pc = min(max(p, lo), hi)
                v = min((key[c] >> B) - c * c + (p - c) ** 2 for c in (pc, lo, hi))
                if v < FINF:
                    w = int(v ** 0.5) + 2
                    lo, hi = max(lo, p - w), min(hi, p + w)
                    assert lo <= hi
            G = max(1, 16 >> l)
            best = min(scan(p, lo, hi, u, G) for u in range(G))
            a[ip] = best & mask
            assert lo <= a[ip] <= hi
            j += 1
    D = [None] * L
    for i in range(M):
        p0 = 8 * i
        a0 = a[i + 1]
        a8 = a[i + 2] if i + 2 <= M else qmax

        # every position of the chunk against all of [a0, a8] (pairs: a8 + 1 may be read); the kernel has no other path
        for k in range(8):
            if p0 + k < L:
                best = min(scan(p0 + k, a0, a8), 0xFFFFFFFF)
                D[p0 + k] = best >> B
    return [d if d < FINF else INF for d in D], evals[0]


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


def main():
    rng = random.Random(1)
    total = 0
    for trial in range(3000):
        L = rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 40, 64, 100, 127, 128, 200, 255, 256, 257, 512])
        dens = rng.choice([0.0, 0.01, 0.05, 0.3, 1.0])
        vmax = rng.choice([1, 4, 50, 1000, 200000])
        kind = rng.choice(["rand", "smooth", "ties", "runs"])
        F = []
        c = rng.randrange(-50, L + 50)
        hh = rng.randrange(0, 300)
        for q in range(L):
            if kind == "smooth":
                F.append(hh * hh + (q - c) ** 2 if rng.random() < max(dens, 0.3) else INF)
            elif kind == "ties":
                F.append(rng.choice([0, 1, 4]) if rng.random() < dens else INF)
            elif kind == "runs":
                F.append(0 if (q // 7) % 3 == 0 and dens > 0 else INF)
            else:
                F.append(rng.randrange(0, vmax + 1) if rng.random() < dens else INF)
        FINF = max([v for v in F if v < INF] + [0]) + (L - 1) ** 2 + 1      # above every real result
        got, ev = dc_line(F, FINF)
        want = brute(F)
        assert got == want, (trial, L, kind, F, got, want)
        total += ev
    # cost on a smooth far-field line of 512 (every site on the envelope)
    F = [150 * 150 + (q - 700) ** 2 for q in range(512)]
    got, ev = dc_line(F, 700 ** 2 + 150 ** 2 + 511 ** 2 + 1)
    assert got == brute(F)
    print("ok; smooth 512-line: %d candidate evaluations (%.1f per position)" % (ev, ev / 512.0))
    F = [INF] * 100 + [150 * 150 + (q - 200) ** 2 for q in range(100, 300)] + [INF] * 212
    got, ev = dc_line(F, 3 * 511 ** 2 + 1)
    assert got == brute(F)
    print("ok; sites in [100, 300) of 512: %d candidate evaluations (%.1f per position)" % (ev, ev / 512.0))


if __name__ == "__main__":
    sys.exit(main())
