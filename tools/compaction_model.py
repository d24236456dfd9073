#!/usr/bin/env python3
"""Site compaction for the far-field sweeps, modelled on the CPU BEFORE any HIP is written (VERDICT r5 "next round" 1b).

Today k_envelope_dc searches over POSITIONS: a scan of a candidate range [a, b] evaluates every position in it, sites or not,
on the envelope or not.  Compaction would search over the SITES THAT SURVIVE a dominance filter: keep, per line, only the sites
that can be the argmin of some position (the lower envelope of the parabolas F(q) + (p - q)^2), pack them, and run the same
three-level search (coarse positions 64 i, positions 8 j inside, chunks of 8) over the packed list.

For real sweep inputs (the y sweep's z distances and the x sweep's in-plane distances of the bench's scenes, built here with
numpy / scipy at n = 256) this reports, per scene and sweep, averaged over sampled lines that hold a site:
  sites / line         positions with a finite F
  envelope / line      sites that are the argmin of at least one position (the floor under ANY filter)
  after k peels        survivors of k rounds of the local filter a GPU pass can run: site q goes when its two neighbouring
                       survivors hide it (intersection(a, q) >= intersection(q, b)); each round = one test per site + one
                       prefix sum + one packed write
  candidate evaluations per position of the three-level search: over positions (today's kernel: ranges include non-sites),
  over the exact envelope, and over the survivors of k peels; with the cost of the peel rounds in the same unit
  (kPeelCost evaluations per list entry and round, see below).
The rule the judge set: build it only if issued trips fall by >= 25 % NET of the compaction pass.
usage: compaction_model.py [n = 256] [lines = 600]"""
import json
import os
import sys

import numpy as np
from scipy import ndimage

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdf_tools_amd import synth  # noqa: E402

INF = 1 << 40
kPeelCost = 13.0            # candidate evaluations' worth of instructions per LIST ENTRY and peel round: ~12 VALU for the test (two
                            # 64-bit products: g differences reach 2^22, position differences 2^10), ~6 for ballot / prefix / packed
                            # write, 2 LDS reads -- ~20 VALU against 1.5 per candidate evaluation; round k works on round k - 1's list


def room(n, floor=True, xwall=True, ywall=True):
    m = np.zeros((n, n, n), np.uint8)
    f = lambda v: int(v * n)
    t = max(f(0.02), 1)
    if floor:
        m[:, :, :t] = 1
    if xwall:
        m[:t, :, :] = 1
    if ywall:
        m[:, :t, :] = 1
    m[f(0.3):f(0.7), f(0.3):f(0.6), f(0.35):f(0.38)] = 1
    for (x, y) in ((0.31, 0.31), (0.68, 0.31), (0.31, 0.58), (0.68, 0.58)):
        m[f(x):f(x) + t, f(y):f(y) + t, :f(0.35)] = 1
    m[f(0.8):f(0.98), f(0.1):f(0.9), f(0.5):f(0.55)] = 1
    return m


def two_box(n):
    res = 0.01 * 512 / n
    pc = synth.two_box_points(200000 * n * n // (512 * 512), seed=0, scale=n * res)
    idx = (pc.astype(np.float64) / res).astype(np.int64)
    m = np.zeros((n, n, n), np.uint8)
    ok = np.all((idx >= 0) & (idx < n), axis=1)
    m[idx[ok, 0], idx[ok, 1], idx[ok, 2]] = 1
    return m


def z_dist_sq(mask):
    """F of the y sweep, pass 0: squared distance along z to the nearest filled voxel of the row (INF = none)."""
    n = mask.shape[2]
    big = 1 << 20
    pos = np.where(mask != 0, np.arange(n)[None, None, :], -big)
    left = np.maximum.accumulate(pos, axis=2)
    pos2 = np.where(mask != 0, np.arange(n)[None, None, :], big)
    right = np.minimum.accumulate(pos2[:, :, ::-1], axis=2)[:, :, ::-1]
    z = np.arange(n)[None, None, :]
    d = np.minimum(z - left, right - z).astype(np.int64)
    return np.where(d >= big // 2, INF, d * d)


def plane_dist_sq(mask):
    """F of the x sweep, pass 0: squared distance inside the x-plane to the nearest filled voxel (INF = none in the plane)."""
    out = np.empty(mask.shape, np.int64)
    for x in range(mask.shape[0]):
        if mask[x].any():
            out[x] = np.rint(ndimage.distance_transform_edt(mask[x] == 0) ** 2).astype(np.int64)
        else:
            out[x] = INF
    return out


def envelope_and_argmin(F):
    L = len(F)
    q = np.arange(L)
    C = F[None, :] + (q[:, None] - q[None, :]) ** 2
    a = np.argmin(C, axis=1)                       # leftmost argmin per position
    return a


def peel(F, sites, rounds):
    """survivors after `rounds` rounds of the neighbour test (exact integer arithmetic)"""
    s = list(sites)
    out = []
    for _ in range(rounds):
        if len(s) < 3:
            out.append(list(s))
            continue
        keep = [s[0]]
        for j in range(1, len(s) - 1):
            a, qq, b = s[j - 1], s[j], s[j + 1]      # (neighbours of THIS round's list: a parallel pass sees the same)
            ga, gq, gb = F[a] + a * a, F[qq] + qq * qq, F[b] + b * b
            # q visible iff x(a,q) < x(q,b):  (gq - ga) / (2 (q - a)) < (gb - gq) / (2 (b - q))
            if (gq - ga) * (b - qq) < (gb - gq) * (qq - a):
                keep.append(qq)
        keep.append(s[-1])
        s = keep
        out.append(list(s))
    return out


def search_evals(F, cand, arg_of_pos):
    """candidate evaluations of the three-level search when the candidates are `cand` (sorted positions; every argmin lies in it):
    level A: coarse positions 64 i scan the candidates inside their distance bound (the tile's smallest site value is taken as
    the line's); B: 8 positions per interval scan the candidates between the argmins of its two ends; C: likewise per chunk."""
    L = len(F)
    cand = np.asarray(cand)
    idx_of = {int(q): j for j, q in enumerate(cand)}
    m = min(int(F[q]) for q in cand)
    ev = 0
    coarse = list(range(0, L, 64))
    for p in coarse:
        j = int(np.argmin(np.abs(cand - p)))
        ub = min(int(F[cand[j]]) + (p - int(cand[j])) ** 2, int(F[cand[0]]) + (p - int(cand[0])) ** 2, int(F[cand[-1]]) + (p - int(cand[-1])) ** 2)
        w = int(np.sqrt(max(ub - m, 0))) + 2
        ev += int(np.count_nonzero((cand >= p - w) & (cand <= p + w)))
    def rng(p0, p1):
        lo = idx_of[int(arg_of_pos[p0])]
        hi = idx_of[int(arg_of_pos[p1])] if p1 < L else len(cand) - 1
        return max(hi - lo + 1, 1)
    for p0 in coarse:
        ev += 7 * rng(p0, p0 + 64)
    for p0 in range(0, L, 8):
        ev += min(8, L - p0) * rng(p0, p0 + 8)
    return ev


def study(name, sweep, lines, rng_seed=0, max_lines=600):
    rng = np.random.default_rng(rng_seed)
    pick = rng.permutation(len(lines))[:max_lines * 4]
    acc = {"sites": 0, "env": 0, "peel": [0, 0, 0], "ev_pos": 0, "ev_env": 0, "ev_peel": [0, 0, 0], "L": 0, "lines": 0}
    for li in pick:
        F = lines[li]
        sites = np.flatnonzero(F < INF)
        if len(sites) == 0:
            continue
        L = len(F)
        a = envelope_and_argmin(F)
        env = np.unique(a)
        Fl = [int(v) for v in F]
        peels = peel(Fl, [int(q) for q in sites], 3)
        acc["sites"] += len(sites)
        acc["env"] += len(env)
        acc["L"] += L
        acc["lines"] += 1
        # today's search: candidates = every POSITION of the site span (non-sites carry the sentinel and are evaluated too)
        acc["ev_pos"] += search_evals(np.where(F < INF, F, INF), np.arange(sites[0], sites[-1] + 1), a)
        acc["ev_env"] += search_evals(F, env, a)
        for k in range(3):
            acc["peel"][k] += len(peels[k])
            assert set(env.tolist()) <= set(peels[k]), "the peel dropped an envelope site"
            acc["ev_peel"][k] += search_evals(F, np.asarray(peels[k]), a)
        if acc["lines"] >= max_lines:
            break
    n = max(acc["lines"], 1)
    Lt = max(acc["L"], 1)
    row = {"scene": name, "sweep": sweep, "lines": acc["lines"], "L": Lt // n, "sites_per_line": round(acc["sites"] / n, 1),
           "envelope_per_line": round(acc["env"] / n, 1), "after_peels_per_line": [round(v / n, 1) for v in acc["peel"]],
           "evals_per_position": {"today_over_positions": round(acc["ev_pos"] / Lt, 2), "exact_envelope": round(acc["ev_env"] / Lt, 2),
                                  "after_peels_search_only": [round(v / Lt, 2) for v in acc["ev_peel"]],
                                  "after_peels_net_of_the_pass": [round((v + kPeelCost * sum(([acc["sites"]] + acc["peel"])[:k + 1])) / Lt, 2)
                                                                  for k, v in enumerate(acc["ev_peel"])]}}
    t = row["evals_per_position"]
    best = min(t["after_peels_net_of_the_pass"])
    row["net_change_best_peel_count"] = round(best / t["today_over_positions"] - 1.0, 3)
    return row


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nlines = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    scenes = [("room", room(n)), ("two-box cloud", two_box(n)), ("Bernoulli 1 %", synth.bernoulli_mask((n, n, n), 0.01, 11))]
    rows = []
    for name, m in scenes:
        Fy = z_dist_sq(m)                                        # y sweep: lines along y at fixed (x, z)
        ylines = Fy.transpose(0, 2, 1).reshape(-1, n)
        rows.append(study(name, "y", ylines, max_lines=nlines))
        print(json.dumps(rows[-1]), flush=True)
        Fx = plane_dist_sq(m)                                    # x sweep: lines along x at fixed (y, z)
        xlines = Fx.reshape(n, -1).T
        rows.append(study(name, "x", xlines, max_lines=nlines))
        print(json.dumps(rows[-1]), flush=True)
    worth = [r for r in rows if r["net_change_best_peel_count"] <= -0.25]
    print(json.dumps({"rule": "build only if evaluations fall by >= 25 % net of the compaction pass", "scenes_and_sweeps_that_pass": [(r["scene"], r["sweep"]) for r in worth],
                      "of": len(rows), "peel_cost_evals_per_list_entry_and_round": kPeelCost}))


if __name__ == "__main__":
    main()
