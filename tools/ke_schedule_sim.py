#!/usr/bin/env python3
"""CPU simulation of the far-field kernel's search schedule on REAL sweep inputs (development aid; numpy + scipy, no GPU).

For sampled tiles (16 memory-adjacent lines) of the y or the x sweep of a 512^3 scene it computes the exact argmins at the
positions of levels A / B / C, the trips every lane of levels B and C needs, and what a wave ISSUES (its slowest lane) under
  plain    : every lane scans its whole range (rounds 2 - 3),
  coop     : wave-uniform blocks 4, 8, 16, ... and the hand-over of long ranges to the wave's four rows (round 4; gate, block size,
             calm rule and the cost of a hand-over as in sdfgpu_envelope_dc.hpp),
  sorted   : level C only -- the tile's 64 (chunk x 16 lines) units dealt to the waves in order of their longest range.
The thresholds of scan8_calm / coop8 were chosen with this; the kernel's trip counters (tools/trip_counts.py on a GPU) read
10.0 / 13.4 / 21.5 issued and 9.6 / 9.6 / 13.1 needed trips per wave for levels A / B / C of the two-box x sweep.

usage: ke_schedule_sim.py <box|room|bernoulli p> <x|y> [tiles = 200]
"""
import sys
import os

import numpy as np
from scipy import ndimage

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdf_tools_amd import synth  # noqa: E402

N, RES = 512, 0.01
TRIP, HANDOVER, CHECK, GATE, SETUP = 29, 60, 12, 4, 12        # VALU wave instructions: one trip, one hand-over round, ...


def scene(kind):
    if kind == "box":
        pts = synth.two_box_points(200000, seed=0, scale=N * RES)
        idx = (pts.astype(np.float64) / RES).astype(np.int64)
        m = np.zeros((N, N, N), np.uint8)
        m[idx[:, 0], idx[:, 1], idx[:, 2]] = 1
        return m
    if kind == "room":
        return synth.room_mask_torch((N, N, N), "cpu").numpy()
    return synth.bernoulli_mask((N, N, N), float(kind), 1)


def zdist2(plane):
    """[y, z] occupancy of one x plane -> squared z distance to the nearest filled voxel of the row (inf: none): the y sweep's input."""
    out = np.full(plane.shape, np.inf)
    for y in range(plane.shape[0]):
        r = plane[y]
        if r.any():
            out[y] = ndimage.distance_transform_edt(r == 0) ** 2
    return out


def wave_cost(unit_trips, kmax, split, gate):
    """Instructions one wave issues for units with these (per-unit maximal) trip counts."""
    mx = int(unit_trips.max())
    if mx < gate:
        return GATE + mx * TRIP
    t, cost, blk = 0, GATE + SETUP, 4
    while t < mx:
        did = min(blk, mx - t)
        blk *= 2
        cost += did * TRIP
        t += did
        if t >= mx:
            break
        cost += CHECK
        if (unit_trips > t).sum() <= kmax and mx - t >= 6:
            return cost + sum(HANDOVER + int(np.ceil((u - t) / split)) * TRIP for u in unit_trips if u > t)
    return cost


def main():
    kind, axis = sys.argv[1], sys.argv[2]
    ntiles = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    m = scene(kind)
    rng = np.random.default_rng(0)
    q = np.arange(N)
    planes = {}
    if axis == "x":                                                 # the x sweep's input: in-plane squared distance of every x plane
        F = np.empty((N, N, N), np.float32)
        for x in range(N):
            F[x] = ndimage.distance_transform_edt(m[x] == 0) ** 2 if m[x].any() else np.inf
    acc = {k: 0.0 for k in ("B plain", "B coop", "B needed", "C plain", "C coop", "C sorted", "C needed")}
    nw = 0
    for _ in range(ntiles):
        z0 = int(rng.integers(0, N // 16)) * 16
        if axis == "x":
            lines = F[:, int(rng.integers(0, N)), z0:z0 + 16].astype(np.float64)
        else:
            x = int(rng.integers(0, N))
            if x not in planes:
                planes[x] = zdist2(m[x])
            lines = planes[x][:, z0:z0 + 16]
        sites = np.where(np.isfinite(lines).any(axis=1))[0]
        if len(sites) == 0:
            continue
        hi_t = sites.max()
        pos = np.arange(0, N, 8)
        am = np.argmin(lines[None, :, :] + ((pos[:, None] - q[None, :]) ** 2)[:, :, None], axis=1)       # [64, 16] argmins at 8 i
        a8 = np.vstack([am[1:], np.full((1, 16), hi_t)])
        tc = np.maximum((a8 - (am & ~1)) // 2 + 1, 1)                                                   # level C trips [chunk, line]
        uc = tc.max(axis=1)
        acc["C needed"] += tc.mean() * 4 * TRIP
        for i0 in range(0, 64, 16):
            for w in range(4):
                u = uc[i0 + 4 * w:i0 + 4 * w + 4]
                acc["C plain"] += u.max() * TRIP / 4
                acc["C coop"] += wave_cost(u, 2, 4, 18) / 4
        us = np.sort(uc)
        acc["C sorted"] += sum(us[4 * g:4 * g + 4].max() for g in range(16)) * TRIP / 4
        amA = am[::8]
        hiA = np.vstack([amA[1:], np.full((1, 16), hi_t)])
        tb = np.where((amA & ~1) <= hiA, (hiA - (amA & ~1)) // 4 + 1, 0)                                 # level B trips of share 0 [interval, line]
        acc["B needed"] += tb.mean() * TRIP
        for w in range(4):
            u = tb[2 * w:2 * w + 2].max(axis=1)
            acc["B plain"] += u.max() * TRIP / 4
            acc["B coop"] += wave_cost(u, 1, 2, 28) / 4
        nw += 1
    print("%s, %s sweep, %d tiles: VALU wave instructions per wave" % (kind, axis, nw))
    for k in acc:
        print("  %-9s %7.1f" % (k, acc[k] / nw))


if __name__ == "__main__":
    main()
