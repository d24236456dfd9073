#!/usr/bin/env python3
"""Flat-positive tiles, counted on the CPU before any HIP is written.

A far-field sweep line is FLAT-POSITIVE when every entry that is not a zero site (F = 0: a filled voxel of pass 0) carries the
same value c (c may be "no site").  Then min_q F(q) + (p - q)^2 = min(c, d0(p)^2), d0 = distance along the line to the nearest
zero site: no search is needed.  A floor under open space makes every y line of the free volume flat (c = height^2), and the
in-plane distances of the x sweep are flat along x wherever the plane's cross-section does not change.
This script reports, per scene and sweep, the share of 16-line tiles (the kernel's unit) whose lines are all flat-positive.
usage: flat_tile_model.py [n = 256]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compaction_model import INF, plane_dist_sq, room, two_box, z_dist_sq  # noqa: E402
from sdf_tools_amd import synth  # noqa: E402


def flat_lines(F):
    """F[..., L]: per line, are all non-zero entries equal?"""
    big = np.where(F == 0, np.int64(-1), F)
    mx = big.max(axis=-1)
    mn = np.where(F == 0, np.int64(1) << 62, F).min(axis=-1)
    return (mx <= 0) | (mn == mx)


def tiles(flat2d):
    """flat2d[o, c]: lines indexed by (outer, z); tiles = 16 consecutive z"""
    o, c = flat2d.shape
    pad = (-c) % 16
    f = np.pad(flat2d, ((0, 0), (0, pad)), constant_values=True).reshape(o, -1, 16)
    return f.all(axis=2)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    scenes = [("room", room(n)), ("two-box cloud", two_box(n)), ("solid boxes", synth.structured_boxes_mask((n, n, n)) if hasattr(synth, "structured_boxes_mask") else None)]
    for name, m in scenes:
        if m is None:
            continue
        Fy = z_dist_sq(m).transpose(0, 2, 1)                 # [x, z, y]: lines along y
        fy = flat_lines(Fy)
        has_site_y = (Fy < INF).any(axis=-1)
        ty = tiles(fy)
        sy = tiles(~has_site_y)                              # tiles with no site at all (already skipped)
        Fx = plane_dist_sq(m).transpose(1, 2, 0)             # [y, z, x]: lines along x
        fx = flat_lines(Fx)
        tx = tiles(fx)
        sx = tiles(~(Fx < INF).any(axis=-1))
        print(json.dumps({"scene": name, "n": n,
                          "y": {"flat_lines": round(float(fy.mean()), 3), "flat_tiles": round(float(ty.mean()), 3), "no_site_tiles": round(float(sy.mean()), 3)},
                          "x": {"flat_lines": round(float(fx.mean()), 3), "flat_tiles": round(float(tx.mean()), 3), "no_site_tiles": round(float(sx.mean()), 3)}}), flush=True)


if __name__ == "__main__":
    main()


def k_valued_lines(F, kmax=3):
    """per line: number of distinct non-zero values, capped at kmax + 1"""
    out = np.zeros(F.shape[:-1], np.int32)
    S = np.sort(np.where(F == 0, np.int64(-1), F), axis=-1)
    d = (np.diff(S, axis=-1) != 0) & (S[..., 1:] > 0)
    first = (S[..., :1] > 0)
    return np.minimum(d.sum(axis=-1) + first[..., 0], kmax + 1)


def main2():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for name, m in (("room", room(n)),):
        Fy = z_dist_sq(m).transpose(0, 2, 1)
        ky = k_valued_lines(Fy)
        Fx = plane_dist_sq(m).transpose(1, 2, 0)
        kx = k_valued_lines(Fx)
        row = {"scene": name, "n": n}
        for lab, k in (("y", ky), ("x", kx)):
            row[lab] = {f"tiles_all_lines_le_{v}_values": round(float(tiles(k <= v).mean()), 3) for v in (1, 2, 3)}
        print(json.dumps(row))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "k":
    main2()
