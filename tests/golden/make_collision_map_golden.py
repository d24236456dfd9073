#!/usr/bin/env python3
"""Writes tests/golden/collision_map_serialized.hex: the bytes CollisionMapGrid::SerializeSelf produces (through pysdf_tools)
for a small fixed grid.  A SELF-golden: it pins this repository's own wire format against accidental change (field order of
/root/reference/src/sdf_tools/collision_map.cpp:21-62 over the in-tree arc_utilities primitives); it says nothing about
byte interoperability with the reference, whose arc_utilities dependency is not vendored."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdf_tools_amd._bindings import load_pysdf_tools  # noqa: E402


def make_grid(m):
    origin = m.Isometry3d([[1, 0, 0, -1.0], [0, 1, 0, 2.5], [0, 0, 1, 0.25], [0, 0, 0, 1]])
    g = m.CollisionMapGrid(origin, "golden_frame", 0.5, 3, 2, 2, m.COLLISION_CELL(-7.5, 9))
    k = 0
    for x in range(3):
        for y in range(2):
            for z in range(2):
                g.SetValue(x, y, z, m.COLLISION_CELL(0.125 * k, k * 3))
                k += 1
    return g


if __name__ == "__main__":
    m = load_pysdf_tools()
    blob = bytes(make_grid(m).SerializeSelf())
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "collision_map_serialized.hex")
    with open(out, "w") as f:
        f.write(blob.hex() + "\n")
    print(out, len(blob), "bytes")
