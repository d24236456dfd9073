#!/usr/bin/env python3
"""Writes tests/golden/tagged_map_serialized.hex: the bytes TaggedObjectCollisionMapGrid::SerializeSelf produces (through
pysdf_tools) for a small fixed grid.  A SELF-golden like collision_map_serialized.hex: it pins this repository's own wire
format against accidental change (field order of /root/reference/src/sdf_tools/tagged_object_collision_map.cpp:23-75 over
the in-tree arc_utilities primitives); it says nothing about byte interoperability with the reference, whose
arc_utilities dependency is not vendored -- wire-format parity stays unpinned."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdf_tools_amd._bindings import load_pysdf_tools  # noqa: E402


def make_grid(m):
    origin = m.Isometry3d([[0, -1, 0, 0.5], [1, 0, 0, -2.0], [0, 0, 1, 4.25], [0, 0, 0, 1]])
    g = m.TaggedObjectCollisionMapGrid(origin, "tagged_golden", 0.25, 2, 3, 2, m.TAGGED_OBJECT_COLLISION_CELL(-3.5, 11, 5, 2))
    k = 0
    for x in range(2):
        for y in range(3):
            for z in range(2):
                g.SetValue(x, y, z, m.TAGGED_OBJECT_COLLISION_CELL(0.25 * k, k % 4, 100 + k, 7 * k))
                k += 1
    return g


if __name__ == "__main__":
    m = load_pysdf_tools()
    blob = bytes(make_grid(m).SerializeSelf())
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tagged_map_serialized.hex")
    with open(out, "w") as f:
        f.write(blob.hex() + "\n")
    print(out, len(blob), "bytes")
