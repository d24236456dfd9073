"""Input scenes shared by the tests: the reference's own known-answer scenes
(restated as data, not code) and the synthetic stress inputs of SURVEY.md 8(d)."""
import numpy as np


def test_bindings_scene():
    """test/test_bindings.py:11-20: 20x40x1 grid, res 0.05, one filled cell at (x=3, y=1)."""
    m = np.zeros((20, 40, 1), np.uint8)
    m[3, 1, 0] = 1
    return m, 0.05


def tutorial_scene():
    """src/sdf_tools_tutorial.cpp:23-59: 10 m cube at 0.25 m -> 40^3, low octant filled."""
    m = np.zeros((40, 40, 40), np.uint8)
    m[:20, :20, :20] = 1
    return m, 0.25


def convex_segments_scene():
    """src/compute_convex_segments_test.cpp:13-41: 100x100x50, res 1: 10-cell wall around x/y,
    20x20 pillar in the middle, then a plus-shaped corridor carved free."""
    nx, ny, nz = 100, 100, 50
    x, y, _ = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    m = np.zeros((nx, ny, nz), np.uint8)
    wall = (x < 10) | (y < 10) | (x >= nx - 10) | (y >= ny - 10)
    pillar = (x >= 40) & (y >= 40) & (x < 60) & (y < 60)
    m[wall | (~wall & pillar)] = 1
    m[((x >= 45) & (x < 55)) | ((y >= 45) & (y < 55))] = 0
    return m, 1.0


def estimate_distance_scene():
    """src/estimate_distance_test.cpp:18-34: 10x10x1 grid, 11 filled cells (grid-frame
    coordinates; the 45-degree origin rotation does not change cell indices)."""
    m = np.zeros((10, 10, 1), np.uint8)
    for (x, y) in [(5, 5), (5, 6), (6, 5), (6, 6), (7, 7), (2, 2), (3, 2), (4, 2), (2, 3), (2, 4), (2, 7)]:
        m[x, y, 0] = 1
    return m, 1.0


def single_voxel(shape, at=None):
    m = np.zeros(shape, np.uint8)
    at = at or tuple(s // 2 for s in shape)
    m[at] = 1
    return m
