"""GPU: the 16-bit plane-field pipeline (int16 saturated plane field + int32 side table, K3/16 with
the packed 16-bit register window and the LDS sqrt table) against the 32-bit pipeline and the exact
oracle -- dense scenes (window path), sparse scenes (exact outward scan), saturated planes
(side table), virtual border, large distances (fp64 finish beyond the table)."""
import numpy as np
import pytest

import scenes
from oracle import oracle as O
from sdf_tools_amd import synth

pytestmark = pytest.mark.gpu


def _run(gpu, m, res, vb, **opts):
    try:
        gpu.set_option("dense", 0)
        for k, v in opts.items():
            gpu.set_option(k, v)
        sdf, ext = gpu.build(m, res, vb)
        info = gpu.last_build_info()
        yz = gpu.debug_yzsweep(tuple([1] * (3 - len([s for s in m.shape if s > 1])) + [s for s in m.shape if s > 1]))
    finally:
        for k, v in {"plane16": 1, "x16_voxels_per_lane": 4, "x16_window": 3, "fused_zy": 1, "dense": 1}.items():
            gpu.set_option(k, v)
    return sdf, ext, yz, info


CASES = [((16, 16, 16), 0.5, False), ((24, 20, 16), 0.5, True), ((9, 7, 8), 0.3, False), ((33, 10, 64), 0.05, False),
         ((40, 40, 40), 0.002, True), ((6, 40, 512), 0.5, False), ((5, 30, 512), 0.001, True),
         ((3, 12, 1024), 0.5, False), ((70, 66, 80), 0.0005, False), ((20, 40, 1), 0.1, False), ((2, 3, 256), 0.97, True)]


@pytest.mark.parametrize("shape,p,vb", CASES)
@pytest.mark.parametrize("variant", [(4, 3), (4, 2), (8, 3)])
def test_plane16_equals_plane32_and_exact(gpu, shape, p, vb, variant):
    m = synth.bernoulli_mask(shape, p, 23)
    a, ea, yz_a, ia = _run(gpu, m, 0.05, vb, plane16=1, x16_voxels_per_lane=variant[0], x16_window=variant[1])
    b, eb, yz_b, ib = _run(gpu, m, 0.05, vb, plane16=0)
    dims = [s for s in shape if s > 1]
    eligible = (dims[-1] % 4 == 0) and ((dims[-1] * (dims[-2] if len(dims) > 1 else 1)) % 8 == 0)
    assert ia["plane16"] == eligible and not ib["plane16"]
    assert np.array_equal(yz_a, yz_b)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ea == eb
    ex, ex_ext, _ = O.exact_sdf(m, 0.05, vb)
    assert np.array_equal(a.view(np.uint32), ex.view(np.uint32)) and ea == ex_ext


def test_saturated_planes_use_the_side_table(gpu):
    """In-plane squared distances beyond 32767 (and planes with no opposite voxel at all)."""
    shape = (6, 400, 512)
    for m in (scenes.single_voxel(shape, (0, 0, 0)), 1 - scenes.single_voxel(shape, (5, 399, 511)),
              scenes.single_voxel(shape, (3, 200, 256)), np.zeros(shape, np.uint8)):
        for fused in (2, 0):
            a, ea, yz_a, ia = _run(gpu, m, 1.0, False, plane16=1, fused_zy=fused)
            assert ia["plane16"] and ia["fused_zy"] == bool(fused)
            ex, ex_ext, dsq = O.exact_sdf(m, 1.0)
            assert np.array_equal(a.view(np.uint32), ex.view(np.uint32)) and ea == ex_ext
    assert np.abs(yz_a).max() >= 1 << 30          # the all-free grid: every plane saturated


def test_reference_scenes_through_plane16(gpu):
    for scene in (scenes.tutorial_scene, scenes.convex_segments_scene, scenes.test_bindings_scene):
        m, res = scene()
        a, ea, _, ia = _run(gpu, m, res, False, plane16=1)
        ref, ref_ext = O.reference_sdf(m, res)
        assert np.array_equal(a, ref) and ea == ref_ext
