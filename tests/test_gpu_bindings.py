"""GPU: the reference's own Python test (test/test_bindings.py:11-33) restated against this build's
pysdf_tools + utils_2d / utils_3d, plus C++-side queries (gradient, EstimateDistance, file / message
round trips) on a field produced by the HIP path."""
import os

import numpy as np
import pytest

import scenes
from oracle import oracle as O
from sdf_tools_amd import synth, utils_2d, utils_3d
from sdf_tools_amd._bindings import load_pysdf_tools

pytestmark = pytest.mark.gpu
m = load_pysdf_tools()
IDENT = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]


def test_reference_test_bindings_restated():
    res = 0.05
    x_width, y_height = 20, 40
    grid_world = np.zeros([y_height, x_width], dtype=np.uint8)
    grid_world[1, 3] = 1
    sdf_origin = [0 - x_width / 2, 0 - y_height / 2]
    sdf, sdf_gradient = utils_2d.compute_sdf_and_gradient(grid_world, res, sdf_origin)
    assert sdf[1, 3] == pytest.approx(-res, abs=1e-7)
    assert sdf[2, 3] == pytest.approx(res, abs=1e-7)
    assert sdf[0, 3] == pytest.approx(res, abs=1e-7)
    assert sdf[1, 2] == pytest.approx(res, abs=1e-7)
    assert sdf[1, 4] == pytest.approx(res, abs=1e-7)
    assert sdf[3, 6] > 3 * res
    assert sdf.shape == (y_height, x_width)
    assert sdf_gradient.shape == (y_height, x_width, 2)
    np.testing.assert_allclose(sdf_gradient[1, 4], [1.5, 0], atol=1e-6)


def test_utils_3d_conventions_against_oracle():
    env = (synth.bernoulli_mask((12, 9, 10), 0.4, 3)).astype(np.float32)          # env is [y, x, z]
    sdf, grad = utils_3d.compute_sdf_and_gradient(env, 0.1, [0.0, 0.0, 0.0])
    mask_xyz = np.transpose(env == 1, [1, 0, 2]).astype(np.uint8)
    want, _ = O.reference_sdf(mask_xyz, 0.1)
    assert np.array_equal(sdf, np.transpose(want, [1, 0, 2]))
    assert grad.shape == env.shape + (3,) and grad.dtype == np.float32
    obj = utils_3d.compute_sdf(env, 0.1, [0.0, 0.0, 0.0])
    assert np.array_equal(np.array(obj.GetRawData(), np.float32).reshape(9, 12, 10), want)
    g64 = utils_3d.get_gradient(obj)
    assert np.array_equal(np.transpose(g64, [1, 0, 2, 3]).astype(np.float32), grad)


def test_collision_map_extract_both_seams_and_extrema():
    mask, res = scenes.tutorial_scene()
    origin = m.Isometry3d([[1, 0, 0, -5.0], [0, 1, 0, -5.0], [0, 0, 1, -5.0], [0, 0, 0, 1]])
    g = m.CollisionMapGrid(origin, "tutorial_frame", res, 40, 40, 40, m.COLLISION_CELL(0.0))
    g.SetOccupancyFromNumpy(mask.astype(np.float32))
    sdf, (mx, mn) = g.ExtractSignedDistanceField(0.0, False, False)
    want, want_ext = O.reference_sdf(mask, res)
    assert np.array_equal(sdf.GetRawDataNumpy(), want) and (mx, mn) == want_ext
    assert sdf.GetFrame() == "tutorial_frame" and sdf.GetResolution() == res
    sdf2, ext2 = g.ExtractSignedDistanceFieldViaPredicate(0.0, False, False)
    assert np.array_equal(sdf2.GetRawDataNumpy(), want) and ext2 == want_ext
    v, ok = sdf.GetValueByIndex(10, 10, 10)
    assert ok and v == -2.5
    v, ok = sdf.GetValueByCoordinates(-5.0 + 0.25 * 20.5, -5.0 + 0.25 * 20.5, -5.0 + 0.25 * 20.5)
    assert ok and v == pytest.approx(0.4330127, abs=1e-7)
    v, ok = sdf.GetValueByIndex(40, 0, 0)
    assert not ok and v == 0.0
    # interior gradient = central differences (sdf.hpp:447-458); +x of the box face points along +x
    gx = sdf.GetGradient(25, 10, 10, False)
    assert gx == pytest.approx([1.0, 0.0, 0.0], abs=1e-6)
    assert sdf.GetGradient(0, 0, 0, False) == []                          # edge gradients disabled
    assert len(sdf.GetGradient(0, 0, 0, True)) == 3
    # trilinear estimate at a cell centre = value shrunk by half a cell (sdf.hpp:773-796)
    est, ok = sdf.EstimateDistance(-5.0 + 0.25 * 30.5, -5.0 + 0.25 * 10.5, -5.0 + 0.25 * 10.5)
    assert ok and est == pytest.approx(float(want[30, 10, 10]) - 0.125, abs=1e-6)


@pytest.mark.parametrize("unknown_is_filled", [False, True])
def test_config1_64cube_random_occupancy_through_the_class(unknown_is_filled):
    """BASELINE.json configs[0] end to end: a 64^3 CollisionMapGrid with random occupancy ->
    CollisionMapGrid.ExtractSignedDistanceField (collision_map.hpp:680-712) -> bit-identical to the reference algorithm
    (oracle) on the same grid, extrema included; cells are set one by one through the reference-shaped SetValue and in
    bulk through the numpy fast path."""
    n, res = 64, 0.02
    rng = np.random.default_rng(64)
    occ = rng.choice(np.array([0.0, 0.5, 1.0], np.float32), size=(n, n, n), p=[0.45, 0.1, 0.45])
    origin = m.Isometry3d([[1, 0, 0, -0.64], [0, 1, 0, -0.64], [0, 0, 1, -0.64], [0, 0, 0, 1]])
    g = m.CollisionMapGrid(origin, "world", res, n, n, n, m.COLLISION_CELL(0.0))
    g.SetOccupancyFromNumpy(occ)
    for (x, y, z) in ((0, 0, 0), (63, 63, 63), (17, 5, 40)):              # reference-shaped writes land in the same cells
        assert g.SetValue(x, y, z, m.COLLISION_CELL(1.0))
        occ[x, y, z] = 1.0
    assert g.GetNumXCells() == n and g.GetNumYCells() == n and g.GetNumZCells() == n
    mask = ((occ > 0.5) | (unknown_is_filled & (occ == 0.5))).astype(np.uint8)
    for vb in (False, True):
        sdf, ext = g.ExtractSignedDistanceField(float("inf"), unknown_is_filled, vb)
        want, want_ext = O.reference_sdf(mask, res, vb)
        assert np.array_equal(sdf.GetRawDataNumpy().view(np.uint32), want.view(np.uint32)), (unknown_is_filled, vb)
        assert tuple(ext) == want_ext
        assert np.array_equal(np.signbit(sdf.GetRawDataNumpy()), mask != 0)      # occupancy / sign bit-exact
    v, ok = sdf.GetValueByIndex(17, 5, 40)
    assert ok and v < 0.0


def test_device_resident_field_answers_batched_queries():
    """Round 4, N1 for callers of the class API: CollisionMapGrid.ExtractSignedDistanceFieldDevice leaves the field in HBM;
    QueryBatch = n x (EstimateDistance3d, GetGradient3d) (sdf.hpp:947-953, :395-403) from one kernel, no download; Host()
    then yields the same container ExtractSignedDistanceField returns."""
    n, res = 40, 0.05
    mask = synth.bernoulli_mask((n, n, n), 0.08, 9)
    origin = m.Isometry3d([[0, -1, 0, 0.3], [1, 0, 0, -0.2], [0, 0, 1, 0.1], [0, 0, 0, 1]])       # rotated + shifted frame
    g = m.CollisionMapGrid(origin, "world", res, n, n, n, m.COLLISION_CELL(0.0))
    g.SetOccupancyFromNumpy(mask.astype(np.float32))
    dsdf, ext = g.ExtractSignedDistanceFieldDevice(float("inf"), False, False)
    host, host_ext = g.ExtractSignedDistanceField(float("inf"), False, False)
    assert tuple(ext) == tuple(host_ext) == tuple(dsdf.GetExtrema())
    assert (dsdf.GetNumXCells(), dsdf.GetResolution(), dsdf.GetFrame()) == (n, res, "world")
    rng = np.random.default_rng(3)
    grid_pts = rng.uniform(-0.1, n * res + 0.1, size=(4000, 3))
    R, t = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]), np.array([0.3, -0.2, 0.1])
    pts = grid_pts @ R.T + t                                          # world-frame points
    dist, grad, flags = dsdf.QueryBatch(pts, True)
    assert not dsdf.HostCopyExists()                                  # nobody asked for the field yet
    inside = 0
    for i in range(0, len(pts), 7):
        e, ok = host.EstimateDistance(*pts[i])
        assert ok == bool(flags[i] & 1)
        if ok:
            inside += 1
            assert dist[i] == pytest.approx(e, abs=1e-9)
        else:
            assert np.isinf(dist[i])
    assert inside > 300
    back = dsdf.Host()
    assert dsdf.HostCopyExists() and np.array_equal(back.GetRawDataNumpy(), host.GetRawDataNumpy())
    assert back.GetFrame() == "world"
    with pytest.raises(ValueError):
        dsdf.QueryBatch(np.zeros((4, 2)), False)


def test_file_and_message_round_trip(tmp_path):
    mask = synth.bernoulli_mask((9, 8, 7), 0.5, 2)
    g = m.CollisionMapGrid(m.Isometry3d(IDENT), "world", 0.5, 9, 8, 7, m.COLLISION_CELL(0.0))
    g.SetOccupancyFromNumpy(mask.astype(np.float32))
    sdf, _ = g.ExtractSignedDistanceField(-1.0, False, True)
    for compress in (True, False):
        path = os.path.join(str(tmp_path), "f.sdf")
        sdf.SaveToFile(path, compress)
        assert open(path, "rb").read(4) == (b"SDFZ" if compress else b"SDFR")
        back = m.SignedDistanceField.LoadFromFile(path)
        assert np.array_equal(back.GetRawDataNumpy(), sdf.GetRawDataNumpy())
        assert back.GetFrame() == "world" and back.GetResolution() == 0.5
    msg = sdf.GetMessageRepresentation()
    assert msg.is_compressed and msg.frame_id == "world"
    back = m.SignedDistanceField.LoadFromMessageRepresentation(msg)
    assert np.array_equal(back.GetRawDataNumpy(), sdf.GetRawDataNumpy())
    with pytest.raises(ValueError):
        m.SignedDistanceField.LoadFromFile(os.path.join(str(tmp_path), "missing.sdf"))


def test_tagged_object_cells_match_oracle_per_filter():
    """N4: the three TaggedObjectCollisionMapGrid predicates (tagged_object_collision_map.hpp:736-749,
    :757-775, :817-827) evaluated on the device from raw 16-byte cells == the oracle on the numpy mask."""
    from sdf_tools_amd import capi
    rng = np.random.default_rng(11)
    shape = (24, 20, 28)
    cells = np.zeros(shape, dtype=np.dtype([("occupancy", "<f4"), ("component", "<u4"), ("object_id", "<u4"),
                                            ("convex_segment", "<u4")]))
    cells["occupancy"] = rng.choice(np.array([0.0, 0.5, 1.0], dtype=np.float32), size=shape, p=[0.6, 0.1, 0.3])
    cells["object_id"] = rng.integers(0, 5, size=shape)
    cells["component"] = rng.integers(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
    g = capi.SdfGpu(0)
    for unknown in (False, True):
        occ = (cells["occupancy"] > 0.5) | (unknown & (cells["occupancy"] == 0.5))
        for mode, ids, mask in ((0, (), occ), (1, (), occ & (cells["object_id"] > 0)),
                                (2, (2, 4), occ & np.isin(cells["object_id"], (2, 4))), (2, (), occ),
                                (2, (77,), np.zeros(shape, bool))):
            for vb in (False, True):
                got, ext = g.build_tagged_cells(cells, shape, object_mode=mode, object_ids=ids, unknown_is_filled=unknown,
                                                resolution=0.25, add_virtual_border=vb)
                want, want_ext, _ = O.exact_sdf(mask.astype(np.uint8), 0.25, vb)
                np.testing.assert_array_equal(got, want)
                assert ext == tuple(float(v) for v in want_ext)
                # cells = NULL: the records of the call above are still on the device (MakeObjectSDFs uploads once)
                again, ext2 = g.build_tagged_cells(None, shape, object_mode=mode, object_ids=ids, unknown_is_filled=unknown,
                                                   resolution=0.25, add_virtual_border=vb)
                np.testing.assert_array_equal(again, want)
                assert ext2 == ext
    # no records of that size on the handle: refused, and the handle stays usable
    with pytest.raises(capi.SdfGpuError):
        g.build_tagged_cells(None, (8, 8, 8))
    g.build(np.zeros((8, 8, 32), np.uint8))                   # another host-buffer entry point takes the staging buffer
    with pytest.raises(capi.SdfGpuError):
        g.build_tagged_cells(None, shape)
    got, _ = g.build_tagged_cells(cells, shape, object_mode=1)
    np.testing.assert_array_equal(got, O.exact_sdf(((cells["occupancy"] > 0.5) & (cells["object_id"] > 0)).astype(np.uint8), 1.0)[0])


def test_cell_predicate_overload_matches_oracle():
    """A8: the cell-predicate overload (reference sdf_generation.hpp:422-441) instantiated with a caller's predicate --
    here a Python lambda on COLLISION_CELL with its own threshold and a component filter."""
    rng = np.random.default_rng(5)
    shape = (14, 11, 16)
    occ = rng.random(shape).astype(np.float32)
    g = m.CollisionMapGrid(m.Isometry3d(IDENT), "world", 0.2, *shape, m.COLLISION_CELL(0.0))
    g.SetOccupancyFromNumpy(occ)
    calls = []

    def is_filled(cell):
        calls.append(1)
        return cell.occupancy > 0.7 and cell.component == 0

    sdf, ext = g.ExtractSignedDistanceFieldCellPredicate(is_filled, 0.0)
    assert len(calls) == occ.size                                      # evaluated exactly once per voxel
    want, want_ext = O.reference_sdf((occ > 0.7).astype(np.uint8), 0.2)
    assert np.array_equal(sdf.GetRawDataNumpy(), want) and ext == want_ext


@pytest.mark.parametrize("shape", [(9, 12, 10), (33, 40, 64), (1, 20, 40)])
def test_full_gradient_gpu_fast_path_equals_host_loop(shape):
    """N1 behind the reference API: GetFullGradientNumpy (one kernel through sdfgpu_gradient) == the per-voxel
    GetGradient loop of sdf.hpp:341-358, bit for bit, with and without edge gradients, identity and rotated frames."""
    mask = synth.bernoulli_mask(shape, 0.3, 7)
    rot = [[0.0, -1.0, 0.0, 0.3], [1.0, 0.0, 0.0, -0.2], [0.0, 0.0, 1.0, 0.1], [0, 0, 0, 1]]
    for frame in (IDENT, rot):
        g = m.CollisionMapGrid(m.Isometry3d(frame), "world", 0.05, *shape, m.COLLISION_CELL(0.0))
        g.SetOccupancyFromNumpy(mask.astype(np.float32))
        sdf, _ = g.ExtractSignedDistanceField(123.0, False, False)
        for edge in (True, False):
            fast = sdf.GetFullGradientNumpy(edge)
            slow = sdf.GetFullGradientNumpyHost(edge)
            assert fast.shape == shape + (3,)
            assert np.array_equal(fast, slow)


def test_full_gradient_256_is_fast():
    """utils_3d.compute_sdf_and_gradient at 256^3 end to end (< 100 ms for the gradient part)."""
    import time
    n = 256
    env = synth.bernoulli_mask((n, n, n), 0.5, 1).astype(np.float32)
    obj = utils_3d.compute_sdf(env, 0.01, [0.0, 0.0, 0.0])
    obj.GetFullGradientNumpy(True)
    t0 = time.perf_counter()
    grad = obj.GetFullGradientNumpy(True)
    dt = time.perf_counter() - t0
    assert grad.shape == (n, n, n, 3)
    print("GetFullGradientNumpy 256^3: %.1f ms" % (dt * 1e3))
    assert dt < 1.0                                                    # the per-voxel host loop takes ~10 s
    # spot-check against the per-voxel API
    for (x, y, z) in [(0, 0, 0), (5, 7, 9), (255, 255, 255), (128, 0, 3)]:
        assert list(grad[x, y, z]) == obj.GetGradient(x, y, z, True)


def test_tagged_object_filter_takes_many_ids():
    """More object ids than the old 4096 cap; ids unsorted."""
    from sdf_tools_amd import capi
    rng = np.random.default_rng(3)
    shape = (16, 16, 32)
    cells = np.zeros(shape, dtype=np.dtype([("occupancy", "<f4"), ("component", "<u4"), ("object_id", "<u4"),
                                            ("convex_segment", "<u4")]))
    cells["occupancy"] = (rng.random(shape) < 0.3).astype(np.float32)
    cells["object_id"] = rng.integers(0, 20000, size=shape)
    ids = rng.permutation(np.arange(0, 20000, 2, dtype=np.uint32))     # 10000 even ids, shuffled
    g = capi.SdfGpu(0)
    got, ext = g.build_tagged_cells(cells, shape, object_mode=2, object_ids=ids, resolution=0.5)
    mask = (cells["occupancy"] > 0.5) & (cells["object_id"] % 2 == 0)
    want, want_ext, _ = O.exact_sdf(mask.astype(np.uint8), 0.5)
    np.testing.assert_array_equal(got, want)
    assert ext == tuple(float(v) for v in want_ext)
