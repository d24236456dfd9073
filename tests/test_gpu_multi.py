"""GPU: libsdfgpu_multi.so (include/sdfgpu_multi.h) -- the x-slab multi-GPU build behind the C ABI.  On a single-GPU box
every logical rank lives on device 0 and the messages of each exchange (bit-plane halos, int32 halos, the x-slab ->
y-slab re-partition and its way back) travel as device-to-device copies; with one rank per GPU the same messages go
through RCCL.  Results must be bit-identical to the single-GPU ABI and the exact oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from sdf_tools_amd import capi, synth

pytestmark = pytest.mark.gpu


def _boxes(shape):
    nx, ny, nz = shape
    m = np.zeros(shape, np.uint8)
    m[nx // 10: nx // 10 + max(2, nx // 8), ny // 2: ny // 2 + max(2, ny // 6), : max(2, nz // 3)] = 1
    m[nx // 2: nx // 2 + max(2, nx // 5), ny // 8: ny // 8 + max(2, ny // 5), nz // 4: nz // 4 + max(2, nz // 4)] = 1
    return m


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_multi_matches_exact_on_every_tier(world):
    mg = capi.MultiSdfGpu(world, [0] * world)
    assert mg.n_ranks == world
    try:
        shape = (48, 40, 64)
        cases = [
            ("dense", synth.bernoulli_mask(shape, 0.5, 1), False, dict(dense_certified=True, whole_lines=False)),
            ("mid", synth.bernoulli_mask(shape, 0.06, 2), False, dict(dense_certified=False)),
            ("far", _boxes(shape), False, dict(dense_certified=False, whole_lines=True)),
            ("far vb", _boxes(shape), True, dict(whole_lines=True)),
            ("dense vb", synth.bernoulli_mask(shape, 0.5, 3), True, dict(dense_certified=False)),   # (slabs: no vb dense tier)
            ("empty", np.zeros(shape, np.uint8), False, dict(whole_lines=True)),
            ("odd shape", synth.bernoulli_mask((29, 18, 20), 0.02, 4), True, {}),
        ]
        for name, m, vb, expect in cases:
            got, ext = mg.build(m, 0.05, vb)
            want, want_ext, _ = O.exact_sdf(m, 0.05, vb)
            bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
            assert len(bad) == 0, (name, world, len(bad), bad[:3].tolist())
            assert ext == want_ext, (name, world, ext, want_ext)
            path = mg.last_path()
            for k, v in expect.items():
                assert path[k] == v, (name, world, path)
            assert not path["rccl"] or world == 1
    finally:
        mg.close()


def test_multi_cells_and_device_entry_points(gpu):
    import torch
    world = 3
    mg = capi.MultiSdfGpu(world, [0] * world)
    try:
        shape = (30, 16, 32)
        rng = np.random.default_rng(2)
        cells = np.zeros(shape + (2,), np.float32)
        cells[..., 0] = rng.choice(np.array([0.0, 0.5, 1.0], np.float32), size=shape, p=[0.5, 0.2, 0.3])
        for unknown in (False, True):
            got, ext = mg.build_cells(cells, shape, 8, 0, unknown, 0.1, False)
            want, want_ext = gpu.build_cells(cells, shape, 8, 0, unknown, 0.1, False)
            assert np.array_equal(got, want) and ext == want_ext
        m = _boxes(shape)
        dev = torch.device("cuda", 0)
        masks, outs = [], []
        for r in range(world):
            a, b = mg.slab_range(shape[0], r)
            masks.append(torch.from_numpy(m[a:b]).to(dev))
            outs.append(torch.empty((b - a,) + shape[1:], dtype=torch.float32, device=dev))
        ext = mg.build_device([t.data_ptr() for t in masks], shape, [t.data_ptr() for t in outs], 0.1, True)
        want, want_ext, _ = O.exact_sdf(m, 0.1, True)
        assert np.array_equal(torch.cat(outs).cpu().numpy(), want) and ext == want_ext
    finally:
        mg.close()


def test_multi_512_far_field_matches_single_gpu(gpu):
    """The streaming scene at 256 x 512 x 512 on 4 logical ranks == the single-GPU ABI, bit for bit."""
    shape = (256, 512, 512)
    pts = synth.two_box_points(100000, seed=1, scale=2.56)
    idx = (pts.astype(np.float64) / 0.01).astype(np.int64)
    idx = idx[(idx[:, 0] < shape[0]) & (idx[:, 1] < shape[1]) & (idx[:, 2] < shape[2])]
    m = np.zeros(shape, np.uint8)
    m[idx[:, 0], idx[:, 1], idx[:, 2]] = 1
    mg = capi.MultiSdfGpu(4, [0] * 4)
    try:
        got, ext = mg.build(m, 0.01)
        assert mg.last_path()["whole_lines"]
    finally:
        mg.close()
    want, want_ext = gpu.build(m, 0.01)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and ext == want_ext


def cavity_scene(shape=(512, 32, 512), cav=84, p=0.3, seed=5):
    """Near-field clutter (Bernoulli p) around ONE free cavity, cav voxels long in x and z and spanning the whole (short) y
    extent: the y sweep's outward scans end at the grid faces, so the y axis stays near-field, the x probe sees a
    near-field scene (the cavity is 3 % of the voxels, < 1/24), and the marching x sweep meets voxels that are still
    undecided after its 40-row scan bound (in-plane distance up to cav / 2 = 42), raises the far flag and leaves them to
    the far-field kernel -- whose values, not the marching sweep's upper bounds, must reach the extrema (ADVICE r2
    medium; sdfgpu_kernels.hpp `inexact`)."""
    nx, ny, nz = shape
    m = synth.bernoulli_mask(shape, p, seed)
    ax, az = (nx - cav) // 2, (nz - cav) // 2
    m[ax:ax + cav, :, az:az + cav] = 0
    return m


def test_cavity_in_clutter_extrema_on_the_whole_line_path(gpu):
    """The advisor's regression (VERDICT r3 missing #6) through sdfgpu_multi_build: 2 logical ranks, the halo path
    raises 'unresolved' inside the cavity, the re-partitioned complete-lines x sweep runs near-field + scan bound +
    far-field kernel; field and extrema bit-identical to the exact oracle and to the single-GPU ABI."""
    m = cavity_scene()
    want, want_ext, dsq = O.exact_sdf(m, 0.01)
    assert np.abs(dsq).max() > 41 * 41                               # the cavity is deeper than the scan bound
    mg = capi.MultiSdfGpu(2, [0, 0])
    try:
        got, ext = mg.build(m, 0.01)
        path = mg.last_path()
        assert path["whole_lines"] and not path["dense_certified"], path
    finally:
        mg.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert ext == want_ext, (ext, want_ext)
    one, one_ext = gpu.build(m, 0.01)
    assert np.array_equal(one.view(np.uint32), want.view(np.uint32)) and one_ext == want_ext
    lp = gpu.last_path()
    assert lp["far_x"] and not lp["far_y"], lp                       # ... there: marching y sweep; the far-field kernel redid the x sweep
