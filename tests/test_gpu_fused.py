"""GPU: the fused z+y kernel (K12, nz = 512 / 1024) against the unfused K1 + K2 path and the exact
oracle, on dense, sparse and degenerate rows (every far-search path of the row bitmap code)."""
import numpy as np
import pytest

import scenes
from oracle import oracle as O
from sdf_tools_amd import synth

pytestmark = pytest.mark.gpu


def _both(gpu, m, res=1.0, vb=False, window=0):
    shape = m.shape
    try:
        gpu.set_option("dense", 0)
        gpu.set_option("fused_zy", 2)                       # force K12 (the policy would pick K1 + K2 here)
        gpu.set_option("fused_window", window)
        a, ea = gpu.build(m, res, vb)
        assert gpu.last_build_fused_zy()
        yz_a = gpu.debug_yzsweep(shape)
        gpu.set_option("fused_zy", 0)
        b, eb = gpu.build(m, res, vb)
        assert not gpu.last_build_fused_zy()
        yz_b = gpu.debug_yzsweep(shape)
    finally:
        gpu.set_option("fused_zy", 1)
        gpu.set_option("fused_window", 0)
        gpu.set_option("dense", 1)
    return a, ea, yz_a, b, eb, yz_b


CASES = [((6, 40, 512), 0.5), ((5, 23, 512), 0.03), ((4, 50, 512), 0.0008), ((3, 9, 512), 0.995),
         ((3, 20, 1024), 0.5), ((2, 33, 1024), 0.002), ((1, 1, 512), 0.3), ((2, 6, 1024), 0.9)]


@pytest.mark.parametrize("shape,p", CASES)
@pytest.mark.parametrize("window", [0])
def test_fused_equals_unfused_and_exact(gpu, shape, p, window):
    m = synth.bernoulli_mask(shape, p, 17)
    a, ea, yz_a, b, eb, yz_b = _both(gpu, m, 0.25, False, window)
    bad = np.argwhere(yz_a != yz_b)
    assert len(bad) == 0, "yz field differs at %s: fused %s unfused %s" % (
        bad[:5].tolist(), yz_a[tuple(bad[0])], yz_b[tuple(bad[0])])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ea == eb
    ex, ex_ext, _ = O.exact_sdf(m, 0.25)
    assert np.array_equal(a.view(np.uint32), ex.view(np.uint32)) and ea == ex_ext


def test_fused_degenerate_rows(gpu):
    for shape in ((3, 12, 512), (2, 5, 1024)):
        nz = shape[2]
        cases = [np.zeros(shape, np.uint8), np.ones(shape, np.uint8), scenes.single_voxel(shape),
                 1 - scenes.single_voxel(shape), scenes.single_voxel(shape, (0, 0, 0)),
                 scenes.single_voxel(shape, (shape[0] - 1, shape[1] - 1, nz - 1))]
        half = np.zeros(shape, np.uint8)
        half[:, :, : nz // 2] = 1                       # one class boundary in the middle of every row
        cases.append(half)
        stripes = np.zeros(shape, np.uint8)
        stripes[:, :, 63::64] = 1                       # sites on word boundaries
        cases.append(stripes)
        edge = np.zeros(shape, np.uint8)
        edge[:, ::3, 0] = 1
        edge[:, 1::3, nz - 1] = 1
        cases.append(edge)
        for m in cases:
            for vb in (False, True):
                a, ea, yz_a, b, eb, yz_b = _both(gpu, m, 1.0, vb)
                assert np.array_equal(yz_a, yz_b)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ea == eb
                ex, ex_ext, _ = O.exact_sdf(m, 1.0, vb)
                assert np.array_equal(a.view(np.uint32), ex.view(np.uint32)) and ea == ex_ext


def test_fused_slab_entry_point(gpu):
    """sdfgpu_sweep_zy_device takes the fused kernel too (1024-wide rows = the 8-GPU configuration)."""
    import torch
    shape = (5, 24, 1024)
    m = synth.bernoulli_mask(shape, 0.5, 3)
    out = torch.empty(shape, dtype=torch.int32, device="cuda")
    gpu.sweep_zy_device(torch.from_numpy(m).cuda().data_ptr(), shape, out.data_ptr())
    torch.cuda.synchronize()
    gpu.set_option("fused_zy", 0)
    try:
        out2 = torch.empty(shape, dtype=torch.int32, device="cuda")
        gpu.sweep_zy_device(torch.from_numpy(m).cuda().data_ptr(), shape, out2.data_ptr())
        torch.cuda.synchronize()
    finally:
        gpu.set_option("fused_zy", 1)
    assert bool(torch.equal(out, out2))
