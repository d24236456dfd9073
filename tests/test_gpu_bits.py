"""GPU: the bits-in entry points (round 6, VERDICT r5 "next round" 4) -- sdfgpu_build_bits_device / sdfgpu_build_bits /
sdfgpu_voxelize_points_bits_device -- against the byte-mask entry points and the oracle: one bit per voxel in, the same field,
extrema and tier choices out; and the host-buffer builds, which now feed their host-classified bits straight into the build."""
import numpy as np
import pytest

from oracle import oracle as O
from sdf_tools_amd import capi, synth

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64, 64), (32, 48, 128), (48, 40, 50), (20, 40, 1), (33, 17, 96), (8, 8, 1024), (25, 20, 15), (1, 1, 77), (16, 24, 36)]


def _bits_tensor(torch, mask_np):
    return torch.from_numpy(capi.pack_bits_host(mask_np).view(np.int32)).cuda()


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("p", [0.5, 0.04, 0.002])
def test_bits_in_equals_mask_in_and_the_oracle(gpu, shape, p):
    import torch
    res = 0.013
    for vb in (False, True):
        m = synth.bernoulli_mask(shape, p, 7)
        mt = torch.from_numpy(m).cuda()
        bt = _bits_tensor(torch, m)
        a = torch.empty(shape, dtype=torch.float32, device="cuda")
        b = torch.full(shape, 7.0, dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        gpu.set_option("policy_reset", 1)
        gpu.build_device(mt.data_ptr(), shape, a.data_ptr(), res, vb, s)
        ea, pa = gpu.get_extrema(), gpu.last_path()
        gpu.set_option("policy_reset", 1)
        gpu.build_bits_device(bt.data_ptr(), shape, b.data_ptr(), res, vb, s)
        eb, pb = gpu.get_extrema(), gpu.last_path()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and ea == eb
        assert {k: pa[k] for k in ("dense_certified", "far_y", "far_x")} == {k: pb[k] for k in ("dense_certified", "far_y", "far_x")}
        ex, ex_ext, _ = O.exact_sdf(m, res, vb)
        assert np.array_equal(b.cpu().numpy(), ex) and eb == ex_ext


def test_bits_in_through_every_tier_and_an_unaligned_field(gpu):
    """dense tier (in place and, for a field that is only 4-byte aligned, through the device copy), the sweep tier, the far-field
    pair, the stand-by behind a trusted dense tier -- each forced by options -- at 128^3."""
    import torch
    shape, res = (128, 128, 128), 0.01
    s = torch.cuda.current_stream().cuda_stream
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    for p, opts in ((0.5, {}), (0.5, {"dense": 0}), (0.03, {}), (0.002, {"dense": 0, "envelope_mode": 1}), (0.002, {"far_predict": 2}),
                    (0.5, {"expect_dense": 1}), (0.001, {"expect_dense": 1})):
        m = synth.bernoulli_mask(shape, p, 3)
        ex, ex_ext, _ = O.exact_sdf(m, res)
        words = capi.pack_bits_host(m).view(np.int32)
        for shift in (0, 1):                        # shift 1: the field starts 4 bytes into an allocation
            buf = torch.zeros(words.size + 4, dtype=torch.int32, device="cuda")
            buf[shift:shift + words.size] = torch.from_numpy(words).cuda()
            gpu.set_option("policy_reset", 1)
            for k, v in opts.items():
                gpu.set_option(k, v)
            gpu.build_bits_device(buf.data_ptr() + 4 * shift, shape, out.data_ptr(), res, False, s)
            ext = gpu.get_extrema()
            for k in opts:
                gpu.set_option(k, {"dense": 1, "envelope_mode": 0, "far_predict": 1, "expect_dense": 0}[k])
            assert np.array_equal(out.cpu().numpy(), ex) and ext == ex_ext, (p, opts, shift)


def test_host_bits_and_host_builds_feed_bits(gpu):
    """sdfgpu_build_bits (host bit field in, host field out) and the mask / cell host builds, whose host-classified bits are now
    the build's input (no unpack -> pack round trip on the device): same fields as the device-resident mask build."""
    shape, res = (96, 80, 64), 0.02
    for p in (0.5, 0.01):
        m = synth.bernoulli_mask(shape, p, 5)
        ex, ex_ext, _ = O.exact_sdf(m, res)
        got, ext = gpu.build_bits(capi.pack_bits_host(m), shape, res)
        assert np.array_equal(got, ex) and ext == ex_ext
        gpu.set_option("host_pack", 2)              # classify on the host whatever the size
        got2, ext2 = gpu.build(m, res)
        cells = np.zeros(shape + (2,), np.float32)
        cells[..., 0] = m
        got3, ext3 = gpu.build_cells(cells, shape, 8, 0, False, res, False)
        gpu.set_option("host_pack", 1)
        assert np.array_equal(got2, ex) and ext2 == ex_ext and np.array_equal(got3, ex) and ext3 == ex_ext
    with pytest.raises(capi.SdfGpuError):
        gpu._check(gpu._lib.sdfgpu_build_bits(gpu._h, None, 4, 4, 4, 1.0, 0, None, None, None))


def test_voxelize_into_bits_equals_voxelize_into_a_mask(gpu):
    import torch
    shape, res, origin = (25, 20, 15), 0.04, (0.0, 0.0, 0.0)
    rng = np.random.RandomState(0)
    pc = np.concatenate([rng.uniform([0.5, 0.5, 0], [0.7, 0.6, 0.5], [100, 3]), rng.uniform([0.5, 0.2, 0.25], [0.75, 0.4, 0.5], [100, 3]),
                         np.array([[-0.5, 0.1, 0.1], [0.1, 0.1, 5.0], [np.nan, 0, 0], [0.99, 0.79, 0.59], [-0.01, 0.0, 0.0]])]).astype(np.float32)
    pt = torch.from_numpy(pc).cuda()
    mask = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    n = int(np.prod(shape))
    bits = torch.full(((n + 31) // 32,), -1, dtype=torch.int32, device="cuda")
    gpu.voxelize_points_device(pt.data_ptr(), len(pc), origin, res, shape, mask.data_ptr())
    gpu.voxelize_points_bits_device(pt.data_ptr(), len(pc), origin, res, shape, bits.data_ptr())
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), capi.pack_bits_host(mask.cpu().numpy()))
    half = len(pc) // 2                             # without clearing the points accumulate
    gpu.voxelize_points_bits_device(pt.data_ptr(), half, origin, res, shape, bits.data_ptr(), clear_first=True)
    gpu.voxelize_points_bits_device(pt[half:].contiguous().data_ptr(), len(pc) - half, origin, res, shape, bits.data_ptr(), clear_first=False)
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), capi.pack_bits_host(mask.cpu().numpy()))
