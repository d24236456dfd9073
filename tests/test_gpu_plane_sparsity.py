"""GPU: plane sparsity (round 6) -- builds that go straight to the far-field pair skip x-planes and z rows without a filled voxel.
Same fields bit for bit as the oracle and as the same build with the option off, intermediates included (the debug copies fill the
skipped rows / planes in), on scenes that have empty planes, planes where every row holds a filled voxel, and both."""
import numpy as np
import pytest

from oracle import oracle as O
from sdf_tools_amd import synth

pytestmark = pytest.mark.gpu


def _scenes(shape):
    nx, ny, nz = shape
    rng = np.random.default_rng(3)
    out = {}
    m = np.zeros(shape, np.uint8)                       # a few points in a slab of x: most planes empty, most rows of the others too
    for _ in range(40):
        m[rng.integers(nx // 3, nx // 2), rng.integers(0, ny), rng.integers(0, nz)] = 1
    out["points_in_a_slab"] = m
    m = np.zeros(shape, np.uint8)                       # a floor: every row of every plane holds a filled voxel (state 2 everywhere)
    m[:, :, :2] = 1
    m[nx // 2, ny // 2, nz // 2] = 1
    out["floor"] = m
    m = np.zeros(shape, np.uint8)                       # a wall across some planes + a box: empty planes, full planes, mixed planes
    m[nx // 4:nx // 4 + 3, :, :] = 1
    m[nx // 2:nx // 2 + 5, ny // 3:ny // 2, nz // 4:nz // 2] = 1
    out["wall_and_box"] = m
    out["single_voxel_last_plane"] = np.zeros(shape, np.uint8)
    out["single_voxel_last_plane"][nx - 1, ny - 1, nz - 1] = 1
    out["inverted_points"] = 1 - out["points_in_a_slab"]   # nearly everything filled: no empty plane, no empty row
    out["noise"] = synth.bernoulli_mask(shape, 0.002, 5)
    return out


@pytest.mark.parametrize("shape", [(64, 48, 64), (40, 33, 128), (130, 20, 64), (24, 600, 64), (16, 1024, 64)], ids=lambda s: "x".join(map(str, s)))
def test_far_field_pair_with_and_without_plane_sparsity(gpu, shape):
    import torch
    res = 0.01
    s = torch.cuda.current_stream().cuda_stream
    n = int(np.prod(shape))
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    for name, m in _scenes(shape).items():
        for vb in (False, True):
            ex, ex_ext, _ = O.exact_sdf(m, res, vb)
            mt = torch.from_numpy(m).cuda()
            fields = {}
            for skip in (1, 0):
                gpu.set_option("policy_reset", 1)
                gpu.set_option("dense", 0)
                gpu.set_option("far_predict", 2)
                gpu.set_option("plane_skip", skip)
                gpu.build_device(mt.data_ptr(), shape, out.data_ptr(), res, vb, s)
                ext = gpu.get_extrema()
                got = out.cpu().numpy()
                assert gpu.last_build_info()["far_predicted"]
                assert np.array_equal(got, ex) and ext == ex_ext, (name, vb, skip)
                fields[skip] = (gpu.debug_zsweep(shape).copy(), gpu.debug_yzsweep(shape).copy())
            # ... and through the bits entry point (the z sweep reads the bit field and writes the same row bytes)
            from sdf_tools_amd import capi
            bt = torch.from_numpy(capi.pack_bits_host(m).view(np.int32)).cuda()
            gpu.set_option("policy_reset", 1)
            gpu.set_option("plane_skip", 1)
            gpu.build_bits_device(bt.data_ptr(), shape, out.data_ptr(), res, vb, s)
            assert gpu.last_build_info()["far_predicted"] and np.array_equal(out.cpu().numpy(), ex) and gpu.get_extrema() == ex_ext, (name, vb, "bits")
            gpu.set_option("dense", 1)
            gpu.set_option("far_predict", 1)
            gpu.set_option("plane_skip", 1)
            assert np.array_equal(fields[1][0], fields[0][0]), (name, vb, "z field")
            assert np.array_equal(fields[1][1], fields[0][1]), (name, vb, "plane field")
    del n
