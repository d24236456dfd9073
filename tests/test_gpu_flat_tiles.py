"""GPU: two-valued tiles (round 6) -- the far-field y sweep of builds that go straight to the far-field pair skips its three search
levels in tiles whose 16 lines all carry at most TWO values outside their zero sites (the open volume above a floor; lines over or
under a table top) and takes min(mx, d0^2, mn + d1^2) instead.  Same fields bit for bit as the oracle and as the same build with the option off, the y sweep's own output included,
on scenes made of such tiles, scenes that mix them with ordinary ones, and scenes that only look flat.
(The row flags the test rests on exist for nz = 64 ... 1024, powers of two: the shapes vary x and y.)"""
import numpy as np
import pytest

from oracle import oracle as O
from sdf_tools_amd import synth

pytestmark = pytest.mark.gpu


def _scenes(shape):
    nx, ny, nz = shape
    rng = np.random.default_rng(11)
    out = {}
    m = np.zeros(shape, np.uint8)                       # floor + a wall at low y: every y line = zero sites, then one value
    m[:, :, :2] = 1
    m[:, :max(ny // 16, 1), :] = 1
    out["floor_and_y_wall"] = m
    m = np.zeros(shape, np.uint8)                       # floor + pillars: zero sites in the middle of lines, several per line, at chunk edges
    m[:, :, 0] = 1
    for y in sorted(v for v in {0, 7, 8, 9, ny // 2, ny - 1, 63, 64} if v < ny):
        m[nx // 3:nx // 3 + 2, y, :] = 1
    m[nx // 2, ::5, :] = 1
    for x in sorted(v for v in {0, 7, 8, 9, 63, 64, nx - 1} if v < nx):       # ... and across x lines
        m[x, ny // 4:ny // 4 + 2, :] = 1
    out["floor_and_pillars"] = m
    m = np.zeros(shape, np.uint8)                       # floor and ceiling: still one value per line; thick floor = filled voxels deep inside
    m[:, :, :max(nz // 8, 1)] = 1
    m[:, :, nz - 1] = 1
    out["floor_and_ceiling"] = m
    out["room"] = synth.room_mask_torch(shape, "cpu").numpy()      # flat tiles next to tiles the furniture breaks
    m = np.zeros(shape, np.uint8)                       # a floor with a step: two values per line in the planes of the step (not flat), one elsewhere
    m[:, :, 0] = 1
    m[nx // 4:nx // 2, ny // 3:, :max(nz // 5, 2)] = 1
    out["floor_with_step"] = m
    m = np.zeros(shape, np.uint8)                       # a plate floating over the floor: two values on the lines over / under it,
    m[:, :, 0] = 1                                      # its edge at a chunk boundary and off one
    m[nx // 4:, 8:max(ny // 2, 9) + 3, nz // 2] = 1
    out["floor_and_plate"] = m
    m = m.copy()                                        # ... and a second, smaller one at another height: three values where both are crossed
    m[nx // 2:, max(ny // 2, 9) + 5:, nz // 4] = 1
    out["floor_and_two_plates"] = m
    m = np.zeros(shape, np.uint8)                       # floor, plate and pillars through both: zero sites between the two values
    m[:, :, 0] = 1
    m[:, ny // 3:, nz // 3] = 1
    m[::3, ::7, :nz // 2] = 1
    out["floor_plate_pillars"] = m
    m = np.zeros(shape, np.uint8)                       # walls of whole rows across the lines, thin and thicker than the 31 voxels pass 0
    m[:, :, 0] = 1                                      # finishes itself (deeper ones are the second pass's), at both ends and in the middle
    m[:, :min(3, ny), :] = 1
    m[:, max(ny - 5, 0):, :] = 1
    if ny >= 100:
        m[:, ny // 3:ny // 3 + 70, :] = 1
        m[nx // 2, ny // 3 + 30:ny // 3 + 33, nz // 2:] = 0     # ... with a pocket inside: filled voxels that are NOT saturated next to it
    out["walls_thin_and_thick"] = m
    m = np.zeros(shape, np.uint8)                       # a floor with one voxel missing under one line: "every row holds a filled voxel" fails there
    m[:, :, 0] = 1
    m[nx // 2, ny // 2, 0] = 0
    m[nx // 2 + 1, ny // 2, nz // 2] = 1
    out["floor_with_hole"] = m
    m = np.zeros(shape, np.uint8)                       # a floor under noise: nearly flat lines
    m[:, :, 0] = 1
    m |= synth.bernoulli_mask(shape, 0.0005, 9)
    out["floor_under_noise"] = m
    m = np.ones(shape, np.uint8)                        # everything filled but a few voxels: all-zero lines and the reverse class
    for _ in range(5):
        m[rng.integers(0, nx), rng.integers(0, ny), rng.integers(0, nz)] = 0
    out["nearly_solid"] = m
    return out


@pytest.mark.parametrize("shape", [(32, 64, 64), (24, 61, 64), (16, 512, 64), (12, 515, 64), (8, 1024, 64), (10, 777, 64), (9, 8, 64), (20, 130, 128),
                                   (512, 16, 64), (515, 9, 64), (1024, 8, 64), (130, 20, 64)],
                         ids=lambda s: "x".join(map(str, s)))
def test_far_field_pair_with_and_without_flat_tiles(gpu, shape):
    import torch
    res = 0.01
    s = torch.cuda.current_stream().cuda_stream
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    try:
        for name, m in _scenes(shape).items():
            for vb in (False, True):
                ex, ex_ext, _ = O.exact_sdf(m, res, vb)
                mt = torch.from_numpy(m).cuda()
                fields = {}
                for flat in (2, 1, 0):            # 2: every candidate tile tries; 1: as long as the device-side habit says it pays
                    gpu.set_option("policy_reset", 1)
                    gpu.set_option("dense", 0)
                    gpu.set_option("far_predict", 2)
                    gpu.set_option("flat_tiles", flat)
                    gpu.build_device(mt.data_ptr(), shape, out.data_ptr(), res, vb, s)
                    ext = gpu.get_extrema()
                    got = out.cpu().numpy()
                    assert gpu.last_build_info()["far_predicted"]
                    assert np.array_equal(got, ex) and ext == ex_ext, (name, vb, flat, int((got != ex).sum()))
                    fields[flat] = gpu.debug_yzsweep(shape).copy()
                assert np.array_equal(fields[2], fields[0]) and np.array_equal(fields[1], fields[0]), (name, vb, "plane field")
    finally:
        gpu.set_option("dense", 1)
        gpu.set_option("far_predict", 1)
        gpu.set_option("flat_tiles", 1)


def test_device_side_habit_follows_the_scene(gpu):
    """A floor under noise: no tile qualifies, the votes close the gate within a build or two (every 64th tile keeps trying); the room
    after it: every tile qualifies and the gate is open again a build or two later.  Every build along the way is exact."""
    import torch
    shape, res = (64, 256, 128), 0.01
    s = torch.cuda.current_stream().cuda_stream
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    noisy = synth.bernoulli_mask(shape, 0.002, 4)
    noisy[:, :, 0] = 1
    room = synth.room_mask_torch(shape, "cpu").numpy()
    try:
        gpu.set_option("policy_reset", 1)
        gpu.set_option("dense", 0)
        gpu.set_option("far_predict", 2)
        gpu.set_option("flat_tiles", 1)
        seen = []
        for m in (noisy, room, noisy):
            ex, ex_ext, _ = O.exact_sdf(m, res, False)
            mt = torch.from_numpy(m).cuda()
            for _ in range(4):
                gpu.build_device(mt.data_ptr(), shape, out.data_ptr(), res, False, s)
                assert np.array_equal(out.cpu().numpy(), ex) and gpu.get_extrema() == ex_ext
            gpu.build_device(mt.data_ptr(), shape, out.data_ptr(), res, False, s)
            seen.append(gpu.debug_flat_habit())
        assert seen[0][1] == 0 and seen[1][1] == 1 and seen[2][1] == 0, seen
    finally:
        gpu.set_option("dense", 1)
        gpu.set_option("far_predict", 1)
