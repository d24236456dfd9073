"""GPU: far-field scenes.  The marching kernels stop scanning after kMaxScan rows, raise a flag, and the
lower-envelope kernels (KE2 / KE3) redo the sweep exactly.  Results must equal the exact oracle and the
unbounded-scan path bit for bit."""
import numpy as np
import pytest

import scenes
from oracle import oracle as O
from sdf_tools_amd import synth

pytestmark = pytest.mark.gpu


def _two_boxes(shape):
    nx, ny, nz = shape
    m = np.zeros(shape, np.uint8)
    m[nx // 10: nx // 10 + max(2, nx // 8), ny // 2: ny // 2 + max(2, ny // 6), : max(2, nz // 3)] = 1
    m[nx // 2: nx // 2 + max(2, nx // 5), ny // 8: ny // 8 + max(2, ny // 5), nz // 4: nz // 4 + max(2, nz // 4)] = 1
    return m


CASES = [
    ("two boxes 64^3", _two_boxes((64, 64, 64)), False),
    ("two boxes 96x80x128 vb", _two_boxes((96, 80, 128)), True),
    ("single voxel 40x48x512", scenes.single_voxel((40, 48, 512), (3, 40, 500)), False),
    ("inverse single voxel", 1 - scenes.single_voxel((33, 20, 64)), False),
    ("sparse cloud", synth.bernoulli_mask((80, 72, 64), 0.0003, 3), False),
    ("sparse + dense half", np.concatenate([synth.bernoulli_mask((40, 48, 64), 0.5, 1), np.zeros((40, 48, 64), np.uint8)]), False),
    ("thin wall far away", np.pad(np.ones((1, 50, 32), np.uint8), ((60, 0), (0, 0), (0, 0))), True),
    ("2-D 300x400", synth.bernoulli_mask((1, 300, 400), 0.0002, 5), False),
    ("all free", np.zeros((20, 24, 32), np.uint8), False),
    # shapes the first two generations of the far-field kernel refused (line groups that are not multiples of 16, rows
    # that are not multiples of 4 elements): partial tiles, scalar loads, the int32 plane field between the two sweeps
    ("two boxes 40x30x20 (partial tiles)", _two_boxes((40, 30, 20)), False),
    ("two boxes 70x66x81 vb (odd rows)", _two_boxes((70, 66, 81)), True),
    ("single voxel 25x20x15 (3d_sdf_demo_rviz.py:107 shape)", scenes.single_voxel((25, 20, 15), (2, 3, 13)), False),
    ("convex-segments scene 100x100x50, dense tier off", scenes.convex_segments_scene()[0], False),
    ("two boxes 1x100x50", _two_boxes((1, 100, 50)), True),
    ("single voxel 3x9x1100 (1100-voxel lines)", scenes.single_voxel((3, 9, 1100), (1, 2, 1000)), False),
    # bench.py's structured leg at test size: floor, walls, table, shelf -- and the reference tutorial's solid boxes and their shells
    ("room 160x128x96", synth.room_mask_torch((160, 128, 96), "cpu").numpy(), False),
    ("room 160x128x96 vb", synth.room_mask_torch((160, 128, 96), "cpu").numpy(), True),
    ("tutorial boxes 128^3, solid", synth.tutorial_boxes_mask_torch((128, 128, 128), "cpu", True).numpy(), False),
    ("tutorial boxes 128^3, shells", synth.tutorial_boxes_mask_torch((128, 128, 128), "cpu", False).numpy(), False),
    # 512-voxel lines in both swept axes: the far-field kernel's instances with the line geometry as compile-time constants (round 5)
    ("two boxes 512x512x16", _two_boxes((512, 512, 16)), False),
    ("room 512x512x32 vb", synth.room_mask_torch((512, 512, 32), "cpu").numpy(), True),
]


@pytest.mark.parametrize("name,m,vb", CASES, ids=[c[0] for c in CASES])
def test_far_field_scenes_are_exact(gpu, name, m, vb):
    res = 0.02
    gpu.set_option("policy_reset", 1)
    sdf, ext = gpu.build(m, res, vb)
    path = gpu.last_path()
    ex, ex_ext, dsq = O.exact_sdf(m, res, vb)
    bad = np.argwhere(sdf.view(np.uint32) != ex.view(np.uint32))
    assert len(bad) == 0, "%s: %d voxels differ, first at %s got %r want %r (path %s)" % (
        name, len(bad), bad[0].tolist(), sdf[tuple(bad[0])], ex[tuple(bad[0])], path)
    assert ext == ex_ext, (name, ext, ex_ext, path)
    if name.startswith("convex-segments"):
        gpu.set_option("dense", 0)                          # (the dense tier certifies this scene; here the general tiers must)
        try:
            sdf0, ext0 = gpu.build(m, res, vb)
        finally:
            gpu.set_option("dense", 1)
        assert np.array_equal(sdf0.view(np.uint32), ex.view(np.uint32)) and ext0 == ex_ext
    if name.startswith(("two boxes 64", "two boxes 96", "two boxes 70", "single voxel 40x48x512", "thin wall")):
        assert path["far_y"] or path["far_x"]              # distances beyond the marching kernels' scan bound: the far-field kernel ran
    gpu.set_option("envelope_mode", 1)                      # every scene also with the far-field kernel as the only sweep of each axis
    try:
        sdf3, ext3 = gpu.build(m, res, vb)
    finally:
        gpu.set_option("envelope_mode", 0)
    assert np.array_equal(sdf.view(np.uint32), sdf3.view(np.uint32)) and ext == ext3, name
    gpu.set_option("envelope", 0)
    try:
        sdf2, ext2 = gpu.build(m, res, vb)                  # unbounded scans
        p2 = gpu.last_path()
    finally:
        gpu.set_option("envelope", 1)
    assert not p2["far_y"] and not p2["far_x"]
    assert np.array_equal(sdf.view(np.uint32), sdf2.view(np.uint32)) and ext == ext2


def test_point_cloud_scene_512(gpu):
    """The streaming configuration's scene: 200 k points in two boxes at 512^3."""
    import torch
    from sdf_tools_amd.streaming import StreamingSdf
    n, res = 512, 0.01
    st = StreamingSdf((n, n, n), res, (0.0, 0.0, 0.0), 0, gradient=False)
    pc = synth.two_box_points(200000, seed=0, scale=n * res)
    sdf, _ = st.frame(torch.from_numpy(pc).cuda())
    torch.cuda.synchronize()
    mask = st.mask.cpu().numpy()
    ex, ex_ext, _ = O.exact_sdf(mask, res)
    assert np.array_equal(sdf.cpu().numpy(), ex) and st.extrema() == ex_ext
    path = st.ctx.last_path()
    assert not path["dense_certified"] and (path["far_y"] or path["far_x"])
    for _ in range(3):                                      # the handle is now in envelope mode: still exact
        sdf, _ = st.frame(torch.from_numpy(pc).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(sdf.cpu().numpy(), ex) and st.extrema() == ex_ext


@pytest.mark.parametrize("shape", [(48, 40, 64), (6, 28, 512), (3, 20, 1024)])
def test_policy_transitions_stay_exact(gpu, shape):
    """The handle adapts its scan bounds / kernel list to the previous build (after a dense-certified build
    the guarded general pipeline is K12 + K3 with unbounded scans and no envelope kernels); every
    combination of previous and current scene type must still be exact."""
    import torch
    scenes_ = {
        "dense": synth.bernoulli_mask(shape, 0.5, 1),
        "far": _two_boxes(shape),
        "mid": synth.bernoulli_mask(shape, 0.01, 2),
        "empty": np.zeros(shape, np.uint8),
    }
    want = {k: O.exact_sdf(m, 0.05) for k, m in scenes_.items()}
    order = ["dense", "dense", "far", "far", "dense", "mid", "far", "mid", "mid", "empty", "dense", "far", "dense", "dense"]
    for name in order:
        sdf, ext = gpu.build(scenes_[name], 0.05)
        torch.cuda.synchronize()
        assert np.array_equal(sdf.view(np.uint32), want[name][0].view(np.uint32)), name
        assert ext == want[name][1], name


@pytest.mark.parametrize("shape", [(40, 33, 50), (25, 20, 15), (17, 64, 96), (9, 7, 130), (64, 64, 64), (5, 300, 33), (20, 40, 1),
                                   (1, 1, 100), (3, 1, 257)])
def test_standby_pair_takes_over_exactly_on_every_shape(gpu, shape):
    """Round 4: behind a TRUSTED dense tier the general pipeline is the stand-by pair -- the far-field y sweep staged from the
    dense tier's BIT FIELD (ragged last word, rows padded to whole words on the generic dense shapes, partial 16-line tiles,
    quad-cooperative row scan) and the far-field x sweep, both in the LOOP form.  Put the handle into that state (option
    `expect_dense`), then build scenes the dense tier cannot certify -- boxes, a sparse cloud, a single voxel, an empty grid,
    an almost-full grid -- with and without the virtual border and through the COLLISION_CELL input: exact, extrema included,
    and the status says the far-field kernels did it."""
    rng = np.random.default_rng(7)
    nx, ny, nz = shape
    scenes_ = {"boxes": _two_boxes(shape), "sparse": synth.bernoulli_mask(shape, 0.004, 3), "empty": np.zeros(shape, np.uint8),
               "single": np.zeros(shape, np.uint8), "almost full": (1 - synth.bernoulli_mask(shape, 0.003, 5)).astype(np.uint8)}
    scenes_["single"][nx // 2, ny - 1, nz - 1] = 1
    dense = (np.indices(shape).sum(axis=0) & 1).astype(np.uint8)      # a checkerboard: certified on any shape, 1-D lines included
    want_dense = O.exact_sdf(dense, 0.05)
    took_over = 0
    for name, m in scenes_.items():
        for vb in (False, True):
            gpu.set_option("policy_reset", 1)
            sdf, ext = gpu.build(dense, 0.05)                        # a certified dense build ...
            assert gpu.last_path()["dense_certified"], (name, shape)
            assert np.array_equal(sdf.view(np.uint32), want_dense[0].view(np.uint32))
            gpu.set_option("expect_dense", 1)                        # ... after which the handle trusts its dense tier
            want, want_ext, _ = O.exact_sdf(m, 0.05, vb)
            if name == "sparse" and not vb:
                cells = np.zeros(shape + (2,), np.float32)
                cells[..., 0] = m
                sdf, ext = gpu.build_cells(cells, shape, 8, 0, False, 0.05, vb)
            else:
                sdf, ext = gpu.build(m, 0.05, vb)
            info, path = gpu.last_build_info(), gpu.last_path()
            assert info["standby_far"] and not info["fused_zy"], (name, shape, vb, info)
            bad = np.argwhere(sdf.view(np.uint32) != want.view(np.uint32))
            assert len(bad) == 0, (name, shape, vb, len(bad), bad[:3].tolist())
            assert ext == want_ext, (name, shape, vb, ext, want_ext)
            # (a thin grid with the virtual border is dense whatever it holds: every voxel lies within 2 of the padded layer)
            assert path["dense_certified"] or (path["far_y"] and path["far_x"]), (name, shape, vb, path)
            took_over += not path["dense_certified"]
    assert took_over >= 5, (shape, took_over)
    gpu.set_option("policy_reset", 1)


def test_envelope_rare_paths_deep_pops_and_dense_advances(gpu):
    """The hot loops of the envelope kernels read only LDS / registers; these scenes force their out-of-line paths:
    a site that pops far more stack entries than the 16-entry LDS ring holds, and runs of positions that each
    advance to a new parabola (more than the register window holds per batch)."""
    gpu.set_option("policy_reset", 1)
    gpu.set_option("dense", 0)
    gpu.set_option("envelope_mode", 1)                      # the envelope kernel is the only sweep of each axis
    try:
        scenes_ = []
        # (a) 40 rows whose only filled voxel is far away in z, then a row with a filled voxel at z = 0: on the lines
        #     near z = 0 the last site dominates (pops) all 40 earlier parabolas at once; mirrored along x as well
        m = np.zeros((48, 64, 64), np.uint8)
        m[:, :40, 63] = 1
        m[:, 40, 0] = 1
        scenes_.append(m)
        m = np.zeros((64, 12, 64), np.uint8)
        m[:40, :, 63] = 1
        m[40, :, 0] = 1
        scenes_.append(m)
        # (b) a staircase: every position along the line gets its own parabola (advance at every step)
        m = np.zeros((40, 40, 64), np.uint8)
        for i in range(40):
            m[i, :, min(63, 8 + i)] = 1
            m[:, i, min(63, 20 + (i * 7) % 40)] = 1
        scenes_.append(m)
        # (c) random smooth height field + its complement
        rng = np.random.default_rng(12)
        hgt = (rng.random((36, 44)) * 30 + 10).astype(int)
        m = (np.arange(64)[None, None, :] >= hgt[:, :, None]).astype(np.uint8)
        scenes_.append(m)
        scenes_.append(1 - m)
        for k, m in enumerate(scenes_):
            for vb in (False, True):
                sdf, ext = gpu.build(m, 0.05, vb)
                ex, ex_ext, _ = O.exact_sdf(m, 0.05, vb)
                bad = np.argwhere(sdf.view(np.uint32) != ex.view(np.uint32))
                assert len(bad) == 0, "scene %d vb=%s: %d voxels differ, first %s" % (k, vb, len(bad), bad[0].tolist())
                assert ext == ex_ext
    finally:
        gpu.set_option("envelope_mode", 0)
        gpu.set_option("dense", 1)


DC_SHAPES = [(40, 100, 48), (1024, 4, 16), (7, 1024, 16), (512, 8, 32), (13, 9, 16), (2, 3, 16), (1, 1, 64), (24, 520, 32),
             (65, 63, 80), (16, 16, 20), (520, 6, 16), (1000, 3, 32), (3, 700, 16)]


@pytest.mark.parametrize("shape", DC_SHAPES, ids=["x".join(map(str, s)) for s in DC_SHAPES])
def test_divide_and_conquer_envelope_kernel_is_exact(gpu, shape):
    """k_envelope_dc (sdfgpu_envelope_dc.hpp) as the ONLY sweep of the y and x axes: line lengths that are not multiples
    of 8, of length 1, 512 and 1024 (keys at their 32-bit limit), lines without sites, both classes, virtual border;
    and the same scenes through the first-generation kernel (option envelope_dc = 0; (16,16,20) always takes it:
    nz % 16 != 0).  Everything must equal the exact oracle bit for bit."""
    gpu.set_option("policy_reset", 1)
    gpu.set_option("dense", 0)
    gpu.set_option("envelope_mode", 1)
    rng = np.random.default_rng(abs(hash(shape)) % (1 << 31))
    nx, ny, nz = shape
    scenes_ = {
        "sparse": synth.bernoulli_mask(shape, 0.002, 3),
        "boxes": _two_boxes(shape),
        "single": scenes.single_voxel(shape),
        "inverse single": 1 - scenes.single_voxel(shape, (nx - 1, 0, nz // 3)),
        "all free": np.zeros(shape, np.uint8),
        "all filled": np.ones(shape, np.uint8),
        "half dense": np.concatenate([synth.bernoulli_mask((nx, ny, nz - nz // 2), 0.4, 9), np.zeros((nx, ny, nz // 2), np.uint8)], axis=2),
    }
    # noisy sheets near both ends of the x and the y lines, nothing in between: the argmin of every long line jumps from one
    # sheet to the other -- the ranges that levels B and C spread over their wave (round 4: scan8_calm / coop8, with one
    # interval per lane for lines above 512 and two lanes per interval below)
    sheets = np.zeros(shape, np.uint8)
    for ax, n_ax in ((0, nx), (1, ny)):
        if n_ax < 64 and max(nx, ny) >= 64:                     # (sheets across the LONG lines only, or they fill each other's gaps)
            continue
        for pos in {min(5, n_ax - 1), max(n_ax - 6, 0), n_ax // 3}:
            sl = [slice(None)] * 3
            sl[ax] = pos
            sheets[tuple(sl)] |= (rng.random(sheets[tuple(sl)].shape) < 0.08).astype(np.uint8)
    # walls and a floor a few voxels thick (their voxels have NO free voxel in their own row / plane: pass 0 of the kernel finishes them
    # with its local search along the line -- up to 15 positions -- instead of sending the tile to the second pass), and a slab too thick for it
    walls = np.zeros(shape, np.uint8)
    walls[:min(10, max(nx // 3, 1))] = 1
    walls[:, :min(7, max(ny // 3, 1))] = 1
    walls[:, :, :3] = 1
    walls[nx // 2:nx // 2 + 40, ny // 2:, nz // 2:] = 1
    scenes_["walls"] = walls
    scenes_["sheets"] = sheets
    scenes_["inverse sheets"] = 1 - sheets
    hgt = (rng.random((nx, ny)) * nz * 0.8).astype(int)
    scenes_["height field"] = (np.arange(nz)[None, None, :] >= hgt[:, :, None]).astype(np.uint8)
    try:
        for name, m in scenes_.items():
            for vb in (False, True):
                ex, ex_ext, _ = O.exact_sdf(m, 0.05, vb)
                # (dc, hand-off): the y sweep hands the x sweep an int32 plane field (default) or p16 + side table
                for dc, ho in ((1, 1), (1, 0), (0, 1)):
                    gpu.set_option("envelope_dc", dc)
                    gpu.set_option("i32_handoff", ho)
                    gpu.set_option("envelope_mode", 1)
                    sdf, ext = gpu.build(m, 0.05, vb)
                    bad = np.argwhere(sdf.view(np.uint32) != ex.view(np.uint32))
                    assert len(bad) == 0, "%s vb=%s dc=%d handoff=%d: %d voxels differ, first %s got %r want %r" % (
                        name, vb, dc, ho, len(bad), bad[0].tolist(), sdf[tuple(bad[0])], ex[tuple(bad[0])])
                    assert ext == ex_ext, (name, vb, dc, ho, ext, ex_ext)
    finally:
        gpu.set_option("envelope_dc", 1)
        gpu.set_option("i32_handoff", 1)
        gpu.set_option("envelope_mode", 0)
        gpu.set_option("dense", 1)


def test_far_field_prediction_skips_the_probes_and_stays_exact(gpu):
    """Round 5: a handle whose recent builds were far-field on both axes enqueues KE2 -> KE3 directly (no probes, no guarded
    marching launches; sdfgpu_last_build_info bit 6), probes again every 16th build, and drops the habit when the scene
    changes.  Every build -- predicted or not, right or wrong -- equals the exact oracle bit for bit."""
    res = 0.02
    far = _two_boxes((96, 80, 64))
    far_want, far_ext, _ = O.exact_sdf(far, res, False)
    near = synth.bernoulli_mask((96, 80, 64), 0.3, 7)
    near_want, near_ext, _ = O.exact_sdf(near, res, False)
    gpu.set_option("dense", 0)                                   # (the general tiers alone: the dense tier would certify `near`)
    predicted = []
    try:
        for i in range(40):
            sdf, ext = gpu.build(far, res, False)
            assert np.array_equal(sdf.view(np.uint32), far_want.view(np.uint32)) and ext == far_ext, i
            predicted.append(gpu.last_build_info()["far_predicted"])
            path = gpu.last_path()
            assert path["far_y"] and path["far_x"], (i, path)
        assert not any(predicted[:4]) and sum(predicted) >= 24, predicted        # learnt after a few reports ...
        assert not all(predicted[8:]), predicted                                   # ... and re-probed every 16th build
        # the scene turns near-field under the handle: at most a few builds still take the far-field pair (exactly), then the probes are back
        predicted = []
        for i in range(40):
            sdf, ext = gpu.build(near, res, False)
            assert np.array_equal(sdf.view(np.uint32), near_want.view(np.uint32)) and ext == near_ext, i
            predicted.append(gpu.last_build_info()["far_predicted"])
        assert not any(predicted[20:]), predicted
        path = gpu.last_path()
        assert not path["far_y"] and not path["far_x"], path
        # forced: every build, whatever the scene
        gpu.set_option("far_predict", 2)
        for m, want, want_ext in ((near, near_want, near_ext), (far, far_want, far_ext)):
            for vb in (False, True):
                w, we = (want, want_ext) if not vb else O.exact_sdf(m, res, True)[:2]
                sdf, ext = gpu.build(m, res, vb)
                assert gpu.last_build_info()["far_predicted"]
                assert np.array_equal(sdf.view(np.uint32), w.view(np.uint32)) and ext == we, vb
        gpu.set_option("far_predict", 0)
        for i in range(8):
            sdf, ext = gpu.build(far, res, False)
            assert not gpu.last_build_info()["far_predicted"]
        assert np.array_equal(sdf.view(np.uint32), far_want.view(np.uint32))
    finally:
        gpu.set_option("far_predict", 1)
        gpu.set_option("dense", 1)


@pytest.mark.parametrize("nx", [9, 13, 21, 3])
def test_virtual_border_x_sweep_writes_nothing_past_the_field(gpu, nx):
    """Round 5 regression (found by the fuzz at 9 x 777 x 64): the far-field x sweep's finish applied the virtual border to the
    positions of a line's last chunk that lie PAST the line's end; from the second such position on the border distance is
    negative, its square as a 24-bit product too, the position stopped being "not mine" and its value was stored x planes
    beyond the end of the field.  Device-resident output with a guard band of 8 x planes behind it: the field equals the exact
    oracle and the band is untouched -- far-field pair forced (probe or not), both hand-off forms."""
    import torch
    shape = (nx, 40, 32)
    m = np.zeros(shape, np.uint8)
    m[nx // 2, 3:6, 4:9] = 1
    m[0, 30, 20] = 1
    res = 0.05
    want, want_ext, _ = O.exact_sdf(m, res, True)
    n, plane = int(np.prod(shape)), shape[1] * shape[2]
    dm = torch.from_numpy(m).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    try:
        for opts in ({"envelope_mode": 1}, {"far_predict": 2}, {"far_predict": 2, "plane16": 0}, {"envelope_mode": 1, "i32_handoff": 0}):
            gpu.set_option("policy_reset", 1)
            gpu.set_option("dense", 0)
            for k, v in opts.items():
                gpu.set_option(k, v)
            out = torch.full((n + 8 * plane,), -12345.0, dtype=torch.float32, device="cuda")
            gpu.build_device(dm.data_ptr(), shape, out.data_ptr(), res, True, stream)
            torch.cuda.synchronize()
            path = gpu.last_path()
            assert path["far_x"], (opts, path)
            got = out[:n].cpu().numpy().reshape(shape)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), opts
            assert bool((out[n:] == -12345.0).all().item()), (opts, int((out[n:] != -12345.0).sum().item()))
            assert gpu.get_extrema() == want_ext, opts
            for k in opts:
                gpu.set_option(k, {"envelope_mode": 0, "far_predict": 1, "plane16": 1, "i32_handoff": 1}[k])
    finally:
        for k, v in (("envelope_mode", 0), ("far_predict", 1), ("plane16", 1), ("i32_handoff", 1), ("dense", 1)):
            gpu.set_option(k, v)
