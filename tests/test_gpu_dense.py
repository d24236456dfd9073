"""GPU: the dense path (K0 pack + KD bit-parallel ball kernel) in front of the general pipeline.
Whatever the scene, the result must be exact; on dense scenes the dense kernel alone must have
decided every voxel (the general kernels exit on their guard flag)."""
import math

import numpy as np
import pytest

import scenes
from oracle import oracle as O
from sdf_tools_amd import synth

pytestmark = pytest.mark.gpu


def _check(gpu, m, res, expect_certified=None):
    sdf, ext = gpu.build(m, res)
    info = gpu.last_build_info()
    cert = gpu.last_dense_certified()
    ex, ex_ext, _ = O.exact_sdf(m, res)
    bad = np.argwhere(sdf.view(np.uint32) != ex.view(np.uint32))
    assert len(bad) == 0, "%d voxels differ, first at %s: got %r want %r (certified=%s)" % (
        len(bad), bad[0].tolist(), sdf[tuple(bad[0])], ex[tuple(bad[0])], cert)
    assert ext == ex_ext
    assert np.array_equal(np.signbit(sdf), m != 0)
    if expect_certified is not None:
        assert cert == expect_certified
    return info, cert


DENSE = [(64, 64, 64), (16, 16, 32), (5, 7, 32), (3, 13, 64), (1, 40, 64), (1, 1, 32), (9, 4, 128),
         (4, 6, 512), (3, 5, 1024), (2, 3, 2048), (33, 21, 96)]


@pytest.mark.parametrize("shape", DENSE)
def test_dense_random_occupancy_is_certified_and_exact(gpu, shape):
    m = synth.bernoulli_mask(shape, 0.5, 31)
    dims = [s for s in shape if s > 1] or [1]
    nz = dims[-1]
    nzw = nz // 32
    tuned = nz % 32 == 0 and (nzw & (nzw - 1)) == 0 and nzw <= 64
    info, cert = _check(gpu, m, 0.01)
    assert info["dense"]                               # tuned kernels, or their generic forms for the other shapes
    if m.size >= 32 * 32:
        assert cert                                    # D <= 8 everywhere at p = 0.5
    gpu.set_option("dense_generic", 0)                 # without the generic forms only the tuned shapes are dense
    try:
        info2, _ = _check(gpu, m, 0.01)
        assert info2["dense"] == tuned
    finally:
        gpu.set_option("dense_generic", 1)
    ref, ref_ext = O.reference_sdf(m, 0.01)
    sdf, ext = gpu.build(m, 0.01)
    assert np.array_equal(sdf, ref) and ext == ref_ext     # the reference algorithm, bit for bit


@pytest.mark.parametrize("p", [0.2, 0.05, 0.001, 0.9, 0.999])
def test_sparser_scenes_fall_back_and_stay_exact(gpu, p):
    m = synth.bernoulli_mask((40, 36, 64), p, 7)
    info, cert = _check(gpu, m, 0.5)
    assert info["dense"]
    if p in (0.001, 0.999):
        assert not cert                                # far voxels exist: the general pipeline did the work


def test_every_level_of_the_ball(gpu):
    """One filled voxel: the free voxels around it realise every d^2 of the ball (1,2,3,4,5,6,8) and
    everything beyond (9, 10, ...), so both the level tables and the fallback hand-over are exercised."""
    for shape in ((9, 9, 32), (7, 7, 64)):
        m = scenes.single_voxel(shape)
        _check(gpu, m, 1.0, expect_certified=False)
        _check(gpu, 1 - m, 1.0, expect_certified=False)
    # lattices of sites keep every voxel inside the ball: every 2nd voxel -> d^2 <= 3
    m = np.zeros((8, 8, 32), np.uint8)
    m[::2, ::2, ::2] = 1
    _check(gpu, m, 1.0, expect_certified=True)
    m = np.zeros((10, 10, 32), np.uint8)
    m[::3, ::3, ::3] = 1                                # sites at 0,3,6,9(,..30): at most 1 away per axis
    _check(gpu, m, 0.25, expect_certified=True)
    m = np.zeros((9, 9, 32), np.uint8)
    m[::4, ::4, ::4] = 1                                # (2,2,2) away -> d^2 = 12 > 8: must fall back
    gpu.set_option("policy_reset", 1)
    gpu.set_option("dense3", 0)                         # (KD alone: with the wide ball kernel behind it the scene is certified)
    try:
        _check(gpu, m, 0.25, expect_certified=False)
    finally:
        gpu.set_option("dense3", 1)
    gpu.set_option("policy_reset", 1)
    _check(gpu, m, 0.25, expect_certified=True)         # ... by KD3, staged behind KD in the same build


def test_grid_edges_replicate_correctly(gpu):
    """Voxels on faces / edges / corners: out-of-grid neighbours must never count as hits."""
    shape = (6, 6, 64)
    for m in (np.zeros(shape, np.uint8), np.ones(shape, np.uint8)):
        sdf, ext = gpu.build(m, 1.0)
        assert np.all(np.isinf(sdf)) and not gpu.last_dense_certified()
    m = np.zeros(shape, np.uint8)
    m[0, 0, 0] = 1
    m[5, 5, 63] = 1
    m[0, 5, 31] = 1
    m[0, 5, 32] = 1                                     # across a word boundary
    _check(gpu, m, 1.0)
    m = synth.bernoulli_mask(shape, 0.5, 3)
    m[:, :, 0] = 1
    m[:, :, 63] = 0
    m[0] = 1
    _check(gpu, m, 1.0)


def test_collision_cells_take_the_dense_path(gpu):
    shape = (12, 10, 64)
    rng = np.random.RandomState(5)
    occ = rng.choice(np.array([0.0, 0.5, 1.0], np.float32), size=shape)
    cells = np.zeros(shape + (2,), np.float32)
    cells[..., 0] = occ
    for unknown in (False, True):
        mask = O.classify_cells(cells, unknown)
        got, ext = gpu.build_cells(cells, shape, 8, 0, unknown, 0.1)
        assert gpu.last_build_info()["dense"]
        want, want_ext = O.reference_sdf(mask, 0.1)
        assert np.array_equal(got, want) and ext == want_ext


def test_virtual_border_folds_into_the_dense_tier(gpu):
    """add_virtual_border (sdf_generation.hpp:287-419) folds into the ball: b^2 = 1 and 4 are levels 0 and 3.  Shapes
    with nz = 32 * 2^k take the tuned kernel (thin grids: an axis of one cell has no border), the others the generic one."""
    for shape in ((8, 8, 32), (64, 64, 64), (2, 5, 64), (1, 9, 32), (3, 1, 128), (40, 33, 256), (20, 17, 45), (1, 30, 70)):
        m = synth.bernoulli_mask(shape, 0.5, 1)
        sdf, ext = gpu.build(m, 1.0, True)
        assert gpu.last_build_info()["dense"] and gpu.last_dense_certified()
        ex, ex_ext, _ = O.exact_sdf(m, 1.0, True)
        assert np.array_equal(sdf, ex) and ext == ex_ext
        ref, ref_ext = O.reference_sdf(m, 1.0, True)       # the reference's pad-twice-and-combine, bit for bit
        assert np.array_equal(sdf, ref) and ext == ref_ext
    # almost dense + virtual border, repeated builds (the fix-up kernel stays out of virtual-border builds), then sparse
    for p in (0.1, 0.03):
        m = synth.bernoulli_mask((48, 40, 64), p, 7)
        ex, ex_ext, _ = O.exact_sdf(m, 0.5, True)
        for _ in range(3):
            sdf, ext = gpu.build(m, 0.5, True)
            assert np.array_equal(sdf, ex) and ext == ex_ext, p
    # one class only + virtual border: voxels deeper than 2 layers are not decided by the ball -> general pipeline
    for fill in (0, 1):
        m = np.full((12, 10, 40), fill, np.uint8)
        sdf, ext = gpu.build(m, 0.5, True)
        ex, ex_ext, _ = O.exact_sdf(m, 0.5, True)
        assert np.array_equal(sdf, ex) and ext == ex_ext and not gpu.last_dense_certified()


REF_SHAPES = [(100, 100, 50), (40, 40, 40), (25, 20, 15), (20, 40, 1), (7, 9, 33), (5, 5, 95), (3, 70, 31)]


@pytest.mark.parametrize("shape", REF_SHAPES, ids=["x".join(map(str, s)) for s in REF_SHAPES])
def test_reference_grid_shapes_take_the_dense_tier(gpu, shape):
    """the reference's own grid shapes (100x100x50 compute_convex_segments_test.cpp:13-41, 40^3
    sdf_tools_tutorial.cpp:23-59, 25x20x15 3d_sdf_demo_rviz.py:107-111, 20x40x1 test_bindings.py:11-20) are dense-tier
    shapes now: certified where every voxel has D <= 8, exact (general pipeline behind the guard) otherwise."""
    for p, vb in ((0.5, False), (0.5, True), (0.2, False), (0.02, True)):
        m = synth.bernoulli_mask(shape, p, 11)
        sdf, ext = gpu.build(m, 0.05, vb)
        assert gpu.last_build_info()["dense"]
        ex, ex_ext, dsq = O.exact_sdf(m, 0.05, vb)
        bad = np.argwhere(sdf.view(np.uint32) != ex.view(np.uint32))
        assert len(bad) == 0, (shape, p, vb, len(bad), bad[:3].tolist())
        assert ext == ex_ext
        assert gpu.last_dense_certified() == bool(np.abs(dsq).max() <= 8)
    # the reference's scenes themselves (walls 10 cells thick / a half-filled cube: D > 8, so the sweeps finish them)
    if shape == (100, 100, 50):
        m, res = scenes.convex_segments_scene()
    elif shape == (40, 40, 40):
        m, res = scenes.tutorial_scene()
    else:
        return
    sdf, ext = gpu.build(m, res)
    ref, ref_ext = O.reference_sdf(m, res)
    ex, ex_ext, _ = O.exact_sdf(m, res)
    assert np.array_equal(sdf, ex) and ext == ex_ext and not gpu.last_dense_certified()


def test_dense_retry_policy_skips_and_retries(gpu):
    """A dense attempt that does not expect to succeed carries the fix-up stage (the wide ball kernel KD3 + the fix-up
    kernel) behind KD in the same build; if that cannot certify the scene either, the dense kernels are left out of the
    next builds and tried again every dense_retry-th build.  Results stay exact all along."""
    gpu.set_option("dense_retry", 4)
    sparse = synth.bernoulli_mask((16, 16, 64), 0.002, 9)
    dense = synth.bernoulli_mask((16, 16, 64), 0.5, 9)
    ex_s, ext_s, _ = O.exact_sdf(sparse, 0.1)
    ex_d, ext_d, _ = O.exact_sdf(dense, 0.1)
    for _ in range(1):                                   # one attempt: KD and, staged behind it, KD3 + KF
        sdf, ext = gpu.build(sparse, 0.1)
        assert np.array_equal(sdf, ex_s) and ext == ext_s
        assert gpu.last_build_info()["dense"] and not gpu.last_dense_certified()
    used = []
    for _ in range(8):                                   # the scene turns dense: skipped 3 times, retried, then kept
        sdf, ext = gpu.build(dense, 0.1)
        assert np.array_equal(sdf, ex_d) and ext == ext_d
        used.append(gpu.last_build_info()["dense"])
    assert used == [False, False, False, True, True, True, True, True]
    assert gpu.last_dense_certified()
    # a scene that stays sparse: the pause doubles after every failed retry (3, 7, 15 builds), results stay exact
    gpu.set_option("policy_reset", 1)
    gpu.set_option("dense_retry", 4)
    used = []
    for _ in range(1 + 3 + 1 + 7 + 1 + 15 + 1):
        sdf, ext = gpu.build(sparse, 0.1)
        assert np.array_equal(sdf, ex_s) and ext == ext_s
        used.append(int(gpu.last_build_info()["dense"]))
    assert used == [1] + [0] * 3 + [1] + [0] * 7 + [1] + [0] * 15 + [1]
    gpu.set_option("dense_retry", 0)


@pytest.mark.parametrize("p", [0.1, 0.07, 0.9, 0.93])
def test_almost_dense_scenes_are_finished_by_the_fixup_kernel(gpu, p):
    """A handful of voxels beyond the d^2 <= 8 ball: the fix-up stage (staged behind KD in a first build, in KD's place
    afterwards) certifies the scene without the general sweeps -- exact every time."""
    shape = (48, 40, 64)
    m = synth.bernoulli_mask(shape, p, 21)
    ex, ex_ext, dsq = O.exact_sdf(m, 0.05)
    assert np.abs(dsq).max() > 8                           # the ball alone cannot decide this scene
    certified = []
    for _ in range(3):
        sdf, ext = gpu.build(m, 0.05)
        assert np.array_equal(sdf.view(np.uint32), ex.view(np.uint32)) and ext == ex_ext
        certified.append(gpu.last_dense_certified())
    assert certified == [True, True, True]                 # (the first build already: the fix-up stage is staged behind KD)
    # forced fix-up mode on a scene that needs distances beyond its cube: it must hand over, not guess
    gpu.set_option("policy_reset", 1)
    gpu.set_option("fixup_mode", 1)
    far = np.zeros(shape, np.uint8)
    far[3, 4, 5] = 1
    far[40, 30, 60] = 1
    sdf, ext = gpu.build(far, 1.0)
    want, want_ext, _ = O.exact_sdf(far, 1.0)
    assert np.array_equal(sdf, want) and ext == want_ext and not gpu.last_dense_certified()


def test_fixup_kernel_edges_and_every_shape(gpu):
    """Grid faces / word boundaries / odd tile shapes with the fix-up kernel forced on."""
    for shape in ((5, 7, 32), (9, 4, 128), (3, 5, 1024), (33, 21, 96), (1, 40, 64), (2, 3, 2048)):
        for p in (0.12, 0.9):
            m = synth.bernoulli_mask(shape, p, 5)
            m[0, 0, 0] = 1 - m[0, 0, 0]
            gpu.set_option("policy_reset", 1)
            gpu.set_option("fixup_mode", 1)
            sdf, ext = gpu.build(m, 0.3)
            ex, ex_ext, _ = O.exact_sdf(m, 0.3)
            assert np.array_equal(sdf.view(np.uint32), ex.view(np.uint32)), shape
            assert ext == ex_ext


def test_wide_ball_kernel_every_level_edges_and_shapes(gpu):
    """KD3 (|offset| <= 3, levels d^2 in {1..6, 8..14}) forced in KD's place: lattices that realise every level, a single
    voxel (hand-over beyond the ball), grid faces / word boundaries / odd tile shapes, both classes."""
    def run(m, res, expect_certified=None):
        gpu.set_option("policy_reset", 1)
        gpu.set_option("dense3_mode", 1)
        return _check(gpu, m, res, expect_certified)
    try:
        m = np.zeros((9, 9, 32), np.uint8)
        m[::4, ::4, ::4] = 1                                # (2,2,2) away -> d^2 = 12: inside the wide ball
        run(m, 0.25, True)
        run(1 - m, 0.25, True)
        m = np.zeros((12, 12, 64), np.uint8)
        m[::6, ::6, ::6] = 1                                # (3,3,3) away -> d^2 = 27 > 14 for a third of the voxels: too many for the
        run(m, 1.0)                                         # fix-up kernel alone (rounds 3 - 4: KD3 said so itself, kBall3MaxUndecided per
                                                            # wave); since round 5 the shell pass KD6 behind KD3 takes them
        gpu.set_option("dense_shell", 0)
        try:
            run(m, 1.0, False)
        finally:
            gpu.set_option("dense_shell", 1)
        m = synth.bernoulli_mask((24, 24, 128), 0.08, 11)
        m[8:14, 8:14, 40:46] = 0                            # a 6^3 cavity: d^2 up to 27 for a few voxels -> KF finishes them
        run(m, 1.0, True)
        m = np.zeros((11, 11, 64), np.uint8)
        m[::5, ::5, ::5] = 1                                # up to (2,2,2) = 12 inside; (2,2,3) = 17 at z = 63: a few voxels for KF
        run(m, 0.5, True)
        for shape in ((9, 9, 32), (7, 7, 64)):
            one = scenes.single_voxel(shape)                # every d^2 of the ball and everything beyond
            run(one, 1.0, False)
            run(1 - one, 1.0, False)
        for shape in ((5, 7, 32), (9, 4, 128), (3, 5, 1024), (33, 21, 96), (1, 40, 64), (2, 3, 2048), (16, 16, 512)):
            for p in (0.05, 0.03, 0.95):
                m = synth.bernoulli_mask(shape, p, 7)
                m[0, 0, 0] = 1 - m[0, 0, 0]
                run(m, 0.3)
    finally:
        gpu.set_option("policy_reset", 1)


def test_shell_pass_between_the_wide_ball_and_the_fixup_kernel(gpu):
    """Round 5, KD6 (sdfgpu_dense6.hpp): the bit-parallel shell pass 16 <= d^2 <= 36 over the words KD3 leaves undecided voxels in,
    with KF behind it for what lies beyond.  Forced fix-up stage (KD3 + KD6 + KF) on noise down to p = 0.01 and up to 0.99 (both
    classes), cavities that realise every level of the shell and the hand-over to KF (d^2 <= 64) and to the sweeps (beyond),
    grid faces / word boundaries / every tile shape; every voxel and the extrema bit for bit against the exact oracle, and the
    same bits with the pass switched off (option dense_shell = 0: KF takes everything, as in rounds 3 - 4)."""
    def run(m, res, expect_certified=None):
        out = []
        for shell in (1, 0):
            gpu.set_option("dense_shell", shell)
            gpu.set_option("policy_reset", 1)
            gpu.set_option("dense3_mode", 1)
            out.append(_check(gpu, m, res, expect_certified if shell else None))
        return out
    try:
        for shape, p, seed in (((64, 64, 128), 0.02, 1), ((48, 40, 256), 0.015, 2), ((64, 48, 128), 0.01, 3), ((40, 40, 64), 0.985, 4),
                               ((32, 32, 512), 0.012, 5), ((24, 24, 1024), 0.99, 6)):
            run(synth.bernoulli_mask(shape, p, seed), 0.05)
        # cavities in clutter: a c^3 hole has voxels up to d^2 = 3 ((c + 1) / 2)^2 from the clutter around it -- c = 6: 27 (shell),
        # 7: 48 (KF), 8: 48 .. 60 (KF), 10: 75 .. 90 (beyond KF: the sweeps); the levels in between come from the clutter's own noise
        for c, cert in ((5, True), (6, True), (7, None), (8, None), (10, None)):
            for cls in (0, 1):
                m = synth.bernoulli_mask((32, 32, 128), 0.1, 20 + c)
                m[10:10 + c, 12:12 + c, 50:50 + c] = 0
                run(m if cls == 0 else 1 - m, 1.0, cert)
        # every level of the shell at least once: one voxel pair per offset (dx, dy, dz) with 16 <= d^2 <= 36 in a clutter-free
        # box inside clutter would need 668 scenes; the same coverage cheaply: a solid slab with pits of every depth 4 .. 6 and
        # a lattice of single voxels whose Voronoi cells reach every d^2 up to 36
        m = synth.bernoulli_mask((40, 40, 128), 0.15, 31)
        m[4:36, 4:36, 20:60] = 0
        m[::5, ::5, 20:60:5] |= 1                            # a 5-lattice in the cleared box: offsets up to (2,2,2) .. (3,3,3)
        run(m, 0.5)
        m = synth.bernoulli_mask((36, 36, 128), 0.15, 32)
        m[2:34, 2:34, 10:100] = 0
        m[2:34:8, 2:34:8, 10:100:8] |= 1                     # an 8-lattice: offsets up to (4,4,4) = 48 (levels 16 .. 36 from KD6, beyond from KF)
        run(m, 0.5)
        run(1 - m, 0.5)
        for shape in ((5, 7, 32), (9, 4, 128), (3, 5, 1024), (33, 21, 96), (1, 40, 64), (2, 3, 2048), (16, 16, 512), (7, 9, 256)):
            for p in (0.02, 0.01, 0.98):
                m = synth.bernoulli_mask(shape, p, 9)
                m[0, 0, 0] = 1 - m[0, 0, 0]
                run(m, 0.3)
    finally:
        gpu.set_option("dense_shell", 1)
        gpu.set_option("policy_reset", 1)


def test_policy_escalates_to_the_wide_ball_kernel(gpu):
    """Bernoulli p = 0.03: KD leaves 6 % of the voxels undecided (more than KF takes), KD3 leaves 2e-3: the first build runs
    KD, then KD3 + KF on KD's verdict, the later ones KD3 + KF alone; certified and exact every time.  With the option off (KD + KF) it
    backs off to the sweeps instead."""
    shape = (64, 64, 128)
    m = synth.bernoulli_mask(shape, 0.03, 3)
    ex, ex_ext, _ = O.exact_sdf(m, 0.05)
    gpu.set_option("policy_reset", 1)
    certified = []
    for _ in range(5):
        sdf, ext = gpu.build(m, 0.05)
        assert np.array_equal(sdf.view(np.uint32), ex.view(np.uint32)) and ext == ex_ext
        certified.append(gpu.last_dense_certified())
    assert certified == [True, True, True, True, True], certified
    gpu.set_option("dense3", 0)
    try:
        gpu.set_option("policy_reset", 1)
        certified = []
        for _ in range(4):
            sdf, ext = gpu.build(m, 0.05)
            assert np.array_equal(sdf.view(np.uint32), ex.view(np.uint32)) and ext == ex_ext
            certified.append(gpu.last_dense_certified())
        assert not any(certified)
    finally:
        gpu.set_option("dense3", 1)
        gpu.set_option("policy_reset", 1)


def test_profiling_levels_and_stage_entry_points_fold(gpu):
    """Profiling levels 1 / 2 / 3 (every stage / ball kernel only / ball kernel on every 4th build) and the deferred
    fold of the stage entry points."""
    import torch
    shape = (32, 32, 64)
    m = torch.from_numpy(synth.bernoulli_mask(shape, 0.5, 2)).cuda()
    out = torch.empty(shape, dtype=torch.float32, device="cuda")

    def run(n):
        for _ in range(n):
            gpu.build_device(m.data_ptr(), shape, out.data_ptr(), 0.1, False, torch.cuda.current_stream().cuda_stream)
    run(2)
    gpu.get_stage_times()
    for level, builds in ((1, 8), (2, 8), (3, 2)):
        gpu.set_profiling(level)
        run(8)
        ms, n = gpu.get_stage_times()
        assert n == builds and ms[1] > 0.0                  # the ball kernel is timed at every level
        assert (ms[0] > 0.0) == (level == 1)                # the pack kernel only at level 1
    gpu.set_profiling(0)
    # stage entry points: with "defer_fold" the maxima stay in the slot array until sdfgpu_fold_extrema_device
    bits = torch.zeros((shape[0], shape[1], shape[2] // 32), dtype=torch.int32, device="cuda")
    small = torch.zeros(4, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    gpu.pack_bits_device(m.data_ptr(), shape[0] * shape[1], shape[2], bits.data_ptr(), s)
    gpu.set_option("defer_fold", 1)
    gpu.dense_ball_device(bits.data_ptr(), shape[0], 0, shape[0], shape[1], shape[2], 0.1, out.data_ptr(), small.data_ptr(),
                          small.data_ptr() + 12, s)
    assert small.tolist()[:2] == [0, 0]
    gpu.fold_extrema_device(small.data_ptr(), s)
    gpu.set_option("defer_fold", 0)
    mf, mq, _, unc = small.tolist()
    want, want_ext, dsq = O.exact_sdf(m.cpu().numpy(), 0.1)
    assert unc == 0 and mf == int(dsq.max()) and mq == int(-dsq.min())
    assert np.array_equal(out.cpu().numpy(), want)
