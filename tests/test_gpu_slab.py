"""GPU: the slab-mode stage entry points (sdfgpu_sweep_zy_device / sdfgpu_sweep_x_device) that the
multi-GPU path is built from.  One GPU plays every rank in turn; the "exchange" is a tensor copy.
The RCCL exchange itself (sdf_tools_amd/slab.py) is covered on CPU with gloo (test_slab_gloo.py)
and runs for real in `bench.py --gpus N`."""
import numpy as np
import pytest

from oracle import oracle as O
from sdf_tools_amd import capi, slab, synth

pytestmark = pytest.mark.gpu


def _emulate(shape, m, world, halo, res, vb):
    import torch
    dev = torch.device("cuda", 0)
    stages = slab.HipStages(0)
    nx, ny, nz = shape
    ranges = [slab.slab_range(nx, r, world) for r in range(world)]
    halo = min(halo, min(b - a for a, b in ranges))
    planes = []
    for (a, b) in ranges:                                   # step 1 on every "rank"
        rows = torch.empty((b - a, ny, nz), dtype=torch.int32, device=dev)
        stages.sweep_zy(torch.from_numpy(m[a:b]).to(dev), rows)
        planes.append(rows)
    full_field = torch.cat(planes)
    out = np.empty(shape, np.float32)
    maxima = np.zeros(2, np.int64)
    status_any = 0
    for r, (a, b) in enumerate(ranges):                     # steps 2-3
        lo = halo if r > 0 else 0
        hi = halo if r < world - 1 else 0
        ext = full_field[a - lo:b + hi].contiguous()        # what the halo exchange would deliver
        o = torch.empty((b - a, ny, nz), dtype=torch.float32, device=dev)
        small = torch.zeros(4, dtype=torch.int32, device=dev)
        stages.sweep_x(ext, lo, b - a, hi, a - lo > 0, b + hi < nx, a, nx, res, vb, o, small)
        stages.fold(small)                                  # HipStages defers the fold of the maxima to the caller
        mf, mq, st, _ = small.tolist()
        if st:                                              # step 5: whole lines
            status_any = 1
            small.zero_()
            stages.sweep_x(full_field, a, b - a, nx - b, False, False, a, nx, res, vb, o, small)
            stages.fold(small)
            mf, mq, st, _ = small.tolist()
            assert st == 0
        out[a:b] = o.cpu().numpy()
        maxima = np.maximum(maxima, [mf, mq])
    return out, capi.extrema_from_dsq(int(maxima[0]), int(maxima[1]), res), status_any


def _emulate_repartition(shape, m, world, res, vb):
    """The far-field general path of slab.py with one GPU playing every rank: tiered z / y sweeps per x slab (int32 plane
    field + far hint), the x-slab -> y-slab re-partition as tensor slicing, sdfgpu_sweep_x_lines_device per y slab."""
    import torch
    dev = torch.device("cuda", 0)
    stages = slab.HipStages(0)
    nx, ny, nz = shape
    field = torch.empty(shape, dtype=torch.int32, device=dev)
    hints = []
    for r in range(world):
        a, b = slab.slab_range(nx, r, world)
        if b > a:
            hint = torch.zeros(1, dtype=torch.int32, device=dev)
            stages.sweep_zy(torch.from_numpy(m[a:b]).to(dev), field[a:b], hint)
            hints.append(int(hint.item()))
    out = np.empty(shape, np.float32)
    maxima = np.zeros(2, np.int64)
    for r in range(world):
        ya, yb = slab.slab_range(ny, r, world)
        if yb == ya:
            continue
        lines = field[:, ya:yb].contiguous()                  # what the re-partition delivers to rank r
        o = torch.empty((nx, yb - ya, nz), dtype=torch.float32, device=dev)
        small = torch.zeros(4, dtype=torch.int32, device=dev)
        stages.sweep_x_lines(lines, ya, ny, res, vb, o, small)
        stages.fold(small)
        out[:, ya:yb] = o.cpu().numpy()
        maxima = np.maximum(maxima, small.cpu().numpy()[:2])
    return out, capi.extrema_from_dsq(int(maxima[0]), int(maxima[1]), res), hints


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("vb", [False, True])
def test_repartitioned_whole_line_sweep_is_exact(gpu, world, vb):
    """the exact general-scene multi-GPU path -- no all-gather of the field, every 'rank' holds
    1/G of every array; far-field scene (two boxes), a sparse cloud and a dense scene."""
    for shape, m in (((64, 48, 64), None), ((40, 24, 32), synth.bernoulli_mask((40, 24, 32), 0.002, 3)),
                     ((32, 16, 48), synth.bernoulli_mask((32, 16, 48), 0.5, 1))):
        if m is None:
            m = np.zeros(shape, np.uint8)
            m[5:12, 20:30, :20] = 1
            m[40:50, 5:15, 16:40] = 1
        got, ext, hints = _emulate_repartition(shape, m, world, 0.05, vb)
        want, want_ext, _ = O.exact_sdf(m, 0.05, vb)
        bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        assert len(bad) == 0, (shape, world, vb, len(bad), bad[:3].tolist())
        assert ext == want_ext
        if shape == (64, 48, 64):
            assert any(hints)                                   # the probe recognises the far-field slabs


def test_cavity_in_clutter_extrema_through_whole_lines(gpu):
    """ADVICE r2 medium / VERDICT r3 missing #6: near-field clutter + one cavity deeper than the marching sweep's scan
    bound.  (a) the stage entry point sdfgpu_sweep_x_lines_device on complete lines of y slabs (probe: near-field ->
    marching sweep, bounded -> far flag -> far-field kernel): the marching sweep's upper bounds for the voxels it left
    undecided must not reach the extrema; (b) SlabSdfBuilder._whole_lines end to end at world = 1."""
    import torch
    from test_gpu_multi import cavity_scene
    m = cavity_scene()
    shape = m.shape
    want, want_ext, dsq = O.exact_sdf(m, 0.01)
    assert np.abs(dsq).max() > 41 * 41
    got, ext, hints = _emulate_repartition(shape, m, 2, 0.01, False)
    assert not any(hints)                                        # the y probes saw a near-field scene
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert ext == want_ext, (ext, want_ext)
    # the builder at world = 1: the halo sweep sees the whole grid (nothing truncated, exact by itself); with the x sweep
    # predicted "complete lines" the same build goes through _whole_lines -- both exact, extrema included
    for predict_far, fallbacks in ((False, 0), (True, 1)):
        b = slab.SlabSdfBuilder(slab.HipStages(0), shape, 0.01, False, rank=0, world=1)
        b.predict_far = predict_far
        o, bext = b.build(torch.from_numpy(m).cuda())
        assert np.array_equal(o.cpu().numpy().view(np.uint32), want.view(np.uint32)) and bext == want_ext, predict_far
        assert b.general_builds == 1 and b.fallbacks == fallbacks and b.host_reads == 1
        assert not b.predict_far                                 # no far hint came back: the next build tries the halo path


def test_slab_builder_far_field_world1(gpu):
    """SlabSdfBuilder end to end on a far-field scene at world = 1: dense attempt uncertified -> tiered sweeps ->
    x sweep (halo path first, complete lines once the far hint has come back; the re-partition degenerates to a copy)."""
    import torch
    shape = (64, 64, 64)
    m = np.zeros(shape, np.uint8)
    m[10:14, 30:40, 5:25] = 1
    b = slab.SlabSdfBuilder(slab.HipStages(0), shape, 0.02, False, rank=0, world=1)
    o, ext = b.build(torch.from_numpy(m).cuda())
    ex, ex_ext, _ = O.exact_sdf(m, 0.02)
    assert np.array_equal(o.cpu().numpy(), ex) and ext == ex_ext
    # round 5: the first build predicts "near"; at world = 1 nothing is truncated, the halo-path sweep resolves
    # every voxel, so its result is KEPT (a far hint alone no longer forces the redo) and only steers the next prediction ...
    assert b.general_builds == 1 and b.fallbacks == 0 and b.mispredictions == 0 and b.predict_far
    # ... and the next build of the same scene goes straight to complete lines
    o, ext = b.build(torch.from_numpy(m).cuda())
    assert np.array_equal(o.cpu().numpy(), ex) and ext == ex_ext
    assert b.general_builds == 2 and b.fallbacks == 1 and b.mispredictions == 0


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dense_slabs_halo_is_enough(gpu, world):
    shape = (64, 32, 48)
    m = synth.bernoulli_mask(shape, 0.5, 3)
    got, ext, status = _emulate(shape, m, world, 4, 0.01, False)
    want, want_ext = O.reference_sdf(m, 0.01)
    assert status == 0
    assert np.array_equal(got, want) and ext == want_ext


@pytest.mark.parametrize("vb", [False, True])
def test_sparse_slabs_raise_status_and_fallback_is_exact(gpu, vb):
    shape = (48, 20, 24)
    m = synth.bernoulli_mask(shape, 0.002, 9)
    assert m.sum() > 0
    got, ext, status = _emulate(shape, m, 4, 3, 1.0, vb)
    want, want_ext, _ = O.exact_sdf(m, 1.0, vb)
    assert status == 1
    assert np.array_equal(got, want) and ext == want_ext


def test_single_rank_slab_equals_whole_path(gpu):
    shape = (40, 24, 32)
    m = synth.bernoulli_mask(shape, 0.3, 4)
    got, ext, status = _emulate(shape, m, 1, 8, 0.5, True)
    want, want_ext = gpu.build(m, 0.5, True)
    assert status == 0 and np.array_equal(got, want) and ext == want_ext


@pytest.mark.parametrize("world", [2, 4])
def test_dense_stages_on_slabs(gpu, world):
    """sdfgpu_pack_bits_device + sdfgpu_dense_ball_device with bit-plane halos, one GPU playing every rank."""
    import torch
    dev = torch.device("cuda", 0)
    stages = slab.HipStages(0)
    shape = (32, 24, 64)
    nx, ny, nz = shape
    for p, expect_cert in ((0.5, True), (0.003, False)):
        m = synth.bernoulli_mask(shape, p, 6)
        bits_all = torch.zeros((nx, ny, nz // 32), dtype=torch.int32, device=dev)
        stages.pack_bits(torch.from_numpy(m).to(dev), bits_all)
        out = np.empty(shape, np.float32)
        small_all = np.zeros(4, np.int64)
        for r in range(world):
            a, b = slab.slab_range(nx, r, world)
            lo = slab.BALL_HALO if r > 0 else 0
            hi = slab.BALL_HALO if r < world - 1 else 0
            ext = bits_all[a - lo:b + hi].contiguous()          # what the bit-plane exchange would deliver
            o = torch.empty((b - a, ny, nz), dtype=torch.float32, device=dev)
            small = torch.zeros(4, dtype=torch.int32, device=dev)
            n, h = b - a, slab.BALL_HALO
            if world == 4:
                stages.dense_ball(ext, lo, lo + n, nz, 0.5, o, small)
            else:
                # the builder's schedule: planes that need no neighbour data first, border planes afterwards
                i_lo, i_hi = (h if lo else 0), (n - h if hi else n)
                stages.dense_ball(ext, lo + i_lo, lo + i_hi, nz, 0.5, o[i_lo:i_hi], small)
                if i_lo:
                    stages.dense_ball(ext, lo, lo + i_lo, nz, 0.5, o[:i_lo], small)
                if i_hi < n:
                    stages.dense_ball(ext, lo + i_hi, lo + n, nz, 0.5, o[i_hi:], small)
            stages.fold(small)                                  # one fold for the 1-3 launches of this "rank"
            out[a:b] = o.cpu().numpy()
            small_all = np.maximum(small_all, small.cpu().numpy())
        assert bool(small_all[3] == 0) == expect_cert
        if expect_cert:
            want, want_ext = O.reference_sdf(m, 0.5)
            assert np.array_equal(out, want)
            assert capi.extrema_from_dsq(int(small_all[0]), int(small_all[1]), 0.5) == want_ext
        else:
            ex, _, dsq = O.exact_sdf(m, 0.5)                      # certified voxels (d^2 <= 8) are still exact
            near = np.abs(dsq) <= 8
            assert np.array_equal(out[near], ex[near])


def test_slab_builder_single_rank_pipelined(gpu):
    """SlabSdfBuilder + HipStages end to end at world = 1: dense path certified, deferred validation,
    double-buffered slots; and a sparse grid that must take the general path."""
    import torch
    dev = torch.device("cuda", 0)
    stages = slab.HipStages(0)
    shape = (24, 20, 64)
    m = synth.bernoulli_mask(shape, 0.5, 8)
    b = slab.SlabSdfBuilder(stages, shape, 0.1, False, rank=0, world=1)
    assert b.dense
    mt = torch.from_numpy(m).to(dev)
    prev, outs = None, []
    for _ in range(4):
        t = b.build_async(mt)
        if prev is not None:
            outs.append(b.finish(prev))
        prev = t
    outs.append(b.finish(prev))
    want, want_ext = O.reference_sdf(m, 0.1)
    for o, ext in outs:
        assert np.array_equal(o.cpu().numpy(), want) and ext == want_ext
    assert b.general_builds == 0 and b.fallbacks == 0
    ms = synth.bernoulli_mask(shape, 0.002, 8)
    o, ext = b.build(torch.from_numpy(ms).to(dev))
    ex, ex_ext, _ = O.exact_sdf(ms, 0.1)
    assert np.array_equal(o.cpu().numpy(), ex) and ext == ex_ext and b.general_builds == 1
    # virtual border disables the dense path inside the builder
    bv = slab.SlabSdfBuilder(stages, shape, 0.1, True, rank=0, world=1)
    assert not bv.dense
    o, ext = bv.build(mt)
    ex, ex_ext, _ = O.exact_sdf(m, 0.1, True)
    assert np.array_equal(o.cpu().numpy(), ex) and ext == ex_ext


def test_gradient_matches_reference_definition(gpu):
    """N1: sdfgpu_gradient_device vs a numpy restatement of GetGridAlignedGradient (sdf.hpp:432-526)."""
    import torch
    shape = (12, 9, 10)
    res = 0.25
    m = synth.bernoulli_mask(shape, 0.3, 2)
    sdf, _ = gpu.build(m, res)
    f = torch.from_numpy(sdf).cuda()
    g = torch.empty(shape + (3,), dtype=torch.float64, device="cuda")
    gpu.gradient_device(f.data_ptr(), shape, g.data_ptr(), res, True, True)
    got = g.cpu().numpy()
    want = np.zeros(shape + (3,), np.float64)
    nx, ny, nz = shape
    for x in range(nx):
        for y in range(ny):
            for z in range(nz):
                interior = 0 < x < nx - 1 and 0 < y < ny - 1 and 0 < z < nz - 1
                idx = [x, y, z]
                for ax, n in enumerate(shape):
                    lo, hi = list(idx), list(idx)
                    if interior:
                        lo[ax] -= 1
                        hi[ax] += 1
                        want[x, y, z, ax] = float(np.float32(sdf[tuple(hi)] - sdf[tuple(lo)])) * (1.0 / (2.0 * res))
                    else:
                        lo[ax] = max(0, idx[ax] - 1)
                        hi[ax] = min(n - 1, idx[ax] + 1)
                        inc = (hi[ax] - lo[ax]) * res
                        if inc > 0:
                            want[x, y, z, ax] = (float(sdf[tuple(hi)]) - float(sdf[tuple(lo)])) * (1.0 / inc)
    assert np.array_equal(got, want)
    # edge gradients disabled -> NaN on the boundary shell, unchanged interior
    gpu.gradient_device(f.data_ptr(), shape, g.data_ptr(), res, False, True)
    got2 = g.cpu().numpy()
    assert np.array_equal(got2[1:-1, 1:-1, 1:-1], want[1:-1, 1:-1, 1:-1])
    assert np.all(np.isnan(got2[0])) and np.all(np.isnan(got2[:, :, -1]))
    # fp32 output (vectorised kernel when nz % 4 == 0) == fp64 output narrowed once
    # (res = 0.25, 0.01: 1 / (2 res) is an fp32 number -> the kernel scales in fp32; 0.03, 0.007: fp64 scale)
    for shp in ((10, 9, 16), (3, 5, 8), (12, 9, 10), (1, 7, 12), (40, 33, 64), (1, 6, 8), (2, 1, 16), (5, 2, 4)):
        mm = synth.bernoulli_mask(shp, 0.3, 4)
        if shp == (1, 6, 8):
            mm[:] = 0                              # all free: +inf everywhere (inf - inf = NaN, but 0 on the singleton axis)
        for r2 in (res, 0.01, 0.03, 0.007):
            ss, _ = gpu.build(mm, r2)
            ft = torch.from_numpy(ss).cuda()
            for edge in (True, False):
                g64 = torch.empty(shp + (3,), dtype=torch.float64, device="cuda")
                g32 = torch.empty(shp + (3,), dtype=torch.float32, device="cuda")
                gpu.gradient_device(ft.data_ptr(), shp, g64.data_ptr(), r2, edge, True)
                gpu.gradient_device(ft.data_ptr(), shp, g32.data_ptr(), r2, edge, False)
                assert np.array_equal(g32.cpu().numpy(), g64.cpu().numpy().astype(np.float32), equal_nan=True), (shp, r2, edge)
    # test_bindings.py:33 -- gradient at (x=4, y=1) of the 20x40x1 scene is [1.5, 0]
    m2 = np.zeros((20, 40, 1), np.uint8)
    m2[3, 1, 0] = 1
    s2, _ = gpu.build(m2, 0.05)
    f2 = torch.from_numpy(s2).cuda()
    g2 = torch.empty((20, 40, 1, 3), dtype=torch.float64, device="cuda")
    gpu.gradient_device(f2.data_ptr(), (20, 40, 1), g2.data_ptr(), 0.05, True, True)
    np.testing.assert_allclose(g2[4, 1, 0].cpu().numpy(), [1.5, 0.0, 0.0], rtol=0, atol=1e-6)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_native_dense_phases_on_slabs(gpu, world):
    """sdfgpu_slab_dense_phase: the three-call schedule of a dense slab build, one GPU playing every rank (the halo
    planes are copied between the ranks' buffers where the bit-plane exchange would run)."""
    import torch
    dev = torch.device("cuda", 0)
    stages = slab.HipStages(0)
    shape = (36, 24, 64)
    nx, ny, nz = shape
    h = slab.BALL_HALO
    m = synth.bernoulli_mask(shape, 0.5, 8)
    want, want_ext = O.reference_sdf(m, 0.5)
    ranks = []
    for r in range(world):
        a, b = slab.slab_range(nx, r, world)
        lo = h if r > 0 else 0
        hi = h if r < world - 1 else 0
        ranks.append(dict(a=a, b=b, lo=lo, hi=hi, mask=torch.from_numpy(m[a:b]).to(dev),
                          bits=torch.zeros((lo + b - a + hi, ny, nz // 32), dtype=torch.int32, device=dev),
                          out=torch.empty((b - a, ny, nz), dtype=torch.float32, device=dev),
                          small=torch.full((4,), 7, dtype=torch.int32, device=dev)))
    for k in ranks:                                             # phase 0 everywhere, then "exchange"
        stages.dense_phase(0, k["mask"], k["bits"], k["lo"], k["hi"], 0.5, k["out"], k["small"])
    for r, k in enumerate(ranks):
        n = k["b"] - k["a"]
        if k["lo"]:
            p = ranks[r - 1]
            k["bits"][:h] = p["bits"][p["lo"] + (p["b"] - p["a"]) - h:p["lo"] + (p["b"] - p["a"])]
        if k["hi"]:
            q = ranks[r + 1]
            k["bits"][k["lo"] + n:] = q["bits"][q["lo"]:q["lo"] + h]
    out = np.empty(shape, np.float32)
    small_all = np.zeros(4, np.int64)
    for r, k in enumerate(ranks):
        if r % 2:                                               # both forms of phase 1
            stages.dense_phase(10, k["mask"], k["bits"], k["lo"], k["hi"], 0.5, k["out"], k["small"])
            stages.dense_phase(11, k["mask"], k["bits"], k["lo"], k["hi"], 0.5, k["out"], k["small"])
        else:
            stages.dense_phase(1, k["mask"], k["bits"], k["lo"], k["hi"], 0.5, k["out"], k["small"])
        stages.dense_phase(2, k["mask"], k["bits"], k["lo"], k["hi"], 0.5, k["out"], k["small"])
        out[k["a"]:k["b"]] = k["out"].cpu().numpy()
        small_all = np.maximum(small_all, k["small"].cpu().numpy())
    assert small_all[3] == 0 and small_all[2] == 0
    assert np.array_equal(out, want)
    assert capi.extrema_from_dsq(int(small_all[0]), int(small_all[1]), 0.5) == want_ext
