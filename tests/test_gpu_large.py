"""GPU: BASELINE.json's full sizes through size-independent properties plus a full exact check."""
import numpy as np
import pytest

from oracle import oracle as O
from sdf_tools_amd import synth

pytestmark = pytest.mark.gpu


def _torch_build(gpu, m_t, res=1.0, vb=False):
    import torch
    out = torch.empty(m_t.shape, dtype=torch.float32, device=m_t.device)
    gpu.build_device(m_t.data_ptr(), tuple(m_t.shape), out.data_ptr(), res, vb,
                     torch.cuda.current_stream().cuda_stream)
    ext = gpu.get_extrema()
    return out, ext


def test_512_cube_properties_and_exact(gpu):
    """BASELINE configs[2]: 512^3, device-resident in and out."""
    import torch
    n = 512
    m_t = synth.bernoulli_mask_torch((n, n, n), 0.5, 1, device="cuda")
    sdf, ext = _torch_build(gpu, m_t)
    # sign == occupancy, bit exact
    assert bool(torch.equal(sdf < 0, m_t != 0))
    # every voxel has an opposite-class voxel at distance >= 1
    assert float(sdf.abs().min()) >= 1.0
    # squared distances are integers
    d2 = (sdf.double() ** 2)
    assert float((d2 - d2.round()).abs().max()) < 1e-4
    # 1-Lipschitz along every axis between same-class neighbours (|grad| <= 1 for an EDT)
    for ax in range(3):
        a = sdf.narrow(ax, 0, n - 1)
        b = sdf.narrow(ax, 1, n - 1)
        same = (a < 0) == (b < 0)
        assert float(((a - b).abs() * same).max()) <= 1.0 + 1e-6
    # extrema agree with the field
    assert ext[0] == pytest.approx(float(sdf.max()), abs=1e-6) and ext[1] == pytest.approx(float(sdf.min()), abs=1e-6)
    # idempotence: rebuilding from the sign of the result reproduces it
    sdf2, _ = _torch_build(gpu, (sdf < 0).to(torch.uint8))
    assert bool(torch.equal(sdf, sdf2))
    # slabs are consistent with an independent exact CPU transform on a 512x512x64... sub-block:
    # interior voxels of a 96^3 crop whose distance is below the crop margin must agree exactly
    c0, cs, margin = 200, 96, 8
    crop = m_t[c0:c0 + cs, c0:c0 + cs, c0:c0 + cs].cpu().numpy()
    ex, _, _ = O.exact_sdf(crop, 1.0)
    got = sdf[c0:c0 + cs, c0:c0 + cs, c0:c0 + cs].cpu().numpy()
    inner = (slice(margin, cs - margin),) * 3
    assert np.all(np.abs(ex[inner]) < margin)
    assert np.array_equal(got[inner], ex[inner])


def test_512_cube_matches_exact_everywhere(gpu):
    import torch
    n = 512
    m = synth.bernoulli_mask((n, n, n), 0.5, 2)
    m_t = torch.from_numpy(m).cuda()
    sdf, ext = _torch_build(gpu, m_t, 0.01)
    ex, ex_ext, _ = O.exact_sdf(m, 0.01)
    assert np.array_equal(sdf.cpu().numpy(), ex)
    assert ext == ex_ext


def test_dense_to_far_field_transition_build_is_bounded():
    """VERDICT r3 "next round" 1: config 5 is a STREAM -- scenes change under a handle.  One handle at 512^3: Bernoulli(0.5)
    until the dense tier is trusted (the guarded stand-by behind it is then the far-field pair, three small-grid launches),
    then the two-box cloud.  The transition build -- dense kernels wasted, stand-by does the work -- must be exact and take
    at most 3x the steady-state far-field build (round 3: the stand-by was fused K12 + K3/16 with unbounded scans, tens of
    ms on this scene), and the handle must come back to the dense scene exactly."""
    import torch
    from sdf_tools_amd import capi
    n, res = 512, 0.01
    dev = torch.device("cuda", 0)
    ctx = capi.SdfGpu(0)                                         # own handle, default policy
    try:
        dense = synth.bernoulli_mask_torch((n, n, n), 0.5, 1, device=dev)
        pts = torch.from_numpy(synth.two_box_points(200000, seed=0, scale=n * res)).to(dev)
        far = torch.zeros((n, n, n), dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        ctx.voxelize_points_device(pts.data_ptr(), pts.shape[0], (0.0, 0.0, 0.0), res, (n, n, n), far.data_ptr(), True, s)
        out = torch.empty((n, n, n), dtype=torch.float32, device=dev)

        def timed(mask):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)

        for _ in range(6):                                       # synchronised: the policy sees every report
            timed(dense)
        assert ctx.last_path()["dense_certified"]
        info = ctx.last_build_info()
        assert info["standby_far"] and not info["fused_zy"], info
        t_dense = min(timed(dense) for _ in range(5))
        want_dense = out.clone()
        ext_dense = ctx.get_extrema()
        t_trans = timed(far)                                     # the transition build
        path = ctx.last_path()
        assert path["far_y"] and path["far_x"] and not path["dense_certified"], path
        assert ctx.last_build_info()["standby_far"]
        got_trans, ext_trans = out.clone(), ctx.get_extrema()
        for _ in range(4):
            timed(far)
        assert not ctx.last_build_info()["standby_far"]
        t_far = min(timed(far) for _ in range(5))
        assert bool(torch.equal(out, got_trans)) and ctx.get_extrema() == ext_trans
        ex, ex_ext, _ = O.exact_sdf(far.cpu().numpy(), res)
        assert np.array_equal(got_trans.cpu().numpy().view(np.uint32), ex.view(np.uint32)) and ext_trans == ex_ext
        print("dense %.3f ms, transition %.3f ms, far-field steady state %.3f ms" % (t_dense, t_trans, t_far))
        assert t_trans <= 3.0 * t_far, (t_trans, t_far)
        timed(dense)                                             # and back: exact whatever the handle has learned
        assert bool(torch.equal(out, want_dense)) and ctx.get_extrema() == ext_dense
    finally:
        ctx.close()


def test_sparse_256_far_scans(gpu):
    """Sparse 256^3 (p = 1e-4): distances of tens of voxels, the outward-scan path dominates."""
    m = synth.bernoulli_mask((256, 256, 256), 1e-4, 3)
    sdf, ext = gpu.build(m, 1.0)
    ex, ex_ext, _ = O.exact_sdf(m, 1.0)
    assert np.array_equal(sdf, ex) and ext == ex_ext
    sdf, ext = gpu.build(1 - m, 1.0, add_virtual_border=True)
    ex, ex_ext, _ = O.exact_sdf(1 - m, 1.0, True)
    assert np.array_equal(sdf, ex) and ext == ex_ext


def test_builds_on_two_streams_are_ordered(gpu):
    """builds on one handle from different streams share the handle's status block and scratch fields;
    the library orders them (event wait) so that alternating streams without any host synchronisation stays exact."""
    import torch
    n = 192
    scenes_ = [synth.bernoulli_mask((n, n, n), p, s) for p, s in ((0.5, 1), (0.02, 2), (0.5, 3), (0.001, 4))]
    wants = [O.exact_sdf(m, 0.01) for m in scenes_]
    dev = torch.device("cuda", 0)
    masks = [torch.from_numpy(m).to(dev) for m in scenes_]
    outs = [torch.empty((n, n, n), dtype=torch.float32, device=dev) for _ in scenes_]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    for rep in range(3):
        for k, (mk, o) in enumerate(zip(masks, outs)):
            s = streams[(k + rep) % 2]
            gpu.build_device(mk.data_ptr(), (n, n, n), o.data_ptr(), 0.01, False, s.cuda_stream)
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert np.array_equal(o.cpu().numpy(), wants[k][0]), (rep, k)
    assert gpu.get_extrema() == wants[-1][1]


def test_1024_cube_single_gpu_and_slab_builder(gpu):
    """BASELINE configs[3] on one GPU: 1024^3 Bernoulli(0.5) through the whole-grid ABI and through the multi-GPU
    slab builder at world = 1; size-independent properties plus random crops against the exact oracle."""
    import torch
    from sdf_tools_amd import slab
    n, res = 1024, 0.01
    dev = torch.device("cuda", 0)
    m_t = synth.bernoulli_mask_torch((n, n, n), 0.5, 3, device=dev)
    sdf = torch.empty((n, n, n), dtype=torch.float32, device=dev)
    gpu.build_device(m_t.data_ptr(), (n, n, n), sdf.data_ptr(), res, False, torch.cuda.current_stream().cuda_stream)
    ext = gpu.get_extrema()
    assert gpu.last_path()["dense_certified"]
    # properties, in x chunks to bound temporary memory
    mx, mn = -1e30, 1e30
    for x0 in range(0, n, 128):
        s, mk = sdf[x0:x0 + 128], m_t[x0:x0 + 128]
        assert bool(torch.equal(s < 0, mk != 0))                      # sign == occupancy, bit exact
        a = s.abs()
        assert float(a.min()) >= res * (1 - 1e-6)
        d2 = (a.double() / res) ** 2
        assert float((d2 - d2.round()).abs().max()) < 1e-3          # squared distances are integers
        mx, mn = max(mx, float(s.max())), min(mn, float(s.min()))
        del a, d2
    assert ext[0] == pytest.approx(mx, abs=1e-6) and ext[1] == pytest.approx(mn, abs=1e-6)
    maxd = int(round(max(ext[0], -ext[1]) / res)) + 2
    rng = np.random.default_rng(0)
    C = 40
    corners = [[0, 0, 0], [n - C, n - C, n - C], [0, n - C, 500]] + [[int(rng.integers(0, n - C)) for _ in range(3)] for _ in range(6)]
    crops = []
    for lo in corners:
        a = [max(0, v - maxd) for v in lo]
        b = [min(n, v + C + maxd) for v in lo]
        sub = m_t[a[0]:b[0], a[1]:b[1], a[2]:b[2]].cpu().numpy()
        want, _, _ = O.exact_sdf(sub, res)
        off = [v - aa for v, aa in zip(lo, a)]
        w = want[off[0]:off[0] + C, off[1]:off[1] + C, off[2]:off[2] + C]
        crops.append((lo, w))
        g = sdf[lo[0]:lo[0] + C, lo[1]:lo[1] + C, lo[2]:lo[2] + C].cpu().numpy()
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), lo
    del sdf
    # the same grid through the slab builder (the N > 1 code path with world = 1)
    builder = slab.SlabSdfBuilder(slab.HipStages(0), (n, n, n), res, False, halo=3, rank=0, world=1)
    out, ext2 = builder.build(m_t)
    assert ext2 == ext
    for lo, w in crops:
        g = out[lo[0]:lo[0] + C, lo[1]:lo[1] + C, lo[2]:lo[2] + C].cpu().numpy()
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), lo


def test_1024_long_lines_far_field_against_unbounded_scans(gpu):
    """Maximum line length of the far-field kernel's grid sizes: a structured 1024 x 1024 x 256 scene (walls, floor, table, shelf:
    far-field, thin and thick solids) through the library's own tier selection -- the far-field pair with 1024-voxel y and x
    lines (one interval per lane in level B, wave-cooperative scans) -- must equal, bit for bit, the same build with the far-field
    kernel switched off (marching sweeps with unbounded exact scans: a different algorithm), and satisfy the size-independent
    properties of an exact signed EDT; the extrema agree as well.  Too large for the CPU oracle in test time."""
    import torch
    shape, res = (1024, 1024, 256), 0.01
    dev = torch.device("cuda", 0)
    m_t = synth.room_mask_torch(shape, dev)
    gpu.set_option("policy_reset", 1)
    stream = torch.cuda.current_stream().cuda_stream
    sdf = torch.empty(shape, dtype=torch.float32, device=dev)
    gpu.build_device(m_t.data_ptr(), shape, sdf.data_ptr(), res, False, stream)
    ext = gpu.get_extrema()
    path = gpu.last_path()
    assert path["far_y"] and path["far_x"] and not path["dense_certified"]
    gpu.set_option("envelope", 0)
    try:
        ref = torch.empty(shape, dtype=torch.float32, device=dev)
        gpu.build_device(m_t.data_ptr(), shape, ref.data_ptr(), res, False, stream)
        ext_ref = gpu.get_extrema()
        p2 = gpu.last_path()
    finally:
        gpu.set_option("envelope", 1)
        gpu.set_option("policy_reset", 1)
    assert not p2["far_y"] and not p2["far_x"]
    assert bool(torch.equal(sdf.view(torch.int32), ref.view(torch.int32))) and ext == ext_ref
    del ref
    assert bool(torch.equal(sdf < 0, m_t != 0))                          # sign == occupancy
    assert float(sdf.abs().min()) >= res * (1 - 1e-6)
    for ax in range(3):                                                  # 1-Lipschitz between same-class neighbours
        a, b = sdf.narrow(ax, 0, shape[ax] - 1), sdf.narrow(ax, 1, shape[ax] - 1)
        same = (a < 0) == (b < 0)
        assert float(((a - b).abs() * same).max()) <= res + 2e-6         # (fp32 values up to ~10: differences carry ~1e-6 of rounding)
        del a, b, same
    assert ext[0] == pytest.approx(float(sdf.max()), abs=1e-6) and ext[1] == pytest.approx(float(sdf.min()), abs=1e-6)


def test_1024_lines_in_both_swept_axes_against_the_oracle(gpu):
    """VERDICT r4 "next round" 7a: ORACLE-backed far-field scenes with full-length 1024-voxel lines in BOTH swept axes and a
    non-trivial cross-section -- 1024 x 1024 x 32 slices of the room (two walls, table on legs, shelf; with and without the
    floor) plus a few noise voxels: 33 M voxels, within the CPU oracle's reach -- next to the GPU-vs-GPU 1024 x 1024 x 256 test
    above.  The library's own tier selection must take the far-field pair (512-lane workgroups, one interval per lane in
    level B, wave-cooperative scans, round 5's flat-stretch shortcut beside the walls and over the floor); every voxel and
    the extrema bit for bit against the oracle's exact EDT.  Without the floor the argmins travel up to 83 voxels (to the nearest noise voxel, wall or piece of furniture) along both
    axes; with it (and the virtual border) every line is one plateau with jumps at the furniture."""
    import torch
    shape, res = (1024, 1024, 32), 0.01
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for floor, vb in ((False, False), (True, True)):
        m_t = synth.room_mask_torch(shape, dev, floor=floor)
        noise = synth.bernoulli_mask_torch(shape, 2e-5, 77, device=dev)
        noise[:, :, :8] = 0                                          # (keep the floor's neighbourhood clean: plateaus AND jumps)
        m_t |= noise
        m = m_t.cpu().numpy()
        gpu.set_option("policy_reset", 1)
        sdf = torch.empty(shape, dtype=torch.float32, device=dev)
        gpu.build_device(m_t.data_ptr(), shape, sdf.data_ptr(), res, vb, stream)
        ext = gpu.get_extrema()
        path = gpu.last_path()
        assert path["far_y"] and path["far_x"] and not path["dense_certified"], (floor, path)
        want, want_ext, dsq = O.exact_sdf(m, res, vb)
        got = sdf.cpu().numpy()
        bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        assert len(bad) == 0, (floor, vb, len(bad), bad[:3].tolist())
        assert ext == want_ext, (floor, vb, ext, want_ext)
        if not floor:
            assert np.abs(dsq).max() > 60 * 60                       # far-field indeed (the noise voxels bound it: 83 here)
        del sdf, m_t, noise
