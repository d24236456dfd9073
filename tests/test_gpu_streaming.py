"""GPU: next-row N2 (point cloud -> occupancy) and the streaming pipeline of BASELINE configs[4]."""
import numpy as np
import pytest

from oracle import oracle as O
from sdf_tools_amd import synth
from sdf_tools_amd.streaming import StreamingSdf

pytestmark = pytest.mark.gpu


def _numpy_voxelize(pc, shape, res, origin):
    """scripts/3d_sdf_demo_rviz.py:22-29 restated (with out-of-grid points dropped instead of raising)."""
    idx = ((pc.astype(np.float64) - np.asarray(origin, np.float64)) / res).astype(np.int64)
    f = (pc.astype(np.float64) - np.asarray(origin, np.float64)) / res
    keep = np.all(f > -1.0, axis=1) & np.all(f < np.asarray(shape), axis=1)
    vg = np.zeros(shape, np.uint8)
    vg[idx[keep, 0], idx[keep, 1], idx[keep, 2]] = 1
    return vg


def test_voxelize_matches_the_demo_convention(gpu):
    import torch
    shape, res, origin = (25, 20, 15), 0.04, (0.0, 0.0, 0.0)
    rng = np.random.RandomState(0)                       # the demo's own cloud (3d_sdf_demo_rviz.py:14-19)
    pc = np.concatenate([rng.uniform([0.5, 0.5, 0], [0.7, 0.6, 0.5], [100, 3]),
                         rng.uniform([0.5, 0.2, 0.25], [0.75, 0.4, 0.5], [100, 3])]).astype(np.float32)
    mask = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    gpu.voxelize_points_device(torch.from_numpy(pc).cuda().data_ptr(), len(pc), origin, res, shape, mask.data_ptr())
    assert np.array_equal(mask.cpu().numpy(), _numpy_voxelize(pc, shape, res, origin))
    # points outside the grid (and NaNs) are dropped; an offset origin; no clearing accumulates
    pc2 = np.array([[-0.5, 0.1, 0.1], [0.1, 0.1, 5.0], [np.nan, 0, 0], [0.99, 0.79, 0.59], [-0.01, 0.0, 0.0]], np.float32)
    gpu.voxelize_points_device(torch.from_numpy(pc2).cuda().data_ptr(), len(pc2), (0.0, 0.0, 0.0), res, shape,
                               mask.data_ptr(), clear_first=False)
    want = _numpy_voxelize(pc, shape, res, origin) | _numpy_voxelize(pc2[[3, 4]], shape, res, origin)
    assert np.array_equal(mask.cpu().numpy(), want)


def test_streaming_frame_equals_oracle_pipeline(gpu):
    import torch
    n, res = 64, 0.02
    st = StreamingSdf((n, n, n), res, (0.0, 0.0, 0.0), 0, gradient=True, grad_f64=True)
    for seed in (0, 1):
        pc = synth.two_box_points(3000, seed=seed, scale=n * res)
        sdf, grad = st.frame(torch.from_numpy(pc).cuda())
        torch.cuda.synchronize()
        mask = _numpy_voxelize(pc, (n, n, n), res, (0, 0, 0))
        assert np.array_equal(st.mask.cpu().numpy(), mask)
        ex, ex_ext, _ = O.exact_sdf(mask, res)
        assert np.array_equal(sdf.cpu().numpy(), ex) and st.extrema() == ex_ext
        g = grad.cpu().numpy()
        inner = (slice(1, -1),) * 3
        want_gx = (ex[2:, 1:-1, 1:-1] - ex[:-2, 1:-1, 1:-1]).astype(np.float64) * (1.0 / (2.0 * res))
        assert np.array_equal(g[inner][..., 0], want_gx)


def _numpy_estimate_distance(sdf, res, g):
    """sdf.hpp:773-915 restated for grid-frame points g [n, 3] that lie inside the grid (float64)."""
    shape = np.asarray(sdf.shape)
    idx = np.floor(g * (1.0 / res)).astype(np.int64)
    centre = res * (idx + 0.5)
    off = g - centre
    lower, upper = idx.copy(), idx.copy()
    for ax in range(3):
        n = shape[ax]
        for k in range(len(g)):
            i = idx[k, ax]
            lo = hi = i
            if off[k, ax] >= 0.0:                           # :806-818
                hi = i + 1
                if hi >= n:
                    hi, lo = i, i - 1
                    if lo < 0:
                        lo = i
            else:                                           # :819-831
                lo = i - 1
                if lo < 0:
                    hi, lo = i + 1, i
                    if hi >= n:
                        hi = i
            lower[k, ax], upper[k, ax] = lo, hi
    half = res * 0.5

    def D(ix, iy, iz):                                      # :773-796
        d = sdf[ix, iy, iz].astype(np.float64)
        return np.where(d >= 0.0, d - half, d + half)

    lo_loc = res * (lower + 0.5)
    x0, y0, z0 = lower.T
    x1, y1, z1 = upper.T

    def bilinear(ll, lh, hl, hh):                           # :699-727 with the corner at lo_loc, side = res
        l1, h1, l2, h2 = lo_loc[:, 0], lo_loc[:, 0] + res, lo_loc[:, 1], lo_loc[:, 1] + res
        mult = 1.0 / ((h1 - l1) * (h2 - l2))
        a0, a1 = mult * (h1 - g[:, 0]), mult * (g[:, 0] - l1)
        return (a0 * ll + a1 * hl) * (h2 - g[:, 1]) + (a0 * lh + a1 * hh) * (g[:, 1] - l2)

    with np.errstate(divide="ignore", invalid="ignore"):
        mz = bilinear(D(x0, y0, z0), D(x0, y1, z0), D(x1, y0, z0), D(x1, y1, z0))
        pz = bilinear(D(x0, y0, z1), D(x0, y1, z1), D(x1, y0, z1), D(x1, y1, z1))
        return mz + (g[:, 2] - lo_loc[:, 2]) * ((pz - mz) * (1.0 / res))    # :745-771


def test_batched_point_queries_match_restated_estimate_distance_and_host_mirror(gpu):
    import torch
    from sdf_tools_amd._bindings import load_pysdf_tools
    shape, res, origin = (20, 14, 9), 0.25, (-1.0, 0.5, 2.0)
    m = synth.bernoulli_mask(shape, 0.2, 3)
    sdf_np, _, _ = O.exact_sdf(m, res)
    d_sdf = torch.from_numpy(sdf_np).cuda()
    rng = np.random.default_rng(5)
    extent = np.asarray(shape) * res
    pts = rng.uniform(-0.3, 1.3, size=(4000, 3)) * extent + np.asarray(origin)      # some outside
    # exact cell centres and points on faces / corners of the grid
    centres = (np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1).reshape(-1, 3) + 0.5) * res + origin
    pts = np.concatenate([pts, centres[::7], np.asarray(origin)[None] + 1e-9, np.asarray(origin)[None] + extent - 1e-9])
    d_pts = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    n = len(pts)
    dist = torch.empty(n, dtype=torch.float64, device="cuda")
    grad = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    w2g = (1, 0, 0, -origin[0], 0, 1, 0, -origin[1], 0, 0, 1, -origin[2])
    for edge in (False, True):
        gpu.query_points_device(d_sdf.data_ptr(), shape, res, d_pts.data_ptr(), n, dist.data_ptr(), grad.data_ptr(),
                                flags.data_ptr(), world_to_grid=w2g, oob_value=123.0, enable_edge_gradients=edge)
        torch.cuda.synchronize()
        got_d, got_g, got_f = dist.cpu().numpy(), grad.cpu().numpy(), flags.cpu().numpy()
        g = pts - np.asarray(origin)
        idx = np.floor(g * (1.0 / res)).astype(np.int64)
        inside = np.all(idx >= 0, axis=1) & np.all(idx < np.asarray(shape), axis=1)
        assert inside.sum() > 1000 and (~inside).sum() > 500
        assert np.array_equal((got_f & 1).astype(bool), inside)
        assert np.all(got_d[~inside] == 123.0) and np.all(np.isnan(got_g[~inside]))
        want = _numpy_estimate_distance(sdf_np, res, g[inside])
        assert np.allclose(got_d[inside], want, rtol=0, atol=1e-9)          # double arithmetic; 1e-5 is the stated bar
        # gradient of the containing cell: the full-grid kernel (already pinned to the reference definition)
        full = torch.empty(shape + (3,), dtype=torch.float64, device="cuda")
        gpu.gradient_device(d_sdf.data_ptr(), shape, full.data_ptr(), res, edge, True)
        fg = full.cpu().numpy()[tuple(idx[inside].T)]
        have = ~np.isnan(fg[:, 0])
        assert np.array_equal(((got_f[inside] >> 1) & 1).astype(bool), have)
        assert np.array_equal(got_g[inside][have], fg[have]) and np.all(np.isnan(got_g[inside][~have]))


def test_batched_point_queries_in_a_rotated_frame_match_the_host_mirror(gpu):
    """include/sdf_tools/sdf.hpp (EstimateDistance, GetGradient: written against the same reference lines, with
    the gradient rotated into the world frame, sdf.hpp:405-430) vs the device kernel, origin rotated + shifted."""
    import torch
    from sdf_tools_amd._bindings import load_pysdf_tools
    mod = load_pysdf_tools()
    shape, res = (12, 10, 8), 0.5
    c, s_ = np.cos(0.5), np.sin(0.5)
    T = np.array([[c, -s_, 0, 1.0], [s_, c, 0, -2.0], [0, 0, 1, 0.25], [0, 0, 0, 1]])
    grid = mod.CollisionMapGrid(mod.Isometry3d(T.tolist()), "w", res, shape[0], shape[1], shape[2], mod.COLLISION_CELL(0.0))
    grid.SetOccupancyFromNumpy(synth.bernoulli_mask(shape, 0.15, 8).astype(np.float32))
    host, _ = grid.ExtractSignedDistanceField(55.0, False, False)
    raw = host.GetRawDataNumpy()
    rng = np.random.default_rng(1)
    g = rng.uniform(-0.1, 1.1, size=(600, 3)) * np.asarray(shape) * res            # grid frame, some outside
    pts = np.ascontiguousarray((T[:3, :3] @ g.T).T + T[:3, 3])
    Tinv = np.linalg.inv(T)
    n = len(pts)
    d_sdf, d_pts = torch.from_numpy(raw).cuda(), torch.from_numpy(pts).cuda()
    dist = torch.empty(n, dtype=torch.float64, device="cuda")
    grad = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    gpu.query_points_device(d_sdf.data_ptr(), shape, res, d_pts.data_ptr(), n, dist.data_ptr(), grad.data_ptr(),
                            flags.data_ptr(), world_to_grid=Tinv[:3, :4], rotation=T[:3, :3], oob_value=55.0,
                            enable_edge_gradients=True)
    torch.cuda.synchronize()
    got_d, got_g, got_f = dist.cpu().numpy(), grad.cpu().numpy(), flags.cpu().numpy()
    seen_in = seen_out = 0
    for k in range(n):
        est, ok = host.EstimateDistance(*pts[k])
        if abs(np.floor(g[k] / res) - g[k] / res).min() < 1e-9:
            continue                                    # on a cell face: floor() may differ in the last ulp
        assert ok == bool(got_f[k] & 1)
        assert abs(est - got_d[k]) <= 1e-9
        if ok:
            seen_in += 1
            ix = np.floor(g[k] / res).astype(int)
            hg = host.GetGradient(int(ix[0]), int(ix[1]), int(ix[2]), True)
            assert np.allclose(got_g[k], hg, rtol=0, atol=1e-12)
        else:
            seen_out += 1
            assert est == 55.0
    assert seen_in > 300 and seen_out > 50


def test_streaming_query_after_frame(gpu):
    import torch
    n, res = 64, 0.02
    st = StreamingSdf((n, n, n), res, (0.0, 0.0, 0.0), 0, gradient=False)
    pc = synth.two_box_points(3000, seed=2, scale=n * res)
    sdf, _ = st.frame(torch.from_numpy(pc).cuda())
    q = torch.from_numpy(np.random.default_rng(0).uniform(0.0, n * res, size=(2000, 3))).cuda()
    d, g, f = st.query(q)
    torch.cuda.synchronize()
    assert bool((f.cpu().numpy() == 3).all())
    want = _numpy_estimate_distance(sdf.cpu().numpy(), res, q.cpu().numpy())
    assert np.allclose(d.cpu().numpy(), want, rtol=0, atol=1e-9)


def test_streaming_query_mode_fuses_the_consumers_queries_into_the_frame(gpu):
    """BASELINE configs[4], "fused gradient (EstimateDistance/gradient query) kernel": the default StreamingSdf answers the
    caller's distance + gradient queries on the fresh field (no full-grid gradient is written); same answers as the
    stand-alone query on the full-gradient variant's field, which is the same field."""
    import torch
    n, res = 64, 0.02
    stq = StreamingSdf((n, n, n), res, (0.0, 0.0, 0.0), 0)                       # default: gradient="query"
    stf = StreamingSdf((n, n, n), res, (0.0, 0.0, 0.0), 0, gradient="full")
    assert stq.mode == "query" and stq.gradient is None and stf.gradient is not None
    gen = torch.Generator(device="cuda").manual_seed(5)
    q = torch.rand((5000, 3), dtype=torch.float64, device="cuda", generator=gen) * (n * res * 1.1) - 0.05
    for seed in (0, 1):
        pc = torch.from_numpy(synth.two_box_points(3000, seed=seed, scale=n * res)).cuda()
        sdf_q, ans = stq.frame(pc, q)
        sdf_f, grad = stf.frame(pc)
        torch.cuda.synchronize()
        assert bool(torch.equal(sdf_q, sdf_f)) and stq.extrema() == stf.extrema()
        d2, g2, f2 = stf.query(q)
        assert bool(torch.equal(ans[2], f2))
        assert np.array_equal(ans[0].cpu().numpy(), d2.cpu().numpy(), equal_nan=True)
        assert np.array_equal(ans[1].cpu().numpy(), g2.cpu().numpy(), equal_nan=True)
        assert 0 < int((f2 & 1).sum()) < q.shape[0]                              # some points inside, some outside
    assert stq.frame(pc)[1] is None                                              # no query points: the field only
