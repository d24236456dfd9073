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
