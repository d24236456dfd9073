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


def test_sparse_256_far_scans(gpu):
    """Sparse 256^3 (p = 1e-4): distances of tens of voxels, the outward-scan path dominates."""
    m = synth.bernoulli_mask((256, 256, 256), 1e-4, 3)
    sdf, ext = gpu.build(m, 1.0)
    ex, ex_ext, _ = O.exact_sdf(m, 1.0)
    assert np.array_equal(sdf, ex) and ext == ex_ext
    sdf, ext = gpu.build(1 - m, 1.0, add_virtual_border=True)
    ex, ex_ext, _ = O.exact_sdf(1 - m, 1.0, True)
    assert np.array_equal(sdf, ex) and ext == ex_ext
