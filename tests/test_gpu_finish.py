"""GPU: the fp64-free finish of the far-field x sweep (sdfgpu_finish.hpp, VERDICT r5 "next round" 1c) on the device's own
v_sqrt_f32 / v_rcp_f32: EVERY squared distance a 1024^3 grid can hold, bit for bit against the reference's arithmetic
float(sqrt((double)D) * resolution) (sdf_generation.hpp:254-265).  Tolerance: none (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("res", [1.0, 0.01, 0.013, 0.05, 0.02, 1.0 / 3.0, 0.1, 2.5, 1e-3, 123.456])
def test_fast_finish_is_the_reference_arithmetic_for_every_squared_distance(gpu, res):
    import torch
    n = 3 * 1024 * 1024 + 2                       # 0 .. nx^2 + ny^2 + nz^2 of a 1024^3 grid
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    D = np.arange(n, dtype=np.float64)
    want = (np.sqrt(D) * res).astype(np.float32)
    gpu.set_option("fast_finish", 1)
    slow = gpu.debug_finish_table(out.data_ptr(), n, res)
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.flatnonzero(got != want)[:10]
    assert 0 < slow < n // 2000, slow               # ~1e-4 of the values ask for the fp64 sequence: the fast path is what runs
    gpu.set_option("fast_finish", 0)                # the fp64 sequence for everything: same table
    assert gpu.debug_finish_table(out.data_ptr(), n, res) == 0
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    gpu.set_option("fast_finish", 1)


def test_fast_finish_stays_out_of_unsafe_ranges(gpu):
    """resolutions outside [2^-60, 2^60] (and non-positive ones) keep the fp64 sequence: same bits as numpy either way."""
    import torch
    n = 70000
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    for res in (1e-30, 1e25, 3e-19):
        slow = gpu.debug_finish_table(out.data_ptr(), n, res)
        want = (np.sqrt(np.arange(n, dtype=np.float64)) * res).astype(np.float32)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert slow == 0
