"""GPU: red zones (round 6, VERDICT r5 "next round" 2).  With option "redzone" (or SDFGPU_REDZONE=1 at sdfgpu_create) every device
allocation of the library carries canaries and every entry point ends with a check: a store outside a buffer fails the call
that made it and names the buffer.  The whole `-m gpu` suite and the fuzz are also run with SDFGPU_REDZONE=1 (tools/r06_redzone.sh,
log under profiles/)."""
import numpy as np
import pytest

from oracle import oracle as O
from sdf_tools_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def rz():
    ctx = capi.SdfGpu(0)
    ctx.set_option("redzone", 1)
    yield ctx
    ctx.close()


def test_a_store_behind_a_buffer_fails_the_call_and_names_the_buffer(rz):
    shape, res = (40, 48, 64), 0.01
    n = int(np.prod(shape))
    m = synth.bernoulli_mask(shape, 0.5, 1)
    d_in = rz.device_malloc(n)
    rz.copy_from_host(d_in, m)
    short = rz.device_malloc(n * 4 - 256)               # 64 floats too small: the build's last stores land in the canary
    with pytest.raises(capi.SdfGpuError) as e:
        rz.build_device(d_in, shape, short, res, False, 0)
    assert e.value.code == -6 and "sdfgpu_device_malloc" in str(e.value) and "BEHIND its end" in str(e.value) and "256 canary bytes" in str(e.value)
    rz.redzone_check()                                  # (the zone was repaired: the next look is clean)
    good = rz.device_malloc(n * 4)
    for opts in ({}, {"dense": 0}, {"dense": 0, "envelope_mode": 1}):
        rz.set_option("policy_reset", 1)
        for k, v in opts.items():
            rz.set_option(k, v)
        rz.build_device(d_in, shape, good, res, False, 0)
        got = rz.copy_to_host(np.empty(shape, np.float32), good)
        ex, ex_ext, _ = O.exact_sdf(m, res)
        assert np.array_equal(got, ex) and rz.get_extrema() == ex_ext
        rz.set_option("dense", 1)
        rz.set_option("envelope_mode", 0)
    for p in (d_in, short, good):
        rz.device_free(p)


@pytest.mark.parametrize("shape", [(25, 20, 15), (9, 13, 64), (64, 64, 64), (21, 5, 512), (520, 3, 16), (2, 777, 32), (33, 17, 96)], ids=lambda s: "x".join(map(str, s)))
def test_every_tier_and_entry_point_is_clean_under_red_zones(rz, shape):
    """the reference's own demo grid (25 x 20 x 15, scripts/3d_sdf_demo_rviz.py:107-111 -- the shape whose x sweep stored past the field
    for three rounds), shapes 1 .. 6 mod 8, long lines, dense shapes: every tier forced in turn, with and without the virtual border,
    host builds, bits in, tagged cells, gradient, queries -- exact results and no canary touched."""
    res = 0.02
    n = int(np.prod(shape))
    for p in (0.5, 0.03, 0.002):
        m = synth.bernoulli_mask(shape, p, 4)
        for vb in (False, True):
            ex, ex_ext, _ = O.exact_sdf(m, res, vb)
            d_in, d_out = rz.device_malloc(n), rz.device_malloc(n * 4)
            rz.copy_from_host(d_in, m)
            for opts in ({}, {"dense": 0}, {"dense": 0, "envelope_mode": 1}, {"dense": 0, "plane16": 0}, {"expect_dense": 1}, {"far_predict": 2, "dense": 0},
                         {"dense3_mode": 1}, {"fixup_mode": 1}):
                rz.set_option("policy_reset", 1)
                for k, v in opts.items():
                    rz.set_option(k, v)
                rz.build_device(d_in, shape, d_out, res, vb, 0)
                ext = rz.get_extrema()
                got = rz.copy_to_host(np.empty(shape, np.float32), d_out)
                for k in opts:
                    rz.set_option(k, {"dense": 1, "envelope_mode": 0, "plane16": 1, "expect_dense": 0, "far_predict": 1, "dense3_mode": 0, "fixup_mode": 0}[k])
                assert np.array_equal(got, ex) and ext == ex_ext, (p, vb, opts)
            rz.device_free(d_in)
            got, ext = rz.build(m, res, vb)
            assert np.array_equal(got, ex) and ext == ex_ext
            got, ext = rz.build_bits(capi.pack_bits_host(m), shape, res, vb)
            assert np.array_equal(got, ex) and ext == ex_ext
            if not vb:
                d_g = rz.device_malloc(n * 12)
                rz.gradient_device(d_out, shape, d_g, res, True, False, 0)
                pts = np.random.default_rng(0).random((300, 3)) * (np.asarray(shape) * res)
                rz.query_points(d_out, shape, res, pts, enable_edge_gradients=True)
                rz.device_free(d_g)
            rz.device_free(d_out)
    cells = np.zeros(shape + (4,), np.float32)
    cells[..., 0] = synth.bernoulli_mask(shape, 0.3, 9)
    cells.view(np.uint32)[..., 2] = 3
    got, ext = rz.build_tagged_cells(cells, shape, 1, (), False, res)
    ex, ex_ext, _ = O.exact_sdf(cells[..., 0] > 0.5, res)
    assert np.array_equal(got, ex) and ext == ex_ext


def test_multi_rank_builds_are_clean_under_red_zones(monkeypatch):
    """libsdfgpu_multi with 1 .. 4 logical ranks (contexts created with SDFGPU_REDZONE=1): slabs, bit planes, halos, the re-partition."""
    monkeypatch.setenv("SDFGPU_REDZONE", "1")
    shape, res = (37, 24, 64), 0.01
    for ranks in (1, 2, 3, 4):
        mg = capi.MultiSdfGpu(ranks, [0] * ranks)
        mg.set_option("dense_retry", 0)
        for p, vb in ((0.5, False), (0.01, False), (0.002, True), (0.3, True)):
            m = synth.bernoulli_mask(shape, p, 6)
            ex, ex_ext, _ = O.exact_sdf(m, res, vb)
            for halo in (1, 8):
                mg.set_option("halo", halo)
                got, ext = mg.build(m, res, vb)
                assert np.array_equal(got, ex) and ext == ex_ext, (ranks, p, vb, halo)
        mg.close()
