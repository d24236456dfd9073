"""GPU parity tests proper: the HIP path, called through the C ABI (libsdfgpu.so), against the
CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): occupancy/sign bit-exact; float distances within 1e-5 of the
reference CPU BuildDistanceField path on dense random occupancy -- in fact bit-identical there
because the final sqrt/multiply is done in fp64 like sdf_generation.hpp:254-265.  On sparse
scenes the reference's propagation over-estimates a few voxels; there the GPU must
equal the exact EDT bit for bit and every disagreement with the reference must be a reference
over-estimate."""
import json
import math
import os

import numpy as np
import pytest

import scenes
from oracle import oracle as O
from sdf_tools_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-5          # absolute, north_star


def _signed_dsq_to_stage(dsq, inf_mag):
    out = np.where(np.abs(dsq) == np.iinfo(np.int64).max, np.sign(dsq) * inf_mag, dsq)
    return out


def _report(name, got, want):
    bad = np.argwhere(got != want)
    msg = "%s: %d / %d voxels differ" % (name, len(bad), got.size)
    for idx in bad[:8]:
        idx = tuple(idx)
        msg += "\n   at %s got %r want %r" % (idx, got[idx], want[idx])
    return msg


def _exact_stage_fields(m):
    """Exact z-sweep and yz-sweep fields in the kernels' signed conventions."""
    nx, ny, nz = m.shape
    zf = np.empty(m.shape, np.int64)
    yz = np.empty(m.shape, np.int64)
    for x in range(nx):
        plane = m[x:x + 1]
        for y in range(ny):
            row = plane[:, y:y + 1]
            dfil, dfre = O.exact_edt(row, 1), O.exact_edt(row, 0)
            d = np.where(row != 0, dfre, dfil)
            zf[x, y] = np.where(d < 0, 32767, np.sqrt(np.maximum(d, 0)).round()).reshape(-1)
        dfil, dfre = O.exact_edt(plane, 1), O.exact_edt(plane, 0)
        d = np.where(plane != 0, dfre, dfil)
        yz[x] = np.where(d < 0, 1 << 30, d)[0]
    sign = np.where(m != 0, -1, 1)
    return (zf * sign).astype(np.int16), (yz * sign).astype(np.int32)


SHAPES_P = [
    ((16, 16, 16), 0.5), ((8, 12, 32), 0.5), ((24, 20, 17), 0.5), ((5, 7, 3), 0.5),
    ((33, 9, 64), 0.3), ((10, 70, 48), 0.5), ((3, 3, 130), 0.1), ((40, 40, 40), 0.02),
    ((20, 40, 1), 0.1), ((1, 1, 50), 0.2), ((7, 1, 9), 0.5), ((1, 1, 1), 0.0), ((2, 3, 256), 0.97),
    ((64, 64, 64), 0.5), ((70, 66, 80), 0.001),
    # nz in {64 .. 1024}: the wave-private z sweep, with row counts that leave its last wave step ragged
    ((5, 7, 64), 0.3), ((3, 11, 128), 0.05), ((9, 5, 256), 0.5), ((3, 5, 512), 0.01), ((2, 3, 1024), 0.002),
]


@pytest.mark.parametrize("shape,p", SHAPES_P)
def test_stages_and_field_match_exact_edt(gpu, shape, p):
    m = synth.bernoulli_mask(shape, p, seed=hash(shape) % 1000 + 1)
    want, want_ext, _ = O.exact_sdf(m, 1.0)
    sdf, ext = gpu.build(m, 1.0)                      # default path (dense kernel first where eligible)
    assert np.array_equal(sdf.view(np.uint32), want.view(np.uint32)), _report("sdf (default path)", sdf, want)
    assert ext == want_ext
    gpu.set_option("dense", 0)                        # general pipeline: check every stage
    try:
        sdf, ext = gpu.build(m, 1.0)
    finally:
        gpu.set_option("dense", 1)
    # canonicalised (singleton-free) dims are what the kernels ran on
    dims = [s for s in shape if s > 1]
    cshape = tuple([1] * (3 - len(dims)) + dims)
    zs = gpu.debug_zsweep(cshape)
    yzs = gpu.debug_yzsweep(cshape)
    ez, eyz = _exact_stage_fields(m.reshape(cshape))
    assert np.array_equal(zs, ez), _report("z sweep", zs, ez)
    assert np.array_equal(yzs, eyz), _report("yz sweep", yzs, eyz)
    assert np.array_equal(sdf.view(np.uint32), want.view(np.uint32)), _report("sdf", sdf, want)
    assert ext == want_ext
    assert np.array_equal(np.signbit(sdf), m != 0)            # occupancy / sign bit-exact


@pytest.mark.parametrize("n,seed,res", [(32, 1, 1.0), (64, 2, 1.0), (64, 3, 0.01), (96, 1, 0.01)])
def test_dense_random_occupancy_matches_reference_algorithm(gpu, n, seed, res):
    """BASELINE gating inputs: Bernoulli p = 0.5, every voxel within 1e-5 of the reference path."""
    m = synth.bernoulli_mask((n, n, n), 0.5, seed)
    sdf, ext = gpu.build(m, res)
    ref, ref_ext = O.reference_sdf(m, res)
    assert np.max(np.abs(sdf.astype(np.float64) - ref)) <= TOL
    assert np.array_equal(sdf.view(np.uint32), ref.view(np.uint32))      # in fact bit-identical
    assert ext == ref_ext


def test_256_cube_config_matches_reference_algorithm(gpu):
    """BASELINE configs[1]: 256^3 occupancy grid, fp32 distances, single MI355X."""
    m = synth.bernoulli_mask((256, 256, 256), 0.5, 1)
    sdf, ext = gpu.build(m, 1.0)
    ref, ref_ext = O.reference_sdf(m, 1.0)
    n_bad = int(np.sum(np.abs(sdf.astype(np.float64) - ref) > TOL))
    assert n_bad == 0
    assert np.array_equal(np.signbit(sdf), m != 0)
    assert ext == ref_ext


def test_reference_known_answer_scenes(gpu):
    ka = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))
    m, res = scenes.test_bindings_scene()
    sdf, ext = gpu.build(m, res)
    for key, want in ka["test_bindings"]["sdf"].items():
        assert sdf[tuple(int(t) for t in key.split(","))] == pytest.approx(want, abs=1e-7)
    assert sdf[6, 3, 0] > 3 * res and ext[1] == -0.05 and ext[0] == pytest.approx(2.06155, abs=1e-5)
    m, res = scenes.tutorial_scene()
    sdf, ext = gpu.build(m, res)
    for key, want in ka["tutorial"]["sdf"].items():
        assert sdf[tuple(int(t) for t in key.split(","))] == pytest.approx(want, abs=1e-7)
    assert ext == (pytest.approx(8.66025, abs=1e-5), -5.0)
    ref, ref_ext = O.reference_sdf(m, res)
    assert np.array_equal(sdf, ref) and ext == ref_ext
    for scene in (scenes.convex_segments_scene, scenes.estimate_distance_scene):
        m, res = scene()
        sdf, ext = gpu.build(m, res)
        ref, ref_ext = O.reference_sdf(m, res)
        assert np.array_equal(sdf, ref) and ext == ref_ext
    m, res = scenes.convex_segments_scene()
    sdf, ext = gpu.build(m, res, add_virtual_border=True)      # compute_convex_segments_test.cpp:84-95
    ex, ex_ext, _ = O.exact_sdf(m, res, True)
    assert np.array_equal(sdf, ex) and ext == ex_ext


def test_golden_oracle_vectors(gpu):
    z = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        shape = tuple(int(v) for v in z[name + "/shape"])
        m = np.unpackbits(z[name + "/mask"])[:int(np.prod(shape))].reshape(shape)
        res, vb = z[name + "/res_vb"]
        sdf, ext = gpu.build(m, float(res), bool(vb))
        ref = z[name + "/sdf"]
        if np.array_equal(sdf, ref):
            assert np.array_equal(np.array(ext), z[name + "/extrema"]), name
            continue
        # sparse vectors: the reference over-estimates a few voxels; GPU must equal the exact EDT
        ex, ex_ext, _ = O.exact_sdf(m, float(res), bool(vb))
        assert np.array_equal(sdf, ex), name
        bad = sdf != ref
        assert np.all(np.abs(ref[bad]) > np.abs(sdf[bad])), name
        assert bad.mean() < 2e-2, name
        print("%s: %d voxels where the reference propagation over-estimates" % (name, int(bad.sum())))


def test_uniform_grids_and_single_voxels(gpu):
    inf = math.inf
    s, ext = gpu.build(np.zeros((8, 8, 8), np.uint8), 1.0)
    assert np.all(np.isposinf(s)) and ext == (inf, inf)
    s, ext = gpu.build(np.ones((8, 8, 8), np.uint8), 1.0)
    assert np.all(np.isneginf(s)) and ext == (-inf, -inf)
    s, ext = gpu.build(np.zeros((16, 16, 16), np.uint8), 1.0, True)
    assert s.min() == 1.0 and s.max() == 8.0 and ext == (8.0, inf)
    s, ext = gpu.build(np.ones((16, 16, 16), np.uint8), 1.0, True)
    assert s.min() == -8.0 and s.max() == -1.0 and ext == (-inf, -8.0)
    for shape in ((33, 20, 48), (1, 1, 1), (9, 1, 1), (1, 40, 3)):
        for m in (scenes.single_voxel(shape), 1 - scenes.single_voxel(shape)):
            for vb in (False, True):
                sdf, ext = gpu.build(m, 0.1, vb)
                ex, ex_ext, _ = O.exact_sdf(m, 0.1, vb)
                assert np.array_equal(sdf.view(np.uint32), ex.view(np.uint32)), (shape, vb)
                assert ext == ex_ext, (shape, vb)


@pytest.mark.parametrize("shape,p", [((16, 12, 20), 0.5), ((64, 64, 64), 0.5), ((30, 31, 33), 0.1), ((20, 40, 1), 0.05)])
def test_virtual_border(gpu, shape, p):
    m = synth.bernoulli_mask(shape, p, 11)
    sdf, ext = gpu.build(m, 0.5, add_virtual_border=True)
    ex, ex_ext, _ = O.exact_sdf(m, 0.5, True)
    assert np.array_equal(sdf.view(np.uint32), ex.view(np.uint32)), _report("vb", sdf, ex)
    assert ext == ex_ext
    if p == 0.5:
        ref, ref_ext = O.reference_sdf(m, 0.5, True)          # sdf_generation.hpp:287-419 literally
        assert np.array_equal(sdf, ref) and ext == ref_ext


def test_sparse_scenes_equal_exact_and_reference_only_overestimates(gpu):
    """Stress inputs of SURVEY 8(d): report-and-prove instead of hiding the reference's inexactness."""
    cases = {
        "p=0.01": synth.bernoulli_mask((96, 96, 96), 0.01, 1),
        "p=0.99": synth.bernoulli_mask((96, 96, 96), 0.99, 1),
        "spheres96": synth.spheres_mask((96, 96, 96), 12, (3, 9), 0),
    }
    for name, m in cases.items():
        sdf, ext = gpu.build(m, 1.0)
        ex, ex_ext, dsq = O.exact_sdf(m, 1.0)
        assert np.array_equal(sdf.view(np.uint32), ex.view(np.uint32)), name
        assert ext == ex_ext, name
        ref, _, df, de = O.reference_sdf(m, 1.0, want_dsq=True)
        ref_d2 = np.where(m != 0, de, df)
        bad = np.abs(sdf.astype(np.float64) - ref) > TOL
        assert np.all(ref_d2[bad] > np.abs(dsq)[bad]), name      # every mismatch: reference too large
        assert bad.mean() < 2e-2, name
        if bad.any():
            assert np.abs(dsq)[bad].min() >= 8, name
        print("%s: %d voxels differ from the reference propagation (all reference over-estimates)" % (name, int(bad.sum())))


def test_collision_cell_fast_path(gpu):
    """sdfgpu_build_cells == collision_map.hpp:680-712 predicate + the mask path."""
    rng = np.random.RandomState(3)
    shape = (20, 18, 24)
    occ = rng.choice(np.array([0.0, 0.25, 0.5, 0.75, 1.0, -10000.0], np.float32), size=shape)
    cells = np.zeros(shape + (2,), np.float32)
    cells[..., 0] = occ
    cells[..., 1] = rng.rand(*shape)          # garbage in the component field must be ignored
    for unknown in (False, True):
        mask = O.classify_cells(cells, unknown)
        want, want_ext = O.reference_sdf(mask, 0.05)
        got, ext = gpu.build_cells(cells, shape, 8, 0, unknown, 0.05)
        assert np.array_equal(got, want) and ext == want_ext
        assert np.array_equal(np.signbit(got), mask != 0)
    # occupancy at a non-zero offset in a wider record
    wide = np.zeros(shape + (3,), np.float32)
    wide[..., 1] = occ
    got, _ = gpu.build_cells(wide, shape, 12, 4, False, 0.05)
    assert np.array_equal(got, O.reference_sdf(O.classify_cells(cells, False), 0.05)[0])


def test_error_codes(gpu):
    from sdf_tools_amd.capi import SdfGpuError
    with pytest.raises(SdfGpuError) as ei:
        gpu.build(np.zeros((0, 4, 4), np.uint8))
    assert ei.value.code == -1
    with pytest.raises(SdfGpuError) as ei:
        gpu.build_cells(np.zeros((4, 4, 4, 6), np.uint8), (4, 4, 4), 6, 0)
    assert ei.value.code == -1
    # the context stays usable after an error
    s, _ = gpu.build(np.zeros((4, 4, 4), np.uint8))
    assert np.all(np.isposinf(s))


def test_device_resident_entry_points_through_the_abi(gpu):
    """Round 4: sdfgpu_device_malloc / sdfgpu_build_to_device / sdfgpu_build_cells_to_device (host input, field left in HBM) and
    sdfgpu_query_points (host points -> host answers against that field): same field and extrema as sdfgpu_build, same
    answers as the device-pointer query on it; error codes for null / bad arguments."""
    import ctypes
    import torch
    from sdf_tools_amd.capi import SdfGpuError
    shape, res = (40, 33, 48), 0.05
    m = synth.bernoulli_mask(shape, 0.1, 11)
    want, want_ext = gpu.build(m, res, True)
    n = int(np.prod(shape))
    d_field = gpu.device_malloc(n * 4)
    try:
        ext = gpu.build_to_device(m, d_field, res, True)
        assert ext == want_ext
        back = np.empty(shape, np.float32)
        gpu._check(gpu._lib.sdfgpu_copy_to_host(gpu._h, back.ctypes.data, ctypes.c_void_p(d_field), n * 4, None))
        assert np.array_equal(back.view(np.uint32), want.view(np.uint32))
        # cells form: 8-byte COLLISION_CELL records, unknown (0.5) filled on request
        cells = np.zeros(shape + (2,), np.float32)
        cells[..., 0] = np.where(m != 0, 1.0, 0.0)
        cells[3, 4, 5, 0] = 0.5
        ext_c = (ctypes.c_double * 2)()
        c = np.ascontiguousarray(cells)
        gpu._check(gpu._lib.sdfgpu_build_cells_to_device(gpu._h, c.ctypes.data, 8, 0, 1, *shape, res, 0, ctypes.c_void_p(d_field),
                                                         ctypes.byref(ext_c, 0), ctypes.byref(ext_c, 8)))
        want_c, want_c_ext = gpu.build_cells(cells, shape, 8, 0, True, res, False)
        gpu._check(gpu._lib.sdfgpu_copy_to_host(gpu._h, back.ctypes.data, ctypes.c_void_p(d_field), n * 4, None))
        assert np.array_equal(back.view(np.uint32), want_c.view(np.uint32)) and (ext_c[0], ext_c[1]) == want_c_ext
        # host-point queries == device-pointer queries on the same field (a rotated, shifted frame)
        rng = np.random.default_rng(1)
        pts = rng.uniform(-0.2, 2.6, size=(3000, 3))
        w2g = [0, 1, 0, 0.1, -1, 0, 0, 2.0, 0, 0, 1, -0.05]
        rot = [0, -1, 0, 1, 0, 0, 0, 0, 1]
        dist, grad, flags = gpu.query_points(d_field, shape, res, pts, world_to_grid=w2g, rotation=rot, oob_value=7.5,
                                             enable_edge_gradients=True)
        tp = torch.from_numpy(pts).cuda()
        td, tg, tf = (torch.empty(3000, dtype=torch.float64, device="cuda"), torch.empty((3000, 3), dtype=torch.float64, device="cuda"),
                      torch.empty(3000, dtype=torch.uint8, device="cuda"))
        gpu.query_points_device(d_field, shape, res, tp.data_ptr(), 3000, td.data_ptr(), tg.data_ptr(), tf.data_ptr(),
                                world_to_grid=w2g, rotation=rot, oob_value=7.5, enable_edge_gradients=True)
        torch.cuda.synchronize()
        assert np.array_equal(flags, tf.cpu().numpy()) and 0 < int((flags & 1).sum()) < 3000
        assert np.array_equal(dist, td.cpu().numpy()) and np.array_equal(grad, tg.cpu().numpy(), equal_nan=True)
        assert np.all(dist[(flags & 1) == 0] == 7.5)
        # zero points / no outputs are no-ops; bad arguments are INVALID_ARGUMENT
        d0, g0, f0 = gpu.query_points(d_field, shape, res, np.zeros((0, 3)))
        assert d0.shape == (0,)
        with pytest.raises(SdfGpuError) as ei:
            gpu.query_points(0, shape, res, pts)
        assert ei.value.code == -1
        with pytest.raises(SdfGpuError) as ei:
            gpu.build_to_device(m, 0, res)
        assert ei.value.code == -1
        with pytest.raises(SdfGpuError) as ei:
            gpu.query_points(d_field, shape, -1.0, pts)
        assert ei.value.code == -1
    finally:
        gpu.device_free(d_field)
    gpu.device_free(0)                                          # freeing NULL is fine


def test_tuning_does_not_change_results(gpu):
    m = synth.bernoulli_mask((50, 45, 64), 0.2, 5)
    base, ext = gpu.build(m, 1.0)
    gpu.set_option("dense", 0)
    try:
        for t in (7, 9, 16, 100):
            gpu.set_tuning(t, t)
            s, e = gpu.build(m, 1.0)
            assert np.array_equal(s, base) and e == ext, t
    finally:
        gpu.set_tuning(0, 0)
        gpu.set_option("dense", 1)


@pytest.mark.parametrize("nz", [64, 128, 256, 512, 1024])
def test_z_sweep_forms_agree(gpu, nz):
    """k_sweep_z_wave16 (whole rows per wave) against k_sweep_z_vec16 (workgroup form): same z field, word for word, on rows
    of one class, rows with a single voxel of the other class at either end, and random rows."""
    rng = np.random.default_rng(nz)
    m = (rng.random((7, 13, nz)) < 0.03).astype(np.uint8)
    m[0, 0] = 0; m[0, 1] = 1                                  # rows of one class
    m[0, 2] = 0; m[0, 2, 0] = 1                               # one filled voxel at the low end
    m[0, 3] = 0; m[0, 3, nz - 1] = 1                          # ... at the high end
    m[0, 4] = 1; m[0, 4, nz // 2] = 0                         # one free voxel in a filled row
    m[1] = (rng.random((13, nz)) < 0.5)
    gpu.set_option("dense", 0)
    try:
        sdf_a, ext_a = gpu.build(m, 1.0)
        za = gpu.debug_zsweep(m.shape).copy()
        gpu.set_option("z_wave", 0)
        sdf_b, ext_b = gpu.build(m, 1.0)
        zb = gpu.debug_zsweep(m.shape).copy()
    finally:
        gpu.set_option("z_wave", 1)
        gpu.set_option("dense", 1)
    ez, _ = _exact_stage_fields(m)
    assert np.array_equal(za, ez), _report("z sweep (wave form)", za, ez)
    assert np.array_equal(zb, ez), _report("z sweep (workgroup form)", zb, ez)
    assert np.array_equal(sdf_a, sdf_b) and ext_a == ext_b


def test_host_copies_chunked_path(gpu):
    """sdfgpu_copy_from_host / sdfgpu_copy_to_host (the copies behind the host-buffer entry points): sizes that span
    several pinned staging chunks with a ragged tail, into untouched destination memory; and the host entry points on a
    grid large enough to take that path give the bytes of the device-resident build."""
    import torch
    rng = np.random.RandomState(11)
    for nbytes in (1 << 20, (32 << 20) + 4096, (100 << 20) + 12345):
        src = rng.randint(0, 256, size=nbytes, dtype=np.uint8)
        d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        for rep in range(4):                      # (round 5: a fill race between neighbouring chunks showed once in ~10 runs)
            d.zero_()
            gpu.copy_from_host(d.data_ptr(), src)
            assert np.array_equal(d.cpu().numpy(), src), (nbytes, rep)
        back = gpu.copy_to_host(np.empty(nbytes, np.uint8), d.data_ptr())
        assert np.array_equal(back, src), nbytes
    shape = (130, 256, 256)                       # 34 MB of output, 68 MB of cells
    m = synth.bernoulli_mask(shape, 0.3, 4)
    dm = torch.from_numpy(m).cuda()
    dout = torch.empty(shape, dtype=torch.float32, device="cuda")
    gpu.build_device(dm.data_ptr(), shape, dout.data_ptr(), 0.02, False, torch.cuda.current_stream().cuda_stream)
    want_ext = gpu.get_extrema()
    want = dout.cpu().numpy()
    got, ext = gpu.build(m, 0.02)
    assert np.array_equal(got, want) and ext == want_ext
    cells = np.zeros(shape + (2,), np.float32)
    cells[..., 0] = m
    got, ext = gpu.build_cells(cells, shape, 8, 0, False, 0.02)
    assert np.array_equal(got, want) and ext == want_ext
    g = gpu.gradient(got, 0.02)
    assert g.shape == shape + (3,) and np.isfinite(g).all()


@pytest.mark.parametrize("shape", [(20, 18, 24), (7, 5, 3), (1, 1, 77), (33, 17, 31), (64, 64, 64), (16, 24, 100)])
def test_host_side_classification_matches_the_device_classifier(gpu, shape):
    """Round 5 (VERDICT r4 "next round" 3): the host-buffer entry points classify the caller's mask / cells into one bit per
    voxel on the host (SSE2 for masks and 8-byte COLLISION_CELL records, a scalar loop for other strides) and upload 1/8 B per
    voxel.  Forced here on grids far below the size at which it switches itself on (option host_pack = 2), with voxel counts
    that are no multiple of 8 / 16 / 32: fields and extrema must equal the upload-and-classify-on-device path (host_pack = 0)
    byte for byte, for both predicates, NaN / negative / 0.5 occupancies, a garbage component field and a wide record."""
    rng = np.random.RandomState(hash(shape) % 1000)
    occ = rng.choice(np.array([0.0, 0.25, 0.5, 0.75, 1.0, -10000.0, np.nan, 0.50000006], np.float32), size=shape)
    cells = np.zeros(shape + (2,), np.float32)
    cells[..., 0] = occ
    cells[..., 1] = rng.rand(*shape)
    swapped = np.ascontiguousarray(cells[..., ::-1])                      # occupancy at offset 4 of the 8-byte record
    wide = np.zeros(shape + (3,), np.float32)
    wide[..., 1] = occ
    mask = (rng.rand(*shape) < 0.3).astype(np.uint8) * rng.randint(1, 256, size=shape).astype(np.uint8)   # any non-zero byte = filled
    try:
        res = {}
        for mode in (0, 2):
            gpu.set_option("host_pack", mode)
            res[mode] = [gpu.build(mask, 0.05), gpu.build(mask, 0.05, True)]
            for unknown in (False, True):
                res[mode].append(gpu.build_cells(cells, shape, 8, 0, unknown, 0.05))
                res[mode].append(gpu.build_cells(swapped, shape, 8, 4, unknown, 0.05))
                res[mode].append(gpu.build_cells(wide, shape, 12, 4, unknown, 0.05, True))
    finally:
        gpu.set_option("host_pack", 1)
    for (a, ea), (b, eb) in zip(res[0], res[2]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ea == eb
    want, want_ext = O.reference_sdf(O.classify_cells(cells, True), 0.05)
    # (and against the oracle's predicate + algorithm: entry 5 = cells, unknown_is_filled, no border)
    assert np.array_equal(res[2][5][0], want) and res[2][5][1] == want_ext
