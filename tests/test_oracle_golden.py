"""Pins the oracle (oracle/sdf_oracle.c) to the reference's known answers.

The expected values are the assertions of the reference's own test
(test/test_bindings.py:24-33) and the known-answer table of SURVEY.md section 4.

The reference-pinning tests run in BOTH tiers ("next round" 3a): once under -m "not gpu" and once under
-m gpu, where the oracle is the library rebuilt / loaded on the GPU box -- the checker every parity test of that run
leans on is pinned in the same run.  (No GPU is touched by them: the `tier` parameter only carries the marker.)"""
import json
import math
import os

import numpy as np
import pytest

import scenes
from oracle import oracle as O
from sdf_tools_amd import synth

both_tiers = pytest.mark.parametrize("tier", ["cpu", pytest.param("gpu_box", marks=pytest.mark.gpu)])

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))


def _f(v):
    return float(v) if not isinstance(v, str) else float(v.replace("inf", "inf"))


def _check_values(sdf, spec):
    for key, want in spec.items():
        idx = tuple(int(t) for t in key.split(","))
        assert sdf[idx] == pytest.approx(want, abs=1e-7), key


@pytest.mark.parametrize("impl", ["reference", "exact"])
@both_tiers
def test_test_bindings_scene(impl, tier):
    m, res = scenes.test_bindings_scene()
    sdf, ext = O.reference_sdf(m, res) if impl == "reference" else O.exact_sdf(m, res)[:2]
    ka = KA["test_bindings"]
    # test_bindings.py:24-29 (numpy view is [y, x]; here indices are [x, y, z])
    _check_values(sdf, ka["sdf"])
    assert sdf[6, 3, 0] > 3 * res            # test_bindings.py:29
    assert sdf.shape == (20, 40, 1)
    assert ext[0] == pytest.approx(ka["extrema"][0], abs=1e-5)
    assert ext[1] == pytest.approx(ka["extrema"][1], abs=1e-12)


@pytest.mark.parametrize("impl", ["reference", "exact"])
@both_tiers
def test_tutorial_scene(impl, tier):
    m, res = scenes.tutorial_scene()
    sdf, ext = O.reference_sdf(m, res) if impl == "reference" else O.exact_sdf(m, res)[:2]
    _check_values(sdf, KA["tutorial"]["sdf"])
    assert ext[0] == pytest.approx(KA["tutorial"]["extrema"][0], abs=1e-5)
    assert ext[1] == pytest.approx(KA["tutorial"]["extrema"][1], abs=1e-12)


@both_tiers
def test_convex_segments_scene(tier):
    m, res = scenes.convex_segments_scene()
    sdf, ext = O.reference_sdf(m, res)
    assert ext[0] == pytest.approx(18.0, abs=1e-12)
    assert ext[1] == pytest.approx(-14.1421, abs=1e-4)
    ex, eext, _ = O.exact_sdf(m, res)
    assert np.array_equal(sdf, ex)            # 0 mismatches vs exact on this scene
    assert eext == ext


@both_tiers
def test_all_free_all_filled(tier):
    s, ext = O.reference_sdf(np.zeros((8, 8, 8), np.uint8), 1.0)
    assert np.all(np.isposinf(s)) and ext == (math.inf, math.inf)
    s, ext = O.reference_sdf(np.ones((8, 8, 8), np.uint8), 1.0)
    assert np.all(np.isneginf(s)) and ext == (-math.inf, -math.inf)
    for fill, want in ((0, (math.inf, math.inf)), (1, (-math.inf, -math.inf))):
        s, ext, _ = O.exact_sdf(np.full((8, 8, 8), fill, np.uint8), 1.0)
        assert ext == want


@both_tiers
def test_virtual_border_uniform_grids(tier):
    s, ext = O.reference_sdf(np.zeros((16, 16, 16), np.uint8), 1.0, True)
    assert s.min() == 1.0 and s.max() == 8.0 and ext == (8.0, math.inf)
    s, ext = O.reference_sdf(np.ones((16, 16, 16), np.uint8), 1.0, True)
    assert s.min() == -8.0 and s.max() == -1.0 and ext == (-math.inf, -8.0)
    for fill in (0, 1):
        m = np.full((16, 16, 16), fill, np.uint8)
        a, ea = O.reference_sdf(m, 1.0, True)
        b, eb, _ = O.exact_sdf(m, 1.0, True)
        assert np.array_equal(a, b) and ea == eb


def test_exact_edt_against_brute_force():
    rng = np.random.RandomState(0)
    for shape in ((9, 7, 11), (1, 13, 6), (5, 1, 1), (12, 12, 1)):
        for p in (0.5, 0.05, 0.95):
            m = (rng.rand(*shape) < p).astype(np.uint8)
            for sv in (0, 1):
                assert np.array_equal(O.brute_edt(m, sv), O.exact_edt(m, sv))


@both_tiers
def test_reference_is_exact_on_dense_random_occupancy(tier):
    # at p = 0.5 the propagation never errs (true d^2 < 8)
    for n, seed in ((32, 1), (48, 2)):
        m = synth.bernoulli_mask((n, n, n), 0.5, seed)
        a, ea = O.reference_sdf(m, 1.0)
        b, eb, _ = O.exact_sdf(m, 1.0)
        assert np.array_equal(a, b) and ea == eb


def test_reference_overestimates_only_on_sparse_scenes():
    # the propagation never under-estimates; errors need true d^2 >= 8
    m = synth.bernoulli_mask((48, 48, 48), 0.02, 1)
    a, _, df, de = O.reference_sdf(m, 1.0, want_dsq=True)
    b, _, dsq = O.exact_sdf(m, 1.0)
    ref_d2 = np.where(m != 0, de, df)
    ex_d2 = np.abs(dsq)
    assert np.all(ref_d2 >= ex_d2)
    bad = ref_d2 != ex_d2
    if bad.any():
        assert ex_d2[bad].min() >= 8
    assert np.all(np.sign(a) == np.sign(b))


def test_virtual_border_matches_clamped_exact_on_dense_grid():
    m = synth.bernoulli_mask((14, 10, 12), 0.5, 7)
    a, ea = O.reference_sdf(m, 0.5, True)
    b, eb, _ = O.exact_sdf(m, 0.5, True)
    assert np.array_equal(a, b) and ea == eb


@both_tiers
def test_oracle_vectors_are_stable(tier):
    z = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) == 5
    for name in names:
        shape = tuple(int(v) for v in z[name + "/shape"])
        m = np.unpackbits(z[name + "/mask"])[:int(np.prod(shape))].reshape(shape)
        res, vb = z[name + "/res_vb"]
        sdf, ext = O.reference_sdf(m, float(res), bool(vb))
        assert np.array_equal(sdf, z[name + "/sdf"]), name
        assert np.array_equal(np.array(ext), z[name + "/extrema"]), name


@both_tiers
def test_classify_cells_predicate(tier):
    # collision_map.hpp:680-712
    occ = np.array([0.0, 0.4999, 0.5, 0.5001, 1.0, -10000.0, np.nan], np.float32)
    cells = np.zeros((occ.size, 2), np.float32)
    cells[:, 0] = occ
    cells = cells.reshape(1, 1, occ.size, 2)
    assert O.classify_cells(cells, False).ravel().tolist() == [0, 0, 0, 1, 1, 0, 0]
    assert O.classify_cells(cells, True).ravel().tolist() == [0, 0, 1, 1, 1, 0, 0]
