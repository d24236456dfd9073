"""CPU: host logic that needs no GPU -- the C-ABI library loads and exports every declared
symbol, fails loudly without a device, and the synthetic-input generators are deterministic."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

from sdf_tools_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sdfgpu.h")).read()
    declared = set(re.findall(r"\b(sdfgpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sdfgpu_context", "sdfgpu_status", "sdfgpu_handle"}
    assert declared == set(capi.EXPORTS)
    lib = capi.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.sdfgpu_version()


def test_library_contains_gfx950_code_object():
    data = open(os.path.join(ROOT, "sdf_tools_amd", "libsdfgpu.so"), "rb").read()
    assert b"gfx950" in data and b"k_sweep_march" in data


def test_no_device_fails_loudly_without_fallback():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.SdfGpuError) as ei:
        capi.SdfGpu(0)
    assert ei.value.code == -4 and "no CPU fallback" in str(ei.value)


def test_extrema_from_dsq_semantics():
    # sdf_generation.hpp:246-269 (and :416-418 for the virtual-border pair)
    inf = math.inf
    assert capi.extrema_from_dsq(3, 2, 1.0) == (math.sqrt(3.0), -math.sqrt(2.0))
    assert capi.extrema_from_dsq(1 << 30, 0, 1.0) == (inf, inf)        # all free
    assert capi.extrema_from_dsq(0, 1 << 30, 1.0) == (-inf, -inf)      # all filled
    assert capi.extrema_from_dsq(64, 0, 1.0) == (8.0, inf)             # vb, all free 16^3
    assert capi.extrema_from_dsq(0, 64, 1.0) == (-inf, -8.0)           # vb, all filled 16^3
    mx, mn = capi.extrema_from_dsq(12, 27, 0.01)
    assert mx == math.sqrt(12.0) * 0.01 and mn == 0.0 - math.sqrt(27.0) * 0.01


def test_bernoulli_mask_is_counter_based():
    full = synth.bernoulli_mask((12, 10, 16), 0.5, 3)
    slab = synth.bernoulli_mask((12, 10, 16), 0.5, 3, x_range=(4, 9))
    assert np.array_equal(full[4:9], slab)
    assert abs(full.mean() - 0.5) < 0.05
    assert not np.array_equal(full, synth.bernoulli_mask((12, 10, 16), 0.5, 4))
    sparse = synth.bernoulli_mask((32, 32, 32), 0.01, 1)
    assert 0.003 < sparse.mean() < 0.03


def test_bernoulli_mask_torch_matches_numpy():
    torch = pytest.importorskip("torch")
    a = synth.bernoulli_mask((7, 9, 16), 0.5, 5, x_range=(2, 6))
    b = synth.bernoulli_mask_torch((7, 9, 16), 0.5, 5, x_range=(2, 6), device="cpu").numpy()
    assert np.array_equal(a, b)
    a = synth.bernoulli_mask((5, 5, 8), 0.93, 9)
    b = synth.bernoulli_mask_torch((5, 5, 8), 0.93, 9, device="cpu").numpy()
    assert np.array_equal(a, b)


def test_multi_gpu_library_exports_every_declared_symbol_and_refuses_without_gpu():
    """include/sdfgpu_multi.h (the sdfgpu_init(n_gpus) row of SURVEY 8(b)): libsdfgpu_multi.so loads next to
    libsdfgpu.so + RCCL, exports what the header declares, and has no CPU fallback."""
    hdr = open(os.path.join(ROOT, "include", "sdfgpu_multi.h")).read()
    declared = set(re.findall(r"\b(sdfgpu_multi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sdfgpu_multi_context", "sdfgpu_multi_handle"}
    assert declared == set(capi.MULTI_EXPORTS)
    lib = capi.load_multi_library()
    for name in declared:
        assert hasattr(lib, name), name
    data = open(os.path.join(ROOT, "sdf_tools_amd", "libsdfgpu_multi.so"), "rb").read()
    assert b"librccl.so" in data                      # inter-GPU traffic goes through RCCL
    if capi.device_count() == 0:
        with pytest.raises(capi.SdfGpuError) as ei:
            capi.MultiSdfGpu(2)
        assert ei.value.code == -4 and "no CPU fallback" in str(ei.value)


def test_far_field_schedule_model_is_exact():
    """tools/envelope_dc_model.py restates k_envelope_dc's schedule (centred 32-bit keys, levels A / B / C, scans in aligned
    pairs that read one candidate before and behind the range, distance-bound clipping with a tile-level bound, exhaustive
    chunk phase) line by line on the CPU; it must equal a brute-force min-plus evaluation on random lines incl. ties, runs,
    flat far-field lines, lengths that are not multiples of 8 / 64 and empty lines, with tile-level spans looser than the line's."""
    import importlib.util
    import random
    spec = importlib.util.spec_from_file_location(
        "envelope_dc_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "envelope_dc_model.py"))
    model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(model)
    rng = random.Random(7)
    for trial in range(400):
        L = rng.choice([1, 2, 7, 8, 9, 17, 40, 64, 100, 127, 128, 257, 512])
        dens = rng.choice([0.0, 0.01, 0.05, 0.3, 1.0])
        kind = rng.choice(["rand", "ties", "runs", "smooth"])
        c, hh = rng.randrange(-50, L + 50), rng.randrange(0, 300)
        F = []
        for q in range(L):
            if kind == "smooth":
                F.append(hh * hh + (q - c) ** 2 if rng.random() < max(dens, 0.3) else model.INF)
            elif kind == "ties":
                F.append(rng.choice([0, 1, 4]) if rng.random() < dens else model.INF)
            elif kind == "runs":
                F.append(0 if (q // 7) % 3 == 0 and dens > 0 else model.INF)
            else:
                F.append(rng.randrange(0, 200001) if rng.random() < dens else model.INF)
        finf = max([v for v in F if v < model.INF] + [0]) + (L - 1) ** 2 + 1
        got, _ = model.dc_line(F, finf)
        assert got == model.brute(F), (trial, L, kind)
        model.check_line(rng, F)                                    # again with a looser span / bound and a forced level-A form


def test_wide_ball_level_tables_and_plane_encoding():
    """The arithmetic KD3 (sdf_tools_amd/csrc/sdfgpu_dense3.hpp) rests on, restated: (1) the levels of the 7^3 cube are
    exactly the d^2 <= 14 that are sums of three squares, and every lattice point with such a d^2 lies inside the cube
    (a level is COMPLETE: no offset of that length is missing); 15 is not a sum of three squares and 16 = 4^2 leaves the
    cube, so 14 is where completeness ends; (2) ball3_level's closed form enumerates them in order; (3) the "not found
    at level l" sets are nested, and bit k of their count is the parity of every 2^(k+1)-th of them starting at 2^k - 1 --
    the four bit-planes the kernel stores."""
    import re
    R = 3
    sums = sorted({x * x + y * y + z * z for x in range(-6, 7) for y in range(-6, 7) for z in range(-6, 7)} - {0})
    levels = [d for d in sums if d <= 14]
    assert levels == [1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14] and 15 not in sums
    for d2 in levels:                                         # complete: every offset of that length is inside the cube
        pts = [(x, y, z) for x in range(-6, 7) for y in range(-6, 7) for z in range(-6, 7) if x * x + y * y + z * z == d2]
        assert all(max(abs(c) for c in p) <= R for p in pts)
    assert any(max(abs(c) for c in (4, 0, 0)) > R for _ in [0])   # d^2 = 16 is the first level the cube would miss

    def ball3_level(d2):                                      # the header's closed form
        return d2 - 1 if 1 <= d2 <= 6 else d2 - 2 if 8 <= d2 <= 14 else -1
    assert [ball3_level(d) for d in levels] == list(range(13))
    assert all(ball3_level(d) == -1 for d in (0, 7, 15, 16, 27))
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                            "sdf_tools_amd", "csrc", "sdfgpu_dense3.hpp")).read()
    m = re.search(r"kBall3D2\[kBall3Levels\] = \{([^}]*)\}", src)
    assert m and [int(v) for v in m.group(1).split(",")] == levels   # the kernel's table is this list
    NL = 13
    for c in range(NL + 1):                                   # c = level index of a voxel = number of levels it was NOT found at
        U = [1 if l < c else 0 for l in range(NL)]            # nested sets
        planes = [0, 0, 0, 0]
        planes[0] = sum(U[l] for l in range(0, NL)) & 1       # the XOR chains written out in the kernel
        planes[1] = sum(U[l] for l in (1, 3, 5, 7, 9, 11)) & 1
        planes[2] = sum(U[l] for l in (3, 7, 11)) & 1
        planes[3] = U[7]
        assert planes[0] | planes[1] << 1 | planes[2] << 2 | planes[3] << 3 == c
