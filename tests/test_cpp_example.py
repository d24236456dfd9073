"""The C++ side of the drop-in boundary: examples/tutorial.cpp is written like reference client code
(src/sdf_tools_tutorial.cpp) against include/sdf_tools/*.hpp and links libsdfgpu.so.
CPU: it must compile, link and fail loudly without a GPU.  GPU: it must reproduce the pinned values."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "tutorial_example")


def _build():
    from sdf_tools_amd import build as b
    b.build_libsdfgpu()
    src = os.path.join(ROOT, "examples", "tutorial.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src,
                               "-o", EXE, "-L", os.path.join(ROOT, "sdf_tools_amd"), "-lsdfgpu",
                               "-Wl,-rpath," + os.path.join(ROOT, "sdf_tools_amd"), "-lz"])
    return EXE


def test_cpp_client_compiles_links_and_refuses_without_gpu():
    from sdf_tools_amd import capi
    exe = _build()
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_cpp_client_reproduces_tutorial_scene():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tutorial scene OK" in r.stdout


C_EXE = os.path.join(ROOT, "examples", "c_client_example")


def _build_c():
    """examples/c_client.c with a C compiler: the two ABI headers must be plain C (what cgo / JNI / ctypes bind)."""
    from sdf_tools_amd import build as b
    b.build_libsdfgpu_multi()
    src = os.path.join(ROOT, "examples", "c_client.c")
    if not os.path.exists(C_EXE) or os.path.getmtime(C_EXE) < os.path.getmtime(src):
        lib = os.path.join(ROOT, "sdf_tools_amd")
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(ROOT, "include"), src,
                               "-o", C_EXE, "-L", lib, "-lsdfgpu_multi", "-lsdfgpu", "-Wl,-rpath," + lib,
                               "-Wl,-rpath-link," + lib, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"])
    return C_EXE


def test_c_client_compiles_as_c11_and_refuses_without_gpu():
    from sdf_tools_amd import capi
    exe = _build_c()
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no CPU fallback" in r.stdout and "sdfgpu_multi_create" in r.stdout


@pytest.mark.gpu
def test_c_client_single_and_multi_rank_fields_are_identical():
    exe = _build_c()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fields identical" in r.stdout


DQ_EXE = os.path.join(ROOT, "examples", "device_queries_example")


def _build_dq():
    """examples/device_queries.cpp: the device-resident seam (CollisionMapGrid::ExtractSignedDistanceFieldDevice,
    DeviceSignedDistanceField::QueryBatch / EstimateDistanceBatch / GetGradientBatch / Host)."""
    from sdf_tools_amd import build as b
    b.build_libsdfgpu()
    src = os.path.join(ROOT, "examples", "device_queries.cpp")
    hdr = os.path.join(ROOT, "include", "sdf_tools", "device_sdf.hpp")
    if not os.path.exists(DQ_EXE) or os.path.getmtime(DQ_EXE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src,
                               "-o", DQ_EXE, "-L", os.path.join(ROOT, "sdf_tools_amd"), "-lsdfgpu",
                               "-Wl,-rpath," + os.path.join(ROOT, "sdf_tools_amd"), "-lz"])
    return DQ_EXE


def test_cpp_device_queries_compile_and_refuse_without_gpu():
    from sdf_tools_amd import capi
    exe = _build_dq()
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("n,points", [(96, 200000), (512, 1 << 20)])
def test_cpp_device_queries_without_downloading_the_field(n, points):
    """VERDICT r3 "next round" 7: a C++ caller builds n^3, answers its queries from HBM (no n^3 x 4-byte download),
    and the answers equal the reference-shaped host calls on the lazily downloaded field (<= 1e-9)."""
    exe = _build_dq()
    r = subprocess.run([exe, str(n), str(points)], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "device queries OK" in r.stdout and "no field download" in r.stdout


def _build_seam():
    from sdf_tools_amd import build as b
    return b.build_example("class_seam")


def test_cpp_class_seam_compiles_and_refuses_without_gpu():
    from sdf_tools_amd import capi
    exe = _build_seam()
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("n,budget_ms", [(96, None), (512, 60.0)])
def test_cpp_class_seam_is_bit_identical_and_bounded(n, budget_ms):
    """VERDICT r4 "next round" 3: CollisionMapGrid::ExtractSignedDistanceField (collision_map.hpp:680-712) end to end from a
    C++ client -- host cells in, host SignedDistanceField out -- equals sdfgpu_build on the same predicate's mask bit for bit
    (the client checks it, and that the returned field was MOVED, not copied), and at 512^3 stays inside a latency budget
    (round 4: 241 ms; the PCIe floor of the 512 MiB download alone is ~10 ms)."""
    import json
    exe = _build_seam()
    r = subprocess.run([exe, str(n), "3", "0.5"], capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0 and "class seam OK" in r.stdout, r.stdout + r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["bit_identical_to_sdfgpu_build"] is True
    if budget_ms is not None:
        assert line["min_ms"] <= budget_ms, line
