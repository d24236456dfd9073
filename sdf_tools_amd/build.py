"""In-tree builds of the native pieces (no network, no cmake needed).

  libsdfgpu.so   HIP kernels + C ABI (include/sdfgpu.h), hipcc --offload-arch=gfx950
  pysdf_tools    pybind11 module mirroring the reference's src/sdf_tools/bindings.cpp

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU container;
the built .so files travel to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
import sysconfig

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libsdfgpu.so")
LIB_MULTI = os.path.join(PKG, "libsdfgpu_multi.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def build_libsdfgpu(force=False, verbose=False, out=None):
    """Three translation units -> objects (compiled side by side) -> libsdfgpu.so: sdfgpu.hip (C ABI, host orchestration, most
    kernels), sdfgpu_envelope_tu.hip (the far-field kernel's instantiations) and sdfgpu_dense6_tu.hip (the shell pass).  Each object is rebuilt when its
    source or ANY header it can include is newer (a stale library after a header-only edit is the kind of bug that
    invalidates measurements without failing anything)."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("sdfgpu") and f.endswith(".hpp")) + [
        os.path.join(INCLUDE, "sdfgpu.h")]
    env_hdrs = [os.path.join(CSRC, f) for f in ("sdfgpu_envelope_dc.hpp", "sdfgpu_kernels.hpp", "sdfgpu_sweep_x16.hpp")]
    d6_hdrs = [os.path.join(CSRC, f) for f in ("sdfgpu_dense6.hpp", "sdfgpu_dense3.hpp", "sdfgpu_dense.hpp", "sdfgpu_kernels.hpp")]
    units = [(os.path.join(CSRC, "sdfgpu.hip"), hdrs), (os.path.join(CSRC, "sdfgpu_envelope_tu.hip"), env_hdrs),
             (os.path.join(CSRC, "sdfgpu_dense6_tu.hip"), d6_hdrs)]
    extra = os.environ.get("SDFGPU_EXTRA_FLAGS", "").split()
    objdir = os.path.join(CSRC, ".obj" + ("_" + "".join(c for c in "".join(extra) if c.isalnum()) if extra else ""))
    os.makedirs(objdir, exist_ok=True)
    todo, objs = [], []
    for src, deps in units:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + deps):
            todo.append([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-I", INCLUDE, "-c", src,
                         "-o", obj] + extra)
    lib = out or LIB
    if not todo and not _newer(lib, objs):
        return lib
    if verbose:
        for cmd in todo:
            print(" ".join(cmd))
    with ThreadPoolExecutor(max_workers=3) as pool:
        list(pool.map(subprocess.check_call, todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


def build_profiling_variant(name, flags, verbose=False):
    """A profiling build of the same library (never the shipped one): tools/probe/libsdfgpu_<name>.so compiled with extra
    flags -- "hooks": -DSDFGPU_DEBUG_HOOKS (switches that skip work; results are then wrong), "trips":
    -DSDFGPU_PHASE_CLOCKS -DSDFGPU_TRIP_COUNTS.  Selected with SDFGPU_LIB=<path> by the tools/ scripts."""
    out = os.path.join(ROOT, "tools", "probe", "libsdfgpu_%s.so" % name)
    old = os.environ.get("SDFGPU_EXTRA_FLAGS")
    os.environ["SDFGPU_EXTRA_FLAGS"] = flags
    try:
        return build_libsdfgpu(False, verbose, out)
    finally:
        if old is None:
            del os.environ["SDFGPU_EXTRA_FLAGS"]
        else:
            os.environ["SDFGPU_EXTRA_FLAGS"] = old


def build_libsdfgpu_multi(force=False, verbose=False):
    """libsdfgpu_multi.so: the multi-GPU C ABI (include/sdfgpu_multi.h) over libsdfgpu.so + RCCL.  Host code only."""
    src = os.path.join(CSRC, "sdfgpu_multi.cpp")
    deps = [src, os.path.join(INCLUDE, "sdfgpu_multi.h"), os.path.join(INCLUDE, "sdfgpu.h")]
    build_libsdfgpu(force=False, verbose=verbose)
    if not force and not _newer(LIB_MULTI, deps + [LIB]):
        return LIB_MULTI
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", INCLUDE, "-I", "/opt/rocm/include",
           src, "-o", LIB_MULTI, "-L", PKG, "-lsdfgpu", "-L", "/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN",
           "-Wl,-rpath,/opt/rocm/lib"]          # (librccl.so is found without LD_LIBRARY_PATH: pysdf_tools links this library)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_MULTI


def pysdf_tools_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(PKG, "pysdf_tools" + suffix)


def build_pysdf_tools(force=False, verbose=False):
    import pybind11

    src = os.path.join(CSRC, "pysdf_tools.cpp")
    if not os.path.exists(src):
        return None
    out = pysdf_tools_path()
    hdr_dir = os.path.join(INCLUDE, "sdf_tools")
    deps = [src, os.path.join(INCLUDE, "sdfgpu.h"), os.path.join(INCLUDE, "sdfgpu_multi.h")] + \
        [os.path.join(hdr_dir, f) for f in sorted(os.listdir(hdr_dir))] + \
        [os.path.join(INCLUDE, "arc_utilities", f) for f in sorted(os.listdir(os.path.join(INCLUDE, "arc_utilities")))]
    if not force and not _newer(out, deps):
        return out
    build_libsdfgpu_multi(force=False, verbose=verbose)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-DSDF_TOOLS_MULTI_GPU", "-DSDF_TOOLS_VECTOR_ADOPT",
           "-I", INCLUDE, "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
           src, "-o", out, "-L", PKG, "-lsdfgpu_multi", "-lsdfgpu", "-Wl,-rpath,$ORIGIN", "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_example(name, force=False, verbose=False):
    """examples/<name>.cpp -> examples/<name>_example with g++ against include/ and the in-tree libsdfgpu.so (the C++ side of
    the drop-in boundary: client code written like the reference's, tests/test_cpp_example.py and bench.py run these)."""
    src = os.path.join(ROOT, "examples", name + ".cpp")
    exe = os.path.join(ROOT, "examples", name + "_example")
    hdr_dir = os.path.join(INCLUDE, "sdf_tools")
    deps = [src, os.path.join(INCLUDE, "sdfgpu.h")] + [os.path.join(hdr_dir, f) for f in sorted(os.listdir(hdr_dir))] + \
        [os.path.join(INCLUDE, "arc_utilities", f) for f in sorted(os.listdir(os.path.join(INCLUDE, "arc_utilities")))]
    build_libsdfgpu(force=False, verbose=verbose)
    if not force and not _newer(exe, deps):
        return exe
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-DSDF_TOOLS_VECTOR_ADOPT", "-I", INCLUDE, src, "-o", exe, "-L", PKG, "-lsdfgpu",
           "-Wl,-rpath," + PKG, "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return exe


def build_all(force=False, verbose=False):
    out = [build_libsdfgpu(force, verbose), build_libsdfgpu_multi(force, verbose)]
    p = build_pysdf_tools(force, verbose)
    if p:
        out.append(p)
    out.append(build_example("class_seam", force, verbose))     # (bench.py's host_api.class_seam runs it on the GPU box)
    return out


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv, verbose=True)))
