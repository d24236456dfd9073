"""Loads the in-tree pysdf_tools extension (built by sdf_tools_amd/build.py)."""
import importlib.util
import os
import sys

from . import build as _build


def load_pysdf_tools():
    if "pysdf_tools" in sys.modules:
        return sys.modules["pysdf_tools"]
    path = _build.pysdf_tools_path()
    if not os.path.exists(path):
        raise ImportError("pysdf_tools is not built (run `python -m sdf_tools_amd.build`)")
    try:   # bind libsdfgpu.so to the HIP runtime torch uses, if torch is around (see capi.load_library)
        import torch  # noqa: F401
    except Exception:
        pass
    spec = importlib.util.spec_from_file_location("pysdf_tools", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["pysdf_tools"] = mod
    return mod
