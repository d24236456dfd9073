import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def _gpu_context():
    """One libsdfgpu context on device 0.  Fails loudly (no skip, no fallback) when the HIP
    library or the GPU is missing: a gpu-marked test must exercise the native path."""
    from sdf_tools_amd import capi

    ctx = capi.SdfGpu(0)
    yield ctx
    ctx.close()


@pytest.fixture
def gpu(_gpu_context):
    """The shared context with its adaptive policy reset (what it learned from another test's scenes only moves
    work between kernels, but several tests assert WHICH kernels ran)."""
    _gpu_context.set_option("policy_reset", 1)
    _gpu_context.set_option("dense_retry", 0)        # always try the dense kernels (the default retries every 16th build)
    return _gpu_context
