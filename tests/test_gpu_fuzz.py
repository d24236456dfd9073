"""GPU: a short differential fuzz (tools/fuzz_parity.py): random shapes, densities, scene types, borders and
kernel options through the C ABI, every voxel bit for bit against the oracle's exact EDT."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_differential_fuzz():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "8", "3"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fuzz OK" in r.stdout
