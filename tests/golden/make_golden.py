"""Regenerates tests/golden/oracle_vectors.npz: seeded inputs and the outputs of the CPU
oracle (oracle/sdf_oracle.c, the restatement of sdf_generation.hpp:95-420) on them.
The reference itself cannot be built in this image (Eigen / arc_utilities / ROS absent), so
these vectors pin the *oracle's* behaviour across refactors; the oracle in turn is pinned to
the reference by known_answers.json.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from sdf_tools_amd import synth  # noqa: E402

cases = {
    "bern50_24x20x17": (synth.bernoulli_mask((24, 20, 17), 0.5, 1), 1.0, False),
    "bern30_32x32x32_res001": (synth.bernoulli_mask((32, 32, 32), 0.3, 2), 0.01, False),
    "bern05_28x28x28": (synth.bernoulli_mask((28, 28, 28), 0.05, 3), 1.0, False),
    "bern50_vb_16x12x20": (synth.bernoulli_mask((16, 12, 20), 0.5, 4), 0.5, True),
    "spheres_48": (synth.spheres_mask((48, 48, 48), 6, (3, 8), 0), 1.0, False),
}
out = {}
for name, (m, res, vb) in cases.items():
    sdf, ext = O.reference_sdf(m, res, vb)
    out[name + "/mask"] = np.packbits(m.reshape(-1))
    out[name + "/shape"] = np.array(m.shape)
    out[name + "/res_vb"] = np.array([res, float(vb)])
    out[name + "/sdf"] = sdf
    out[name + "/extrema"] = np.array(ext)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "oracle_vectors.npz"), **out)
print("wrote", len(cases), "cases")
