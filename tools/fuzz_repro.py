#!/usr/bin/env python3
"""Re-run one fuzz scene (gpurun_out/fuzz_last_mask.npy, written by FUZZ_VERBOSE=1 tools/fuzz_parity.py) on a fresh context with the
given options, against the exact oracle:  fuzz_repro.py res vb name=value ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from sdf_tools_amd import capi
m = np.load(os.environ.get("FUZZ_MASK", "gpurun_out/fuzz_last_mask.npy"))
res, vb = float(sys.argv[1]), sys.argv[2] == "1"
ctx = capi.SdfGpu(0)
reps = 1
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    if k == "reps": reps = int(v)
    else: ctx.set_option(k, int(v))
want, want_ext, _ = O.exact_sdf(m, res, vb)
for r in range(reps):
    got, ext = ctx.build(m, res, vb)
    ok = np.array_equal(got.view(np.uint32), want.view(np.uint32)) and tuple(ext) == tuple(float(v) for v in want_ext)
    print("build", r, "ok" if ok else "MISMATCH", ctx.last_build_info(), ctx.last_path(), flush=True)
