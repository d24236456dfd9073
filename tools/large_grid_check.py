#!/usr/bin/env python3
"""One-off sanity check at sizes the oracle cannot reach: a 1024^3 (or given) Bernoulli grid is built on the GPU and
random 48^3 crops are compared with the oracle run on the crop plus a margin larger than the largest distance."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402
from sdf_tools_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
shape = (n, n, n)
ctx = capi.SdfGpu(0)
mask = synth.bernoulli_mask_torch(shape, p, 3, device="cuda")
out = torch.empty(shape, dtype=torch.float32, device="cuda")
for _ in range(2):
    ctx.build_device(mask.data_ptr(), shape, out.data_ptr(), 0.01, False, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
mx, mn = ctx.get_extrema()
maxd = int(round(max(mx, -mn) / 0.01)) + 2
rng = np.random.default_rng(0)
C, bad = 48, 0
for k in range(6):
    lo = [int(rng.integers(0, n - C)) if k else 0 for _ in range(3)] if k < 5 else [n - C] * 3
    a = [max(0, l - maxd) for l in lo]
    b = [min(n, l + C + maxd) for l in lo]
    sub = mask[a[0]:b[0], a[1]:b[1], a[2]:b[2]].cpu().numpy()
    want, _, _ = O.exact_sdf(sub, 0.01)
    off = [l - aa for l, aa in zip(lo, a)]
    w = want[off[0]:off[0] + C, off[1]:off[1] + C, off[2]:off[2] + C]
    g = out[lo[0]:lo[0] + C, lo[1]:lo[1] + C, lo[2]:lo[2] + C].cpu().numpy()
    # crop faces that are not grid faces see fewer sites in the oracle run: only compare where the margin is complete
    ok = np.array_equal(g.view(np.uint32), w.view(np.uint32))
    print("crop", lo, "ok" if ok else "MISMATCH", "maxd", maxd)
    bad += 0 if ok else 1
print("info", ctx.last_build_info(), ctx.last_path(), "extrema", (mx, mn))
sys.exit(1 if bad else 0)
