#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a few 512^3 SDF builds plus a torch copy of known size
(512 MiB read + 512 MiB written) that calibrates FETCH_SIZE / WRITE_SIZE in the same pass.
usage: pmc_workload.py [n] [p=<Bernoulli density>] [builds=<count>] [name=value library options ...]
Every build is synchronised so that the handle's policy has seen it before the next one (p = 0.03: the steady state is
KD3 + the fix-up kernel from the second build on)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdf_tools_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = capi.SdfGpu(0)
p, builds = 0.5, 4
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "p":
        p = float(v)
    elif k == "builds":
        builds = int(v)
    else:
        ctx.set_option(k, int(v))
mask = synth.bernoulli_mask_torch((n, n, n), p, 1, device="cuda")
out = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
src = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
for _ in range(builds):
    ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), 0.01, False, torch.cuda.current_stream().cuda_stream)
    dst.copy_(src)
    torch.cuda.synchronize()
print("extrema", ctx.get_extrema())
