#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a few 512^3 SDF builds plus a torch copy of known size
(512 MiB read + 512 MiB written) that calibrates FETCH_SIZE / WRITE_SIZE in the same pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdf_tools_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = capi.SdfGpu(0)
for kv in sys.argv[2:]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
mask = synth.bernoulli_mask_torch((n, n, n), 0.5, 1, device="cuda")
out = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
src = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
for _ in range(4):
    ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), 0.01, False, torch.cuda.current_stream().cuda_stream)
    dst.copy_(src)
torch.cuda.synchronize()
print("extrema", ctx.get_extrema())
