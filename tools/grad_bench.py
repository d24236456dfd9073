#!/usr/bin/env python3
"""Full-grid gradient kernel at 512^3 (fp32 out): ms per launch on the streaming scene's field."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdf_tools_amd import capi, synth
n = 512; res = 0.01; shape = (n, n, n)
ctx = capi.SdfGpu(0); dev = torch.device("cuda", 0)
m = synth.bernoulli_mask_torch(shape, 0.5, 1, device=dev)
sdf = torch.empty(shape, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
ctx.build_device(m.data_ptr(), shape, sdf.data_ptr(), res, False, s)
g = torch.empty(shape + (3,), dtype=torch.float32, device=dev)
for r in (res, 0.03):
    for _ in range(3): ctx.gradient_device(sdf.data_ptr(), shape, g.data_ptr(), r, True, False, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ctx.gradient_device(sdf.data_ptr(), shape, g.data_ptr(), r, True, False, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("res %g: %.4f ms  (%.0f GB/s of 16 B/voxel)" % (r, ms, n ** 3 * 16 / ms / 1e6))
