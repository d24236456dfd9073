#!/usr/bin/env python3
"""Where the first build of a fresh process goes: context creation, first / second device-resident build, per grid size."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdf_tools_amd import capi, synth
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
out = {}
for n in (64, 512):
    m = synth.bernoulli_mask_torch((n, n, n), 0.5, 1, device=dev)
    o = torch.empty((n, n, n), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ctx = capi.SdfGpu(0); t1 = time.perf_counter()
    times = []
    for i in range(3):
        a = time.perf_counter(); ctx.build_device(m.data_ptr(), (n, n, n), o.data_ptr(), 0.01, False, s); torch.cuda.synchronize(); times.append((time.perf_counter() - a) * 1e3)
    out[n] = {"create_ms": round((t1 - t0) * 1e3, 3), "builds_ms": [round(t, 3) for t in times]}
    ctx.close()
print(json.dumps(out))
