#!/usr/bin/env python3
"""ms per 512^3 build over Bernoulli occupancy p (steady state and fresh context), through the C ABI: the table of DESIGN.md 4.2."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdf_tools_amd import capi, synth
n = 512
shape = (n, n, n)
dev = torch.device("cuda", 0)
out = torch.empty(shape, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
rows = []
for p in ([float(a) for a in sys.argv[1:]] or [0.5, 0.2, 0.1, 0.05, 0.03, 0.02, 0.015, 0.01, 0.003, 0.001, 0.0001]):
    masks = [synth.bernoulli_mask_torch(shape, p, 1 + k, device=dev) for k in range(2)]
    ctx = capi.SdfGpu(0)
    torch.cuda.synchronize()                      # (the mask generators above are asynchronous)
    t0 = time.perf_counter(); ctx.build_device(masks[0].data_ptr(), shape, out.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    for i in range(30):
        ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    ctx.get_stage_times(); ctx.set_profiling(1)
    for i in range(8):
        ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s)
    torch.cuda.synchronize()
    st, b = ctx.get_stage_times(); ctx.set_profiling(0)
    names = ["pack", "ball", "z", "y", "env_y", "x", "env_x"]
    rows.append({"p": p, "ms": round(ms, 3), "first_ms": round(first, 3), "Gvox_s": round(n ** 3 / ms / 1e6, 1), "path": ctx.last_path(),
                 "stages": {k: round(v / b, 3) for k, v in zip(names, st) if v / b > 0.002}})
    ctx.close()
    print(json.dumps(rows[-1]), flush=True)
# virtual border through the generic dense kernels, 512^3 p = 0.5
ctx = capi.SdfGpu(0)
m = synth.bernoulli_mask_torch(shape, 0.5, 1, device=dev)
for vb in (False, True):
    for i in range(5):
        ctx.build_device(m.data_ptr(), shape, out.data_ptr(), 0.01, vb, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        ctx.build_device(m.data_ptr(), shape, out.data_ptr(), 0.01, vb, s)
    torch.cuda.synchronize()
    print(json.dumps({"p": 0.5, "vb": vb, "ms": round((time.perf_counter() - t0) / 20 * 1e3, 3), "path": ctx.last_path(), "info": ctx.last_build_info()}))
