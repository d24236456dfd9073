#!/usr/bin/env python3
"""First build of a fresh context at 512^3 (what a one-shot caller of the reference API pays), in the order bench.py's legs
use: context created, a throw-away context builds the scene and is closed, then the first build of the fresh one."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdf_tools_amd import capi, synth
n = 512; shape = (n, n, n); dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
out = torch.empty(shape, dtype=torch.float32, device=dev)
out2 = torch.empty(shape, dtype=torch.float32, device=dev)
m = synth.bernoulli_mask_torch(shape, 0.01, 1, device=dev)
main = capi.SdfGpu(0)
main.build_device(m.data_ptr(), shape, out.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
for trial in range(3):
    for order in ("ctx_first", "warm_first"):
        if order == "ctx_first":
            ctx = capi.SdfGpu(0)
        warm = capi.SdfGpu(0); warm.build_device(m.data_ptr(), shape, out2.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
        warm.close()
        if order == "warm_first":
            ctx = capi.SdfGpu(0)
        t2 = time.perf_counter()
        ctx.build_device(m.data_ptr(), shape, out2.data_ptr(), 0.01, False, s); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
        ctx.build_device(m.data_ptr(), shape, out2.data_ptr(), 0.01, False, s); torch.cuda.synchronize(); t5 = time.perf_counter()
        print(json.dumps({"order": order, "first_enqueue_ms": round((t3 - t2) * 1e3, 2), "first_sync_ms": round((t4 - t3) * 1e3, 2), "second_ms": round((t5 - t4) * 1e3, 2)}))
        ctx.close()
