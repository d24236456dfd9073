import json, os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from sdf_tools_amd import capi, synth
n = 512; shape = (n, n, n); dev = torch.device("cuda", 0)
out = torch.empty(shape, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for p in [0.06, 0.05, 0.04, 0.03, 0.025, 0.02, 0.015]:
    masks = [synth.bernoulli_mask_torch(shape, p, 1 + k, device=dev) for k in range(2)]
    row = {"p": p}
    for label, opts in (("default", {}), ("dense0", {"dense": 0}), ("farfield", {"dense": 0, "envelope_mode": 1}), ("thr_y16", {"dense": 0, "far_threshold_y": 16}), ("thr_y9", {"dense": 0, "far_threshold_y": 9})):
        ctx = capi.SdfGpu(0)
        for k, v in opts.items(): ctx.set_option(k, v)
        for i in range(20):
            ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s)
        torch.cuda.synchronize()
        row[label] = round((time.perf_counter() - t0) / 30 * 1e3, 3)
        ctx.close()
    print(json.dumps(row), flush=True)
