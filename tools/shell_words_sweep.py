#!/usr/bin/env python3
"""ms per 512^3 build over the KD6 group threshold (option "shell_min_words") at the mid densities: picks the constant kShellMinWords."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdf_tools_amd import capi, synth
n = 512; shape = (n, n, n); dev = torch.device("cuda", 0)
out = torch.empty(shape, dtype=torch.float32, device=dev); s = torch.cuda.current_stream().cuda_stream
for p in (0.04, 0.03, 0.025, 0.02, 0.015):
    masks = [synth.bernoulli_mask_torch(shape, p, 1 + k, device=dev) for k in range(2)]
    row = {"p": p}
    for w in [int(a) for a in sys.argv[1:]] or [0, 16, 32, 64, 96, 128, 100000]:
        ctx = capi.SdfGpu(0); ctx.set_option("shell_min_words", w)
        for i in range(12):
            ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s)
        torch.cuda.synchronize()
        row[w] = round((time.perf_counter() - t0) / 40 * 1e3, 3)
        if not ctx.last_path()["dense_certified"]: row[w] = -row[w]
        ctx.close()
    print(json.dumps(row), flush=True)
