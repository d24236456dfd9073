#!/usr/bin/env python3
"""Why does the dense tier's wide form (KD3 + KF) give up on a scene?  Builds Bernoulli(p) at 512^3 with KD3 + KF forced
(option dense3_mode), prints whether the tier certified the scene, the build time, and the scene's largest squared distance
per class from the (exact) result.   usage: kf_diag.py p [p ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdf_tools_amd import capi, synth
n = 512
shape = (n, n, n)
dev = torch.device("cuda", 0)
out = torch.empty(shape, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for p in [float(a) for a in sys.argv[1:]] or [0.015]:
    for seed in (1, 2):
        m = synth.bernoulli_mask_torch(shape, p, seed, device=dev)
        ctx = capi.SdfGpu(0)
        ctx.set_option("dense_retry", 0)
        ctx.set_option("dense3_mode", 1)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.build_device(m.data_ptr(), shape, out.data_ptr(), 1.0, False, s)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        d2 = (out.double() ** 2).round()
        free_max = float(d2[m == 0].max()); filled_max = float(d2[m != 0].max())
        und = float(((d2 > 14) & (m == 0)).double().mean())
        print(json.dumps({"p": p, "seed": seed, "ms": [round(t, 3) for t in ts], "path": ctx.last_path(), "info": ctx.last_build_info(),
                          "max_d2_free": free_max, "max_d2_filled": filled_max, "share_beyond_14": round(und, 5)}), flush=True)
        ctx.close()
