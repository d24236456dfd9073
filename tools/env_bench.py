#!/usr/bin/env python3
"""Far-field tier micro-benchmark: the streaming scene (200 k points in two boxes, 512^3) with the envelope kernels as the
only y / x sweeps; prints per-stage HIP-event times for the divide-and-conquer kernel and the first-generation kernel.
usage: env_bench.py [n] [builds] [name=value ...]   (options go to sdfgpu_set_option)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdf_tools_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
opts = [a for a in sys.argv[3:] if "=" in a and not a.startswith("--")]
res = 0.01
ctx = capi.SdfGpu(0)
dev = torch.device("cuda", 0)
pts = torch.from_numpy(synth.two_box_points(200000, seed=0, scale=n * res)).to(dev)
mask = torch.zeros((n, n, n), dtype=torch.uint8, device=dev)
out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
ctx.voxelize_points_device(pts.data_ptr(), pts.shape[0], (0.0, 0.0, 0.0), res, (n, n, n), mask.data_ptr(), True, s)
for a in sys.argv:
    if a.startswith("--bernoulli="):          # a uniformly sparse scene instead of the two boxes
        mask = synth.bernoulli_mask_torch((n, n, n), float(a.split("=")[1]), 1, device=dev)
names = ["pack_bits", "dense_ball", "sweep_z", "sweep_y", "envelope_y", "sweep_x", "envelope_x"]
result = {}
for dc in ([1, 0] if "--both" in sys.argv else [1]):      # 0: the far-field kernel off (unbounded marching scans)
    ctx.set_option("policy_reset", 1)
    ctx.set_option("dense", 0)
    ctx.set_option("envelope_dc", dc)
    for kv in opts:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.set_option("envelope_mode", 1)
    for _ in range(3):
        ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
    torch.cuda.synchronize()
    ctx.get_stage_times()
    ctx.set_profiling(1)
    for _ in range(builds):
        ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
    torch.cuda.synchronize()
    ms, b = ctx.get_stage_times()
    ctx.set_profiling(0)
    result["dc=%d" % dc] = {k: round(v / max(b, 1), 4) for k, v in zip(names, ms) if v > 0}
    result["dc=%d" % dc]["extrema"] = ctx.get_extrema()
print(json.dumps(result))
