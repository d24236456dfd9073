#!/usr/bin/env python3
"""Trips of the far-field kernels' scan loops per level (needs a library built with -DSDFGPU_PHASE_CLOCKS -DSDFGPU_TRIP_COUNTS,
SDFGPU_LIB=...): issued by the waves (the slowest lane's count, summed over waves) against needed by the lanes.
usage: trip_counts.py [stream|<bernoulli p>] [name=value ...]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdf_tools_amd import capi, synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "stream"
n, res = 512, 0.01
ctx = capi.SdfGpu(0)
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
if kind == "stream":
    pts = torch.from_numpy(synth.two_box_points(200000, seed=0, scale=n * res)).to(dev)
    mask = torch.zeros((n, n, n), dtype=torch.uint8, device=dev)
    ctx.voxelize_points_device(pts.data_ptr(), pts.shape[0], (0.0, 0.0, 0.0), res, (n, n, n), mask.data_ptr(), True, s)
elif kind == "room":
    mask = synth.room_mask_torch((n, n, n), dev)
elif kind == "boxes":
    mask = synth.tutorial_boxes_mask_torch((n, n, n), dev, True)
elif kind == "spheres":
    mask = synth.solid_spheres_mask_torch((n, n, n), dev)
elif kind == "shells":
    mask = synth.tutorial_boxes_mask_torch((n, n, n), dev, False)
elif kind == "furniture":                 # tools/scene_bench.py's table, legs and shelf without the room around them
    mask = torch.zeros((n, n, n), dtype=torch.uint8, device=dev)
    f = lambda v: int(v * n)
    mask[f(0.3):f(0.7), f(0.3):f(0.6), f(0.35):f(0.38)] = 1
    for (x, y) in ((0.31, 0.31), (0.68, 0.31), (0.31, 0.58), (0.68, 0.58)):
        mask[f(x):f(x) + f(0.02), f(y):f(y) + f(0.02), :f(0.35)] = 1
    mask[f(0.8):f(0.98), f(0.1):f(0.9), f(0.5):f(0.55)] = 1
else:
    mask = synth.bernoulli_mask_torch((n, n, n), float(kind), 1, device=dev)
out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
ctx.set_option("policy_reset", 1)
ctx.set_option("dense", 0)
for kv in sys.argv[2:]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.set_option("envelope_mode", 1)
lib = ctx._lib
lib.sdfgpu_debug_read_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 16)()
ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
torch.cuda.synchronize()
lib.sdfgpu_debug_read_clocks(ctx._h, buf)
ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
torch.cuda.synchronize()
lib.sdfgpu_debug_read_clocks(ctx._h, buf)
waves = n * n * n // 16 // 512 * 4 * (512 // n if n < 512 else 1)
out_ = {}
for st in range(2):
    d = {}
    for l, nm in enumerate("ABC"):
        issued, needed = buf[st * 8 + 2 * l], buf[st * 8 + 2 * l + 1]
        d[nm] = {"issued_per_wave": round(issued / waves, 2), "needed_per_lane": round(needed / waves / 64, 2),
                 "imbalance": round(issued * 64 / max(needed, 1), 2)}
    out_["KE%d" % (st + 2)] = d
print(json.dumps({kind: out_}))
