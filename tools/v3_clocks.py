#!/usr/bin/env python3
"""Phase clocks of the far-field kernels (needs a library built with SDFGPU_EXTRA_FLAGS=-DSDFGPU_PHASE_CLOCKS):
share of the waves' shader-clock time per phase.  usage: v3_clocks.py [stream|room[n]|<bernoulli p>] [name=value ...]
(two-valued tiles: levelA = classification, levelB = nearest-site scans, scanC = distance chains)"""
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
elif kind.startswith("room"):
    n = int(kind[4:] or 512)
    mask = synth.room_mask_torch((n, n, n), dev)
else:
    mask = synth.bernoulli_mask_torch((n, n, n), float(kind), 1, device=dev)
out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
ctx.set_option("policy_reset", 1)
ctx.set_option("dense", 0)
for kv in sys.argv[2:]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if not any(a.startswith("far_predict=") for a in sys.argv[2:]):
    ctx.set_option("envelope_mode", 1)
lib = ctx._lib
lib.sdfgpu_debug_read_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
torch.cuda.synchronize()
lib.sdfgpu_debug_read_clocks(ctx._h, buf)
B = 5
for _ in range(B):
    ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
torch.cuda.synchronize()
lib.sdfgpu_debug_read_clocks(ctx._h, buf)
names = ["stage", "levelA", "levelB", "scanC", "finishC", "local", "endbar", "-"]
res_ = {}
for st in range(2):
    v = [buf[st * 8 + k] / B for k in range(8)]
    tot = sum(v) or 1.0
    res_["KE%d" % (st + 2)] = {"Mcycles_per_wave_sum": round(tot / 1e6, 1), **{nm: round(x / tot, 3) for nm, x in zip(names, v) if x}}
print(json.dumps({kind: res_}))
