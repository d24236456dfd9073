#!/usr/bin/env python3
"""Build times of a few structured 512^3 scenes (solid obstacles in free space -- what CollisionMapGrid callers hold --
next to the synthetic ones of bench.py) with the library's own tier selection: ms per build after warm-up, the path the
handle reports, per-stage HIP-event times.  usage: scene_bench.py [n = 512] [name=value ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdf_tools_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
opts = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
res = 0.01
dev = torch.device("cuda", 0)
ctx = capi.SdfGpu(0)
s = torch.cuda.current_stream().cuda_stream


def boxes(solid):
    return synth.tutorial_boxes_mask_torch((n, n, n), dev, solid)


def room(floor=True, xwall=True, ywall=True):
    return synth.room_mask_torch((n, n, n), dev, floor, xwall, ywall)


def spheres():
    return synth.solid_spheres_mask_torch((n, n, n), dev)


out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
names = ["pack_bits", "dense_ball", "sweep_z", "sweep_y", "envelope_y", "sweep_x", "envelope_x"]
result = {}
SCENES = [("solid boxes", lambda: boxes(True)), ("box shells", lambda: boxes(False)), ("room", room), ("solid spheres", spheres)]
if "--room-variants" in sys.argv:
    SCENES = [("room", room), ("room, no x wall", lambda: room(xwall=False)), ("room, no y wall", lambda: room(ywall=False)),
              ("room, no floor", lambda: room(floor=False)), ("furniture only", lambda: room(False, False, False))]
for name, mk in SCENES:
    mask = mk()
    ctx.set_option("policy_reset", 1)
    for kv in opts:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
    e1.record()
    torch.cuda.synchronize()
    first = e0.elapsed_time(e1)
    for _ in range(4):
        ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
        torch.cuda.synchronize()
    ctx.get_stage_times()
    B = 10
    e0.record()
    for _ in range(B):
        ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / B
    ctx.set_profiling(1)
    for _ in range(4):
        ctx.build_device(mask.data_ptr(), (n, n, n), out.data_ptr(), res, False, s)
    torch.cuda.synchronize()
    st, b = ctx.get_stage_times()
    ctx.set_profiling(0)
    p = ctx.last_path()
    result[name] = {"filled_fraction": round(float(mask.float().mean().item()), 4), "first_build_ms": round(first, 3), "ms_per_build": round(ms, 3),
                    "path": {k: p[k] for k in ("dense_certified", "far_y", "far_x")},
                    "stages_ms": {k: round(v / max(b, 1), 3) for k, v in zip(names, st) if v > 0}, "extrema": ctx.get_extrema(),
                    "checksum": int(out.view(torch.int32).to(torch.int64).sum().item())}
    print(json.dumps({name: result[name]}), flush=True)
