#!/usr/bin/env python3
"""Is the per-voxel cost of the far-field sweeps at 1024^3 a matter of the line length or of the row stride?  The tutorial-box
scene (scaled to the grid) over shapes that vary the two separately; ns per 1000 voxels of the y and x far-field sweeps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdf_tools_amd import capi, synth
dev = torch.device("cuda", 0)
ctx = capi.SdfGpu(0)
s = torch.cuda.current_stream().cuda_stream
names = ["pack_bits", "dense_ball", "sweep_z", "sweep_y", "envelope_y", "sweep_x", "envelope_x"]
for shape in [(512, 512, 512), (1024, 512, 512), (1024, 1024, 256), (512, 1024, 1024), (512, 1024, 512), (1024, 256, 256), (1024, 1024, 1024)]:
    mask = synth.tutorial_boxes_mask_torch(shape, dev, True)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    ctx.set_option("policy_reset", 1)
    for _ in range(4):
        ctx.build_device(mask.data_ptr(), shape, out.data_ptr(), 0.01, False, s)
    torch.cuda.synchronize()
    ctx.get_stage_times(); ctx.set_profiling(1)
    for _ in range(4):
        ctx.build_device(mask.data_ptr(), shape, out.data_ptr(), 0.01, False, s)
    torch.cuda.synchronize()
    st, b = ctx.get_stage_times(); ctx.set_profiling(0)
    n = shape[0] * shape[1] * shape[2]
    d = {k: v / max(b, 1) for k, v in zip(names, st)}
    print(json.dumps({"shape": shape, "x_row_stride_MiB": shape[1] * shape[2] * 4 / 2**20, "y_ps_per_voxel": round(d["envelope_y"] * 1e9 / n, 2),
                      "x_ps_per_voxel": round(d["envelope_x"] * 1e9 / n, 2), "env_y_ms": round(d["envelope_y"], 3), "env_x_ms": round(d["envelope_x"], 3)}), flush=True)
    del mask, out
