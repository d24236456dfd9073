#!/usr/bin/env python3
"""A/B of two builds of libsdfgpu.so in ONE process on ONE box (VERDICT r5 "next round" 3): the separable-sweep tier
(option dense=0: K1 -> probe -> K2/16 -> K3/16) on the headline grids, alternating the libraries repetition by repetition,
so that a box's own drift shows up in both columns.  Only entry points that every round's library has are bound.

usage: sweep_tier_ab.py A=<path> B=<path> [n=512] [reps=6] [steps=50] [scene=...] [opt=value ...] [A:opt=value | B:opt=value ...]
(the same path twice with an A: / B: option = an A/B of that option inside one library)
prints one JSON line per (rep, library) and a summary line."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (first: the libraries then bind to torch's HIP runtime)

from sdf_tools_amd import synth  # noqa: E402

kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
libs = {"A": kv.pop("A"), "B": kv.pop("B")}
n = int(kv.pop("n", 512))
reps = int(kv.pop("reps", 6))
steps = int(kv.pop("steps", 50))
scene_arg = kv.get("scene")
only = {"A": {k[2:]: int(v) for k, v in kv.items() if k.startswith("A:")}, "B": {k[2:]: int(v) for k, v in kv.items() if k.startswith("B:")}}
opts = {k: int(v) for k, v in kv.items() if k != "scene" and k[1:2] != ":"} or {"dense": 0}
kv = {"scene": scene_arg} if scene_arg else {}
vp, i64, dbl, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int
STAGES = ["pack_bits", "dense_ball", "sweep_z", "sweep_y", "envelope_y", "sweep_x", "envelope_x"]


def bind(path, extra):
    L = ctypes.CDLL(os.path.abspath(path))
    L.sdfgpu_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.sdfgpu_destroy.argtypes = [vp]
    L.sdfgpu_set_option.argtypes = [vp, ctypes.c_char_p, ci]
    L.sdfgpu_build_device.argtypes = [vp, vp, i64, i64, i64, dbl, ci, vp, vp]
    L.sdfgpu_set_profiling.argtypes = [vp, ci]
    L.sdfgpu_get_stage_times.argtypes = [vp, vp, vp]
    L.sdfgpu_last_error.argtypes = [vp]
    L.sdfgpu_last_error.restype = ctypes.c_char_p
    h = vp()
    assert L.sdfgpu_create(0, ctypes.byref(h)) == 0
    for k, v in list(opts.items()) + list(extra.items()):
        assert L.sdfgpu_set_option(h, k.encode(), v) == 0, (path, k, L.sdfgpu_last_error(h))
    return L, h


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
shape = (n, n, n)
scene = kv.pop("scene", "bernoulli") if "scene" in kv else "bernoulli"
if scene == "twobox":                       # the streaming scene: 200 k points in two boxes (far-field on both axes)
    import numpy as np
    res0 = 0.01
    masks = []
    for f in range(3):
        pc = synth.two_box_points(200000 * n * n // (512 * 512), seed=f, scale=n * res0)
        idx = (pc.astype(np.float64) / res0).astype(np.int64)
        ok = np.all((idx >= 0) & (idx < n), axis=1)
        m = np.zeros(shape, np.uint8)
        m[idx[ok, 0], idx[ok, 1], idx[ok, 2]] = 1
        masks.append(torch.from_numpy(m).to(dev))
elif scene == "room":
    masks = [synth.room_mask_torch(shape, dev)] * 3
elif scene in ("boxes", "shells", "spheres"):      # tools/scene_bench.py's structured scenes
    m = (synth.solid_spheres_mask_torch(shape, dev) if scene == "spheres" else synth.tutorial_boxes_mask_torch(shape, dev, scene == "boxes"))
    masks = [m] * 3
elif scene == "noisyfloor":                 # a floor under 0.05 % noise: every x-plane "has a filled voxel in every row", no tile is two-valued
    masks = []
    for k in range(3):
        m = synth.bernoulli_mask_torch(shape, 0.0005, 1 + k, device=dev)
        m[:, :, :2] = 1
        masks.append(m)
else:
    masks = [synth.bernoulli_mask_torch(shape, 0.5, 1 + k, device=dev) for k in range(3)]
out = {k: torch.empty(shape, dtype=torch.float32, device=dev) for k in libs}
stream = torch.cuda.current_stream(dev).cuda_stream
ctx = {k: bind(p, only[k]) for k, p in libs.items()}


def run(key, count):
    L, h = ctx[key]
    for i in range(count):
        rc = L.sdfgpu_build_device(h, masks[i % 3].data_ptr(), n, n, n, 0.01, 0, out[key].data_ptr(), stream)
        assert rc == 0, L.sdfgpu_last_error(h)


for key in libs:
    run(key, 10)
torch.cuda.synchronize()
assert torch.equal(out["A"], out["B"]), "the two libraries disagree"
acc = {k: {"ms": [], "stages": []} for k in libs}
for rep in range(reps):
    for key in (("A", "B") if rep % 2 == 0 else ("B", "A")):
        L, h = ctx[key]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run(key, steps)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        L.sdfgpu_set_profiling(h, 1)
        run(key, 12)
        torch.cuda.synchronize()
        t = (dbl * 7)()
        b = i64()
        L.sdfgpu_get_stage_times(h, t, ctypes.byref(b))
        L.sdfgpu_set_profiling(h, 0)
        st = {s: round(t[i] / max(b.value, 1), 4) for i, s in enumerate(STAGES) if t[i] > 0.0005 * max(b.value, 1)}
        acc[key]["ms"].append(ms)
        acc[key]["stages"].append(st)
        print(json.dumps({"rep": rep, "lib": key, "path": libs[key], "ms_per_build": round(ms, 4), "stages_ms": st}), flush=True)
summary = {"n": n, "scene": scene, "options": opts, "only": only, "steps": steps}
for key in libs:
    v = sorted(acc[key]["ms"])
    summary[key] = {"path": libs[key], "ms_min": round(v[0], 4), "ms_median": round(v[len(v) // 2], 4), "ms_max": round(v[-1], 4),
                    "stages_ms_median": {s: sorted(x.get(s, 0.0) for x in acc[key]["stages"])[len(v) // 2] for s in STAGES
                                         if any(s in x for x in acc[key]["stages"])}}
summary["B_over_A_median"] = round(summary["B"]["ms_median"] / summary["A"]["ms_median"], 4)
print(json.dumps(summary))
