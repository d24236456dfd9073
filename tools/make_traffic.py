#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py's roofline.traffic reads) from ONE round's condensed PMC files:
    python tools/make_traffic.py r04
reads profiles/<round>_pmc_traffic_{dense,mid,general,envelope}.json (tools/rocprof_summary.py pmc: HBM bytes per launch
by EXACT kernel name, guard exits left out) and writes profiles/traffic.json with the source of every entry."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
P = os.path.join(ROOT, "profiles")
want = {  # kernel -> (file, scene)
    "k_pack_bits_mask": ("dense", "Bernoulli p=0.5"), "k_ball_dense": ("dense", "Bernoulli p=0.5"),
    "k_ball_dense3": ("mid", "Bernoulli p=0.03"), "k_ball_fixup": ("mid", "Bernoulli p=0.03"),
    "k_sweep_z_wave16": ("general", "Bernoulli p=0.5, dense=0"), "k_sweep_y16": ("general", "Bernoulli p=0.5, dense=0"),
    "k_sweep_x16": ("general", "Bernoulli p=0.5, dense=0"),
    "k_envelope_dc<2>": ("envelope", "two-box cloud, int32 hand-off"), "k_envelope_dc<3>": ("envelope", "two-box cloud, int32 hand-off"),
}
out, src = {}, {}
for kernel, (which, scene) in want.items():
    f = os.path.join(P, "%s_pmc_traffic_%s.json" % (name, which))
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    if kernel in d:
        out[kernel] = d[kernel]
        src[kernel] = {"file": os.path.basename(f), "scene": scene, "launches": d["_raw"][kernel].get("launches")}
if "k_ball_dense3" in out and "k_ball_fixup" in out:
    out["k_ball_dense3+k_ball_fixup"] = out["k_ball_dense3"] + out["k_ball_fixup"]
out["_source"] = {"round": name, "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (--kernel-trace only), 512^3; bytes per "
                  "launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950: FETCH_SIZE under-reports a streaming read 2x; "
                  "MI355X_MICROARCH.md), averaged over the launches that did work (guard exits excluded), kernels matched by exact name",
                  "entries": src}
json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
