#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (run on the GPU box) into the small files committed under profiles/.

  python tools/rocprof_summary.py stats  <dir> <out.md>          kernel-trace --stats summary
  python tools/rocprof_summary.py pmc    <dir>... <out.json>     FETCH_SIZE / WRITE_SIZE per launch

The PMC corrections follow MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE
under-reports a wide coalesced streaming read by exactly 2x; WRITE_SIZE is uncalibrated, so the
calibration kernel (a torch elementwise copy of known size recorded in the same pass) is used to
derive the factor actually observed here.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(dirname, pattern):
    return sorted(glob.glob(os.path.join(dirname, "**", pattern), recursive=True))


def short(name):
    for key, label in (("k_envelope_dc<2", "envelope_y"), ("k_envelope_dc<3", "envelope_x"),
                       ("k_envelope_dcILi2", "envelope_y"), ("k_envelope_dcILi3", "envelope_x"),
                       ("k_envelope<2>", "envelope_y_gen1"), ("k_envelope<3>", "envelope_x_gen1"),
                       ("k_pack_bits", "pack_bits"), ("k_ball_dense", "dense_ball"), ("k_sweep_zy_fused", "sweep_zy"),
                       ("k_sweep_x16", "sweep_x16"), ("k_sweep_z_vec16", "sweep_z"), ("k_sweep_z_generic", "sweep_z_generic"),
                       ("k_sweep_march<2", "sweep_y"), ("k_sweep_march<3", "sweep_x"),
                       ("k_sweep_marchILi2", "sweep_y"), ("k_sweep_marchILi3", "sweep_x"),
                       ("k_fused", "fused"), ("k_gradient", "gradient")):
        if key in name:
            return label
    return None


def stats(dirname, out_md):
    files = find(dirname, "*kernel_stats.csv")
    if not files:
        raise SystemExit("no *kernel_stats.csv under " + dirname)
    rows = list(csv.DictReader(open(files[0])))
    lines = ["# rocprofv3 --kernel-trace --stats (source: %s)" % os.path.basename(files[0]), "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:12]:
        g = lambda *ks: next((r[k] for k in ks if k in r), "")
        lines.append("| %s | %s | %.3f | %.2f | %.2f | %.2f | %s |" % (
            g("Name", "Kernel_Name")[:110], g("Calls"), float(g("TotalDurationNs") or 0) / 1e6,
            float(g("AverageNs") or 0) / 1e3, float(g("MinNs") or 0) / 1e3, float(g("MaxNs") or 0) / 1e3,
            g("Percentage")))
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def pmc(dirs, out_json):
    per = defaultdict(lambda: defaultdict(list))     # label -> counter -> values
    for d in dirs:
        for f in find(d, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name", "")
                label = short(name) or ("copy" if ("direct_copy" in name or "elementwise_kernel" in name) else None)
                if label:
                    per[label][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"_units": "bytes per launch; raw = counter * 1024 (KiB); read_corrected = 2 * raw FETCH_SIZE "
                     "(gfx950 wide-stream under-report, MI355X_MICROARCH.md); write as reported",
           "_raw": {}}
    for label, counters in per.items():
        raw = {c: (sum(v) / len(v)) * 1024.0 for c, v in counters.items()}
        out["_raw"][label] = {c: round(x) for c, x in raw.items()}
        out["_raw"][label]["launches"] = {c: len(v) for c, v in counters.items()}
        rd = raw.get("FETCH_SIZE")
        wr = raw.get("WRITE_SIZE")
        if rd is not None and wr is not None:
            out[label] = round(2.0 * rd + wr)
            out["_raw"][label]["corrected_read"] = round(2.0 * rd)
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
