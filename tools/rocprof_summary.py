#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (run on the GPU box) into the small files committed under profiles/.

  python tools/rocprof_summary.py stats  <dir> <out.md>          kernel-trace --stats summary
  python tools/rocprof_summary.py pmc    <dir>... <out.json>     FETCH_SIZE / WRITE_SIZE per launch, by exact kernel
  python tools/rocprof_summary.py times  <dir> <out.json>        average ms per exact kernel (guard exits left out)

The PMC corrections follow MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE
under-reports a wide coalesced streaming read by exactly 2x; WRITE_SIZE is uncalibrated, so the
calibration kernel (a torch elementwise copy of known size recorded in the same pass) is used to
derive the factor actually observed here.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def find(dirname, pattern):
    return sorted(glob.glob(os.path.join(dirname, "**", pattern), recursive=True))


KERNEL_RE = re.compile(r"(?:void\s+)?(?:sdfgpu::)?(k_[A-Za-z0-9_]+)\s*(?:<([^>]*)>)?")


def exact(name):
    """Exact kernel identity from a (demangled) rocprofv3 kernel name: the function's base name -- `k_ball_dense` and
    `k_ball_dense3` are different kernels (round 3's substring match folded them, and the guarded no-op launches of one
    into the average of the other) -- plus the stage template argument of the two kernels that are one function template
    for two sweeps (k_envelope_dc<2|3, ...>, k_sweep_march<2|3, ...>)."""
    m = KERNEL_RE.search(name)
    if not m:
        if "direct_copy" in name or "elementwise_kernel" in name:
            return "copy"
        return None
    base, targs = m.group(1), (m.group(2) or "")
    if base in ("k_envelope_dc", "k_sweep_march", "k_probe_window"):
        first = targs.split(",")[0].strip()
        return "%s<%s>" % (base, first) if first else base
    return base


# a launch that returned on its guard moves (next to) nothing: it is not a sample of the kernel's traffic
GUARD_EXIT_FRACTION = 0.05


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
    per = defaultdict(lambda: defaultdict(list))     # exact kernel -> counter -> values of every launch
    for d in dirs:
        for f in find(d, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                label = exact(r.get("Kernel_Name", ""))
                if label:
                    per[label][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"_units": "bytes per launch; raw = counter * 1024 (KiB); read_corrected = 2 * raw FETCH_SIZE "
                     "(gfx950 wide-stream under-report, MI355X_MICROARCH.md); write as reported.  Kernels by EXACT name; "
                     "launches whose counter is below %g of the kernel's largest launch are guard exits (the kernel "
                     "returned on its guard word) and are counted, not averaged" % GUARD_EXIT_FRACTION,
           "_raw": {}}
    for label, counters in sorted(per.items()):
        raw, info = {}, {}
        for c, v in counters.items():
            top = max(v)
            work = [x for x in v if x >= GUARD_EXIT_FRACTION * top] if top > 0 else v
            raw[c] = (sum(work) / len(work)) * 1024.0
            info[c] = {"launches": len(v), "worked": len(work), "guard_exits": len(v) - len(work)}
        out["_raw"][label] = {c: round(x) for c, x in raw.items()}
        out["_raw"][label]["launches"] = info
        rd = raw.get("FETCH_SIZE")
        wr = raw.get("WRITE_SIZE")
        if rd is not None and wr is not None:
            out[label] = round(2.0 * rd + wr)
            out["_raw"][label]["corrected_read"] = round(2.0 * rd)
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


def kernel_times(stats_csv, out_json):
    """{exact kernel: average ms over the launches that did work} from a rocprofv3 --kernel-trace --stats CSV.  The stats
    file averages ALL launches of a name; where guard exits are mixed in (min << avg) the trace CSV next to it is used."""
    out = {}
    trace = stats_csv.replace("kernel_stats.csv", "kernel_trace.csv")
    if os.path.exists(trace):
        per = defaultdict(list)
        for r in csv.DictReader(open(trace)):
            label = exact(r.get("Kernel_Name", ""))
            if label and label != "copy":
                per[label].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for label, v in per.items():
            top = max(v)
            work = [x for x in v if x >= GUARD_EXIT_FRACTION * top]
            out[label] = {"avg_ms": round(sum(work) / len(work) / 1e6, 5), "launches": len(v), "worked": len(work)}
    else:
        for r in csv.DictReader(open(stats_csv)):
            label = exact(r.get("Name") or r.get("KernelName") or "")
            if label and label != "copy" and label not in out:
                out[label] = {"avg_ms": round(float(r.get("AverageNs") or 0.0) / 1e6, 5), "launches": int(r.get("Calls") or 0),
                              "worked": None}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "times":
        files = find(sys.argv[2], "*kernel_stats.csv")
        if not files:
            raise SystemExit("no *kernel_stats.csv under " + sys.argv[2])
        kernel_times(files[0], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
