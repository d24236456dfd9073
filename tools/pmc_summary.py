#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: python tools/pmc_summary.py <dir>... [filter]"""
import collections, csv, glob, sys
dirs = [a for a in sys.argv[1:] if "/" in a or a.startswith("gpurun")]
flt = [a for a in sys.argv[1:] if a not in dirs]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").replace("sdfgpu::", "")[:40]
            if flt and not any(x in name for x in flt):
                continue
            per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[name]["_VGPR"] = [float(r["VGPR_Count"])]
for k, v in per.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
