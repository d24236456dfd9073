#!/bin/bash
# rows-per-chunk sweep of the marching kernels with rocprofv3 kernel times (run on the GPU box through gpurun)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for t in "$@"; do
  ty=${t%,*}; tx=${t#*,}
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats -d /tmp/tr -o s --output-format csv -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-legs --opt dense=0 --tune $ty $tx > /dev/null 2>&1
  python - "$ty" "$tx" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/tr/**/s_kernel_stats.csv", recursive=True)[0]
out = {}
for r in csv.DictReader(open(f)):
    for k in ("k_sweep_x16", "k_sweep_y16", "k_sweep_z_wave16"):
        if k in r["Name"]: out[k] = round(float(r["AverageNs"]) / 1e3, 1)
print("ty", sys.argv[1], "tx", sys.argv[2], out)
PY
done
