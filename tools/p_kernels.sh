#!/bin/bash
# per-kernel times (rocprofv3) of the default build at given Bernoulli densities: tools/p_kernels.sh 0.04 0.03 ...
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for p in "$@"; do
  rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats -d /tmp/pk -o s --output-format csv -- python $R/bench.py --steps 30 --warmup 20 --no-cpu-baseline --no-legs --p $p > /tmp/pk.log 2>&1
  tail -1 /tmp/pk.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p', '$p', 'ms', d['ms_per_step'])"
  python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pk/**/s_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "sdfgpu" in r["Name"]: print("   %-62s calls %4s avg %8.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
