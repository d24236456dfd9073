#!/bin/bash
# Instruction / time breakdown of the far-field kernels by ablation (profiling library built with -DSDFGPU_DEBUG_HOOKS:
# tools/probe/libsdfgpu_hooks.so; results of the ablated runs are WRONG by construction).  Run from the repo root via gpurun:
#   tools/ke_ablate.sh <tag> [env_bench args]     -> gpurun_out/<tag>/
# An ablated y sweep hands garbage (no sites) to the x sweep, whose levels A and B then do not run at all: the x sweep's numbers
# are only meaningful with dc_debug_stage=3 (and the y sweep's with dc_debug_stage=2 or 0).
tag=${1:-ke}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
export SDFGPU_LIB=$R/tools/probe/libsdfgpu_hooks.so
cd /tmp; export TMPDIR=/tmp
for dbg in ${KE_ABLATE_SET:-0 8 24 1 3 7}; do     # 8: no level-C scan; 24: no level-B / C scans; 1: no search; 3: + no fp64 finish; 7: + no stores
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d $O/dbg_$dbg -o p --output-format csv -- python $R/tools/env_bench.py 512 4 dc_debug=$dbg "$@" > $O/dbg_$dbg.log 2>&1
  echo "== dc_debug=$dbg" | tee -a $O/summary.txt
  (cd $R; python tools/pmc_summary.py gpurun_out/$tag/dbg_$dbg k_envelope | tee -a $O/summary.txt)
  python - $O/dbg_$dbg <<'PY' | tee -a $O/summary.txt
import csv, glob, sys, collections
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_envelope_dc" in n:
            per[n.split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
    v = [x for x in v if x > 0.05 * max(v)]
    print("  time", k, "avg %.1f us over %d" % (sum(v) / len(v), len(v)))
PY
done
