#!/bin/bash
# SQ counters + kernel times of the dense tier's wide form (KD3 + the fix-up kernel KF) on Bernoulli scenes at 512^3
# (run from the repo root through gpurun):   tools/kf_counters.sh <tag> [p ...]   -> gpurun_out/<tag>/
tag=${1:-kf}; shift
ps=${@:-0.02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for p in $ps; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$p -o s --output-format csv -- python $R/tools/pmc_workload.py 512 p=$p builds=8 > $O/stats_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/sq_$p -o p --output-format csv -- python $R/tools/pmc_workload.py 512 p=$p builds=8 > $O/sq_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_ANY -d $O/sq2_$p -o p --output-format csv -- python $R/tools/pmc_workload.py 512 p=$p builds=8 > $O/sq2_$p.log 2>&1
  cd $R
  echo "== p=$p" | tee -a $O/summary.txt
  python tools/rocprof_summary.py times $O/stats_$p $O/times_$p.json | tee -a $O/summary.txt
  python tools/pmc_summary.py gpurun_out/$tag/sq_$p gpurun_out/$tag/sq2_$p k_ball | tee -a $O/summary.txt
  cd /tmp
done
