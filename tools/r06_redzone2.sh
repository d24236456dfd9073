#!/bin/bash
# guarded fuzz under red zones (full log), the multi library's host time per API call at 1 / 2 / 8 logical ranks, the class seam with and without SDF_TOOLS_VECTOR_ADOPT
tag=${1:-r06e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
( export SDFGPU_REDZONE=1 FUZZ_GUARD=1; timeout 800 python tools/fuzz_parity.py ${FUZZ_SECONDS:-620} 11 > $O/fuzz_redzone.log 2>&1; echo "fuzz under red zones rc=$?" | tee -a $O/summary.txt; grep -a "fuzz OK\|MISMATCH\|GUARD\|red zone\|Error" $O/fuzz_redzone.log | cut -c1-500 | tee -a $O/summary.txt )
timeout 300 python tools/fuzz_parity.py 120 12 > $O/fuzz_plain.log 2>&1; echo "plain fuzz rc=$?" | tee -a $O/summary.txt; grep -a "fuzz OK\|MISMATCH\|GUARD\|Error" $O/fuzz_plain.log | cut -c1-500 | tee -a $O/summary.txt
# class seam: the opt-in vector adoption against the safe default
g++ -O2 -std=c++17 -pthread -I include examples/class_seam.cpp -o $O/class_seam_safe -L sdf_tools_amd -lsdfgpu -Wl,-rpath,$R/sdf_tools_amd -lz 2> $O/class_seam_safe.err
g++ -O2 -std=c++17 -pthread -DSDF_TOOLS_VECTOR_ADOPT -I include examples/class_seam.cpp -o $O/class_seam_adopt -L sdf_tools_amd -lsdfgpu -Wl,-rpath,$R/sdf_tools_amd -lz 2>> $O/class_seam_safe.err
for v in safe adopt safe adopt; do echo -n "class_seam $v: " | tee -a $O/summary.txt; timeout 300 $O/class_seam_$v 512 4 0.5 2>/dev/null | grep "^{" | tail -1 | cut -c1-300 | tee -a $O/summary.txt; done
rm -f $O/class_seam_safe $O/class_seam_adopt
# where a rank thread's host time goes when logical ranks share one GPU: HIP API trace (no counters)
cd /tmp; export TMPDIR=/tmp
for ranks in 1 2 8; do
  timeout 300 rocprofv3 --hip-runtime-trace --stats -d $O/hiptrace_$ranks -o t --output-format csv -- python $R/tools/multi_host_trace.py $ranks > $O/hiptrace_$ranks.log 2>&1
  echo "== HIP API stats, $ranks logical rank(s) on one GPU" | tee -a $O/summary.txt
  grep -a "^{" $O/hiptrace_$ranks.log | tail -1 | cut -c1-400 | tee -a $O/summary.txt
  f=$(find $O/hiptrace_$ranks -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 | tee -a $O/summary.txt
done
