#!/bin/bash
tag=${1:-r06q}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_envelope.py tests/test_gpu_streaming.py tests/test_gpu_large.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
timeout 200 python tools/fuzz_parity.py 90 41 > $O/fuzz.log 2>&1; grep -a "fuzz OK\|MISMATCH\|GUARD" $O/fuzz.log | cut -c1-200 | tee -a $O/summary.txt
for sc in twobox room; do
  timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r5.so B=sdf_tools_amd/libsdfgpu.so scene=$sc reps=6 steps=30 dense=0 far_predict=2 > $O/ab_$sc.jsonl 2> $O/ab_$sc.err; tail -1 $O/ab_$sc.jsonl | tee -a $O/summary.txt
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_stream -o s --output-format csv -- python $R/bench_stream.py --frames 30 > $O/stats_stream.log 2>&1
cd $R; python tools/rocprof_summary.py stats $O/stats_stream $O/stream_kernel_stats.md; head -10 $O/stream_kernel_stats.md | cut -c1-160 | tee -a $O/summary.txt
