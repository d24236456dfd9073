#!/bin/bash
# round 5's library (A) against HEAD (B) in one process, alternating repetitions: far-field scenes, the sweep tier, the dense tier
tag=${1:-r06h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_finish.py tests/test_gpu_envelope.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
for sc in twobox room; do
  timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r5.so B=sdf_tools_amd/libsdfgpu.so scene=$sc reps=6 steps=30 dense=0 envelope_mode=1 > $O/ab_$sc.jsonl 2> $O/ab_$sc.err; tail -1 $O/ab_$sc.jsonl | tee -a $O/summary.txt
done
timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r5.so B=sdf_tools_amd/libsdfgpu.so reps=6 steps=50 dense=0 > $O/ab_sweeps.jsonl 2> $O/ab_sweeps.err; tail -1 $O/ab_sweeps.jsonl | tee -a $O/summary.txt
timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r5.so B=sdf_tools_amd/libsdfgpu.so reps=6 steps=100 dense=1 > $O/ab_dense.jsonl 2> $O/ab_dense.err; tail -1 $O/ab_dense.jsonl | tee -a $O/summary.txt
