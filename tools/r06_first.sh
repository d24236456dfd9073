#!/bin/bash
# round 6, first GPU call: the suite on the reworked host side, the r4-vs-HEAD A/B of the sweep tier, and what the
# room scene's pass-0 local search costs (profiling library, dc_debug bit 6 = no local-search rounds: wrong results, timing only)
tag=${1:-r06a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r4.so B=sdf_tools_amd/libsdfgpu.so reps=8 steps=50 > $O/sweep_tier_ab.jsonl 2> $O/sweep_tier_ab.err; tail -1 $O/sweep_tier_ab.jsonl | tee -a $O/summary.txt
export SDFGPU_LIB=$R/tools/probe/libsdfgpu_hooks.so
for st in 2 3; do for dbg in 0 64; do
  echo "== 512 room variants dc_debug=$dbg stage=$st" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py 512 --room-variants dc_debug=$dbg dc_debug_stage=$st 2>&1 | tee -a $O/room_ablate_512.jsonl | cut -c1-400 | tee -a $O/summary.txt
done; done
for st in 2 3; do for dbg in 0 64; do
  echo "== 1024 room variants dc_debug=$dbg stage=$st" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py 1024 --room-variants dc_debug=$dbg dc_debug_stage=$st 2>&1 | tee -a $O/room_ablate_1024.jsonl | cut -c1-400 | tee -a $O/summary.txt
done; done
