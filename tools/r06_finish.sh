#!/bin/bash
# the fp64-free finish of the far-field x sweep: exhaustive device test, then A/B in one process order (option fast_finish)
tag=${1:-r06c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_finish.py tests/test_gpu_envelope.py tests/test_gpu_large.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
for rep in 1 2; do for ff in 1 0; do
  echo "== scene_bench 512 fast_finish=$ff" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py 512 fast_finish=$ff 2>&1 | grep -v amdgpu.ids | tee -a $O/scene_512_ff$ff.jsonl | cut -c1-330 | tee -a $O/summary.txt
  echo "== env_bench two-box 512 fast_finish=$ff" | tee -a $O/summary.txt
  timeout 300 python tools/env_bench.py 512 20 fast_finish=$ff 2>&1 | grep -v amdgpu.ids | tee -a $O/env_512_ff$ff.jsonl | tee -a $O/summary.txt
done; done
for ff in 1 0; do
  echo "== scene_bench 1024 fast_finish=$ff" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py 1024 fast_finish=$ff 2>&1 | grep -v amdgpu.ids | tee -a $O/scene_1024_ff$ff.jsonl | cut -c1-330 | tee -a $O/summary.txt
done
