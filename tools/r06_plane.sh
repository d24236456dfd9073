#!/bin/bash
# plane sparsity (option plane_skip): tests, fuzz, A/B against round 5's library on far-field scenes, the streaming bench
tag=${1:-r06j}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_envelope.py tests/test_gpu_streaming.py tests/test_gpu_large.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout 400 python tools/fuzz_parity.py 240 21 > $O/fuzz.log 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt; grep -a "fuzz OK\|MISMATCH\|GUARD" $O/fuzz.log | cut -c1-300 | tee -a $O/summary.txt
for sc in twobox room; do
  timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r5.so B=sdf_tools_amd/libsdfgpu.so scene=$sc reps=6 steps=30 dense=0 far_predict=2 > $O/ab_$sc.jsonl 2> $O/ab_$sc.err; tail -1 $O/ab_$sc.jsonl | tee -a $O/summary.txt
done
for ps in 1 0 1 0; do echo "== scene_bench 512 plane_skip=$ps" | tee -a $O/summary.txt; timeout 300 python tools/scene_bench.py 512 plane_skip=$ps 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): print('   %-14s build %.3f  z %.3f y %.3f  x %.3f  checksum %d' % (k, v['ms_per_build'], v['stages_ms'].get('sweep_z',0), v['stages_ms'].get('envelope_y',0), v['stages_ms'].get('envelope_x',0), v['checksum']))
" | tee -a $O/summary.txt; done
timeout 300 python bench_stream.py 2>&1 | grep -a "^{" | cut -c1-1200 | tee -a $O/summary.txt
