#!/bin/bash
tag=${1:-r06v}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt; grep -a "passed\|failed" $O/pytest_gpu.log | tail -2 | tee -a $O/summary.txt
( export SDFGPU_REDZONE=1; timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_redzone.log 2>&1; echo "SDFGPU_REDZONE=1 pytest -m gpu rc=$?" | tee -a $O/summary.txt; grep -a "passed\|failed" $O/pytest_gpu_redzone.log | tail -2 | tee -a $O/summary.txt )
( export SDFGPU_REDZONE=1 FUZZ_GUARD=1; timeout 500 python tools/fuzz_parity.py 300 31 > $O/fuzz_redzone.log 2>&1; echo "fuzz under red zones rc=$?" | tee -a $O/summary.txt; grep -a "fuzz OK\|MISMATCH\|GUARD\|red zone" $O/fuzz_redzone.log | cut -c1-300 | tee -a $O/summary.txt )
for ps in 1 0; do echo "== scene_bench 1024 plane_skip=$ps" | tee -a $O/summary.txt; timeout 300 python tools/scene_bench.py 1024 plane_skip=$ps 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): print('   %-14s build %.3f  z %.3f y %.3f  x %.3f  checksum %d' % (k, v['ms_per_build'], v['stages_ms'].get('sweep_z',0), v['stages_ms'].get('envelope_y',0), v['stages_ms'].get('envelope_x',0), v['checksum']))
" | tee -a $O/summary.txt; done
