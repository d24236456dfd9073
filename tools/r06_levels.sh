#!/bin/bash
# far-field kernel: share of levels A / B / C and of the floor at 512^3 and 1024^3 on the structured scenes (profiling library;
# dc_debug 8 = no level-C scans, 24 = no B / C scans, 1 = no search; results of those runs are wrong by construction)
tag=${1:-r06f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_finish.py tests/test_gpu_envelope.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest.log | tee -a $O/summary.txt
export SDFGPU_LIB=$R/tools/probe/libsdfgpu_hooks.so
for n in 512 1024; do for st in 3 2; do for dbg in 0 8 24 1; do
  echo "== $n stage=$st dc_debug=$dbg" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py $n dc_debug=$dbg dc_debug_stage=$st 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): print('   %-14s build %.3f  y %.3f  x %.3f' % (k, v['ms_per_build'], v['stages_ms'].get('envelope_y',0), v['stages_ms'].get('envelope_x',0)))
" | tee -a $O/summary.txt
done; done; done
