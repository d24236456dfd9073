#!/bin/bash
# x sweep of the far-field pair: what its floor is made of (profiling library; dc_debug 1 = no search, 3 = + no fp64 finish, 7 = + one store per chunk)
tag=${1:-r06xf}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
export SDFGPU_LIB=$R/tools/probe/libsdfgpu_hooks.so
for n in 512; do for dbg in 0 1 3 7 2; do
  echo "== $n stage=3 dc_debug=$dbg" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py $n dc_debug=$dbg dc_debug_stage=3 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): print('   %-14s build %.3f  y %.3f  x %.3f' % (k, v['ms_per_build'], v['stages_ms'].get('envelope_y',0), v['stages_ms'].get('envelope_x',0)))
" | tee -a $O/summary.txt
done; done
