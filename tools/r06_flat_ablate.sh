#!/bin/bash
# two-valued tiles: where the path's own time goes (profiling library; results of the ablated runs are wrong by construction)
tag=${1:-r06fa}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
export SDFGPU_LIB=$R/tools/probe/libsdfgpu_hooks.so
for n in 512 1024; do for dbg in 0 128 256 512 768 1792 1; do
  echo "== $n stage=2 dc_debug=$dbg" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py $n dc_debug=$dbg dc_debug_stage=2 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items():
        if k == 'room': print('   %-14s build %.3f  y %.3f  x %.3f' % (k, v['ms_per_build'], v['stages_ms'].get('envelope_y',0), v['stages_ms'].get('envelope_x',0)))
" | tee -a $O/summary.txt
done; done
