#!/bin/bash
# two-valued tiles tried on ANY far-field scene (not only floor-like grids): flat_tiles 1 (floor-like only) vs 3 (any scene, habit) vs 4 (any scene, forced)
tag=${1:-r06fany}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for n in 512 1024; do for f in 1 3 4; do
  echo "== $n flat_tiles=$f" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py $n flat_tiles=$f 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): print('   %-14s build %.3f  y %.3f  x %.3f  checksum %d' % (k, v['ms_per_build'], v['stages_ms'].get('envelope_y',0), v['stages_ms'].get('envelope_x',0), v['checksum']))
" | tee -a $O/summary.txt
done; done
for f in 1 3; do
  timeout 600 python tools/sweep_tier_ab.py A=sdf_tools_amd/libsdfgpu.so B=sdf_tools_amd/libsdfgpu.so n=512 scene=twobox reps=6 steps=40 dense=0 far_predict=2 A:flat_tiles=0 B:flat_tiles=$f 2>&1 | tail -1 | cut -c1-700 | tee -a $O/summary.txt
done
