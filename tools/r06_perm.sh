#!/bin/bash
# far-field kernel: tiles in a scattered order (option dc_tile_perm: bit 0 y sweep, bit 1 x sweep) -- A/B
tag=${1:-r06g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_envelope.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
for n in 512 1024; do for rep in 1 2; do for pm in 0 1 2 3; do
  echo "== $n dc_tile_perm=$pm" | tee -a $O/summary.txt
  timeout 300 python tools/scene_bench.py $n dc_tile_perm=$pm 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): print('   %-14s build %.3f  y %.3f  x %.3f  checksum %d' % (k, v['ms_per_build'], v['stages_ms'].get('envelope_y',0), v['stages_ms'].get('envelope_x',0), v['checksum']))
" | tee -a $O/summary.txt
done; done; done
