#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 1; do
  echo "== dc_fixed=$v"
  timeout 120 python tools/scene_bench.py 1024 dc_fixed=$v 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    for k, v in d.items(): print('  %-14s %.3f ms  y %.3f x %.3f  sum %d' % (k, v['ms_per_build'], v['stages_ms'].get('envelope_y', 0), v['stages_ms'].get('envelope_x', 0), v['checksum']))
"
done
timeout 200 python -u -m pytest tests/test_gpu_large.py -m gpu -x -q -k "1024" 2>&1 | tail -2
