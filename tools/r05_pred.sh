#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 120 python -m pytest tests/test_gpu_envelope.py -m gpu -x -q -k "prediction or transitions or standby" 2>&1 | tail -5
for v in 0 1 0 1; do
  echo "== far_predict=$v"
  timeout 100 python tools/scene_bench.py 512 far_predict=$v 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    for k, v in d.items(): print('  %-14s %.3f ms  %s  sum %d' % (k, v['ms_per_build'], v['stages_ms'], v['checksum']))
"
done
