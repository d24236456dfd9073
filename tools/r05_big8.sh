#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python bench.py --force-slab --steps 20 --no-cpu-baseline 2>/dev/null | head -1 > gpurun_out/r05x_slab.json
python -c "import json;d=json.loads(open('gpurun_out/r05x_slab.json').readline());print(d['value'], d.get('vs_1gpu_same_grid'))"
for v in 16 8; do
  echo "== dc_big_lines=$v"
  python tools/scene_bench.py 1024 dc_big_lines=$v 2>/dev/null | tee gpurun_out/r05x_big${v}.jsonl | cut -c1-420
done
