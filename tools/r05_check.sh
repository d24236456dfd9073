#!/bin/bash
# Round-5 GPU check (run from the repo root through gpurun): the GPU suite, the bench line (with host_api.class_seam), the
# slab bench with the multi-rank host-time block, the structured scenes at 512^3 and 1024^3.
#   tools/r05_check.sh <tag>    -> gpurun_out/<tag>/
tag=${1:-r05}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150 > $O/tests.txt; tail -3 $O/tests.txt
SDFGPU_HOST_TIMING=1 timeout 300 ./examples/class_seam_example 512 3 0.5 > $O/class_seam.txt 2>&1; tail -8 $O/class_seam.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d = [json.loads(l) for l in open("$O/bench.json") if l.startswith("{")][0]
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d.get("roofline", {}).get("frac"), "host_api", json.dumps(d.get("host_api"))[:900])
PY
timeout 600 python bench.py --force-slab --steps 20 --no-cpu-baseline > $O/bench_slab.json 2> $O/bench_slab.err; python - <<PY
import json
d = [json.loads(l) for l in open("$O/bench_slab.json") if l.startswith("{")][0]
print("slab value", d["value"], json.dumps(d.get("multi_native_on_one_gpu"))[:2500])
PY
timeout 300 python tools/scene_bench.py 512 > $O/scene512.jsonl 2>&1; cat $O/scene512.jsonl | cut -c1-400
timeout 600 python tools/scene_bench.py 1024 > $O/scene1024.jsonl 2>&1; cat $O/scene1024.jsonl | cut -c1-400
