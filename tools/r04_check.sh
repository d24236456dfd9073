#!/bin/bash
# Round-4 check on the GPU box (run from the repo root through gpurun): the GPU suite, the driver-style bench line, and the
# A/B of the headline step with the old (K12 + K3/16) and the new (far-field pair) stand-by behind the dense tier.
#   tools/r04_check.sh <tag>    -> gpurun_out/<tag>/
tag=${1:-r04a}; export TAG=$tag
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -s 2>&1 | tail -60 > $O/pytest.txt
tail -5 $O/pytest.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
for i in 1 2; do
  python bench.py --no-legs --no-cpu-baseline --steps 200 --warmup 10 > $O/bench_standby_far_$i.json 2>> $O/bench_ab.err
  python bench.py --no-legs --no-cpu-baseline --steps 200 --warmup 10 --opt standby_far=0 > $O/bench_standby_old_$i.json 2>> $O/bench_ab.err
done
python - <<'PY'
import json, glob, os
O = os.environ.get("O_DIR")
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ.get("TAG", "r04a"), "bench_*.json"))):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["value"], d["ms_per_step"], d.get("value_repeats", {}).get("median"), d.get("config", {}).get("guarded_general_pipeline_ms"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
