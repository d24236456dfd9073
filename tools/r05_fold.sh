#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for o in 1 0; do
  echo "== standby_fold=$o"
  python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-legs --opt standby_fold=$o 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('value_repeats'))"
  python bench.py --size 256 256 256 --steps 400 --warmup 30 --no-cpu-baseline --no-legs --opt standby_fold=$o 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('256^3', d['value'], d['ms_per_step'], d.get('value_repeats'))"
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
