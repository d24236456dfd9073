#!/bin/bash
tag=${1:-r06z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt; grep -a "passed\|failed" $O/pytest_gpu.log | tail -2 | tee -a $O/summary.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -a "smoke" | tee -a $O/summary.txt
timeout 300 python bench_stream.py 2>&1 | grep -a "^{" | cut -c1-1500 | tee -a $O/summary.txt
( t0=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt; tail -1 $O/bench_driver_style.json | cut -c1-2500 | tee -a $O/summary.txt )
