#!/bin/bash
# round 6, two-valued tiles: the whole GPU suite, then the fuzz (with its floor scenes) on three seeds, one of them with red zones
tag=${1:-r06ff}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest.log | tee -a $O/summary.txt
for seed in 601 602; do
  timeout 400 python tools/fuzz_parity.py 240 $seed > $O/fuzz_$seed.log 2>&1; grep -a "fuzz OK\|MISMATCH\|GUARD" $O/fuzz_$seed.log | cut -c1-260 | tee -a $O/summary.txt
done
SDFGPU_REDZONE=1 timeout 400 python tools/fuzz_parity.py 240 603 > $O/fuzz_603.log 2>&1; grep -a "fuzz OK\|MISMATCH\|GUARD" $O/fuzz_603.log | cut -c1-260 | tee -a $O/summary.txt
