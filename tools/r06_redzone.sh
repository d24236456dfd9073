#!/bin/bash
# round 6: the whole GPU suite and >= 10 minutes of guarded fuzz under red zones (VERDICT r5 "next round" 2)
tag=${1:-r06d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_redzone.py -x -q > $O/pytest_redzone.log 2>&1; echo "pytest redzone tests rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest_redzone.log | tee -a $O/summary.txt
( export SDFGPU_REDZONE=1; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_under_redzones.log 2>&1; echo "SDFGPU_REDZONE=1 pytest -m gpu rc=$?" | tee -a $O/summary.txt; tail -8 $O/pytest_gpu_under_redzones.log | tee -a $O/summary.txt )
( export SDFGPU_REDZONE=1 FUZZ_GUARD=1; timeout 800 python tools/fuzz_parity.py ${FUZZ_SECONDS:-620} 11 2>&1 | grep -v amdgpu.ids | tail -5 | tee -a $O/summary.txt )
timeout 300 python tools/fuzz_parity.py 120 12 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/summary.txt
