#!/bin/bash
tag=${1:-r06b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_bits.py tests/test_gpu_streaming.py -x -q > $O/pytest_bits.log 2>&1; echo "pytest bits rc=$?" | tee -a $O/summary.txt; tail -15 $O/pytest_bits.log | tee -a $O/summary.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout 300 python bench_stream.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_stream.log | cut -c1-1500 | tee -a $O/summary.txt
timeout 900 python bench.py --no-cpu-baseline --steps 100 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-3000 | tee -a $O/summary.txt
