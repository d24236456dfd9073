#!/bin/bash
# Far-field tier check on the GPU box (run from the repo root through gpurun): correctness tests of the far-field kernels,
# stage times on the streaming scene and three Bernoulli densities, and the SQ instruction counters of the streaming run.
#   tools/ff_check.sh <tag> [name=value ...]    -> gpurun_out/<tag>/
tag=${1:-ff}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_envelope.py tests/test_gpu_streaming.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
for sc in stream 0.03 0.01 0.001 0.0001; do
  if [ $sc = stream ]; then a=""; else a="--bernoulli=$sc"; fi
  echo -n "$sc " ; python tools/env_bench.py 512 10 $a "$@" 2>/dev/null | tee $O/env_$sc.json
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_SQ -o p --output-format csv -- python $R/tools/env_bench.py 512 4 "$@" > $O/pmc_SQ.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY -d $O/pmc_SQ2 -o p --output-format csv -- python $R/tools/env_bench.py 512 4 "$@" > $O/pmc_SQ2.log 2>&1
cd $R; python tools/pmc_summary.py gpurun_out/$tag/pmc_SQ gpurun_out/$tag/pmc_SQ2 envelope | tee $O/sq_counters.txt
