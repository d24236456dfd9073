#!/bin/bash
# Collects the round's evidence on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
# bench line, kernel-trace stats of the bench and of the streaming / far-field workloads, PMC passes (HBM traffic and SQ
# counters; counters are collected in their own runs, with --kernel-trace only).
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python bench_stream.py > $O/stream_bench.json 2> $O/stream_bench.err
python bench.py --force-slab --no-cpu-baseline --no-legs --steps 100 > $O/bench_slab_world1.json 2> $O/bench_slab.err
timeout 400 python tools/p_sweep.py > $O/psweep.jsonl 2> $O/psweep.err
timeout 300 python tools/scene_bench.py 512 2>/dev/null | grep '^{' > $O/scene_bench.jsonl
timeout 300 python tools/scene_bench.py 1024 2>/dev/null | grep '^{' > $O/scene_bench_1024.jsonl
( t0=$(date +%s); python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench wall $(( $(date +%s) - t0 )) s" >> $O/bench_driver_style.err )
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_dense -o s --output-format csv -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-legs > $O/stats_dense.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_stream -o s --output-format csv -- python $R/bench_stream.py --frames 30 > $O/stats_stream.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_general -o s --output-format csv -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-legs --opt dense=0 > $O/stats_general.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_dense_$c -o p --output-format csv -- python $R/tools/pmc_workload.py 512 > $O/pmc_dense_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_general_$c -o p --output-format csv -- python $R/tools/pmc_workload.py 512 dense=0 > $O/pmc_general_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_mid_$c -o p --output-format csv -- python $R/tools/pmc_workload.py 512 p=0.03 builds=8 > $O/pmc_mid_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_env_$c -o p --output-format csv -- python $R/tools/env_bench.py 512 4 > $O/pmc_env_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_env_SQ -o p --output-format csv -- python $R/tools/env_bench.py 512 4 > $O/pmc_env_SQ.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY -d $O/pmc_env_SQ2 -o p --output-format csv -- python $R/tools/env_bench.py 512 4 > $O/pmc_env_SQ2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_general_SQ -o p --output-format csv -- python $R/tools/pmc_workload.py 512 dense=0 > $O/pmc_general_SQ.log 2>&1
ls $O
