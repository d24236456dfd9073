#!/bin/bash
# Condenses gpurun_out/<tag> (written by tools/profile_round.sh on the GPU box) into the tracked profiles/<name>_* files:
#   tools/condense_round.sh <tag> <name>          e.g.  tools/condense_round.sh r02u r02
set -e
tag=$1; name=$2
G=gpurun_out/$tag; P=profiles
# (round 6: bench.py prints the full detail as "BENCH_FULL {...}" and the contract line behind it)
tail -1 $G/bench.json > $P/${name}_bench.json
grep -a '^BENCH_FULL ' $G/bench.json | tail -1 | cut -c12- > $P/${name}_bench_full.json
[ -f $G/bench_driver_style.json ] && tail -1 $G/bench_driver_style.json > $P/${name}_bench_driver_style.json
[ -f $G/scene_bench.jsonl ] && cp $G/scene_bench.jsonl $P/${name}_scene_bench.jsonl
[ -f $G/scene_bench_1024.jsonl ] && cp $G/scene_bench_1024.jsonl $P/${name}_scene_bench_1024.jsonl
cp $G/stream_bench.json $P/${name}_stream_bench.json
grep -a '^{' $G/bench_slab_world1.json | tail -1 > $P/${name}_bench_slab_world1.json      # (RCCL prints its banner on stdout behind the line)
[ -f $G/psweep.jsonl ] && cp $G/psweep.jsonl $P/${name}_p_sweep.jsonl
python tools/rocprof_summary.py stats $G/stats_dense $P/${name}_dense_kernel_stats.md
python tools/rocprof_summary.py stats $G/stats_stream $P/${name}_stream_kernel_stats.md
python tools/rocprof_summary.py stats $G/stats_general $P/${name}_general_kernel_stats.md
for f in dense stream; do
  s=$(find $G/stats_$f -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp $s $P/${name}_${f}_kernel_stats.csv
done
python tools/rocprof_summary.py pmc $G/pmc_dense_FETCH_SIZE $G/pmc_dense_WRITE_SIZE $P/${name}_pmc_traffic_dense.json
python tools/rocprof_summary.py pmc $G/pmc_general_FETCH_SIZE $G/pmc_general_WRITE_SIZE $P/${name}_pmc_traffic_general.json
python tools/rocprof_summary.py pmc $G/pmc_env_FETCH_SIZE $G/pmc_env_WRITE_SIZE $P/${name}_pmc_traffic_envelope.json
python tools/pmc_summary.py $G/pmc_env_SQ $G/pmc_env_SQ2 k_envelope k_sweep_z > $P/${name}_sq_counters_envelope.txt
[ -d $G/pmc_general_SQ ] && python tools/pmc_summary.py $G/pmc_general_SQ $G/pmc_general_SQ k_sweep_x16 k_sweep_y16 k_sweep_z > $P/${name}_sq_counters_general.txt
python tools/rocprof_summary.py pmc $G/pmc_mid_FETCH_SIZE $G/pmc_mid_WRITE_SIZE $P/${name}_pmc_traffic_mid.json
# average kernel times of the bench command (rocprofv3 --kernel-trace --stats; exact kernel names, guard exits left out), read by
# bench.py next to its HIP-event times; and the HBM bytes per launch of every kernel from THIS round's PMC files only
python tools/rocprof_summary.py times $G/stats_dense $P/kernel_times.json
python tools/make_traffic.py $name
