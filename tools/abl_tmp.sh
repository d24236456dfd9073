cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r02e
for d in 0 8 9 11 15 31; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $R/gpurun_out/r02e/pmc_d$d -o p --output-format csv -- python $R/tools/env_bench.py 512 3 dc_debug=$d > $R/gpurun_out/r02e/pmc_d$d.log 2>&1
  python $R/tools/env_bench.py 512 10 dc_debug=$d > $R/gpurun_out/r02e/env_d$d.json 2>/dev/null; echo "dbg=$d $(cat $R/gpurun_out/r02e/env_d$d.json)"
done
