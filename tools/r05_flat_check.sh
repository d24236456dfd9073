#!/bin/bash
# Round-5 far-field A/B on the GPU box: the shipped library (flat-stretch shortcut) against sdf_tools_amd/libsdfgpu_base.so (the
# same sources with round 4's sdfgpu_envelope_dc.hpp), interleaved on one box.
#   tools/r05_flat_check.sh <tag> [fuzz seconds]    -> gpurun_out/<tag>/
tag=${1:-r05b}; fz=${2:-60}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_envelope.py tests/test_gpu_streaming.py tests/test_gpu_large.py tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_multi.py -q --tb=short 2>&1 | tail -60 > $O/tests.txt; tail -3 $O/tests.txt
timeout 400 python tools/fuzz_parity.py $fz 11 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
for rep in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export SDFGPU_LIB=$R/sdf_tools_amd/libsdfgpu_base.so; else unset SDFGPU_LIB; fi
    echo "== $lib $rep"
    timeout 300 python tools/scene_bench.py 512 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = list(d)[0]; v = d[k]
    print('%-14s %.3f ms  %s' % (k, v['ms_per_build'], {a: v['stages_ms'][a] for a in ('sweep_z', 'envelope_y', 'envelope_x') if a in v['stages_ms']}))
" | tee -a $O/scene512_$lib.txt
    for sc in stream 0.01; do
      if [ $sc = stream ]; then a=""; else a="--bernoulli=$sc"; fi
      echo -n "env $sc " ; timeout 200 python tools/env_bench.py 512 10 $a 2>/dev/null | tee -a $O/env_${sc}_$lib.json
    done
  done
done
for lib in new base; do
  if [ $lib = base ]; then export SDFGPU_LIB=$R/sdf_tools_amd/libsdfgpu_base.so; else unset SDFGPU_LIB; fi
  echo "== 1024 $lib"
  timeout 400 python tools/scene_bench.py 1024 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = list(d)[0]; v = d[k]
    print('%-14s %.3f ms  %s' % (k, v['ms_per_build'], {a: v['stages_ms'][a] for a in ('sweep_z', 'envelope_y', 'envelope_x') if a in v['stages_ms']}))
" | tee -a $O/scene1024_$lib.txt
done
unset SDFGPU_LIB
cd /tmp; export TMPDIR=/tmp
for mode in multi single; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o p --output-format csv -- python $R/tools/multi_far_probe.py $mode 10 > $O/probe_$mode.log 2>&1
  tail -1 $O/probe_$mode.log
  cd $R; python tools/rocprof_summary.py stats gpurun_out/$tag/prof_$mode $O/stats_$mode.md > /dev/null 2>&1; head -16 $O/stats_$mode.md | cut -c1-150; cd /tmp
done
rm -rf $O/prof_multi $O/prof_single
