#!/bin/bash
# two-valued tiles: phase clocks of the y sweep with the path on and off (profiling library)
tag=${1:-r06fc}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
export SDFGPU_LIB=$R/tools/probe/libsdfgpu_clocks.so
for k in room512 room1024; do for f in 1 0; do
  echo "== $k flat_tiles=$f" >> $O/summary.txt
  timeout 300 python tools/v3_clocks.py $k far_predict=2 flat_tiles=$f 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/summary.txt
done; done
