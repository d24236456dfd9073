#!/bin/bash
# trips of the far-field scan loops per level (profiling library with the trip counters), flat-stretch shortcut on / off
tag=${1:-r05d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
export SDFGPU_LIB=$R/sdf_tools_amd/libsdfgpu_trips.so
for sc in room boxes furniture stream 0.01; do
  for fl in 1 0; do
    echo -n "flat=$fl "; timeout 120 python tools/trip_counts.py $sc dc_flat=$fl 2>/dev/null | tee -a $O/trips.jsonl
  done
done
