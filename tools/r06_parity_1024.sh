#!/bin/bash
# full-size parity of the structured scenes: round 5's library and HEAD in one process (the tool asserts that the two fields are bit-equal)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r06p1024}; mkdir -p $out
for sc in boxes shells spheres room; do
  timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r5.so B=sdf_tools_amd/libsdfgpu.so n=1024 scene=$sc reps=4 steps=10 dense=0 far_predict=2 > $out/ab_$sc.txt 2>&1
  echo "== $sc 1024 (bit-equal to round 5: $(grep -c disagree $out/ab_$sc.txt) disagreements)" >> $out/summary.txt; tail -1 $out/ab_$sc.txt | cut -c1-700 >> $out/summary.txt
done
