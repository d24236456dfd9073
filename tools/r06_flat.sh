#!/bin/bash
# round 6: flat-positive tiles -- parity, then A/B against round 5's library and against the option off
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r06flat}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_flat_tiles.py tests/test_gpu_plane_sparsity.py -x -q -m gpu > $out/pytest.txt 2>&1; echo "pytest rc=$?" > $out/summary.txt
tail -5 $out/pytest.txt >> $out/summary.txt
for n in 512 1024; do
  for sc in room twobox noisyfloor; do
    reps=6; steps=40; [ $n = 1024 ] && steps=10
    timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r5.so B=sdf_tools_amd/libsdfgpu.so n=$n scene=$sc reps=$reps steps=$steps dense=0 far_predict=2 > $out/ab_${sc}_$n.txt 2>&1
    echo "== $sc $n vs r5" >> $out/summary.txt; tail -1 $out/ab_${sc}_$n.txt | cut -c1-900 >> $out/summary.txt
  done
done
