#!/bin/bash
# two-valued tiles: the option off against on inside the shipped library (what a scene that never qualifies still pays)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r06fo}; mkdir -p $out
for sc in noisyfloor room twobox; do
  timeout 600 python tools/sweep_tier_ab.py A=sdf_tools_amd/libsdfgpu.so B=sdf_tools_amd/libsdfgpu.so n=512 scene=$sc reps=6 steps=40 dense=0 far_predict=2 A:flat_tiles=0 B:flat_tiles=1 > $out/ab_${sc}.txt 2>&1
  echo "== $sc 512: A = flat_tiles 0, B = flat_tiles 1" >> $out/summary.txt; tail -1 $out/ab_${sc}.txt | cut -c1-900 >> $out/summary.txt
done
for sc in noisyfloor room; do
  timeout 600 python tools/sweep_tier_ab.py A=tools/ab/libsdfgpu_r5.so B=sdf_tools_amd/libsdfgpu.so n=512 scene=$sc reps=6 steps=40 dense=0 far_predict=2 > $out/r5_${sc}.txt 2>&1
  echo "== $sc 512: A = round 5, B = HEAD" >> $out/summary.txt; tail -1 $out/r5_${sc}.txt | cut -c1-900 >> $out/summary.txt
done
timeout 900 python -m pytest tests/test_gpu_flat_tiles.py tests/test_gpu_plane_sparsity.py -x -q -m gpu > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/summary.txt
tail -3 $out/pytest.txt >> $out/summary.txt
