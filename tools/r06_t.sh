#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06t; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_plane_sparsity.py -x -q 2>&1 | tail -15 | tee $O/summary.txt
