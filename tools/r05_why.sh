#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
cat > /tmp/why.py <<'PY'
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from sdf_tools_amd import capi, synth
n = 512; shape = (n, n, n); dev = torch.device("cuda", 0)
out = torch.empty(shape, dtype=torch.float32, device=dev); s = torch.cuda.current_stream().cuda_stream
for p in (0.015, 0.01, 0.007):
    m = synth.bernoulli_mask_torch(shape, p, 1, device=dev)
    ctx = capi.SdfGpu(0); ctx.set_option("dense_retry", 0)
    for i in range(3):
        ctx.set_option("dense3_mode", 1)
        ctx.get_stage_times(); ctx.set_profiling(1)
        ctx.build_device(m.data_ptr(), shape, out.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
        st, b = ctx.get_stage_times(); ctx.set_profiling(0)
        print(p, i, ctx.last_path(), [round(v, 3) for v in st])
    ctx.close()
PY
python /tmp/why.py 2>/dev/null
