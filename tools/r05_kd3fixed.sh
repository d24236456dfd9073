#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
cat > /tmp/k3.py <<'PY'
import os, sys, json, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from sdf_tools_amd import capi, synth
n = 512; shape = (n, n, n); dev = torch.device("cuda", 0)
out = torch.empty(shape, dtype=torch.float32, device=dev); s = torch.cuda.current_stream().cuda_stream
for p in (0.1, 0.05, 0.03, 0.02, 0.015, 0.01):
    masks = [synth.bernoulli_mask_torch(shape, p, 1 + k, device=dev) for k in range(2)]
    row = {"p": p}
    for rep in range(2):
        for v in (0, 1):
            ctx = capi.SdfGpu(0); ctx.set_option("dense3_fixed", v)
            for i in range(30):
                ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40):
                ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s)
            torch.cuda.synchronize()
            row["fixed=%d #%d" % (v, rep)] = round((time.perf_counter() - t0) / 40 * 1e3, 3)
            row["sum%d" % v] = int(out.view(torch.int32).to(torch.int64).sum().item())
            ctx.close()
    print(json.dumps(row), flush=True)
PY
timeout 200 python /tmp/k3.py 2>/dev/null
