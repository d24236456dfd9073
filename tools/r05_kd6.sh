#!/bin/bash
# KD6 check: dense-tier tests + fuzz, then the density sweep with the shell pass on / off
tag=${1:-r05f}; fz=${2:-45}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_parity.py -q --tb=short 2>&1 | tail -40 > $O/tests.txt; tail -3 $O/tests.txt
timeout 300 python tools/fuzz_parity.py $fz 21 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
cat > /tmp/ps.py <<'PY'
import json, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from sdf_tools_amd import capi, synth
n = 512; shape = (n, n, n); dev = torch.device("cuda", 0)
out = torch.empty(shape, dtype=torch.float32, device=dev); s = torch.cuda.current_stream().cuda_stream
for p in (0.03, 0.02, 0.015, 0.01, 0.007, 0.005, 0.003):
    masks = [synth.bernoulli_mask_torch(shape, p, 1 + k, device=dev) for k in range(2)]
    for shell in (1, 0):
        ctx = capi.SdfGpu(0); ctx.set_option("dense_shell", shell)
        for i in range(40):
            ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        ctx.get_stage_times(); ctx.set_profiling(1)
        for i in range(8):
            ctx.build_device(masks[i % 2].data_ptr(), shape, out.data_ptr(), 0.01, False, s)
        torch.cuda.synchronize()
        st, b = ctx.get_stage_times(); ctx.set_profiling(0)
        names = ["pack", "ball", "z", "y", "env_y", "x", "env_x"]
        print(json.dumps({"p": p, "shell": shell, "ms": round(ms, 3), "cert": ctx.last_path().get("dense_certified"), "why": ctx.last_path().get("dense_gave_up"),
                          "stages": {k: round(v / b, 3) for k, v in zip(names, st) if v / b > 0.002}}), flush=True)
        ctx.close()
PY
timeout 600 python /tmp/ps.py 2>/dev/null | tee $O/psweep_shell.jsonl
