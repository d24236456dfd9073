#!/bin/bash
tag=${1:-r05g}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cat > /tmp/w.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from sdf_tools_amd import capi, synth
p = float(sys.argv[1]); n = 512; shape = (n, n, n); dev = torch.device("cuda", 0)
out = torch.empty(shape, dtype=torch.float32, device=dev); s = torch.cuda.current_stream().cuda_stream
m = synth.bernoulli_mask_torch(shape, p, 1, device=dev)
ctx = capi.SdfGpu(0); ctx.set_option("dense_retry", 0)
for i in range(8):
    ctx.set_option("dense3_mode", 1)
    ctx.build_device(m.data_ptr(), shape, out.data_ptr(), 0.01, False, s); torch.cuda.synchronize()
print(p, ctx.last_path())
PY
cd /tmp; export TMPDIR=/tmp
for p in 0.015 0.01; do
  rocprofv3 --kernel-trace --stats -d $O/st_$p -o s --output-format csv -- python /tmp/w.py $p > $O/w_$p.log 2>&1
  grep "dense_certified" $O/w_$p.log | cut -c1-300
  cd $R; python tools/rocprof_summary.py stats gpurun_out/$tag/st_$p $O/stats_$p.md > /dev/null 2>&1; head -12 $O/stats_$p.md | cut -c1-160; cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_$p -o p --output-format csv -- python /tmp/w.py $p > $O/pmc_$p.log 2>&1
  cd $R; python tools/pmc_summary.py gpurun_out/$tag/pmc_$p gpurun_out/$tag/pmc_$p k_ball 2>/dev/null | head -20; cd /tmp
  rm -rf $O/st_$p $O/pmc_$p
done
