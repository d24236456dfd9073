#!/usr/bin/env python3
"""libsdfgpu_multi, dense 512^3 builds with N logical ranks on ONE GPU: wall time per synchronous build and the host time the rank
threads spent inside API calls (sdfgpu_multi_last_host_us).  Run under `rocprofv3 --hip-runtime-trace --stats` to see WHICH HIP calls
that time is (16 us at one rank, 195 - 214 us at 2 / 8: the shared device, or a lock in the runtime?).
usage: multi_host_trace.py <ranks> [builds = 60]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdf_tools_amd import capi, synth  # noqa: E402

ranks = int(sys.argv[1])
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
shape, res = (512, 512, 512), 0.01
dev = torch.device("cuda", 0)
m = synth.bernoulli_mask_torch(shape, 0.5, 1, device=dev)
mg = capi.MultiSdfGpu(ranks, [0] * ranks)
slabs = [m[a:b] for a, b in (mg.slab_range(shape[0], r) for r in range(ranks))]
outs = [torch.empty(tuple(t.shape), dtype=torch.float32, device=dev) for t in slabs]
pm, po = [t.data_ptr() for t in slabs], [t.data_ptr() for t in outs]
for _ in range(5):
    mg.build_device(pm, shape, po, res, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
hmax = hsum = 0.0
for _ in range(builds):
    mg.build_device(pm, shape, po, res, False)
    st = mg.last_stats()
    hmax += st["host_us_max_rank"]
    hsum += st["host_us_sum"]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / builds
print(json.dumps({"logical_ranks": ranks, "ms_per_build": round(dt * 1e3, 4), "host_us_slowest_rank_thread": round(hmax / builds, 1),
                  "host_us_sum_over_rank_threads": round(hsum / builds, 1), "path": mg.last_path()}))
