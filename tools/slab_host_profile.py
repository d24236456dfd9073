#!/usr/bin/env python3
"""Host cost of one pipelined slab build (sdf_tools_amd/slab.py, world = 1): a small grid, so that the GPU is never the
bottleneck, 3000 builds, cProfile by total time.  usage: slab_host_profile.py [n]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdf_tools_amd import slab, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shape = (n, n, n)
dev = torch.device("cuda", 0)
stages = slab.HipStages(0)
b = slab.SlabSdfBuilder(stages, shape, 0.01, False, rank=0, world=1)
masks = [synth.bernoulli_mask_torch(shape, 0.5, 1 + k, device=dev) for k in range(3)]
pending = []


def step(i):
    pending.append(b.build_async(masks[i % 3]))
    if len(pending) > 1:
        b.finish(pending.pop(0))


for i in range(200):
    step(i)
torch.cuda.synchronize()
K = 3000
t0 = time.perf_counter()
for i in range(K):
    step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("host-bound step: %.1f us per build (world = 1, %d^3)" % (dt / K * 1e6, n))
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
