#!/usr/bin/env python3
"""The two-box scene at 512^3 through libsdfgpu_multi with ONE rank, or through the single-GPU ABI -- the same work by
construction -- for a kernel trace of each (rocprofv3 --kernel-trace --stats): which launches make the difference.
usage: multi_far_probe.py multi|single [builds = 10]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdf_tools_amd import capi, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "multi"
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, res = 512, 0.01
shape = (n, n, n)
dev = torch.device("cuda", 0)
one = capi.SdfGpu(0)
s = torch.cuda.current_stream().cuda_stream
pts = torch.from_numpy(synth.two_box_points(200000, seed=0, scale=n * res)).to(dev)
mask = torch.zeros(shape, dtype=torch.uint8, device=dev)
one.voxelize_points_device(pts.data_ptr(), pts.shape[0], (0.0, 0.0, 0.0), res, shape, mask.data_ptr(), True, s)
out = torch.empty(shape, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
if mode == "multi":
    mg = capi.MultiSdfGpu(1, [0])

    def build():
        mg.build_device([mask.data_ptr()], shape, [out.data_ptr()], res, False)
else:
    def build():
        one.build_device(mask.data_ptr(), shape, out.data_ptr(), res, False, s)
        one.get_extrema()
for _ in range(3):
    build()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(builds):
    build()
torch.cuda.synchronize()
print("%s: %.4f ms per build" % (mode, (time.perf_counter() - t0) / builds * 1e3))
