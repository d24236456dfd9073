#!/usr/bin/env python3
"""Times the host-buffer entry points (what the reference's C++ classes call): sdfgpu_build / sdfgpu_build_cells
host -> host, next to the device-resident build and to plain pinned / pageable copies of the same sizes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sdf_tools_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
ctx = capi.SdfGpu(0)
mask = synth.bernoulli_mask(shape, 0.5, 1)
res = {}


def timeit(fn, reps=3):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append((time.perf_counter() - t0) * 1e3)
    return round(min(t), 3)


res["build_host_ms"] = timeit(lambda: ctx.build(mask, 0.01))
cells = np.zeros(shape + (2,), np.float32)
cells[..., 0] = mask
res["build_cells_host_ms"] = timeit(lambda: ctx.build_cells(cells, shape, resolution=0.01))
out = np.empty(shape, np.float32)
res["numpy_empty_plus_touch_ms"] = timeit(lambda: np.empty(shape, np.float32).fill(0))
d_mask = torch.from_numpy(mask).cuda()
d_out = torch.empty(shape, dtype=torch.float32, device="cuda")
h_pin = torch.empty(shape, dtype=torch.float32).pin_memory()
h_page = torch.empty(shape, dtype=torch.float32)
m_pin = torch.from_numpy(mask).pin_memory()
m_page = torch.from_numpy(mask)


def sync(fn):
    def g():
        fn()
        torch.cuda.synchronize()
    return g


res["d2h_pinned_ms"] = timeit(sync(lambda: h_pin.copy_(d_out, non_blocking=True)))
res["d2h_pageable_ms"] = timeit(sync(lambda: h_page.copy_(d_out)))
res["h2d_mask_pinned_ms"] = timeit(sync(lambda: d_mask.copy_(m_pin, non_blocking=True)))
res["h2d_mask_pageable_ms"] = timeit(sync(lambda: d_mask.copy_(m_page)))
res["memcpy_pinned_to_pageable_ms"] = timeit(lambda: h_page.copy_(h_pin))
res["device_build_ms"] = timeit(sync(lambda: ctx.build_device(d_mask.data_ptr(), shape, d_out.data_ptr(), 0.01, False,
                                                              torch.cuda.current_stream().cuda_stream)))
res["grid"] = list(shape)
print(json.dumps(res))
