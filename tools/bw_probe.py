#!/usr/bin/env python3
"""Yardstick for the write-bound kernels: what does this GPU sustain for a pure 512 MiB fp32 write,
a copy, and a read-reduction?  (torch kernels, timed with events.)"""
import torch

n = 128 * 1024 * 1024
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn, nbytes in (("fill (write 512 MiB)", lambda: b.fill_(1.5), 4 * n),
                         ("zero_ (write 512 MiB)", lambda: b.zero_(), 4 * n),
                         ("copy (read + write 1 GiB)", lambda: b.copy_(a), 8 * n),
                         ("sum (read 512 MiB)", lambda: a.sum(), 4 * n),
                         ("add_ in place (r+w 1 GiB)", lambda: a.add_(1.0), 8 * n)):
    ms = timeit(fn)
    print("%-28s %.4f ms  %.0f GB/s" % (name, ms, nbytes / ms / 1e6))
