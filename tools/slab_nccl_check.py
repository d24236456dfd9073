#!/usr/bin/env python3
"""One process per GPU over RCCL (torch.distributed backend "nccl"): the x-slab builder of sdf_tools_amd/slab.py on every
tier -- dense (2 bit-plane halos), near-field general path (int32 halos), far-field (re-partition x slabs -> y slabs and
back), virtual border -- each rank's slab compared bit for bit with the same rows of a single-GPU build of the whole grid
made on that rank's own GPU through the C ABI.  Launched by tests/test_gpu_zz_multi_gpu_hardware.py under torch.distributed.run when the
box has >= 2 GPUs; prints "SLAB_NCCL_OK world=<n>" from rank 0 on success, exits non-zero on the first mismatch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from sdf_tools_amd import capi, slab, synth  # noqa: E402


def boxes(shape):
    nx, ny, nz = shape
    m = np.zeros(shape, np.uint8)
    m[nx // 10: nx // 10 + max(2, nx // 8), ny // 2: ny // 2 + max(2, ny // 6), : max(2, nz // 3)] = 1
    m[nx // 2: nx // 2 + max(2, nx // 5), ny // 8: ny // 8 + max(2, ny // 5), nz // 4: nz // 4 + max(2, nz // 4)] = 1
    return m


def main():
    world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ["RANK"]); local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    stages = slab.HipStages(local)
    single = capi.SdfGpu(local)
    shape = (16 * world + 16, 48, 64)
    cases = [("dense", synth.bernoulli_mask(shape, 0.5, 1), False), ("mid", synth.bernoulli_mask(shape, 0.06, 2), False),
             ("far", boxes(shape), False), ("far vb", boxes(shape), True), ("empty", np.zeros(shape, np.uint8), False),
             ("sparse", synth.bernoulli_mask(shape, 0.002, 3), True)]
    ok = True
    for name, m, vb in cases:
        want, want_ext = single.build(m, 0.05, vb)
        b = slab.SlabSdfBuilder(stages, shape, 0.05, vb, halo=3, rank=rank, world=world)
        x0, x1 = slab.slab_range(shape[0], rank, world)
        got, ext = b.build(torch.from_numpy(np.ascontiguousarray(m[x0:x1])).to(dev))
        torch.cuda.synchronize(dev)
        same = np.array_equal(got.cpu().numpy().view(np.uint32), want[x0:x1].view(np.uint32)) and tuple(ext) == tuple(want_ext)
        flag = torch.tensor([0 if same else 1], device=dev)
        dist.all_reduce(flag)
        if int(flag.item()):
            ok = False
            if rank == 0:
                print("MISMATCH in case %r (world %d)" % (name, world), flush=True)
    dist.barrier()
    if rank == 0 and ok:
        print("SLAB_NCCL_OK world=%d" % world, flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
