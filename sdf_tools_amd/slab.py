"""x-slab partitioned SDF build across the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  The grid is cut
along x, the slowest memory axis, so every rank owns a contiguous ``[nxs, ny, nz]`` block of the
reference's VoxelGrid layout.  The z and y sweeps are slab-local; only the x sweep couples slabs:

  1. ``sweep_zy``   mask slab -> signed in-plane d^2 (int32), written straight into the middle of
                    an extended buffer ``[halo_lo + nxs + halo_hi, ny, nz]``
  2. halo exchange  the first / last ``halo`` planes go to the x-neighbours (grouped
                    send/recv: one message per direct xGMI link, 4 MiB per plane at 1024^2); the
                    boundary planes are swept first so the exchange overlaps the interior sweeps
  3. ``sweep_x``    x sweep + signed merge over the extended buffer; the kernel itself checks that
                    no voxel needed a plane beyond the halo and raises a status bit otherwise
  4. all-reduce     (MAX) of {max d^2 free, max d^2 filled, status}: 3 integers
  5. if any rank was unresolved (sparse scenes): all-gather the whole plane field along x and
                    redo step 3 on complete lines -- exact for any input

No CUDA-style ring emulation: the only bulk traffic is nearest-neighbour planes, which on xGMI's
point-to-point links is one message per link and direction.

The stage executor is pluggable: the product one (:class:`HipStages`) calls the C ABI
(``sdfgpu_sweep_zy_device`` / ``sdfgpu_sweep_x_device``).  The CPU tests inject their own executor
(built on the oracle) to exercise partitioning, halo exchange and the fallback with ``gloo``.
"""
import torch
import torch.distributed as dist

DSQ_INF = 1 << 30


def slab_range(nx, rank, world):
    """Balanced contiguous x range of `rank`."""
    return (rank * nx) // world, ((rank + 1) * nx) // world


class HipStages:
    """Stage executor on the rank's GPU via libsdfgpu.so (no CPU fallback)."""

    def __init__(self, device_index):
        from . import capi

        self.ctx = capi.SdfGpu(device_index)
        self.device = torch.device("cuda", device_index)

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def sweep_zy(self, mask_slab, plane_dsq_rows):
        assert mask_slab.is_contiguous() and plane_dsq_rows.is_contiguous()
        self.ctx.sweep_zy_device(mask_slab.data_ptr(), tuple(mask_slab.shape), plane_dsq_rows.data_ptr(),
                                 self.stream())

    def sweep_x(self, ext, halo_lo, nxs, halo_hi, lo_trunc, hi_trunc, x_global, nx_global, resolution, vb,
                out, small):
        ny, nz = ext.shape[1], ext.shape[2]
        base = small.data_ptr()
        self.ctx.sweep_x_device(ext.data_ptr(), halo_lo, nxs, halo_hi, ny, nz, lo_trunc, hi_trunc, x_global,
                                nx_global, resolution, vb, out.data_ptr(), base, base + 8, self.stream())


class SlabSdfBuilder:
    """Owns the per-rank buffers and runs steps 1-5 for one grid shape.

    ``build(mask_slab)`` takes this rank's ``[nxs, ny, nz]`` uint8 occupancy (1 = filled) and returns
    ``(sdf_slab float32 [nxs, ny, nz], (max, min))`` with the extrema of the *whole* grid.
    """

    def __init__(self, stages, shape, resolution=1.0, add_virtual_border=False, halo=3,
                 rank=None, world=None, group=None, device=None):
        self.stages = stages
        self.nx, self.ny, self.nz = (int(s) for s in shape)
        self.resolution = float(resolution)
        self.vb = bool(add_virtual_border)
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.device = device if device is not None else getattr(stages, "device", torch.device("cpu"))
        self.x0, self.x1 = slab_range(self.nx, self.rank, self.world)
        self.nxs = self.x1 - self.x0
        if self.nxs <= 0:
            raise ValueError("grid has fewer x planes (%d) than ranks (%d)" % (self.nx, self.world))
        min_slab = min(slab_range(self.nx, r, self.world)[1] - slab_range(self.nx, r, self.world)[0]
                       for r in range(self.world))
        self.halo = max(0, min(int(halo), min_slab))
        self.halo_lo = self.halo if self.rank > 0 else 0
        self.halo_hi = self.halo if self.rank < self.world - 1 else 0
        rows = self.halo_lo + self.nxs + self.halo_hi
        self.ext = torch.empty((rows, self.ny, self.nz), dtype=torch.int32, device=self.device)
        self.out = torch.empty((self.nxs, self.ny, self.nz), dtype=torch.float32, device=self.device)
        self.small = torch.zeros(4, dtype=torch.int32, device=self.device)   # maxF, maxQ, status, pad
        self.full = None            # all-gather target, allocated on first fallback
        self.fallbacks = 0

    # -- step 2 ------------------------------------------------------------------------------
    def _start_halo_exchange(self):
        """Posts the grouped nearest-neighbour send/recv (asynchronous on RCCL's stream)."""
        if self.world == 1 or self.halo == 0:
            return []
        h, lo, n = self.halo, self.halo_lo, self.nxs
        ops = []
        if self.rank > 0:          # lower neighbour: send my first h planes, receive its last h
            ops.append(dist.P2POp(dist.isend, self.ext[lo:lo + h], self._peer(self.rank - 1), self.group))
            ops.append(dist.P2POp(dist.irecv, self.ext[0:lo], self._peer(self.rank - 1), self.group))
        if self.rank < self.world - 1:
            ops.append(dist.P2POp(dist.isend, self.ext[lo + n - h:lo + n], self._peer(self.rank + 1), self.group))
            ops.append(dist.P2POp(dist.irecv, self.ext[lo + n:lo + n + h], self._peer(self.rank + 1), self.group))
        return dist.batch_isend_irecv(ops)

    def _peer(self, group_rank):
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    # -- step 5 ------------------------------------------------------------------------------
    def _gather_full(self):
        if self.full is None:
            self.full = torch.empty((self.nx, self.ny, self.nz), dtype=torch.int32, device=self.device)
        own = self.ext[self.halo_lo:self.halo_lo + self.nxs]
        chunks = [self.full[slice(*slab_range(self.nx, r, self.world))] for r in range(self.world)]
        even = all(c.shape[0] == chunks[0].shape[0] for c in chunks)
        if even:
            dist.all_gather_into_tensor(self.full, own.contiguous(), group=self.group)
        else:
            dist.all_gather(chunks, own.contiguous(), group=self.group)
        return self.full

    def build(self, mask_slab):
        assert tuple(mask_slab.shape) == (self.nxs, self.ny, self.nz), (mask_slab.shape, self.nxs)
        lo, n, hi, h = self.halo_lo, self.nxs, self.halo_hi, self.halo
        own = self.ext[lo:lo + n]
        if self.world > 1 and 0 < h and 2 * h < n:
            # boundary planes first, so their exchange over xGMI overlaps the interior z/y sweeps
            self.stages.sweep_zy(mask_slab[:h], own[:h])
            self.stages.sweep_zy(mask_slab[n - h:], own[n - h:])
            works = self._start_halo_exchange()
            self.stages.sweep_zy(mask_slab[h:n - h], own[h:n - h])
        else:
            self.stages.sweep_zy(mask_slab, own)
            works = self._start_halo_exchange()
        for w in works:
            w.wait()
        self.small.zero_()
        self.stages.sweep_x(self.ext, lo, n, hi, self.x0 - lo > 0, self.x1 + hi < self.nx, self.x0, self.nx,
                            self.resolution, self.vb, self.out, self.small)
        if self.world > 1:
            dist.all_reduce(self.small, op=dist.ReduceOp.MAX, group=self.group)
        max_f, max_q, status, _ = (int(v) for v in self.small.tolist())
        if status:
            # some voxel anywhere needed a plane beyond its halo: redo the x sweep on complete lines
            self.fallbacks += 1
            full = self._gather_full()
            self.small.zero_()
            self.stages.sweep_x(full, self.x0, n, self.nx - self.x1, False, False, self.x0, self.nx,
                                self.resolution, self.vb, self.out, self.small)
            if self.world > 1:
                dist.all_reduce(self.small, op=dist.ReduceOp.MAX, group=self.group)
            max_f, max_q, status, _ = (int(v) for v in self.small.tolist())
            assert status == 0
        from . import capi

        return self.out, capi.extrema_from_dsq(max_f, max_q, self.resolution)
