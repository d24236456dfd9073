"""x-slab partitioned SDF build across the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  The grid is cut
along x, the slowest memory axis, so every rank owns a contiguous ``[nxs, ny, nz]`` block of the
reference's VoxelGrid layout.  Two paths, tried in this order:

Dense path (scenes whose every voxel has an opposite-class voxel within squared distance 8)
  1. ``pack_bits``   occupancy bytes -> 1 bit / voxel, boundary planes first
  2. halo exchange   2 bit-planes (ny*nz/8 bytes each: 128 KiB at 1024^2) to / from each x neighbour,
                     one grouped isend/irecv per direct xGMI link, overlapped with packing the interior
  3. ``dense_ball``  bit-parallel ball kernel -> fp32 SDF, integer extrema, and an "uncertified" flag if some
                     voxel is farther than d^2 = 8 from the other class; the planes that need no neighbour
                     data are launched before the exchange completes, the 2 + 2 border planes after it
  4. all-reduce(MAX) of {max d^2 free, max d^2 filled, -, uncertified}: 4 integers, posted behind the NEXT
                     build's halo exchange (or by ``finish``) so it never delays that exchange

General path (only if some rank raised the flag; exact for any input)
  1. ``sweep_zy``    mask slab -> signed in-plane d^2 (int32); the y sweep's tier (marching / envelope kernel) is
                     chosen on the device, and a "far" hint is left in the status block
  Which x sweep follows is PREDICTED from the previous general build (no host read in the middle of a build):
  predicted near-field:
  2. halo exchange   `halo` int32 planes per neighbour
  3. ``sweep_x``     x sweep + signed merge; raises a status bit if a voxel needed planes beyond the halo;
                     all-reduce(MAX) of {max d^2 free, max d^2 filled, status, far hint} -- the build's ONE host read;
                     status or hint set: steps 4-6 on top (a misprediction: the scene turned far-field)
  predicted far-field (or redoing a mispredicted build):
  4. re-partition    x slabs -> y slabs: every rank sends the rows of every other rank's y slab, ONE message per peer
                     and direction (grouped isend/irecv = one message per direct xGMI link; 64 MiB per peer at
                     1024^3 on 8 GPUs), and receives complete x lines of its own y slab: [nx, ny/G, nz]
  5. ``sweep_x_lines``  exact x sweep on complete lines (marching or envelope kernel, chosen on the device)
  6. re-partition back to x slabs (fp32), all-reduce(MAX) of the integer extrema and the hint (-> next prediction)
  No rank ever holds more than 1/G of any field.

A build is validated one step late (:meth:`SlabSdfBuilder.build_async` / :meth:`finish`): the
all-reduced status lands in pinned host memory through a side stream, so the GPU never idles waiting
for the host between consecutive builds; results are double-buffered.

No CUDA-style ring emulation: the only bulk traffic is nearest-neighbour planes, one message per link
and direction.  The stage executor is pluggable: the product one (:class:`HipStages`) calls the C ABI;
the CPU tests inject an oracle-backed executor to exercise all of the logic above with ``gloo``.
"""
import torch
import torch.distributed as dist

DSQ_INF = 1 << 30
BALL_HALO = 2                   # planes of context the dense kernel needs on each side


def slab_range(nx, rank, world):
    """Balanced contiguous x range of `rank`."""
    return (rank * nx) // world, ((rank + 1) * nx) // world


def dense_shape_ok(nz):
    nzw = nz // 32
    return nz % 32 == 0 and 1 <= nzw <= 64 and (nzw & (nzw - 1)) == 0


class HipStages:
    """Stage executor on the rank's GPU via libsdfgpu.so (no CPU fallback)."""

    def __init__(self, device_index):
        from . import capi

        self.ctx = capi.SdfGpu(device_index)
        self.ctx.set_option("defer_fold", 1)      # several stage calls per build: fold their maxima once (fold())
        self.device = torch.device("cuda", device_index)

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- general path --------------------------------------------------------------------------
    def sweep_zy(self, mask_slab, plane_dsq_rows, far=None):
        """far: optional int32[1] device tensor, set to 1 when the envelope kernel did the y sweep (far-field slab)."""
        assert mask_slab.is_contiguous() and plane_dsq_rows.is_contiguous()
        self.ctx.sweep_zy_tiered_device(mask_slab.data_ptr(), tuple(mask_slab.shape), plane_dsq_rows.data_ptr(),
                                        far.data_ptr() if far is not None else 0, self.stream())

    def sweep_x_lines(self, lines, y_global, ny_global, resolution, vb, out, small):
        """Exact x sweep on complete lines of a y slab: lines [nx, nys, nz] int32 -> out [nx, nys, nz] fp32."""
        assert lines.is_contiguous() and out.is_contiguous()
        nx, nys, nz = lines.shape
        self.ctx.sweep_x_lines_device(lines.data_ptr(), nx, nys, nz, y_global, ny_global, resolution, vb,
                                      out.data_ptr(), small.data_ptr(), self.stream())

    def sweep_x(self, ext, halo_lo, nxs, halo_hi, lo_trunc, hi_trunc, x_global, nx_global, resolution, vb,
                out, small):
        ny, nz = ext.shape[1], ext.shape[2]
        base = small.data_ptr()
        self.ctx.sweep_x_device(ext.data_ptr(), halo_lo, nxs, halo_hi, ny, nz, lo_trunc, hi_trunc, x_global,
                                nx_global, resolution, vb, out.data_ptr(), base, base + 8, self.stream())

    def dense_phase(self, phase, mask_slab, bits_ext, halo_lo, halo_hi, resolution, out, small, stream=None):
        """One of the three native phases of a dense slab build (see sdfgpu_slab_dense_phase in include/sdfgpu.h)."""
        nxs, ny, nz = mask_slab.shape
        self.ctx.slab_dense_phase(phase, mask_slab.data_ptr(), nxs, ny, nz, bits_ext.data_ptr(), halo_lo, halo_hi,
                                  resolution, out.data_ptr(), small.data_ptr(), self.stream() if stream is None else stream)

    def fold(self, small):
        """Fold the maxima the stage kernels left in the context's slot array into small[0:2]."""
        self.ctx.fold_extrema_device(small.data_ptr(), self.stream())

    # ---- dense path ----------------------------------------------------------------------------
    def pack_bits(self, mask_rows, bits_rows):
        """mask_rows [r, ny, nz] uint8 -> bits_rows [r, ny, nz/32] int32 (bit i of word w: z = 32w+i)."""
        assert mask_rows.is_contiguous() and bits_rows.is_contiguous()
        r, ny, nz = mask_rows.shape
        self.ctx.pack_bits_device(mask_rows.data_ptr(), r * ny, nz, bits_rows.data_ptr(), self.stream())

    def dense_ball(self, bits_ext, out_lo, out_hi, nz, resolution, out, small):
        rows_x, ny = bits_ext.shape[0], bits_ext.shape[1]
        base = small.data_ptr()
        self.ctx.dense_ball_device(bits_ext.data_ptr(), rows_x, out_lo, out_hi, ny, nz, resolution,
                                   out.data_ptr(), base, base + 12, self.stream())


class _Slot:
    """Per-build buffers (double-buffered so a build can be validated while the next one runs)."""

    def __init__(self, nxs, ny, nz, device):
        self.out = torch.empty((nxs, ny, nz), dtype=torch.float32, device=device)
        self.small = torch.zeros(4, dtype=torch.int32, device=device)    # maxF, maxQ, status, uncertified
        self.host = torch.zeros(4, dtype=torch.int32)
        if device.type == "cuda":
            self.host = self.host.pin_memory()
        self.event = torch.cuda.Event() if device.type == "cuda" else None
        self.mask = None
        self.pending = False
        self.dense = False
        self.reduce_deferred = False


class SlabSdfBuilder:
    """Owns the per-rank buffers and runs the dense-first / general-fallback schedule for one grid shape.

    ``build(mask_slab)`` takes this rank's ``[nxs, ny, nz]`` uint8 occupancy (1 = filled) and returns
    ``(sdf_slab float32 [nxs, ny, nz], (max, min))`` with the extrema of the *whole* grid.
    ``build_async`` / ``finish`` split that into enqueue and validate (see the module docstring); the
    caller must keep ``mask_slab`` unchanged until the matching ``finish``.
    """

    def __init__(self, stages, shape, resolution=1.0, add_virtual_border=False, halo=3,
                 rank=None, world=None, group=None, device=None, dense=True):
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
        self.ext = None                 # int32 plane field of the general path, allocated on first use
        self.ext_rows = rows
        self.y0, self.y1 = slab_range(self.ny, self.rank, self.world)      # this rank's y slab (whole-line x sweep)
        self.lines = None               # [nx, nys, nz] int32: complete x lines of the y slab, allocated on first use
        self.out_y = None               # [nx, nys, nz] fp32
        self.predict_far = False        # general path: the x sweep predicted for the next build (complete lines / halo)
        self.whole_hold = 0             # builds left on complete lines after a halo sweep came back unresolved
        self.host_reads = 0             # host round trips of the general path (one per build when the prediction holds)
        self.mispredictions = 0         # general builds whose predicted x sweep had to be redone
        self.general_exchange = "halo (near-field) / all-to-all re-partition to y slabs (far-field)"
        self.fallbacks = 0              # whole-line (re-partitioned) x sweeps of the general path
        self.general_builds = 0         # builds the dense path could not certify
        # dense path
        self.dense = (bool(dense) and not self.vb and dense_shape_ok(self.nz) and min_slab >= BALL_HALO
                      and hasattr(stages, "dense_ball"))
        self.bh_lo = BALL_HALO if self.rank > 0 else 0
        self.bh_hi = BALL_HALO if self.rank < self.world - 1 else 0
        if self.dense:
            self.bits = torch.zeros((self.bh_lo + self.nxs + self.bh_hi, self.ny, self.nz // 32),
                                    dtype=torch.int32, device=self.device)
        self._p2p_cache = {}
        self._timing_every = 0          # > 0: bracket the interior ball launch of every k-th build with events
        self._timings = []              # (event0, event1, voxels)
        self._builds = 0
        self.slots = [_Slot(self.nxs, self.ny, self.nz, self.device) for _ in range(2)]
        self.cur = 0
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    # -- communication helpers ---------------------------------------------------------------------
    def _peer(self, group_rank):
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def _exchange(self, buf, lo, n, h):
        """Grouped nearest-neighbour send/recv of `h` boundary planes of buf[lo:lo+n] (asynchronous).
        The P2POp list of a buffer is built once and re-used by every build."""
        if self.world == 1 or h == 0:
            return []
        key = (buf.data_ptr(), lo, n, h)
        ops = self._p2p_cache.get(key)
        if ops is None:
            ops = []
            if self.rank > 0:          # lower neighbour: send my first h planes, receive its last h
                ops.append(dist.P2POp(dist.isend, buf[lo:lo + h], self._peer(self.rank - 1), self.group))
                ops.append(dist.P2POp(dist.irecv, buf[lo - h:lo], self._peer(self.rank - 1), self.group))
            if self.rank < self.world - 1:
                ops.append(dist.P2POp(dist.isend, buf[lo + n - h:lo + n], self._peer(self.rank + 1), self.group))
                ops.append(dist.P2POp(dist.irecv, buf[lo + n:lo + n + h], self._peer(self.rank + 1), self.group))
            self._p2p_cache[key] = ops
        return dist.batch_isend_irecv(ops)

    def _allreduce_small(self, small, async_op=False):
        if self.world > 1:
            return dist.all_reduce(small, op=dist.ReduceOp.MAX, group=self.group, async_op=async_op)
        return None

    # -- dense path --------------------------------------------------------------------------------
    def _enqueue_dense(self, mask_slab, slot):
        if hasattr(self.stages, "dense_phase"):
            return self._enqueue_dense_native(mask_slab, slot)
        n, lo, h = self.nxs, self.bh_lo, BALL_HALO
        own = self.bits[lo:lo + n]
        slot.small.zero_()
        if self.world > 1 and 4 * h < n:
            # boundary planes first: their exchange over xGMI runs while the interior is packed AND while the
            # ball kernel works on every plane that needs no neighbour data; only the 2 + 2 border planes
            # wait for the messages
            self.stages.pack_bits(mask_slab[:h], own[:h])
            self.stages.pack_bits(mask_slab[n - h:], own[n - h:])
            works = self._exchange(self.bits, lo, n, h)
            self._flush_deferred(exclude=slot)      # previous build's all-reduce goes behind this exchange
            self.stages.pack_bits(mask_slab[h:n - h], own[h:n - h])
            i_lo = h if self.bh_lo else 0
            i_hi = n - h if self.bh_hi else n
            self.stages.dense_ball(self.bits, lo + i_lo, lo + i_hi, self.nz, self.resolution, slot.out[i_lo:i_hi],
                                   slot.small)
            for w in works:
                w.wait()
            if i_lo:
                self.stages.dense_ball(self.bits, lo, lo + i_lo, self.nz, self.resolution, slot.out[:i_lo], slot.small)
            if i_hi < n:
                self.stages.dense_ball(self.bits, lo + i_hi, lo + n, self.nz, self.resolution, slot.out[i_hi:], slot.small)
        else:
            self.stages.pack_bits(mask_slab, own)
            works = self._exchange(self.bits, lo, n, h)
            self._flush_deferred(exclude=slot)
            for w in works:
                w.wait()
            self.stages.dense_ball(self.bits, lo, lo + n, self.nz, self.resolution, slot.out, slot.small)
        if hasattr(self.stages, "fold"):
            self.stages.fold(slot.small)
        self._builds += 1
        slot.reduce_deferred = True

    def _enqueue_dense_native(self, mask_slab, slot):
        """The same schedule through the three native phases (3 host calls per build instead of ~10: at 0.15 ms of
        GPU work per build the binding's per-call cost decides whether a rank stays GPU-bound)."""
        st = self.stages
        stream = st.stream()
        args = (mask_slab, self.bits, self.bh_lo, self.bh_hi, self.resolution, slot.out, slot.small, stream)
        st.dense_phase(0, *args)                    # clear the status words, pack the boundary planes
        works = self._exchange(self.bits, self.bh_lo, self.nxs, BALL_HALO)
        self._flush_deferred(exclude=slot)          # previous build's all-reduce goes behind this exchange
        timed = self._timing_every > 0 and self._builds % self._timing_every == 0
        if timed:                                   # events around the ball kernel of the interior planes only
            st.dense_phase(10, *args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st.dense_phase(11, *args)
            e1.record()
            split = (self.bh_lo or self.bh_hi) and 4 * BALL_HALO < self.nxs
            i_lo = BALL_HALO if (split and self.bh_lo) else 0
            i_hi = self.nxs - BALL_HALO if (split and self.bh_hi) else self.nxs
            if split:
                self._timings.append((e0, e1, (i_hi - i_lo) * self.ny * self.nz))
        else:
            st.dense_phase(1, *args)                # pack the interior, ball kernel where no neighbour data is needed
        for w in works:
            w.wait()
        if timed and not ((self.bh_lo or self.bh_hi) and 4 * BALL_HALO < self.nxs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st.dense_phase(2, *args)                # unsplit slab: the whole ball kernel (+ fold) is in this phase
            e1.record()
            self._timings.append((e0, e1, self.nxs * self.ny * self.nz))
        else:
            st.dense_phase(2, *args)                # border planes + fold of the maxima
        self._builds += 1
        slot.reduce_deferred = True

    def _flush_deferred(self, exclude=None, only=None):
        """Issue the status all-reduce (+ copy to pinned host memory) of every build that still owes one.
        It is deferred until the NEXT build has posted its halo exchange, so that on the communicator's
        stream the latency-critical exchange is never queued behind the previous build's all-reduce; every
        rank defers identically, so the collective order stays the same everywhere."""
        for slot in self.slots:
            if slot is exclude or (only is not None and slot is not only) or not slot.reduce_deferred:
                continue
            slot.reduce_deferred = False
            work = self._allreduce_small(slot.small, async_op=True)
            if self.side is not None:
                # through a side stream: the main stream never waits for the collective or the copy, so the
                # next build's kernels follow back to back
                self.side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.side):
                    if work is not None:
                        work.wait()
                    slot.host.copy_(slot.small, non_blocking=True)
                    slot.event.record(self.side)
            else:
                if work is not None:
                    work.wait()
                slot.host.copy_(slot.small)

    def time_ball_kernel(self, every):
        """Bracket the (interior) ball launch of every `every`-th build with events on the launch stream
        (0 = off).  GPU only."""
        self._timing_every = int(every) if self.device.type == "cuda" else 0
        self._timings = []

    def pop_ball_timings(self):
        """(launches, total ms, voxels per launch) of the bracketed launches since the last call; synchronises."""
        t = self._timings
        self._timings = []
        if not t:
            return 0, 0.0, 0
        t[-1][1].synchronize()
        return len(t), sum(e0.elapsed_time(e1) for e0, e1, _ in t), t[0][2]

    # -- general path (synchronous; exact for any input) -----------------------------------------------
    def _p2p(self, sends, recvs):
        """One grouped batch of point-to-point messages: sends / recvs = [(tensor, group rank)]."""
        ops = [dist.P2POp(dist.isend, t, self._peer(r), self.group) for t, r in sends]
        ops += [dist.P2POp(dist.irecv, t, self._peer(r), self.group) for t, r in recvs]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()

    def _whole_lines(self, own, slot):
        """Steps 4-6: x slabs -> y slabs, exact x sweep on complete lines, back to x slabs.  Returns the maxima and the
        all-reduced far hint of the y sweeps (status word 3)."""
        nys = self.y1 - self.y0
        if self.lines is None:
            self.lines = torch.empty((self.nx, max(nys, 1), self.nz), dtype=torch.int32, device=self.device)
            self.out_y = torch.empty((self.nx, max(nys, 1), self.nz), dtype=torch.float32, device=self.device)
        small = slot.small
        ranges_x = [slab_range(self.nx, r, self.world) for r in range(self.world)]
        ranges_y = [slab_range(self.ny, r, self.world) for r in range(self.world)]
        # forward: my rows of peer r's y slab -> peer r; peer s's rows of MY y slab land at x in [x0_s, x1_s)
        sends, recvs = [], []
        for r in range(self.world):
            ya, yb = ranges_y[r]
            xa, xb = ranges_x[r]
            if r == self.rank:
                if nys > 0:
                    self.lines[self.x0:self.x1].copy_(own[:, self.y0:self.y1])
            else:
                if yb > ya:
                    sends.append((own[:, ya:yb].contiguous(), r))
                if nys > 0:
                    recvs.append((self.lines[xa:xb], r))
        self._p2p(sends, recvs)
        small[:3].zero_()                   # (word 3 keeps the far hint of the y sweep)
        if nys > 0:
            self.stages.sweep_x_lines(self.lines, self.y0, self.ny, self.resolution, self.vb, self.out_y, small)
            if hasattr(self.stages, "fold"):
                self.stages.fold(small)     # (HipStages sets defer_fold: the stage call leaves its maxima in the slots)
        # backward: the x rows of peer r's slab (contiguous) -> peer r; placed into my y columns of peer s
        sends, recvs, tmp = [], [], {}
        for r in range(self.world):
            xa, xb = ranges_x[r]
            ya, yb = ranges_y[r]
            if r == self.rank:
                if nys > 0:
                    slot.out[:, self.y0:self.y1].copy_(self.out_y[self.x0:self.x1])
            else:
                if nys > 0:
                    sends.append((self.out_y[xa:xb], r))
                if yb > ya:
                    tmp[r] = torch.empty((self.nxs, yb - ya, self.nz), dtype=torch.float32, device=self.device)
                    recvs.append((tmp[r], r))
        self._p2p(sends, recvs)
        for r, t in tmp.items():
            ya, yb = ranges_y[r]
            slot.out[:, ya:yb].copy_(t)
        self._allreduce_small(small)
        max_f, max_q, _, hinted = (int(v) for v in small.tolist())
        self.host_reads += 1
        return max_f, max_q, hinted

    def _build_general(self, mask_slab, slot):
        """General path with ONE host read per build in the steady state (round 4).  Which x sweep runs -- halo planes +
        slab-local sweep, or the re-partition to complete lines -- is PREDICTED from the previous general build instead of
        read back from the GPUs in the middle of the build (a collective plus a host synchronisation with every GPU idle);
        the far hint rides in word 3 of the status block (unused on this path) and comes back, all-reduced, with the
        maxima at the end.  A wrong "near" costs the re-partition on top, a wrong "far" costs nothing: the whole-line
        sweep is exact on any scene.  Every rank sees the same all-reduced block, so every rank predicts alike."""
        if self.ext is None:
            self.ext = torch.empty((self.ext_rows, self.ny, self.nz), dtype=torch.int32, device=self.device)
        lo, n, hi, h = self.halo_lo, self.nxs, self.halo_hi, self.halo
        own = self.ext[lo:lo + n]
        small = slot.small
        small.zero_()
        self.stages.sweep_zy(mask_slab, own, small[3:4])
        far = self.predict_far or self.whole_hold > 0
        if not far:
            for w in self._exchange(self.ext, lo, n, h):
                w.wait()
            self.stages.sweep_x(self.ext, lo, n, hi, self.x0 - lo > 0, self.x1 + hi < self.nx, self.x0, self.nx,
                                self.resolution, self.vb, slot.out, small)
            if hasattr(self.stages, "fold"):
                self.stages.fold(small)
            self._allreduce_small(small)
            max_f, max_q, status, hinted = (int(v) for v in small.tolist())
            self.host_reads += 1
            # Only an UNRESOLVED voxel (status) forces the redo: with every voxel resolved inside the halo the result is exact
            # already, whatever the hint says -- the hint then only steers the next build's prediction.
            self.predict_far = bool(hinted)
            if not status:
                return max_f, max_q
            # some voxel needed a plane beyond its halo: sweep complete lines after all
            self.mispredictions += 1
            if not hinted:
                self.whole_hold = 8             # near-field clutter with a cavity deeper than the halo: do not flap
        elif self.whole_hold > 0:
            self.whole_hold -= 1
        self.fallbacks += 1
        max_f, max_q, hinted = self._whole_lines(own, slot)
        self.predict_far = bool(hinted)
        return max_f, max_q

    # -- public API --------------------------------------------------------------------------------
    def build_async(self, mask_slab):
        """Enqueue one build; returns the slot index to pass to :meth:`finish`."""
        assert tuple(mask_slab.shape) == (self.nxs, self.ny, self.nz), (mask_slab.shape, self.nxs)
        idx = self.cur
        self.cur ^= 1
        slot = self.slots[idx]
        if slot.pending:
            self.finish(idx)                     # never overwrite an unvalidated result
        slot.mask = mask_slab
        slot.dense = self.dense
        if self.dense:
            self._enqueue_dense(mask_slab, slot)
            if self.world == 1:
                self._flush_deferred()           # nothing to order against: read the flags back right away
        slot.pending = True
        return idx

    def finish(self, idx):
        """Validate build `idx`: returns (sdf_slab, (max, min)).  Falls back to the general path if the
        dense kernel could not certify every voxel (collective decision: all ranks take it together)."""
        from . import capi

        slot = self.slots[idx]
        assert slot.pending
        need_general = True
        if slot.dense:
            self._flush_deferred(only=slot)      # no later build posted it for us (e.g. build(), last step)
            if slot.event is not None:
                slot.event.synchronize()
            max_f, max_q, _, uncert = (int(v) for v in slot.host.tolist())
            need_general = uncert != 0
        if need_general:
            if slot.dense:
                self.general_builds += 1
            max_f, max_q = self._build_general(slot.mask, slot)
        slot.pending = False
        slot.mask = None
        return slot.out, capi.extrema_from_dsq(max_f, max_q, self.resolution)

    def build(self, mask_slab):
        return self.finish(self.build_async(mask_slab))
