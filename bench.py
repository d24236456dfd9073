#!/usr/bin/env python3
"""bench.py -- SDF build throughput on MI355X (BASELINE.json metric: Mvoxels/s, % HBM roofline).

A "step" is one pass of the hot path over one synthetic occupancy grid: device-resident uint8
mask -> device-resident fp32 signed distance field + extrema, through the C ABI
(libsdfgpu.so).  Inputs are i.i.d. Bernoulli(p = 0.5) occupancy,
generated directly in HBM before the timed region.

  N = 1 : 512^3 (the configuration the metric is quoted on)
  N > 1 : one process per GPU (torchrun), the grid cut into x slabs with an RCCL halo exchange
          (sdf_tools_amd/slab.py); weak scaling at 134 Mvoxel per GPU:
          N=2 1024x512x512, N=4 1024x1024x512, N=8 1024^3 (the metric's 8-GPU configuration)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
# Algorithmic bytes per voxel: uint8 mask in, int16 / int32 intermediates, fp32 out
# (K1 1+2, K2 2+4, K3 4+4 = 17).  The fused z+y kernel does K1's and K2's work in one launch, so it is
# priced at their sum (9); the bytes it actually has to move are fewer (1 in + 4 out) and are reported
# next to it as design_bytes_per_voxel.
# The dense path does the whole mask -> fp32 job in K0 (pack) + KD (ball): K0 is priced at the 1 B/voxel
# mask read, KD at the remaining 16 B of the separable formulation it replaces.
B_ALG = {"pack_bits": 1, "dense_ball": 16, "sweep_z": 1 + 2, "sweep_y": 2 + 4, "sweep_zy": 9, "sweep_x": 4 + 4,
         "envelope_y": 2 + 4, "envelope_x": 4 + 4}
B_DESIGN32 = {"pack_bits": 1.125, "dense_ball": 4.625, "sweep_z": 3, "sweep_y": 6, "sweep_zy": 5, "sweep_x": 8,
              "envelope_y": 8, "envelope_x": 10}
B_DESIGN16 = {"pack_bits": 1.125, "dense_ball": 4.625, "sweep_z": 3, "sweep_y": 4, "sweep_zy": 3, "sweep_x": 6,
              "envelope_y": 8, "envelope_x": 10}
B_ALG_TOTAL = 17
KERNEL_NAMES = {"envelope_y": "k_envelope<2>", "envelope_x": "k_envelope<3>", "pack_bits": "k_pack_bits_mask", "dense_ball": "k_ball_dense", "sweep_z": "k_sweep_z_vec16", "sweep_y": "k_sweep_march<2,...>",
                "sweep_zy": "k_sweep_zy_fused", "sweep_x": "k_sweep_march<3,...> / k_sweep_x16"}

GRIDS = {1: (512, 512, 512), 2: (1024, 512, 512), 4: (1024, 1024, 512), 8: (1024, 1024, 1024)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, nargs=3, default=None, help="override grid nx ny nz")
    ap.add_argument("--p", type=float, default=0.5, help="Bernoulli occupancy probability")
    ap.add_argument("--resolution", type=float, default=0.01)
    ap.add_argument("--halo", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=320, help="edge of the cube timed on the CPU oracle")
    ap.add_argument("--tune", type=int, nargs=2, default=None, help="rows per chunk: y x")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage HIP events in the timed region")
    ap.add_argument("--force-slab", action="store_true", help="run the multi-GPU slab builder even at N = 1")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (sdfgpu_set_option)")
    return ap.parse_args()


def cpu_baseline(sample_n, p, resolution):
    """Times the CPU oracle (kind = "port": the C restatement of the reference's single-threaded
    bucket-queue BuildDistanceField path) on a bounded sample of the same workload."""
    import numpy as np  # noqa: F401

    from oracle import oracle as O
    from sdf_tools_amd import synth

    m = synth.bernoulli_mask((sample_n,) * 3, p, 1)
    t0 = time.perf_counter()
    O.reference_sdf(m, resolution)
    dt = time.perf_counter() - t0
    return {"value": round(m.size / dt / 1e6, 4), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "host_cores": os.cpu_count(), "seconds": round(dt, 2),
            "sample": "%d^3 Bernoulli(p=%g) occupancy grid, same generator and resolution as the GPU workload; "
                      "single-threaded like the reference" % (sample_n, p)}


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC pass (profiles/*_traffic.json), if any."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required for RCCL between processes here
    import torch
    import torch.distributed as dist

    from sdf_tools_amd import capi, slab, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the SDF build path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    shape = tuple(args.size) if args.size else GRIDS.get(world, (128 * world, 1024, 1024))
    nx, ny, nz = shape
    n_total = nx * ny * nz
    res = args.resolution
    stream = torch.cuda.current_stream(dev)

    use_slab = world > 1 or args.force_slab
    if not use_slab:
        ctx = capi.SdfGpu(local_rank)
        if args.tune:
            ctx.set_tuning(*args.tune)
        for kv in args.opt:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        mask = synth.bernoulli_mask_torch(shape, args.p, 1, device=dev)
        out = torch.empty(shape, dtype=torch.float32, device=dev)

        def step():
            ctx.build_device(mask.data_ptr(), shape, out.data_ptr(), res, False, stream.cuda_stream)
    else:
        stages = slab.HipStages(local_rank)
        ctx = stages.ctx
        if args.tune:
            ctx.set_tuning(*args.tune)
        for kv in args.opt:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        x0, x1 = slab.slab_range(nx, rank, world)
        mask = synth.bernoulli_mask_torch(shape, args.p, 1, x_range=(x0, x1), device=dev)
        builder = slab.SlabSdfBuilder(stages, shape, res, False, halo=args.halo, rank=rank, world=world)
        pending = []

        def step():
            # enqueue this build, then validate the previous one (its all-reduced status has already
            # landed in pinned host memory): every build is validated inside the timed region, but the
            # GPUs never wait for the host between builds
            pending.append(builder.build_async(mask))
            if len(pending) > 1:
                builder.finish(pending.pop(0))

        def drain():
            while pending:
                builder.finish(pending.pop(0))

    def fence():
        if use_slab:
            drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
        torch.cuda.synchronize(dev)     # untimed: lets the handle's policy see each warm-up build before the next one
    fence()
    if use_slab and not args.no_profile:
        builder.time_ball_kernel(4)     # events around the dominant kernel of every 4th build, on its launch stream
    if not use_slab:
        ctx.get_stage_times()           # drop anything recorded so far
        # HIP events on the launch stream inside the timed region: around the dominant kernel only when the
        # warm-up builds were dense-certified (2 events on every 4th build), around every stage otherwise
        try:
            dominant_only = bool(ctx.last_path().get("dense_certified")) and not args.no_profile
        except Exception:               # no warm-up build yet
            dominant_only = False
        ctx.set_profiling(0 if args.no_profile else (3 if dominant_only else 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = n_total / (dt / args.steps) / 1e6
    result = {
        "metric": "Mvoxels/sec SDF build", "value": round(value, 2), "unit": "Mvoxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": "%dx%dx%d uint8 occupancy grid, Bernoulli(p=%g) seed 1, resolution %g, "
                               "no virtual border; device-resident mask -> device-resident fp32 SDF + extrema"
                               % (nx, ny, nz, args.p, res),
                   "grid": list(shape), "voxels": n_total,
                   "partition": "single GPU" if world == 1 else
                   "x-slab x%d; RCCL halo exchange: 2 bit-planes (dense path) / %d int32 planes (general path)"
                   % (world, args.halo)},
    }

    if not use_slab:
        ms_sum, builds = ctx.get_stage_times()
        ctx.set_profiling(False)
        if not args.no_profile and dominant_only:
            # per-stage breakdown from a second, untimed pass of K steps with an event behind every stage
            dominant_ms = ms_sum[1] / max(builds, 1)
            fence()
            ctx.set_profiling(1)
            for _ in range(args.steps):
                step()
            fence()
            ms_sum, builds = ctx.get_stage_times()
            ctx.set_profiling(False)
            ms_sum = list(ms_sum)
            result["config"]["stage_breakdown"] = "untimed second pass with an event behind every stage; dense_ball from the timed pass"
            ms_sum[1] = dominant_ms * builds
        if not args.no_profile:
            # the same K steps again without the per-stage HIP events (they cost a few us per build):
            # reported next to `value`, never instead of it
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            result["value_without_stage_events"] = round(n_total / ((time.perf_counter() - t1) / args.steps) / 1e6, 2)
        mx, mn = ctx.get_extrema()
        result["extrema"] = [mx, mn]
        if builds:
            info = ctx.last_build_info()
            B_DESIGN = B_DESIGN16 if info["plane16"] else B_DESIGN32
            info.update(ctx.last_path())
            result["config"]["kernels"] = info
            avg = [v / builds for v in ms_sum]
            stage_ms = {}
            if info["dense"]:
                stage_ms["pack_bits"], stage_ms["dense_ball"] = avg[0], avg[1]
            if not info["dense_certified"]:
                if info["fused_zy"]:
                    stage_ms["sweep_zy"] = avg[3]
                else:
                    stage_ms["sweep_z"], stage_ms["sweep_y"] = avg[2], avg[3]
                stage_ms["sweep_x"] = avg[5]
                if info["far_y"] or avg[4] > 0.05:
                    stage_ms["envelope_y"] = avg[4]
                if info["far_x"] or avg[6] > 0.05:
                    stage_ms["envelope_x"] = avg[6]
            else:
                result["config"]["guarded_general_pipeline_ms"] = round(sum(avg[2:]), 4)
            dom = max(stage_ms, key=stage_ms.get)
            achieved = n_total * B_ALG[dom] / (stage_ms[dom] * 1e-3) / 1e9
            traffic = load_traffic()
            kernel_ms = sum(avg)
            result["roofline"] = {
                "bound": "hbm", "kernel": KERNEL_NAMES[dom], "stage": dom,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": (traffic or {}).get(dom),
                "hbm_frac_traffic": (round((traffic or {}).get(dom) / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                     if (traffic or {}).get(dom) and n_total == 512 ** 3 else None),
                "alg_bytes_per_voxel": B_ALG[dom], "design_bytes_per_voxel": B_DESIGN[dom],
                "avg_ms": round(stage_ms[dom], 4),
                "stages_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                "note": ("achieved = SURVEY 8(d) algorithmic bytes of the separable formulation attributed to this kernel "
                         "x voxels / its HIP-event duration; a bit-parallel kernel moves far fewer bytes than that "
                         "(design_bytes_per_voxel), so frac can exceed 1 -- hbm_frac_traffic = measured PMC bytes "
                         "(traffic, from profiles/) / duration / peak is the hardware utilisation"),
                "pipeline": {"alg_bytes_per_voxel": B_ALG_TOTAL, "kernel_ms": round(kernel_ms, 4),
                             "achieved": round(n_total * B_ALG_TOTAL / (kernel_ms * 1e-3) / 1e9, 1),
                             "frac": round(n_total * B_ALG_TOTAL / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            }
    else:
        launches, ms_total, vox = builder.pop_ball_timings()
        if launches and rank == 0:
            avg_ms = ms_total / launches
            achieved = vox * B_ALG["dense_ball"] / (avg_ms * 1e-3) / 1e9
            traffic = load_traffic()
            per_voxel = ((traffic or {}).get("dense_ball") or 0) / float(512 ** 3)
            result["roofline"] = {
                "bound": "hbm", "kernel": KERNEL_NAMES["dense_ball"], "stage": "dense_ball",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": int(per_voxel * vox) if per_voxel else None,
                "hbm_frac_traffic": round(per_voxel * vox / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if per_voxel else None,
                "alg_bytes_per_voxel": B_ALG["dense_ball"], "avg_ms": round(avg_ms, 4),
                "launches_timed": launches, "voxels_per_launch": vox,
                "note": "rank 0, interior planes of its slab, every 4th build; traffic scaled from the 512^3 PMC pass",
            }
        result["config"]["dense_path"] = builder.dense
        result["config"]["builds_needing_general_path"] = builder.general_builds
        result["config"]["whole_line_fallbacks"] = builder.fallbacks

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.force_slab:
        # what a caller of the reference's C++ API sees: host mask -> host SDF through sdfgpu_build (PCIe both ways,
        # fresh output buffer).  Reported for context only -- never `value`.
        host_mask = mask.cpu().numpy()
        ctx.build(host_mask, res)
        t_host = []
        for _ in range(2):
            t1 = time.perf_counter()
            ctx.build(host_mask, res)
            t_host.append(time.perf_counter() - t1)
        result["host_api"] = {"call": "sdfgpu_build (host -> host, PCIe inclusive)", "ms": round(min(t_host) * 1e3, 2),
                              "Mvoxels_per_s": round(n_total / min(t_host) / 1e6, 1),
                              "note": "includes numpy output allocation; not the benchmark metric"}
        result["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.p, res)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
