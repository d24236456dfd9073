#!/usr/bin/env python3
"""Streaming configuration (BASELINE.json configs[4]): 200 k points / frame -> 512^3 occupancy @ 1 cm -> SDF ->
EstimateDistance + gradient queries at 1 M points (the fused query kernel), sustained frame rate on one MI355X.
Target: >= 30 Hz.  --full-gradient times the variant that writes the gradient of every voxel instead (GetFullGradient
callers; what rounds 1-3 reported as the frame).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-gradient", action="store_true", help="field only: no queries, no gradient")
    ap.add_argument("--full-gradient", action="store_true", help="write the grid-aligned gradient of every voxel per frame")
    ap.add_argument("--queries", type=int, default=1 << 20, help="EstimateDistance + gradient query points per frame")
    args = ap.parse_args()
    import torch

    from sdf_tools_amd import synth
    from sdf_tools_amd.streaming import StreamingSdf

    n = args.size
    res = 0.01
    mode = None if args.no_gradient else ("full" if args.full_gradient else "query")
    st = StreamingSdf((n, n, n), res, (0.0, 0.0, 0.0), 0, gradient=mode)
    gen = torch.Generator(device="cuda").manual_seed(0)
    q = torch.rand((args.queries, 3), dtype=torch.float64, device="cuda", generator=gen) * (n * st.resolution)
    qf = q if mode == "query" else None
    # a different cloud every frame (the two-box pattern of scripts/3d_sdf_demo_rviz.py:15-19, scaled to the grid)
    clouds = [torch.from_numpy(synth.two_box_points(args.points, seed=f, scale=n * res)).cuda() for f in range(4)]
    for f in range(3):
        st.frame(clouds[f % 4], qf)
    torch.cuda.synchronize()
    # per-stage breakdown of one profiled frame (HIP events inside the library), outside the timed loop
    st.ctx.get_stage_times()
    st.ctx.set_profiling(True)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    if st.bits is not None:                     # (the frame's own voxelisation: points -> bit field)
        st.ctx.voxelize_points_bits_device(clouds[0].data_ptr(), clouds[0].shape[0], st.origin, st.resolution, st.shape,
                                           st.bits.data_ptr(), True, torch.cuda.current_stream().cuda_stream)
    else:
        st.ctx.voxelize_points_device(clouds[0].data_ptr(), clouds[0].shape[0], st.origin, st.resolution, st.shape,
                                      st.mask.data_ptr(), True, torch.cuda.current_stream().cuda_stream)
    e1.record()
    st.frame(clouds[0], qf)
    e2.record()
    torch.cuda.synchronize()
    ms, _ = st.ctx.get_stage_times()
    st.ctx.set_profiling(False)
    names = ["pack_bits", "dense_ball", "sweep_z", "sweep_y", "envelope_y", "sweep_x", "envelope_x"]
    breakdown = {k: round(v, 3) for k, v in zip(names, ms)}
    breakdown["voxelize"] = round(e0.elapsed_time(e1), 3)
    breakdown["frame_total"] = round(e1.elapsed_time(e2), 3)
    t0 = time.perf_counter()
    for f in range(args.frames):
        st.frame(clouds[f % 4], qf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    occ = float(st.mask.float().mean())
    # batched EstimateDistance + gradient queries on the last field (what a planner does with it), timed alone
    st.query(q)
    q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    q0.record()
    for _ in range(10):
        st.query(q)
    q1.record()
    torch.cuda.synchronize()
    breakdown["query_kernel_alone"] = round(q0.elapsed_time(q1) / 10, 3)
    what = {None: "", "full": " + full-grid gradient", "query": " + %d distance / gradient queries" % args.queries}[mode]
    print(json.dumps({"metric": "streaming frames/sec (points -> occupancy -> SDF%s)" % what, "mode": mode or "field only",
                      "value": round(args.frames / dt, 2), "unit": "Hz", "ms_per_frame": round(dt / args.frames * 1e3, 3),
                      "grid": [n, n, n], "points_per_frame": args.points, "occupancy": occ,
                      "kernels": st.ctx.last_build_info(), "path": st.ctx.last_path(),
                      "extrema": st.extrema(), "target_hz": 30,
                      "stage_ms_one_frame": breakdown}))


if __name__ == "__main__":
    main()
