#!/usr/bin/env python3
"""bench.py -- NDT registrations/sec on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE config 4's batch
of 512 independent registrations (8 submaps x 64 scans; 2000-point radar scans vs 100x100-slot
0.5 m submaps, indoor parameter set), each registration = NDT build from the raw points +
association against its submap + the full GNC / Levenberg-Marquardt solve, all on the GPU through
the C ABI of librandt_hip.so.  Inputs are resident in HBM before the timed region starts.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): independent registrations shard
with no data-path collective -- every rank processes its own 512-registration batch per step
("weak" scaling).  The submap tables are built once on rank 0 and broadcast over RCCL at set-up.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SUBMAPS, SCANS_PER_SUBMAP, N_KEYFRAMES = 8, 64, 34
N_POINTS, N_SLOTS = 2000, 100 * 100
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # vector fp64 (half the 157.3 TF fp32 vector rate)
# HBM bytes of one 512-registration step measured with PMC counters (profiles/r01_pmc_summary.csv):
# FETCH_SIZE KB x2 (gfx950 correction) of k_ndt_build + k_associate + k_solve, plus their WRITE_SIZE KB
PMC_TRAFFIC_BYTES_PER_STEP = int(((8156.6 + 2082.4 + 1835.4) * 2 + (1834.9 + 609.2 + 48.0)) * 1024)  # profiles/r01_f_pmc_summary.csv


def algorithmic_bytes(n_points, n_slots, m_cells, k):
    """bytes one registration must move (points in, dense submap table + index grid,
    scan cells out + back in, correspondences, pose/stat out)."""
    return n_points * 16 + n_slots * 48 + n_slots * 4 + m_cells * 48 * 2 + m_cells * k * 4 + 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-scale", type=int, default=1,
                    help="diagnostic: registrations per step = 512 x this (the headline workload is 1)")
    ap.add_argument("--only", choices=["build", "associate", "solve"], default=None,
                    help="diagnostic: time ONE stage alone at saturation (the printed value is then not the headline metric)")
    ap.add_argument("--streams", type=int, default=16, help="in-flight batches: step i runs on HIP stream i %% streams "
                    "(GPU_MAX_HW_QUEUES is raised to match unless already set: HIP maps streams onto 4 hardware queues by default)")
    ap.add_argument("--odometry-scans", type=int, default=200, help="BASELINE config 3 side measurement (0 = skip)")
    ap.add_argument("--polar-scans", type=int, default=16, help="BASELINE config 5 side measurement: polar filter (0 = skip)")
    ap.add_argument("--slam-scans", type=int, default=300,
                    help="side measurement: whole SLAM call pattern (odometry + loop closure + pose graph) on a two-lap drive (0 = skip)")
    ap.add_argument("--polar-odometry-scans", type=int, default=60,
                    help="BASELINE config 5 side measurement: full local-fuser loop on raw polar scans (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="wall-clock budget of the CPU baseline leg")
    args = ap.parse_args()

    # in-flight batches need their own hardware queues to overlap (must be set before the HIP runtime starts)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, min(32, args.streams))))
    import torch
    import torch.distributed as dist

    import randt_slam_amd as R
    from randt_slam_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and os.environ.get("RANDT_BENCH_BACKEND") != "gloo":
        raise SystemExit("rank %d has no GPU (%d visible)" % (local_rank, n_dev))
    local_rank = local_rank % max(1, n_dev)   # only differs in the single-GPU gloo logic test
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # "nccl" IS RCCL on ROCm.  RANDT_BENCH_BACKEND=gloo exists only to exercise the multi-rank control
        # flow on a box with fewer GPUs than ranks (tests); it is never used for reported numbers.
        backend = os.environ.get("RANDT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    # One randt context per HIP stream: consecutive steps (independent batches) alternate streams so
    # that the latency-bound tail of one batch's solve overlaps the next batch's build / association.
    n_streams = max(1, args.streams)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    ctxs = [R.Context(local_rank, st.cuda_stream) for st in streams]
    stream, ctx = streams[0], ctxs[0]
    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    mp = R.default_matcher_params()
    k = mp.n_neighbours
    scans_per_submap = SCANS_PER_SUBMAP * max(1, args.batch_scale)
    B = N_SUBMAPS * scans_per_submap
    scan_cap = 512

    # ---------------- set-up (untimed): synthetic world, scans, submaps ---------------------------
    prob = synth.make_batch_problem(N_SUBMAPS, scans_per_submap, N_KEYFRAMES, scan_seed0=1000 + 100000 * rank,
                                    guess_seed0=2000 + 100000 * rank)
    # submap tables live in torch-owned HBM so that RCCL can broadcast them
    cb, nb, gb = R.Maps.storage_bytes(N_SUBMAPS, mapp, N_SLOTS)
    t_cells = torch.zeros(cb, dtype=torch.uint8, device=dev)
    t_counts = torch.zeros(N_SUBMAPS, dtype=torch.int32, device=dev)
    t_grid = torch.zeros(gb // 4, dtype=torch.int32, device=dev)
    submaps = R.Maps(ctx, N_SUBMAPS, mapp, N_SLOTS, storage=(t_cells, t_counts, t_grid))
    if rank == 0:
        for j, sm in enumerate(prob["submaps"]):
            kf = torch.from_numpy(np.stack(sm["kf_scans"])).to(dev)
            tmp = R.Maps(ctx, kf.shape[0], mapp, scan_cap, with_grid=False)
            R.ndt_build_batch(ctx, kf, clu, tmp)
            submaps.merge(j, tmp, 0, synth.pose3_to_pose4(sm["kf_rel"]))  # rolling-submap path (a9 + a18)
            tmp.close()
    ctx.synchronize()
    if world > 1:
        # the only collective: submap cell tables + index grids from the owner rank, once per submap epoch
        for t in (t_cells, t_counts, t_grid):
            dist.broadcast(t, src=0)
        torch.cuda.synchronize()

    points = torch.from_numpy(prob["scans"]).to(dev)                       # (B, 2000, 4) f32, 16 B / point
    fixed_idx = torch.from_numpy(prob["submap_of"]).to(dev)
    guess4 = torch.from_numpy(synth.pose3_to_pose4(prob["guess"])).to(dev)
    # per-stream working set (outputs + intermediates); inputs and submaps are shared read-only
    poses = [guess4.clone() for _ in range(n_streams)]
    resultss = [torch.zeros((B, 64), dtype=torch.uint8, device=dev) for _ in range(n_streams)]
    corrs = [torch.full((B, scan_cap, k), -1, dtype=torch.int32, device=dev) for _ in range(n_streams)]
    submaps_v = [submaps] + [R.Maps(ctxs[i], N_SUBMAPS, mapp, N_SLOTS, storage=(t_cells, t_counts, t_grid), clear=False)
                             for i in range(1, n_streams)]
    scan_mapss = [R.Maps(ctxs[i], B, mapp, scan_cap, with_grid=False) for i in range(n_streams)]
    # the initial guess is an input and the solve updates it in place (Sophus::SE2d& trans): every timed step gets
    # its own 16 KB copy of the guesses, resident before the timed region starts, instead of a reset copy per step
    step_pose = [guess4.clone() for _ in range(args.steps)]
    pose, results, scan_maps = step_pose[0], resultss[0], scan_mapss[0]
    torch.cuda.synchronize()

    def step(i, events=None):
        j = i % n_streams
        st, cx = streams[j], ctxs[j]
        if events is None:
            with torch.cuda.stream(st):
                poses[j].copy_(guess4)                         # warm-up: reuse the per-stream buffer
            pose_j = poses[j]
        else:
            pose_j = step_pose[i]
        only = args.only if events is not None else None        # warm-up always runs the full path
        if events is not None:
            events[0].record(st)
        if only in (None, "build"):
            R.ndt_build_batch(cx, points, clu, scan_mapss[j])
        if events is not None:
            events[1].record(st)
        if only in (None, "associate"):
            R.associate_batch(cx, submaps_v[j], fixed_idx, scan_mapss[j], 0, B, pose_j, mp, corrs[j])
        if events is not None:
            events[2].record(st)
        if only in (None, "solve"):
            R.solve_batch(cx, submaps_v[j], fixed_idx, scan_mapss[j], 0, B, corrs[j], mp, pose_j, resultss[j])
        if events is not None:
            events[3].record(st)

    for i in range(args.warmup * n_streams):
        step(i)
    torch.cuda.synchronize()

    # ---------------- timed region -----------------------------------------------------------------
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s, ev[s])
    t_enqueued = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stage_ms = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(3)] for s in range(args.steps)]).mean(axis=0)

    if rank == 0:
        res = results.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)
        counts = scan_maps.counts()
        m_mean = float(counts.mean())
        n_res_mean = float(res["n_residuals"].mean())
        evals_mean = float(res["n_evals"].mean())
        value = B * args.steps * world / elapsed
        path_ms = float(stage_ms.sum())
        b_alg = algorithmic_bytes(N_POINTS, N_SLOTS, m_mean, k)
        achieved_gbs = b_alg * B / (path_ms * 1e-3) / 1e9
        solve_gbs = b_alg * B / (float(stage_ms[2]) * 1e-3) / 1e9
        # fp64 work of the solve kernel: SURVEY 8(d) F_alg = C * (E_J*370 + E_c*250); every pass here is a
        # Jacobian pass except the raw-residual one
        flops = n_res_mean * ((evals_mean - 1) * 370 + 250) * B
        out = {
            "metric": "ndt_registrations_per_sec" if args.only is None and args.batch_scale == 1 else "DIAGNOSTIC_only=%s_batch_scale=%d" % (args.only, args.batch_scale), "value": value, "unit": "registrations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "BASELINE config 4 batch: 512 independent registrations per GPU per step "
                            "(8 submaps x 64 scans), 2000-pt synthetic radar scan vs 100x100-slot 0.5 m NDT submap, "
                            "indoor parameters, NDT build + association + GNC/LM solve (estimateLoopConstraint unit)",
                "batch_per_gpu": B, "points_per_scan": N_POINTS, "submap_slots": N_SLOTS, "n_neighbours": k,
                "parameterization": "ambient4 (reference loop-closure behaviour)", "gnc_steps": mp.gnc_steps,
                "streams": n_streams,
                "mean_scan_cells": m_mean, "mean_residuals": n_res_mean, "mean_lm_iterations": float(res["iterations"].mean()),
            },
            "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
            "stage_ms": {"ndt_build": float(stage_ms[0]), "associate": float(stage_ms[1]), "solve": float(stage_ms[2])},
            "roofline": {
                # dominant kernel = k_solve; achieved = SURVEY 8(d) algorithmic bytes per registration x the 512
                # registrations one launch processes / that kernel's average launch duration (HIP events on its stream)
                "bound": "hbm", "achieved": solve_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": solve_gbs / HBM_PEAK_GBS, "traffic": PMC_TRAFFIC_BYTES_PER_STEP,
                "kernel": "k_solve<3,1,64,true,4> (dominant; four registrations = four wavefronts per workgroup); k_ndt_build and k_associate run once per step as well",
                "algorithmic_bytes_per_registration": b_alg, "registrations_per_launch": B,
                "avg_launch_ms": {"k_ndt_build": float(stage_ms[0]), "k_associate": float(stage_ms[1]), "k_solve": float(stage_ms[2])},
                "all_three_kernels": {"achieved": achieved_gbs, "frac": achieved_gbs / HBM_PEAK_GBS},
                "path_effective": {"achieved": value / world * b_alg / 1e9, "frac": value / world * b_alg / 1e9 / HBM_PEAK_GBS,
                                   "note": "SURVEY 8(d) definition: registrations/s x algorithmic bytes (batches overlap on %d streams)" % n_streams},
                "traffic_note": "HBM bytes per step (the three launches) from separate rocprofv3 --pmc passes, FETCH_SIZE x2 "
                                "(gfx950 correction, calibrated on k_ndt_build / k_filter_peaks whose byte counts are known) + "
                                "WRITE_SIZE, profiles/r01_f_pmc_summary.csv; ~10x BELOW the algorithmic bytes because the compact "
                                "cell tables never touch the empty slots of the dense submap grid",
                "note": "path is fp64-VALU / iteration-latency bound (SURVEY 8(d)); see roofline_fp64",
            },
            "roofline_fp64": {
                "bound": "fp64_valu", "kernel": "k_solve", "achieved": flops / (float(stage_ms[2]) * 1e-3) / 1e12,
                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / (float(stage_ms[2]) * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "note": "per launch: useful fp64 flops of one 512-registration solve / its duration while %d batches share the chip" % n_streams,
                # whole-path figure: the solve's useful flops per registration x registrations/s of one GPU
                "path_effective": {"achieved": flops / B * (value / world) / 1e12,
                                   "frac": flops / B * (value / world) / 1e12 / FP64_PEAK_TFLOPS},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out.update(cpu_baseline(prob, mp, pose.cpu().numpy(), args.cpu_seconds))
        if args.odometry_scans > 0 and world == 1:
            out["config3_streaming_odometry"] = streaming_odometry(ctx, args.odometry_scans, not args.no_cpu_baseline)
        if args.polar_scans > 0 and world == 1:
            out["config5_polar_filter"] = polar_filter(ctx, args.polar_scans)
        if args.slam_scans > 0 and world == 1:
            out["slam_loop"] = slam_loop(ctx, args.slam_scans)
        if args.polar_odometry_scans > 0 and world == 1:
            out["config5_polar_odometry"] = polar_odometry(ctx, args.polar_odometry_scans)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def polar_filter(ctx, n_scans):
    """BASELINE config 5 front end (side measurement): RadarPreprocessor::filterScan on Oxford-shaped
    polar scans, 400 azimuths x 3000 bins x 16 B = 19.2 MB per scan read once -> the one genuinely
    HBM-streaming stage of the path; followed by the NDT build of the filtered points."""
    import torch

    import randt_slam_amd as R
    from randt_slam_amd import host, synth

    dev = torch.device("cuda", ctx.device)
    world = synth.make_world()
    tr = synth.make_trajectory(3400, 4)
    base = [torch.from_numpy(synth.make_polar_scan(world, tr[i], 70 + i)).to(dev) for i in range(4)]
    raw = torch.stack([base[i % 4] for i in range(n_scans)]).contiguous()      # distinct HBM copies
    pitch = 6144
    out = torch.zeros((n_scans, pitch, 4), dtype=torch.float32, device=dev)
    counts = torch.zeros(n_scans, dtype=torch.int32, device=dev)
    status = torch.zeros(n_scans, dtype=torch.int32, device=dev)
    fp = host.filter_params()
    maps = R.Maps(ctx, n_scans, R.indoor_map_params(), 1024, with_grid=False)
    st = torch.cuda.current_stream()
    for _ in range(2):
        host.filter_scan_batch(ctx, raw, fp, out, counts, status)
        R.ndt_build_batch(ctx, out, R.indoor_cluster_params(), maps, n_points=counts)
    torch.cuda.synchronize()
    reps = 10
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_f = t_b = 0.0
    for _ in range(reps):
        e[0].record(st)
        host.filter_scan_batch(ctx, raw, fp, out, counts, status)
        e[1].record(st)
        R.ndt_build_batch(ctx, out, R.indoor_cluster_params(), maps, n_points=counts)
        e[2].record(st)
        torch.cuda.synchronize()
        t_f += e[0].elapsed_time(e[1])
        t_b += e[1].elapsed_time(e[2])
    t_f, t_b = t_f / reps * 1e-3, t_b / reps * 1e-3
    nbytes = raw.numel() * 4
    return {"scans_per_launch": n_scans, "raw_bytes_per_scan": nbytes // n_scans, "filter_ms": t_f * 1e3, "ndt_build_ms": t_b * 1e3,
            "filter_GBps": nbytes / t_f / 1e9, "filter_hbm_frac": nbytes / t_f / 1e9 / HBM_PEAK_GBS,
            "scans_per_sec_filter_plus_build": n_scans / (t_f + t_b), "mean_filtered_points": float(counts.float().mean().item()),
            "status_ok": bool((status == 0).all().item())}


def slam_loop(ctx, n_scans):
    """Side measurement: the reference's whole SLAM loop as a call pattern (randt_slam_amd/slam.py) -- fixed-lag odometry
    with submap roll-overs, graph nodes / edges, Scan Context candidates, loop registration against finished submaps, CS
    gate, pose-graph optimisation every 40 scans -- on a circular two-lap drive (160 scans per lap, 40-state submaps)."""
    import torch

    import randt_slam_amd as R
    from randt_slam_amd import odometry, slam, synth

    world = synth.make_world()
    dt, per_lap = 0.25, 160
    th = 2 * np.pi * np.arange(n_scans) / per_lap
    truth = np.stack([5.0 * np.cos(th), 5.0 * np.sin(th), th + np.pi / 2], 1)
    scans = np.stack([synth.make_scan(world, truth[i], 71000 + i) for i in range(n_scans)])
    d_scans = torch.from_numpy(scans).to(torch.device("cuda", ctx.device))
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    loop_mp = R.default_matcher_params(gnc_steps=2)
    wp = R.window_params()

    def run():
        s = slam.Slam(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params(), scan_slots=n_scans // 4 + 64,
                                          submap_slots=n_scans // 40 + 8), mp, wp, loop_mp,
                      params=dict(submap_size_poses=40, submap_overlap=10), sc_params=dict(max_radius=20.0, dist_thresh=0.5),
                      loop_closure_weight=40.0)
        t_loop = t_pg = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_scans):
            pose = s.process_scan(d_scans[i], i * dt)
            t1 = time.perf_counter()
            s.detect_loop_closures()
            t2 = time.perf_counter()
            if i % 40 == 39:
                s.optimize_pose_graph()
            t_loop += t2 - t1
            t_pg += time.perf_counter() - t2
        torch.cuda.synchronize()
        return s, pose, time.perf_counter() - t0, t_loop, t_pg

    run()                                                    # warm-up pass
    s, pose, el, t_loop, t_pg = run()
    rel = synth.se2_mul3(synth.se2_inv3(truth[0]), truth[-1])
    est = synth.pose4_to_pose3(pose)
    loops = [e for e in s.edges if e[0] + 1 != e[1]]
    return {"scans": n_scans, "scans_per_sec": n_scans / el, "ms_per_scan": el / n_scans * 1e3, "graph_nodes": len(s.nodes),
            "loop_constraints": len(loops), "loop_candidates_checked": len(s.loop_log), "pose_graph_optimisations": s.n_optimizations,
            "submaps_finished": s.n_finished_submaps, "loop_closure_ms_total": t_loop * 1e3, "pose_graph_ms_total": t_pg * 1e3,
            "end_pose_error_vs_truth_m": float(np.hypot(est[0] - rel[0], est[1] - rel[1]))}


def polar_odometry(ctx, n_scans):
    """BASELINE config 5, whole loop (side measurement): Oxford-shaped raw polar scans (400 azimuths x 3000 bins,
    19.2 MB each, resident in HBM) -> filterScan -> clustering + NDT -> constant-velocity prediction -> fixed-lag window
    registration -> keyframe merge / submap roll-over, one scan after the other (LocalFuser::processScan call pattern)."""
    import torch

    import randt_slam_amd as R
    from randt_slam_amd import host, odometry, synth

    world = synth.make_world()
    dt = 0.25
    traj = synth.make_trajectory(3300, n_scans, step=0.25)
    dev = torch.device("cuda", ctx.device)
    d_raw = [torch.from_numpy(synth.make_polar_scan(world, traj[i], 61000 + i)).to(dev) for i in range(n_scans)]
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    wp = R.window_params()
    fp = host.filter_params()
    odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp)
    # One untimed pass (kernels, workspaces, first touch of the freshly uploaded scans), then three timed passes over the
    # same drive: this loop keeps the GPU mostly idle between small launches, and the first passes after an idle period run
    # ~1.6x slower than the settled rate (clock ramp); the median pass is reported, all three are listed.
    passes = []
    for rep in range(4):
        odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_scans):
            pose = odo.process_scan(d_raw[i], i * dt, polar_filter=fp)
        torch.cuda.synchronize()
        if rep:
            passes.append(time.perf_counter() - t0)
    el = sorted(passes)[1]
    rel = synth.se2_mul3(synth.se2_inv3(traj[0]), traj[-1])
    est = synth.pose4_to_pose3(pose)
    raw_bytes = int(d_raw[0].numel() * 4)
    return {"scans": n_scans, "scans_per_sec": n_scans / el, "ms_per_scan": el / n_scans * 1e3, "raw_bytes_per_scan": raw_bytes,
            "raw_GBps": n_scans * raw_bytes / el / 1e9, "pass_ms_per_scan": [p / n_scans * 1e3 for p in passes],
            "registrations": odo.n_registrations, "rejected": odo.n_rejected,
            "end_pose_error_vs_truth_m": float(np.hypot(est[0] - rel[0], est[1] - rel[1]))}


def streaming_odometry(ctx, n_scans, with_cpu):
    """BASELINE config 3 (side measurement, not the headline metric): sequential scans through the
    LocalFuser::processScan call pattern -- scan NDT build, constant-velocity prediction, fixed-lag
    window registration (3 states, motion factors), keyframe merge with insertion delay, submap
    roll-over with overlap -- one scan after the other on one GPU (the path does not shard: scan t
    needs pose t-1).  Inputs resident in HBM; every scan costs one device->host pose read-back."""
    import torch

    import randt_slam_amd as R
    from randt_slam_amd import odometry, synth

    world = synth.make_world()
    dt = 0.25
    traj = synth.make_trajectory(3300, n_scans, step=0.25)
    scans = np.stack([synth.make_scan(world, traj[i], 20000 + i) for i in range(n_scans)])
    d_scans = torch.from_numpy(scans).to(torch.device("cuda", ctx.device))
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    wp = R.window_params()
    odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp)
    for i in range(min(8, n_scans)):                       # warm-up (kernels, workspace)
        odo.process_scan(d_scans[i], i * dt)
    odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 0
    for i in range(n_scans):
        pose = odo.process_scan(d_scans[i], i * dt)
        iters += int(odo.last_result["iterations"]) if odo.last_result is not None else 0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    origin_inv = synth.se2_inv3(traj[0])
    rel = synth.se2_mul3(origin_inv, traj[-1])
    est = synth.pose4_to_pose3(pose)
    out = {"scans": n_scans, "scans_per_sec": n_scans / el, "ms_per_scan": el / n_scans * 1e3,
           "mean_lm_iterations_per_scan": iters / max(1, odo.n_registrations), "submaps_finished": odo.n_finished_submaps,
           "end_pose_error_vs_truth_m": float(np.hypot(est[0] - rel[0], est[1] - rel[1]))}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_backend import OracleBackend

        n_cpu = min(n_scans, 40)
        cpu = odometry.Odometry(OracleBackend(), mp, wp)
        t0 = time.perf_counter()
        for i in range(n_cpu):
            pc = cpu.process_scan(scans[i], i * dt)
        out["cpu_oracle_scans_per_sec"] = n_cpu / (time.perf_counter() - t0)
        out["cpu_oracle_sample"] = "first %d scans, 1 thread (sequential path)" % n_cpu
    return out


def cpu_baseline(prob, mp, gpu_pose, budget_s):
    """The CPU oracle (a restatement 'port', not Ceres) on the box's host cores, OpenMP over
    registrations, on a bounded sample of the SAME batch; also yields the pose error GPU-vs-oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from randt_slam_amd import synth

    ip = synth.indoor_params()
    fixed = []
    for sm in prob["submaps"]:
        sub = po.Map(ip["size_x"], ip["size_y"], ip["resolution"], (0, 0), ip["max_neighbour_dist"], ip["min_points_per_cell"])
        for t in range(len(sm["kf_scans"])):
            s = po.Map(ip["size_x"], ip["size_y"], ip["resolution"], (0, 0), ip["max_neighbour_dist"], ip["min_points_per_cell"], 512)
            s.build(sm["kf_scans"][t], ip["n_clusters"], ip["max_range"])
            s.transform(synth.pose3_to_pose4(sm["kf_rel"][t]))
            sub.merge(s)
        fixed.append(sub)
    op = po.default_params()
    for name, _ in mp._fields_:
        if name != "reserved":
            setattr(op, name, getattr(mp, name))
    g4 = synth.pose3_to_pose4(prob["guess"])
    cores = po.num_threads()
    B = len(prob["scans"])
    done, t_total, poses = 0, 0.0, None
    while t_total < budget_s:
        t0 = time.perf_counter()
        fail, poses, cost, iters = po.register_batch(prob["scans"], fixed, prob["submap_of"], op, g4, ip["n_clusters"],
                                                     ip["max_range"], n_threads=cores)
        t_total += time.perf_counter() - t0
        done += B
    err_t = float(np.abs(gpu_pose[:, 2:] - poses[:, 2:]).max())
    dth = np.arctan2(gpu_pose[:, 1], gpu_pose[:, 0]) - np.arctan2(poses[:, 1], poses[:, 0])
    err_r = float(np.abs((dth + np.pi) % (2 * np.pi) - np.pi).max())
    return {
        "cpu_baseline": {
            "value": done / t_total, "unit": "registrations/s", "cores": cores, "kind": "port",
            "sample": "%d passes of the same 512-registration batch through the OpenMP CPU oracle (%.1f s wall, %d threads)" % (done // B, t_total, cores),
        },
        "pose_err_vs_oracle": {"max_abs_translation_m": err_t, "max_abs_rotation_rad": err_r, "tolerance": [1e-4, 1e-4]},
    }


if __name__ == "__main__":
    main()
