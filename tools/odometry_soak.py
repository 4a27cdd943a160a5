"""Longer GPU-vs-oracle drives through the LocalFuser::processScan call pattern (outside the test suite).
Usage: odometry_soak.py [n_drives] [n_scans]   (SOAK_LAGS / SOAK_LONG_LAGS / SOAK_PARAMS / SOAK_IMU = 1 widen the draw)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import odometry, synth  # noqa: E402
from oracle_backend import OracleBackend  # noqa: E402

n_drives = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
for d in (range(n_drives) if os.environ.get('SOAK_DRIVE') is None else [int(os.environ['SOAK_DRIVE'])]):
    rng = np.random.default_rng(100 + d)
    world = synth.make_world(seed=1234 + 7 * d)
    traj = synth.make_trajectory(5000 + d, n_scans, step=float(rng.choice([0.15, 0.25, 0.4])))
    scans = [synth.make_scan(world, traj[i], 40000 + 1000 * d + i) for i in range(n_scans)]
    small = dict(submap_size_poses=int(rng.choice([24, 40])), submap_overlap=8)
    if os.environ.get("SOAK_LAGS") == "1":   # lags beyond the shipped 3: the general window kernel (window_gen.hip)
        small["smoothing_steps"] = int(rng.choice([3, 4, 5, 6, 7] if os.environ.get("SOAK_LONG_LAGS") != "1" else [8, 9, 10, 11, 12]))   # (8..12: window_gen_big.hip, round 4)
    param = R.PARAM_MANIFOLD if os.environ.get("SOAK_PARAMS") != "1" else int(rng.choice([R.PARAM_MANIFOLD, R.PARAM_VECTOR, R.PARAM_ANALYTIC]))
    # SOAK_IMU=1 (round 6: the harness feeds heading increments): gyro increments = the truth's heading steps + noise + a bias,
    # IMU factors on, a non-zero initial bias; without increments the factor stays off (all-zero measurements contradict the
    # drive, costs are ~1e5 and last-bit differences amplify by 1e4 per scan -- chaos, not parity)
    soak_imu = os.environ.get("SOAK_IMU") == "1"
    use_imu, const_vel = (1 if soak_imu else 0), int(rng.random() < 0.6)
    heading = np.unwrap(np.asarray(traj)[:, 2])
    yaw = (np.concatenate([[0.0], np.diff(heading)]) + rng.normal(0.0, 2e-3, n_scans) + 0.003) if soak_imu else np.zeros(n_scans)
    if soak_imu:
        small["initial_imu_bias"] = float(rng.choice([0.0, 0.01]))
    mp = R.default_matcher_params(parameterization=param, gnc_steps=3)
    wp = R.window_params(use_imu=use_imu, const_vel=const_vel)
    gpu = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp, small)
    cpu = odometry.Odometry(OracleBackend(), mp, wp, small)
    # sensitivity reference: the same CPU oracle with ONE parameter moved by one ulp
    wp1 = R.window_params(use_imu=use_imu, const_vel=const_vel)
    wp1.motion_sqrtI[0] = wp1.motion_sqrtI[0] * (1 + 4e-16)
    cpu1 = odometry.Odometry(OracleBackend(), mp, wp1, small)
    worst_t = worst_r = worst_self = 0.0
    for i in range(n_scans):
        pg = gpu.process_scan(scans[i], i * 0.25, imu_yaw_increment=yaw[i])
        pc = cpu.process_scan(scans[i], i * 0.25, imu_yaw_increment=yaw[i])
        p1 = cpu1.process_scan(scans[i], i * 0.25, imu_yaw_increment=yaw[i])
        worst_self = max(worst_self, np.abs(p1[2:] - pc[2:]).max())
        dt_ = np.abs(pg[2:] - pc[2:]).max()
        if os.environ.get("SOAK_VERBOSE") == "1" and dt_ > 10 * max(worst_t, 1e-9):
            print("   scan %d: deviation jumps to %.3e (gpu iters %s term %s | oracle %s)" % (i, dt_, gpu.last_result["iterations"] if gpu.last_result is not None else None,
                  gpu.last_result["termination"] if gpu.last_result is not None else None, getattr(cpu, "last_result", None)))
        worst_t = max(worst_t, dt_)
        worst_r = max(worst_r, abs(synth.wrap_angle(np.arctan2(pg[1], pg[0]) - np.arctan2(pc[1], pc[0]))))
    print("drive %d: %d scans, lag %d param %d, imu %d const_vel %d, submaps %d/%d, rejected %d/%d, worst GPU-vs-oracle %.3e m %.3e rad | oracle vs one-ulp-perturbed oracle %.3e m"
          % (d, n_scans, gpu.smoothing_steps, param, use_imu, const_vel, gpu.n_finished_submaps, cpu.n_finished_submaps, gpu.n_rejected, cpu.n_rejected, worst_t, worst_r, worst_self))
