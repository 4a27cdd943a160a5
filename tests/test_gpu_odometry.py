"""-m gpu: BASELINE config 3 call pattern (LocalFuser::processScan + submap roll-over with overlap)
driven through the C ABI, compared pose by pose with the same loop driven by the CPU oracle."""
import numpy as np
import pytest

import randt_slam_amd as R
from randt_slam_amd import odometry, synth
from oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu


def test_streaming_odometry_matches_oracle_and_truth(built):
    import torch

    world = synth.make_world()
    n_scans, dt = 64, 0.25
    traj = synth.make_trajectory(3200, n_scans, step=0.25)
    scans = [synth.make_scan(world, traj[i], 9000 + i) for i in range(n_scans)]
    small = dict(submap_size_poses=24, submap_overlap=8)        # force two roll-overs inside 64 scans
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    wp = R.window_params()

    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    gpu = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp, small)
    cpu = odometry.Odometry(OracleBackend(), mp, wp, small)
    origin_inv = synth.se2_inv3(traj[0])
    worst_t = worst_r = 0.0
    for i in range(n_scans):
        pg = gpu.process_scan(scans[i], i * dt)
        pc = cpu.process_scan(scans[i], i * dt)
        # north_star tolerance along the whole drive (errors compound through keyframe merges)
        worst_t = max(worst_t, np.abs(pg[2:] - pc[2:]).max())
        worst_r = max(worst_r, abs(synth.wrap_angle(np.arctan2(pg[1], pg[0]) - np.arctan2(pc[1], pc[0]))))
        assert worst_t <= 1e-4 and worst_r <= 1e-4, (i, pg, pc)
        rel = synth.se2_mul3(origin_inv, traj[i])
        est = synth.pose4_to_pose3(pg)
        assert np.all(np.abs(est[:2] - rel[:2]) < 0.25) and abs(synth.wrap_angle(est[2] - rel[2])) < 0.08, (i, est, rel)
    assert gpu.n_finished_submaps == cpu.n_finished_submaps == 2
    assert gpu.n_registrations == cpu.n_registrations and gpu.n_rejected == cpu.n_rejected == 0
    print("max deviation GPU vs oracle over the drive: %.3e m, %.3e rad" % (worst_t, worst_r))


def test_config5_polar_loop_matches_oracle(built):
    """BASELINE config 5: Oxford-shaped polar scans -> filterScan -> NDT -> fixed-lag odometry."""
    import torch
    from randt_slam_amd import host

    world = synth.make_world()
    n_scans, dt = 10, 0.25
    traj = synth.make_trajectory(3500, n_scans, step=0.25)
    raws = [synth.make_polar_scan(world, traj[i], 11000 + i) for i in range(n_scans)]
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    wp = R.window_params()
    fp = host.filter_params()
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    gpu = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp)
    cpu = odometry.Odometry(OracleBackend(), mp, wp)
    origin_inv = synth.se2_inv3(traj[0])
    for i in range(n_scans):
        pg = gpu.process_scan(torch.from_numpy(raws[i]).cuda(), i * dt, polar_filter=fp)
        pc = cpu.process_scan(raws[i], i * dt, polar_filter=fp)
        assert np.abs(pg[2:] - pc[2:]).max() <= 1e-4 and abs(synth.wrap_angle(np.arctan2(pg[1], pg[0]) - np.arctan2(pc[1], pc[0]))) <= 1e-4
        assert np.allclose(pg, pc, atol=1e-7)
        rel = synth.se2_mul3(origin_inv, traj[i])
        est = synth.pose4_to_pose3(pg)
        assert np.all(np.abs(est[:2] - rel[:2]) < 0.25) and abs(synth.wrap_angle(est[2] - rel[2])) < 0.08
