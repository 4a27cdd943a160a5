"""Committed golden vectors (tests/golden/pair_indoor.npz, made by tests/golden/make_golden.py from
the oracle): the oracle must keep reproducing them (CPU), and the HIP path must reproduce them
through the C ABI without needing the oracle binary (GPU)."""
import os
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pair_indoor.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def test_oracle_reproduces_golden(built, gold):
    import pyoracle as po
    from randt_slam_amd import synth
    from util import cells_equal, oracle_scan_map, oracle_map

    sub = oracle_map()
    for t in range(len(gold["kf_scans"])):
        s = oracle_scan_map(gold["kf_scans"][t])
        s.transform(synth.pose3_to_pose4(gold["kf_rel3"][t]))
        sub.merge(s)
    assert cells_equal(sub.cells(), gold["submap_cells"].reshape(-1).view(po.CELL_DTYPE))
    g = sub.grid()
    assert np.array_equal(np.nonzero(g >= 0)[0], gold["submap_grid_slots"]) and np.array_equal(g[g >= 0], gold["submap_grid_vals"])
    for i in range(2):
        scan = oracle_scan_map(gold["scans"][i])
        assert cells_equal(scan.cells(), gold[f"scan{i}_cells"].reshape(-1).view(po.CELL_DTYPE))
        g4 = synth.pose3_to_pose4(gold["guess3"][i])
        corr, _ = po.associate(sub, scan, g4, 4, 1, 1)
        assert np.array_equal(corr, gold[f"scan{i}_corr"])
        for name, param in (("ambient4", po.PARAM_AMBIENT4), ("manifold", po.PARAM_MANIFOLD)):
            rc, p4, cost, st = po.register_pair(sub, scan, po.default_params(parameterization=param), g4)
            assert np.allclose(p4, gold[f"scan{i}_{name}_pose4"], rtol=0, atol=1e-12)
            assert np.allclose(st["trace_cost"], gold[f"scan{i}_{name}_trace_cost"], rtol=1e-12)
            assert np.array_equal(st["trace_flag"], gold[f"scan{i}_{name}_trace_flag"])


@pytest.mark.gpu
def test_hip_path_reproduces_golden(gold):
    import torch

    import randt_slam_amd as R
    from randt_slam_amd import synth
    from util import cells_equal

    dev = torch.device("cuda:0")
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    sub = R.Maps(ctx, 1, mapp, 10000, with_grid=True)
    kf = torch.from_numpy(gold["kf_scans"]).to(dev)
    tmp = R.Maps(ctx, kf.shape[0], mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, kf, clu, tmp)
    sub.merge(0, tmp, 0, synth.pose3_to_pose4(gold["kf_rel3"]))
    cells, grid = sub.download(0)
    assert cells_equal(cells, gold["submap_cells"].reshape(-1).view(R.CELL_DTYPE))          # bit exact
    assert np.array_equal(np.nonzero(grid >= 0)[0], gold["submap_grid_slots"]) and np.array_equal(grid[grid >= 0], gold["submap_grid_vals"])
    pts = torch.from_numpy(gold["scans"]).to(dev)
    scans = R.Maps(ctx, 2, mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, pts, clu, scans)
    fidx = torch.zeros(2, dtype=torch.int32, device=dev)
    for name, param in (("ambient4", R.PARAM_AMBIENT4), ("manifold", R.PARAM_MANIFOLD)):
        mp = R.default_matcher_params(parameterization=param)
        pose = torch.from_numpy(synth.pose3_to_pose4(gold["guess3"])).to(dev)
        corr = torch.full((2, 512, 4), -1, dtype=torch.int32, device=dev)
        res = torch.zeros((2, 64), dtype=torch.uint8, device=dev)
        trace = torch.zeros((2, 3 * 256 + 1), dtype=torch.float64, device=dev)
        R.associate_batch(ctx, sub, fidx, scans, 0, 2, pose, mp, corr)
        ctx.set_trace(trace, trace.shape[1])
        R.solve_batch(ctx, sub, fidx, scans, 0, 2, corr, mp, pose, res)
        ctx.synchronize()
        ctx.set_trace(None, 0)
        pose, corr, trace = pose.cpu().numpy(), corr.cpu().numpy(), trace.cpu().numpy()
        for i in range(2):
            c, _ = scans.download(i)
            assert cells_equal(c, gold[f"scan{i}_cells"].reshape(-1).view(R.CELL_DTYPE))
            assert np.array_equal(corr[i, : len(c)], gold[f"scan{i}_corr"])
            gp = gold[f"scan{i}_{name}_pose4"]
            assert np.abs(pose[i, 2:] - gp[2:]).max() <= 1e-4                             # north_star tolerance
            assert abs(np.arctan2(pose[i, 1], pose[i, 0]) - np.arctan2(gp[1], gp[0])) <= 1e-4
            assert np.allclose(pose[i], gp, rtol=0, atol=1e-7)
            n = int(trace[i, 0])
            t = trace[i, 1 : 1 + 3 * n].reshape(n, 3)
            assert np.allclose(t[:, 0], gold[f"scan{i}_{name}_trace_cost"], rtol=1e-8)
            assert np.array_equal(t[:, 2].astype(int), gold[f"scan{i}_{name}_trace_flag"])


# ------------------------------------------------------------------------------------------------
# fixed-lag path: tests/golden/odometry_drive.npz (made by tests/golden/make_golden_odometry.py)
def _odometry_golden():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_odometry as mg

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "odometry_drive.npz"))
    return mg, z


def test_oracle_reproduces_odometry_golden():
    """the CPU oracle backend through the processScan call pattern still lands on the committed poses (first 16
    scans keep the CPU suite short); libm differences between machines stay far below the bar"""
    import randt_slam_amd as R
    from randt_slam_amd import odometry
    from oracle_backend import OracleBackend

    mg, z = _odometry_golden()
    traj, scans = mg.drive_inputs()
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    odo = odometry.Odometry(OracleBackend(), mp, R.window_params(), mg.SMALL)
    for i in range(16):
        p = odo.process_scan(scans[i], i * mg.DT)
        assert np.abs(p - z["poses4"][i]).max() < 1e-8, i


def test_harness_imu_bookkeeping_on_the_oracle_backend():
    """Matcher::imu_constraints_ in the processScan harness (round 6), on the CPU oracle: one increment per
    predicted state (ndt_matcher.cpp:58), IMU factors only once the vector is longer than the window (the reference reads in
    front of it until then: DESIGN spec decision 12), dropped at a submap roll-over (local_fuser.cpp:51 resetMatcher), first
    state of submap 0 = initial_imu_bias (:36,235), later submaps inherit the last state's bias (:241)."""
    import randt_slam_amd as R
    from randt_slam_amd import odometry
    from oracle_backend import OracleBackend

    mg, _ = _odometry_golden()
    traj, scans = mg.drive_inputs()
    small = dict(submap_size_poses=6, submap_overlap=3, initial_imu_bias=0.02)
    heading = np.unwrap(np.asarray(traj)[:, 2])
    yaw = np.concatenate([[0.0], np.diff(heading)]) + 0.005
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)

    class Spy(OracleBackend):
        calls = []

        def register_window(self, fixed_h, moving_h, states, mp, wp, trans4, imu=None):
            Spy.calls.append(None if imu is None else list(imu))
            return super().register_window(fixed_h, moving_h, states, mp, wp, trans4, imu)

    Spy.calls = []
    odo = odometry.Odometry(Spy(), mp, R.window_params(use_imu=1), small)
    odo.process_scan(scans[0], 0.0, imu_yaw_increment=yaw[0])
    assert odo.trajectory[0]["imu_bias"] == 0.02 and odo.imu_constraints == []
    for i in range(1, 9):
        odo.process_scan(scans[i], i * mg.DT, imu_yaw_increment=yaw[i])
    # scans 1..5 fill submap 0 (windows of 1, 2, 3, 3, 3 states), scan 5 rolls over, scans 6..8 are the new submap's first windows
    assert odo.n_finished_submaps == 1 and len(odo.trajectory) == 4 and len(odo.imu_constraints) == 3
    assert [c is None for c in Spy.calls] == [True, True, True, False, False, True, True, True]
    assert Spy.calls[3] == [yaw[1], yaw[2], yaw[3]] and Spy.calls[4] == [yaw[2], yaw[3], yaw[4]]   # sic: one step older than the states
    assert odo.trajectory[0]["imu_bias"] == odo.last_state["imu_bias"] != 0.02                      # estimated, then inherited
    # without use_imu nothing is passed down, whatever the caller feeds
    Spy.calls = []
    off = odometry.Odometry(Spy(), mp, R.window_params(use_imu=0), small)
    for i in range(6):
        off.process_scan(scans[i], i * mg.DT, imu_yaw_increment=yaw[i])
    assert all(c is None for c in Spy.calls)


@pytest.mark.gpu
def test_hip_path_reproduces_odometry_golden(built):
    """the HIP fixed-lag path against the committed drive (window solve, keyframe merges, one submap roll-over) --
    no oracle involved"""
    import torch
    import randt_slam_amd as R
    from randt_slam_amd import odometry

    mg, z = _odometry_golden()
    traj, scans = mg.drive_inputs()
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, R.window_params(), mg.SMALL)
    worst = 0.0
    for i in range(mg.N_SCANS):
        p = odo.process_scan(scans[i], i * mg.DT)
        worst = max(worst, float(np.abs(p - z["poses4"][i]).max()))
        assert worst <= 1e-4, (i, p, z["poses4"][i])
        assert int(odo.last_result["iterations"]) == int(z["lm_iterations"][i]) if odo.last_result is not None and z["lm_iterations"][i] > 0 else True
    assert odo.n_finished_submaps == int(z["submaps_finished"]) and odo.n_registrations == int(z["registrations"])
    assert worst < 1e-6
