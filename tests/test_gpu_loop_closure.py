"""-m gpu: the loop-closure data path of LocalFuser::detectLoopClosures (local_fuser.cpp:318-410) end to end on the
device -- Scan Context candidate (f-4) -> estimateLoopConstraint from the candidate's yaw (a15) -> CS-divergence gate
(f-2) -- against the same chain on the CPU oracle."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import host, synth
from test_scancontext import _database, osp
from util import IP, oracle_scan_map, to_oracle_params

pytestmark = pytest.mark.gpu


def test_loop_closure_chain_matches_oracle(built):
    import torch

    dev = torch.device("cuda:0")
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    scans, pos, dist = _database(n_db=60, revisit=(50, 5))
    n_db = scans.shape[0]
    sp_o, sp = osp(max_radius=20.0, dist_thresh=0.5), host.sc_params(max_radius=20.0, dist_thresh=0.5)
    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    mp = R.default_matcher_params(gnc_steps=2)                         # loop_closure_gnc_steps (indoor)

    # ---- device: every keyframe gets a scan NDT and a Scan Context node
    kf = R.Maps(ctx, n_db, mapp, 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(scans).to(dev), clu, kf)
    db = host.ScDatabase(ctx, sp)
    for i in range(n_db):
        db.append(scans[i], pos[i], dist[i])
    # ---- oracle twin
    omaps = [oracle_scan_map(s) for s in scans]
    descs, rks = [], []
    for s in scans:
        d, rk, _ = po.sc_make(s, sp_o)
        descs.append(d)
        rks.append(rk)
    descs, rks = np.stack(descs), np.stack(rks)

    n_loops = 0
    for q in (30, 50, 59):
        lid, yaw, _ = db.detect(q)
        olid, oyaw, _ = po.sc_detect(sp_o, descs, rks, pos, dist, q)
        assert lid == olid and yaw == np.float32(oyaw)
        if lid < 0:
            continue
        n_loops += 1
        # estimated_difference = ... * SE2(-yaw, 0)  (local_fuser.cpp:333): the candidate's frame is the fixed map here
        g4 = np.array([np.cos(-yaw), np.sin(-yaw), 0.0, 0.0])
        fidx = torch.tensor([lid], dtype=torch.int32, device=dev)
        pose = torch.from_numpy(g4[None].copy()).to(dev)
        corr = torch.full((1, 512, mp.n_neighbours), -1, dtype=torch.int32, device=dev)
        res = torch.zeros((1, 64), dtype=torch.uint8, device=dev)
        R.associate_batch(ctx, kf, fidx, kf, q, 1, pose, mp, corr)
        R.solve_batch(ctx, kf, fidx, kf, q, 1, corr, mp, pose, res)
        ctx.synchronize()
        p_gpu = pose.cpu().numpy()[0]
        rc, p_o, cost_o, st = po.register_pair(omaps[lid], omaps[q], to_oracle_params(mp), g4)
        assert np.abs(p_gpu - p_o).max() < 1e-6
        cost_gpu = float(res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)[0]["cost"])
        assert abs(cost_gpu - cost_o) <= 1e-8 * max(1.0, abs(cost_o))
        # gate: cs_m_loop_map.transformMap(estimated_difference); f_loop_map.calculateCSDivergence(...)  (:338-339)
        cs_gpu, _ = host.cs_divergence(ctx, kf, lid, kf, q, p_gpu)
        om = oracle_scan_map(scans[q])
        om.transform(p_o)
        cs_o, _ = po.cs_divergence(omaps[lid], om)
        assert np.isclose(cs_gpu, cs_o, rtol=1e-9, atol=1e-9) or (np.isnan(cs_gpu) and np.isnan(cs_o))
        # the refined transform really aligns the two places: compare with the ground-truth relative pose of the scene
        # (the drive of _database revisits node 5 at node 50 with a small offset and a 2-sector rotation)
        if q == 50:
            assert lid in (4, 5, 6)
    assert n_loops >= 1
    db.close()


def test_loop_closures_pull_a_drifting_drive_back(built):
    """The whole back half of the reference's loop: Scan Context candidates -> estimateLoopConstraint -> CS gate ->
    Constraint -> GlobalFuser::optimizePoseGraph (ndt_slam.cpp:351-361), all on the device.  A 1.25-lap circular drive
    whose odometry drifts; the loop constraints found on the second lap must pull the graph back onto the truth."""
    import torch

    dev = torch.device("cuda:0")
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    world = synth.make_world()
    per_lap, n = 48, 60
    th = 2 * np.pi * np.arange(n) / per_lap
    truth = np.stack([5.0 * np.cos(th), 5.0 * np.sin(th), th + np.pi / 2], 1)
    scans = np.stack([synth.make_scan(world, truth[i], 52000 + i) for i in range(n)])
    # drifting odometry: every relative motion over-rotated by 3 mrad and stretched by 1 %
    odom = [truth[0].copy()]
    rels = []
    for i in range(n - 1):
        rel = synth.se2_mul3(synth.se2_inv3(truth[i]), truth[i + 1])
        rel = np.array([rel[0] * 1.01, rel[1] * 1.01, rel[2] + 0.003])
        rels.append(rel)
        odom.append(synth.se2_mul3(odom[-1], rel))
    odom = np.array(odom)
    dist = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(odom[:, :2], axis=0), axis=1))])
    err_before = np.linalg.norm(odom[:, :2] - truth[:, :2], axis=1)
    assert err_before[-1] > 0.5

    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    mp = R.default_matcher_params(gnc_steps=2)
    kf = R.Maps(ctx, n, mapp, 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(scans).to(dev), clu, kf)
    db = host.ScDatabase(ctx, host.sc_params(max_radius=20.0, dist_thresh=0.5))
    ia, ib, meas, sqi = [], [], [], []
    for i in range(n - 1):
        ia.append(i); ib.append(i + 1); meas.append(rels[i]); sqi.append(np.diag([10.0, 10.0, 50.0]))   # local_fuser.cpp:203-205
    n_loops = 0
    for q in range(n):
        db.append(scans[q], odom[q, :2], dist[q])
        lid, yaw, _ = db.detect(q)
        if lid < 0:
            continue
        g4 = np.array([np.cos(-yaw), np.sin(-yaw), 0.0, 0.0])
        fidx = torch.tensor([lid], dtype=torch.int32, device=dev)
        pose = torch.from_numpy(g4[None].copy()).to(dev)
        corr = torch.full((1, 512, mp.n_neighbours), -1, dtype=torch.int32, device=dev)
        res = torch.zeros((1, 64), dtype=torch.uint8, device=dev)
        R.associate_batch(ctx, kf, fidx, kf, q, 1, pose, mp, corr)
        R.solve_batch(ctx, kf, fidx, kf, q, 1, corr, mp, pose, res)
        ctx.synchronize()
        p4 = pose.cpu().numpy()[0]
        cs, _ = host.cs_divergence(ctx, kf, lid, kf, q, p4)
        if not (cs < 3.6):                                   # loop_closure_max_cs_divergence (parameters_indoor.yaml:8; local_fuser.cpp:340)
            continue
        # the registration must have found the true relative pose of the two places
        want = synth.se2_mul3(synth.se2_inv3(truth[lid]), truth[q])
        got = synth.pose4_to_pose3(p4)
        assert np.abs(got[:2] - want[:2]).max() < 0.1 and abs(synth.wrap_angle(got[2] - want[2])) < 0.03
        ia.append(lid); ib.append(q); meas.append(got); sqi.append(np.eye(3) * 40.0)
        n_loops += 1
    db.close()
    assert n_loops >= 5
    x, r = host.pose_graph_optimize(ctx, odom, ia, ib, meas, sqi, n - 1)
    err_after = np.linalg.norm(x[:, :2] - truth[:, :2], axis=1)
    assert r["n_loop_closures"] == n_loops and r["termination"] in (1, 2, 3)
    assert err_after.max() < 0.35 * err_before.max() and err_after[-1] < 0.25 * err_before[-1]
    # and the same graph through the dense CPU oracle lands on the same poses
    xo, ro = po.pose_graph_optimize(odom, ia, ib, meas, sqi, n - 1)
    assert np.abs(x - xo).max() < 1e-7 and ro["iterations"] == r["iterations"]
