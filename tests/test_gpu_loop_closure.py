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
