"""SURVEY row f-3 (Matcher::estimateTransformGlobalBNB): batched cost evaluation and the coarse-to-fine
pose search, oracle on CPU and HIP parity on GPU."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import host, synth
from util import GpuRig, oracle_scan_map, oracle_submap, problem, to_oracle_params


def test_oracle_cost_is_half_sum_rho(built):
    prob = problem()
    sub, scan = oracle_submap(prob["submaps"][0]), oracle_scan_map(prob["scans"][0])
    g = synth.pose3_to_pose4(prob["guess"][0])
    corr, n = po.associate(sub, scan, g, 4)
    poses = synth.pose3_to_pose4(np.array([prob["guess"][0], prob["truth"][0]]))
    cost, n_res = po.eval_cost_batch(sub, scan, corr, poses)
    assert n_res == n and cost[1] < cost[0]                       # the true pose is cheaper than the guess
    # same number as the first LM iteration of a mu = 1, weight = 1 solve at that pose
    rc, p4, st = po.solve_pair(sub, scan, corr, po.default_params(gnc_steps=1), poses[0])
    assert np.isclose(st["trace_cost"][0], cost[0], rtol=1e-12)


def test_oracle_bnb_finds_the_basin(built):
    prob = problem()
    sub, scan = oracle_submap(prob["submaps"][0]), oracle_scan_map(prob["scans"][0])
    bad = prob["truth"][0] + np.array([0.9, -0.7, 0.1])
    mc, t4, n = po.search_global_bnb(sub, scan, po.default_params(), po.bnb_params(cost_threshold=2.0), synth.pose3_to_pose4(bad))
    est = synth.pose4_to_pose3(t4)
    assert np.hypot(*(est[:2] - prob["truth"][0][:2])) < 0.5 and n > 180
    # nothing below the threshold -> identity and the sentinel cost, exactly like the reference (:542,606)
    mc, t4, n = po.search_global_bnb(sub, scan, po.default_params(), po.bnb_params(cost_threshold=0.01), synth.pose3_to_pose4(bad))
    assert mc == 100000.0 and np.array_equal(t4, [1, 0, 0, 0])


@pytest.mark.gpu
def test_hip_cost_batch_and_search_match_oracle(built):
    rig = GpuRig(problem(), scan_cap=512)
    rig.build_submaps()
    rig.build_scans()
    torch = rig.torch
    prob = rig.prob
    sub, scan = oracle_submap(prob["submaps"][0]), oracle_scan_map(prob["scans"][0])
    mp = R.default_matcher_params()
    g = synth.pose3_to_pose4(prob["guess"][:1])
    corr = torch.full((1, 512, 4), -1, dtype=torch.int32, device=rig.dev)
    R.associate_batch(rig.ctx, rig.submaps, rig.fixed_idx[:1], rig.scan_maps, 0, 1, torch.from_numpy(g).to(rig.dev), mp, corr)
    rng = np.random.default_rng(0)
    poses3 = prob["truth"][0] + rng.normal(0, [0.5, 0.5, 0.1], (300, 3))
    poses = synth.pose3_to_pose4(poses3)
    cost = torch.zeros(300, dtype=torch.float64, device=rig.dev)
    nres = torch.zeros(1, dtype=torch.int32, device=rig.dev)
    host.eval_cost_batch(rig.ctx, rig.submaps, 0, rig.scan_maps, 0, corr, mp, 1.5, torch.from_numpy(poses).to(rig.dev), cost, nres)
    rig.ctx.synchronize()
    ocorr, n = po.associate(sub, scan, g[0], 4)
    ocost, on = po.eval_cost_batch(sub, scan, ocorr, poses)
    assert nres.item() == on and np.allclose(cost.cpu().numpy(), ocost, rtol=1e-12)
    # whole search
    bad = synth.pose3_to_pose4(prob["truth"][0] + np.array([0.9, -0.7, 0.1]))
    for thr in (2.0, 0.01):
        mc, t4, ne = host.search_global(rig.ctx, rig.submaps, 0, rig.scan_maps, 0, mp, host.bnb_params(cost_threshold=thr), bad)
        omc, ot4, one = po.search_global_bnb(sub, scan, to_oracle_params(mp), po.bnb_params(cost_threshold=thr), bad)
        assert ne == one and np.array_equal(t4, ot4) and np.isclose(mc, omc, rtol=1e-12)
