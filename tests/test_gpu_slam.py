"""-m gpu: the whole SLAM call pattern (row f-4) -- odometry with submap roll-over, graph nodes / odometry edges, Scan
Context candidates, loop registration against the candidate's finished submap, CS gate, pose-graph optimisation, submap
origin update -- through the C ABI, against the same harness driven by the CPU oracle."""
import numpy as np
import pytest

import randt_slam_amd as R
from randt_slam_amd import odometry, slam, synth
from oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu


def _drive(n_scans, per_lap):
    world = synth.make_world()
    th = 2 * np.pi * np.arange(n_scans) / per_lap
    truth = np.stack([5.0 * np.cos(th), 5.0 * np.sin(th), th + np.pi / 2], 1)
    scans = np.stack([synth.make_scan(world, truth[i], 71000 + i) for i in range(n_scans)])
    return truth, scans


def test_slam_loop_matches_oracle_chain(built):
    import torch

    n_scans, per_lap, dt = 300, 160, 0.25
    truth, scans = _drive(n_scans, per_lap)
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    loop_mp = R.default_matcher_params(gnc_steps=2)               # loop_closure_gnc_steps / loop_closure_scale (indoor)
    wp = R.window_params()
    params = dict(submap_size_poses=40, submap_overlap=10)        # short submaps: several roll-overs within two laps
    sc = dict(max_radius=20.0, dist_thresh=0.5)
    kw = dict(params=params, sc_params=sc, loop_closure_weight=40.0)
    gpu = slam.Slam(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params(), scan_slots=160, submap_slots=16),
                    mp, wp, loop_mp, **kw)
    cpu = slam.Slam(OracleBackend(), mp, wp, loop_mp, **kw)
    d_scans = torch.from_numpy(scans).cuda()
    n_opt = 0
    for i in range(n_scans):
        pg = gpu.process_scan(d_scans[i], i * dt)
        pc = cpu.process_scan(scans[i], i * dt)
        assert np.abs(pg - pc).max() < 1e-6, i
        ag, ac = gpu.detect_loop_closures(), cpu.detect_loop_closures()
        assert ag == ac
        if i % 40 == 39:
            rg, rc = gpu.optimize_pose_graph(), cpu.optimize_pose_graph()
            assert (rg is None) == (rc is None)
            if rg is not None:
                n_opt += 1
                assert rg["iterations"] == rc["iterations"] and rg["termination"] == rc["termination"]
                assert np.abs(gpu.node_positions() - cpu.node_positions()).max() < 1e-6
    # same graph on both sides
    assert len(gpu.nodes) == len(cpu.nodes) > 60 and gpu.submap_idzs == cpu.submap_idzs and gpu.root_nodes == cpu.root_nodes
    assert [(a, b) for a, b, _, _ in gpu.edges] == [(a, b) for a, b, _, _ in cpu.edges]
    for eg, ec in zip(gpu.edges, cpu.edges):
        assert np.abs(eg[2] - ec[2]).max() < 1e-6
    loops = [e for e in gpu.edges if e[0] + 1 != e[1]]
    assert len(loops) >= 3 and n_opt >= 3 and gpu.n_finished_submaps >= 5
    assert [(q, l, ok) for q, l, _, ok in gpu.loop_log] == [(q, l, ok) for q, l, _, ok in cpu.loop_log]
    # every loop constraint ties a second-lap node to the root of a first-lap submap, and agrees with the ground truth
    origin_inv = synth.se2_inv3(truth[0])
    for a, b, trans, _ in loops:
        assert gpu.submap_idzs[a] < gpu.submap_idzs[b]
    # the optimised graph stays on the truth (nodes are keyframes: scan index = first scan of a submap or a multiple of 4)
    est = gpu.node_positions()
    assert np.all(np.isfinite(est))
    end = synth.pose4_to_pose3(gpu.get_transform())
    rel = synth.se2_mul3(origin_inv, truth[-1])
    assert np.hypot(end[0] - rel[0], end[1] - rel[1]) < 0.3
