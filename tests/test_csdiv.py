"""SURVEY row f-2 (Map::calculateCSDivergence): oracle sanity on CPU, HIP parity on GPU."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import host, synth
from util import GpuRig, oracle_scan_map, oracle_submap, problem


def test_oracle_cs_divergence_ranks_poses(built):
    prob = problem()
    sub = oracle_submap(prob["submaps"][0])
    scan = oracle_scan_map(prob["scans"][0])
    vals = []
    for pose in (prob["truth"][0], prob["guess"][0], prob["truth"][0] + np.array([2.0, 1.0, 0.5])):
        m = scan.copy()
        m.transform(synth.pose3_to_pose4(pose))
        v, terms = po.cs_divergence(sub, m)
        vals.append(v)
        assert np.isclose(v, -np.log(terms[0]) + 0.5 * np.log(terms[1]) + 0.5 * np.log(terms[2]))
    assert vals[0] < vals[1] < vals[2]          # better alignment = smaller divergence
    assert vals[0] < 3.6 < vals[2]              # loop_closure_max_cs_divergence (parameters_indoor.yaml:8)
    # independent numpy evaluation of the interaction term for a few cells
    fc, mc = sub.cells(), scan.cells()
    def full(c):
        return np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]], dtype=np.float64)
    m = scan.copy(); m.transform(synth.pose3_to_pose4(prob["truth"][0])); mc = m.cells()
    tot = 0.0
    for f in fc:
        Sf = full(f["cov"])
        if np.linalg.det(Sf) < 1e-5:
            continue
        for q in mc:
            S = Sf + full(q["cov"]); d = (f["mean"] - q["mean"]).astype(np.float64)
            tot += 0.5 / np.sqrt(np.pi ** 2 * np.linalg.det(S)) * np.exp(-0.5 * d @ np.linalg.solve(S, d))
    assert np.isclose(tot, po.cs_divergence(sub, m)[1][0], rtol=1e-4)


@pytest.mark.gpu
def test_hip_cs_divergence_matches_oracle(built):
    rig = GpuRig(problem())
    rig.build_submaps()
    rig.build_scans()
    torch = rig.torch
    osub = [oracle_submap(sm) for sm in rig.prob["submaps"]]
    poses = synth.pose3_to_pose4(rig.prob["truth"])
    d_pose = torch.from_numpy(poses).to(rig.dev)
    out = torch.zeros(rig.B, dtype=torch.float64, device=rig.dev)
    terms = torch.zeros((rig.B, 3), dtype=torch.float64, device=rig.dev)
    host.cs_divergence_batch(rig.ctx, rig.submaps, 0, rig.n_sub, rig.fixed_idx, rig.scan_maps, 0, rig.B, d_pose, out, terms)
    rig.ctx.synchronize()
    out, terms = out.cpu().numpy(), terms.cpu().numpy()
    for i in range(rig.B):
        m = oracle_scan_map(rig.prob["scans"][i])
        m.transform(poses[i])
        v, t = po.cs_divergence(osub[rig.prob["submap_of"][i]], m)
        assert np.allclose(terms[i], t, rtol=1e-10), (i, terms[i], t)     # same fp32 pair terms, different fp64 summation order
        assert np.isclose(out[i], v, rtol=1e-10, atol=1e-10)
    # the one-pair host entry point (Map::calculateCSDivergence as detectLoopClosures calls it) gives the batch's numbers
    for i in (0, rig.B - 1):
        v1, t1 = host.cs_divergence(rig.ctx, rig.submaps, int(rig.prob["submap_of"][i]), rig.scan_maps, i, poses[i])
        assert v1 == out[i] and np.array_equal(t1, terms[i])
