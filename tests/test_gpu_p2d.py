"""north_star's point-to-distribution special case on the D2D path: moving cells with ZERO covariance.

The reference has no P2D functor; with Sigma_m = 0 the D2D residual of ceres_residuals.h:520-552 reduces to
d^T Sigma_f^-1 d and the rotation term of the Jacobian vanishes ("P2D specialisation").  Nothing special is
compiled for it -- this test pins that the general kernels take such cells (uploaded, not built: the NDT build regularises
every covariance it produces) and agree with the oracle on association, solve traces and poses, for both residual
dimensions."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import synth
from util import GpuRig, oracle_map, oracle_scan_map, oracle_submap, problem, to_oracle_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("intensity", [1, 0])
def test_zero_moving_covariance_matches_oracle(built, intensity):
    import torch

    prob = problem()
    rig = GpuRig(prob)
    rig.build_submaps()
    rig.build_scans()
    osub = [oracle_submap(sm) for sm in prob["submaps"]]
    mp = R.default_matcher_params(use_intensity=intensity, parameterization=R.PARAM_MANIFOLD)
    k = mp.n_neighbours
    # scan cells -> points: covariance zeroed (the intensity variance too)
    p2d = R.Maps(rig.ctx, rig.B, rig.mapp, rig.scan_cap, with_grid=True)
    omaps = []
    for i in range(rig.B):
        cells, grid = rig.scan_maps.download(i)
        cells = cells.copy()
        cells["cov"][:] = 0.0
        p2d.upload(i, cells, grid)
        om = oracle_map(rig.scan_cap)
        om.set(cells, grid)
        omaps.append(om)
    g4 = synth.pose3_to_pose4(prob["guess"])
    pose = torch.from_numpy(g4.copy()).to(rig.dev)
    corr = torch.full((rig.B, rig.scan_cap, k), -1, dtype=torch.int32, device=rig.dev)
    res = torch.zeros((rig.B, 64), dtype=torch.uint8, device=rig.dev)
    trace_len = 3 * 512 + 1
    trace = torch.zeros((rig.B, trace_len), dtype=torch.float64, device=rig.dev)
    R.associate_batch(rig.ctx, rig.submaps, rig.fixed_idx, p2d, 0, rig.B, pose, mp, corr)
    rig.ctx.set_trace(trace, trace_len)
    R.solve_batch(rig.ctx, rig.submaps, rig.fixed_idx, p2d, 0, rig.B, corr, mp, pose, res)
    rig.ctx.synchronize()
    rig.ctx.set_trace(None, 0)
    pose, corr, trace = pose.cpu().numpy(), corr.cpu().numpy(), trace.cpu().numpy()
    res = res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)
    op = to_oracle_params(mp)
    for i in range(rig.B):
        fixed = osub[prob["submap_of"][i]]
        oc, _ = po.associate(fixed, omaps[i], g4[i], k, mp.lookup_mahalanobis, intensity)
        assert np.array_equal(corr[i, : omaps[i].n_cells], oc), i
        rc, p4, cost, st = po.register_pair(fixed, omaps[i], op, g4[i])
        assert np.allclose(pose[i], p4, rtol=0, atol=1e-7), (i, pose[i], p4)
        assert res["iterations"][i] == st["n_iterations"] and res["termination"][i] == st["termination"]
        n = int(trace[i, 0])
        assert n == len(st["trace_cost"])
        t = trace[i, 1 : 1 + 3 * n].reshape(n, 3)
        assert np.allclose(t[:, 0], st["trace_cost"], rtol=1e-8) and np.array_equal(t[:, 2].astype(int), st["trace_flag"])
        est = synth.pose4_to_pose3(pose[i])
        assert np.all(np.abs(est[:2] - prob["truth"][i][:2]) < 0.15) and abs(est[2] - prob["truth"][i][2]) < 0.05
