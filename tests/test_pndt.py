"""pNDT cells (`Cell::updateCell` with `use_pndt`, ndt_cell.cpp:67-82 and :102; NDTCellParameters::beam_cov / use_pndt,
ndt_slam_parameters.h:12-15 -- false in every shipped configuration, built for completeness of row a4).

CPU: the oracle's restatement against an independent numpy computation (float64 J S J^T per point).
GPU (-m gpu): `randt_ndt_build_pndt_batch_dev` against the oracle -- means, counts, order and index grid bit-exact,
covariances to 2e-6 relative (the only arithmetic the two sides do not share bit for bit is sin / cos of the beam angle:
both take it in double and round once, see oracle/randt_oracle.c orc_cell_from_points_pndt)."""
import numpy as np
import pytest

import pyoracle as po
from util import IP, oracle_map

F = np.float32
BEAM = np.array([[0.0349208, 0, 0], [0, 0.0225, 0], [0, 0, 4.0]], dtype=F)   # config/ndt_radar_slam_base_parameters.yaml:67-69 (shape)


def _scan_with_polar(seed, n=2000):
    from randt_slam_amd import synth

    w = synth.make_world()
    pts = synth.make_scan(w, synth.make_trajectory(3100, 2)[0], seed)[:n]
    ang = np.arctan2(pts[:, 1], pts[:, 0]).astype(F)                    # filterScan's (angle, dist) (radar_preprocessor.cpp:57,116)
    rng = np.hypot(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)).astype(F)
    return pts, np.stack([ang, rng], axis=1)


def test_oracle_pndt_matches_numpy(built):
    pts, polar = _scan_with_polar(777)
    plain, pn = oracle_map(cap=512), oracle_map(cap=512)
    plain.build(pts, IP["n_clusters"], IP["max_range"])
    pn.build(pts, IP["n_clusters"], IP["max_range"], polar=polar, beam_cov=BEAM)
    a, b = plain.cells(), pn.cells()
    assert len(a) == len(b) > 40
    assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["n"], b["n"]) and np.array_equal(plain.grid(), pn.grid())
    # independent: labels -> clusters in ascending label / input order, float64 J S J^T
    labels = po.grid_labels(pts, IP["n_clusters"], IP["max_range"])
    order = np.lexsort((np.arange(len(pts)), labels))
    S = BEAM.astype(np.float64)
    k = 0
    for lab in np.unique(labels):
        idx = order[labels[order] == lab]
        if len(idx) <= IP["min_points_per_cell"]:
            continue
        q = np.stack([pts[idx, 0], pts[idx, 1], pts[idx, 3]], axis=1).astype(np.float64)
        cov = np.cov(q.T, bias=True)
        a_, r_ = polar[idx, 0].astype(np.float64), polar[idx, 1].astype(np.float64)
        J = np.zeros((len(idx), 3, 3))
        J[:, 0, 0], J[:, 0, 1], J[:, 1, 0], J[:, 1, 1], J[:, 2, 2] = -r_ * np.sin(a_), np.cos(a_), r_ * np.cos(a_), np.sin(a_), 1.0
        P = np.einsum("nij,jk,nlk->il", J, S, J)
        full = cov + P / len(idx)
        got = b[k]["cov"]
        want = np.array([full[0, 0], full[0, 1], full[0, 2], full[1, 1], full[1, 2], full[2, 2]])
        assert np.allclose(got, want, rtol=2e-4, atol=1e-5), (k, got, want)
        # no eigenvalue regularisation: the plain cell differs from cov by it, the pNDT cell carries the beam term instead
        assert got[0] > a[k]["cov"][0] - 1e-6 and got[3] > a[k]["cov"][3] - 1e-6
        k += 1
    assert k == len(b)


@pytest.mark.gpu
def test_hip_pndt_matches_oracle(built):
    import torch

    import randt_slam_amd as R

    dev = torch.device("cuda:0")
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    B = 3
    scans = [_scan_with_polar(800 + s) for s in range(B)]
    n_pts = np.array([2000, 1500, 2000], dtype=np.int32)
    pts = np.stack([s[0] for s in scans])
    polar = np.stack([s[1] for s in scans])
    maps = R.Maps(ctx, B, R.indoor_map_params(), 512, with_grid=True)
    R.ndt_build_pndt_batch(ctx, torch.from_numpy(pts).to(dev), torch.from_numpy(polar).to(dev), BEAM, R.indoor_cluster_params(), maps,
                           n_points=torch.from_numpy(n_pts).to(dev))
    ctx.synchronize()
    worst = 0.0
    for s in range(B):
        cells, grid = maps.download(s)
        om = oracle_map(cap=512)
        om.build(pts[s, :n_pts[s]], IP["n_clusters"], IP["max_range"], polar=polar[s, :n_pts[s]], beam_cov=BEAM)
        oc = om.cells()
        assert len(cells) == len(oc) > 30
        assert np.array_equal(cells["mean"], oc["mean"]) and np.array_equal(cells["n"], oc["n"])
        assert np.array_equal(cells["max_intensity"], oc["max_intensity"]) and np.array_equal(grid, om.grid())
        rel = np.abs(cells["cov"].astype(np.float64) - oc["cov"]) / (np.abs(oc["cov"]) + 1e-6)
        worst = max(worst, float(rel.max()))
    assert worst < 2e-6, worst
    # the plain build of the same scans is untouched by the option (regularised cells, same means)
    plain = R.Maps(ctx, B, R.indoor_map_params(), 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts).to(dev), R.indoor_cluster_params(), plain, n_points=torch.from_numpy(n_pts).to(dev))
    c0, _ = plain.download(0)
    c1, _ = maps.download(0)
    assert np.array_equal(c0["mean"], c1["mean"]) and not np.array_equal(c0["cov"], c1["cov"])
