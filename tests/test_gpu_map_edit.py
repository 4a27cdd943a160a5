"""-m gpu: the cell-by-cell face of the reference's Map through the C ABI -- Map::insertCluster, Map::insertCell and
Map::getClosestCells (both overloads) -- against the oracle and against the batched entries they must agree with."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import synth
from util import IP, cells_equal, oracle_scan_map, problem, oracle_submap

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import torch

    return torch, torch.device("cuda:0"), R.Context(0, torch.cuda.current_stream().cuda_stream)


def _scan(seed):
    w = synth.make_world()
    return synth.make_scan(w, synth.make_trajectory(3000, 2)[0], seed)


def test_insert_cluster_by_cluster_equals_the_batched_build(env):
    """HierarchicalMap::addClusters = insertCluster per label-ordered cluster (ndt_hierarchical_map.cpp:28-33): feeding the
    clusters one by one must give the very map the one-launch build gives -- cells, order and index grid, bit for bit."""
    torch, dev, ctx = env
    pts = _scan(1200)
    labels = po.grid_labels(pts, IP["n_clusters"], IP["max_range"])
    whole = R.Maps(ctx, 1, R.indoor_map_params(), 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts[None]).to(dev), R.indoor_cluster_params(), whole)
    ctx.synchronize()
    step = R.Maps(ctx, 1, R.indoor_map_params(), 512, with_grid=True)
    n_acc = 0
    for lab in np.unique(labels):                       # std::map order = ascending label
        n_acc += step.insert_cluster(0, pts[labels == lab])
    a, ga = whole.download(0)
    b, gb = step.download(0)
    om = oracle_scan_map(pts)
    assert n_acc == len(a) == om.n_cells > 30
    assert cells_equal(a, b) and np.array_equal(ga, gb)
    assert cells_equal(b, om.cells()) and np.array_equal(gb, om.grid())
    # below the acceptance gate (n > min_points_per_cell, ndt_cell.cpp:26): nothing is added
    assert step.insert_cluster(0, pts[:5]) is False and step.counts()[0] == len(a)
    # PCL layout (stride 8, intensity at 4) gives the same cell as packed xyzI
    pcl = np.zeros((40, 8), dtype=np.float32)
    pcl[:, :2] = pts[:40, :2] * 0.01 + [3.3, 1.1]
    pcl[:, 4] = pts[:40, 3]
    packed = np.ascontiguousarray(pcl[:, [0, 1, 2, 4]])
    m8 = R.Maps(ctx, 2, R.indoor_map_params(), 8, with_grid=True)
    assert m8.insert_cluster(0, pcl) and m8.insert_cluster(1, packed)
    assert cells_equal(m8.download(0)[0], m8.download(1)[0])
    # a cluster whose mean lies outside the index grid (the reference's grid_indizes_.at() throws): dropped, like the
    # batched build drops it
    far = pts[:40].copy()
    far[:, :2] = far[:, :2] * 0.01 + [0.0, 400.0]
    assert m8.insert_cluster(0, far) is False and m8.counts()[0] == 1
    # ... while x = 400 m only wraps into another row of the flat index (no row check in the reference either)
    far[:, :2] = pts[:40, :2] * 0.01 + [400.0, 0.0]
    assert m8.insert_cluster(0, far) is True and m8.counts()[0] == 2


def test_insert_cell_appends_without_touching_the_grid(env):
    torch, dev, ctx = env
    m = R.Maps(ctx, 1, R.indoor_map_params(), 4, with_grid=True)
    cells = oracle_scan_map(_scan(1201)).cells()[:5]
    m.insert_cells(0, cells[:2])                                   # Map::insertCell: grid_.push_back only
    got, grid = m.download(0)
    assert cells_equal(got, cells[:2]) and (grid == -1).all()
    m.insert_cells(0, cells[2:3], set_grid=True)
    got, grid = m.download(0)
    assert len(got) == 3 and (grid >= 0).sum() == 1 and grid[grid >= 0][0] == 2
    with pytest.raises(R.RandtError):                              # capacity 4
        m.insert_cells(0, cells[3:5])
    assert m.counts()[0] == 4


def test_closest_cells_match_the_oracle_search(env):
    torch, dev, ctx = env
    pr = problem()
    sub = oracle_submap(pr["submaps"][0])
    fixed = R.Maps(ctx, 1, R.indoor_map_params(), max(sub.n_cells, 1), with_grid=True)
    fixed.upload(0, sub.cells(), sub.grid())
    scan = oracle_scan_map(pr["scans"][0])
    q = scan.cells()
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    for maha in (True, False):
        want, _ = po.associate(sub, scan, ident, k=4, lookup_mahalanobis=maha, use_intensity=True)
        got = fixed.closest_cells(0, q, k=4, lookup_mahalanobis=maha)
        assert np.array_equal(got, want)
    # fewer neighbours than asked for, and a query far from everything
    lone = np.zeros(1, dtype=R.CELL_DTYPE)
    lone["mean"][0] = [24.0, 24.0, 0.0]
    assert (fixed.closest_cells(0, lone, k=4, lookup_mahalanobis=False) == -1).all()
    got7 = fixed.closest_cells(0, q[:8], k=7)
    want7, _ = po.associate(sub, scan, ident, k=7, lookup_mahalanobis=True, use_intensity=True)
    assert np.array_equal(got7, want7[:8])
