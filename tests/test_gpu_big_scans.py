"""-m gpu: scans beyond one workgroup's LDS (> 7168 points) through the multi-workgroup build (csrc/ndt_build_big.hip):
`ClusterGenerator::labelClouds` has no size limit (radar_preprocessor.cpp:151-169).  Same bar as the one-workgroup kernel:
cell statistics, compact order and index grid bit-identical to the oracle -- the fp32 sums run in the reference's
sequential point order whatever the scan size."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import synth
from util import IP, cells_equal, oracle_map, oracle_scan_map

pytestmark = pytest.mark.gpu
F = np.float32


def _dense_scan(n, seed0):
    """n points: several synthetic scans from nearby poses stacked (azimuth order inside each)."""
    w = synth.make_world()
    tr = synth.make_trajectory(3000, 2)
    parts = []
    k = 0
    while sum(len(p) for p in parts) < n:
        pose = tr[0] + np.array([0.01 * k, -0.005 * k, 0.002 * k])
        parts.append(synth.make_scan(w, pose, seed0 + k))
        k += 1
    return np.concatenate(parts)[:n]


@pytest.fixture(scope="module")
def env(built):
    import torch

    return torch, torch.device("cuda:0"), R.Context(0, torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("n", [7169, 8192, 20000, 65536])
def test_big_scans_bit_exact(env, n):
    torch, dev, ctx = env
    pts = _dense_scan(n, 9000 + n)
    maps = R.Maps(ctx, 1, R.indoor_map_params(), 2048, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts[None]).to(dev), R.indoor_cluster_params(), maps)
    cells, grid = maps.download(0)
    om = oracle_scan_map(pts, cap=2048)
    assert om.n_cells > 50 and len(cells) == om.n_cells
    assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())
    assert int(cells["n"].max()) > 253                        # chains far longer than anything a 2000-point scan has


def test_big_batch_ragged_and_pcl_stride(env):
    torch, dev, ctx = env
    n_pts = np.array([12000, 0, 9000, 5, 7000], dtype=np.int32)   # a ragged batch whose pitch forces the tiled path
    base = _dense_scan(12000, 9500)
    pcl = np.zeros((len(n_pts), 12000, 8), dtype=F)               # pcl::PointXYZI layout
    for s in range(len(n_pts)):
        sc = np.roll(base, 37 * s, axis=0)
        pcl[s, :, :2], pcl[s, :, 4] = sc[:, :2], sc[:, 3]
    maps = R.Maps(ctx, len(n_pts), R.indoor_map_params(), 2048, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pcl).to(dev), R.indoor_cluster_params(), maps, n_points=torch.from_numpy(n_pts).to(dev))
    counts = maps.counts()
    for s, n in enumerate(n_pts):
        om = oracle_map(2048)
        if n:
            om.build(pcl[s, :n], IP["n_clusters"], IP["max_range"], ioff=4)
        cells, grid = maps.download(s)
        assert counts[s] == om.n_cells
        assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid()), s
    assert counts[1] == 0 and counts[3] == 0


def test_tiled_path_equals_one_workgroup_kernel(built, monkeypatch):
    """RANDT_BUILD_TILED=1 sends ordinary 2000-point scans through the tiled path: the maps must be bit-identical to what the
    LDS kernel builds (and to the oracle), incl. slot collisions of a misaligned clustering grid."""
    import torch

    dev = torch.device("cuda:0")
    w = synth.make_world()
    tr = synth.make_trajectory(3000, 6)
    scans = np.stack([synth.make_scan(w, tr[i], 9700 + i) for i in range(6)])
    outs = {}
    for tiled in ("0", "1"):
        monkeypatch.setenv("RANDT_BUILD_TILED", tiled)
        ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)    # the knob is read when the context is created
        maps = R.Maps(ctx, 6, R.indoor_map_params(), 512, with_grid=True)
        R.ndt_build_batch(ctx, torch.from_numpy(scans).to(dev), R.indoor_cluster_params(), maps)
        outs[tiled] = [maps.download(i) for i in range(6)]
        # outdoor parameters: clustering grid 1.2308 m vs map grid 1.2 m
        mapp = R.MapParams(41, 41, 1.2, 0.0, 0.0, 4.0, 3, 0)
        clu = R.ClusterParams(int((2 * 16 / 1.2) ** 2), 16.0)
        rng = np.random.default_rng(0)
        pts = np.zeros((3000, 4), dtype=F)
        pts[:, :2] = rng.uniform(-14, 14, (3000, 2))
        pts[:, 3] = rng.uniform(20, 80, 3000)
        m2 = R.Maps(ctx, 1, mapp, 1024, with_grid=True)
        R.ndt_build_batch(ctx, torch.from_numpy(pts[None]).to(dev), clu, m2)
        outs[tiled].append(m2.download(0))
    for (c0, g0), (c1, g1) in zip(outs["0"], outs["1"]):
        assert cells_equal(c0, c1) and np.array_equal(g0, g1)
    for i in range(6):
        om = oracle_scan_map(scans[i])
        assert cells_equal(outs["1"][i][0], om.cells())


def test_cluster_means_outside_the_map_are_dropped_in_order(env):
    """A map smaller than the scan: clusters whose mean leaves the index grid are dropped (reference: vector::at throws) and
    the later cells move down -- same compact order as the oracle."""
    torch, dev, ctx = env
    pts = _dense_scan(9000, 9800)
    mapp = R.MapParams(24, 24, 0.5, 0.0, 0.0, 4.0, 5, 0)          # 12 m x 12 m around the sensor, the scan reaches 12 m
    maps = R.Maps(ctx, 1, mapp, 2048, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts[None]).to(dev), R.indoor_cluster_params(), maps)
    om = po.Map(24, 24, 0.5, (0.0, 0.0), 4.0, 5, 2048)
    om.build(pts, IP["n_clusters"], IP["max_range"])
    cells, grid = maps.download(0)
    full = oracle_scan_map(pts, cap=2048)
    assert 0 < om.n_cells < full.n_cells                          # some clusters really were dropped
    assert len(cells) == om.n_cells and cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())


@pytest.mark.parametrize("case", ["far_outliers", "fine_cluster_grid", "batch_of_mixed_scans", "ragged_counts"])
def test_label_ranges_beyond_a_tile_take_the_sorting_path(env, case):
    """Scans above 7168 points whose cluster labels span more than the 8192 bins a tile of the counting sort holds -- garbage
    returns thousands of max_range away, or a cluster grid of more than ~7900 clusters (grid.cpp:8-11 takes any n_clusters) --
    were refused until round 4; they now go through a stable device radix sort of (label, point index), which IS labelClouds'
    order: same cells, same compact order, same index grid as the oracle."""
    torch, dev, ctx = env
    clu = R.indoor_cluster_params()
    if case == "far_outliers":
        scans = [_dense_scan(8000, 9900).copy()]
        scans[0][::97, 0] += 30000.0                              # garbage returns 2500 x max_range away
        scans[0][5::131, 1] -= 17000.0
    elif case == "fine_cluster_grid":
        clu = R.ClusterParams(16384, 24.0)                        # 128 x 128 clusters
        scans = [_dense_scan(12000, 9910)]
    elif case == "batch_of_mixed_scans":                          # one ordinary scan between two wide ones: only those take the sort
        a, b, c = _dense_scan(9000, 9920).copy(), _dense_scan(9000, 9921), _dense_scan(9000, 9922).copy()
        a[::61, 1] += 9000.0
        c[3::73, 0] -= 12345.0
        scans = [a, b, c]
    else:
        a, b = _dense_scan(9000, 9930).copy(), _dense_scan(9000, 9931).copy()
        a[::89, 0] += 20000.0
        b[::89, 0] += 20000.0
        scans = [a, b]
    pts = np.stack(scans)
    n_pts = None
    if case == "ragged_counts":
        n_pts = np.array([8123, 7400], dtype=np.int32)
    maps = R.Maps(ctx, len(scans), R.indoor_map_params(), 4096, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts).to(dev), clu, maps, n_points=None if n_pts is None else torch.from_numpy(n_pts).to(dev))
    ctx.synchronize()
    for i, sc in enumerate(scans):
        n = len(sc) if n_pts is None else int(n_pts[i])
        om = po.Map(IP["size_x"], IP["size_y"], IP["resolution"], (0.0, 0.0), IP["max_neighbour_dist"], IP["min_points_per_cell"], 4096)
        om.build(sc[:n], clu.n_clusters, clu.max_range)
        cells, grid = maps.download(i)
        assert om.n_cells > 50 and len(cells) == om.n_cells, (case, i, len(cells), om.n_cells)
        assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid()), (case, i)
