"""-m gpu edge cases through the C ABI: empty / ragged / tiny / maximum-size scans, PCL point
stride, misaligned clustering grid (slot collisions), map upload/download/copy/clear round trips,
scans with no usable cells (no residuals -> pose unchanged), run-to-run determinism."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import host, synth
from util import IP, cells_equal, oracle_map, oracle_scan_map

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def env(built):
    import torch

    dev = torch.device("cuda:0")
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    return torch, dev, ctx


def _scan(seed):
    w = synth.make_world()
    tr = synth.make_trajectory(3000, 2)
    return synth.make_scan(w, tr[0], seed)


def test_ragged_and_empty_scans(env):
    torch, dev, ctx = env
    base = _scan(1000)
    n_pts = np.array([2000, 0, 5, 6, 1234], dtype=np.int32)   # empty, below / at the min-points gate, ragged
    pts = np.stack([base] * len(n_pts))
    pts[3, :6, :2] = [3.2, 1.2]                                 # 6 coincident-ish points -> exactly one cell
    pts[3, :6, 0] += np.linspace(0, 0.05, 6)
    maps = R.Maps(ctx, len(n_pts), R.indoor_map_params(), 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts).to(dev), R.indoor_cluster_params(), maps, n_points=torch.from_numpy(n_pts).to(dev))
    ctx.synchronize()
    counts = maps.counts()
    for i, n in enumerate(n_pts):
        om = oracle_scan_map(pts[i, :n]) if n else oracle_map(512)
        cells, grid = maps.download(i)
        assert counts[i] == om.n_cells
        assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())
    assert counts[1] == 0 and counts[2] == 0 and counts[3] == 1


def test_pcl_stride_and_host_entry_point(env):
    torch, dev, ctx = env
    base = _scan(1001)
    pcl = np.zeros((2000, 8), dtype=F)                          # pcl::PointXYZI: x y z pad I pad pad pad
    pcl[:, :2] = base[:, :2]
    pcl[:, 4] = base[:, 3]
    maps = R.Maps(ctx, 2, R.indoor_map_params(), 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pcl[None]).to(dev), R.indoor_cluster_params(), maps, first_map=0)
    host.ndt_build_host(ctx, base, R.indoor_cluster_params(), maps, 1)   # randt_ndt_build (host pointer, packed xyzI)
    a, ga = maps.download(0)
    b, gb = maps.download(1)
    om = oracle_scan_map(base)
    assert cells_equal(a, om.cells()) and cells_equal(b, om.cells()) and np.array_equal(ga, gb)


def test_largest_one_workgroup_scans(env):
    torch, dev, ctx = env
    maps = R.Maps(ctx, 1, R.indoor_map_params(), 2048, with_grid=True)
    for n in (4096, 7168):                                      # 7168 = the one-workgroup (LDS) build kernel's maximum
        big = np.concatenate([_scan(1002 + i) for i in range(4)])[:n]
        R.ndt_build_batch(ctx, torch.from_numpy(big[None]).to(dev), R.indoor_cluster_params(), maps)
        cells, grid = maps.download(0)
        om = oracle_scan_map(big, cap=2048)
        assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())
    # larger scans take the multi-workgroup path: tests/test_gpu_big_scans.py


def test_out_of_range_points_take_the_fallback_sort(env):
    """labels far outside the clustering grid (points beyond max_range) exceed the LDS label bins:
    the kernel's rank-by-counting fallback must give the same cells."""
    torch, dev, ctx = env
    pts = _scan(1006).copy()
    pts[::97, 0] += 300.0                                       # garbage returns far away
    pts[5::101, 1] -= 450.0
    maps = R.Maps(ctx, 1, R.indoor_map_params(), 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts[None]).to(dev), R.indoor_cluster_params(), maps)
    cells, grid = maps.download(0)
    om = oracle_scan_map(pts)
    assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())


def test_slot_collisions_misaligned_grids(env):
    """outdoor parameter set: clustering grid 1.2308 m vs map grid 1.2 m (quirk A.7-5)."""
    torch, dev, ctx = env
    mapp = R.MapParams(41, 41, 1.2, 0.0, 0.0, 4.0, 3, 0)
    clu = R.ClusterParams(int((2 * 16 / 1.2) ** 2), 16.0)
    rng = np.random.default_rng(0)
    pts = np.zeros((3000, 4), dtype=F)
    pts[:, :2] = rng.uniform(-14, 14, (3000, 2))
    pts[:, 3] = rng.uniform(20, 80, 3000)
    maps = R.Maps(ctx, 1, mapp, 1024, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts[None]).to(dev), clu, maps)
    om = po.Map(41, 41, 1.2, (0, 0), 4.0, 3, 1024)
    om.build(pts, clu.n_clusters, clu.max_range)
    cells, grid = maps.download(0)
    assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())
    assert (grid >= 0).sum() < len(cells)                      # collisions really happened


def test_map_roundtrip_copy_clear(env):
    torch, dev, ctx = env
    om = oracle_scan_map(_scan(1004))
    maps = R.Maps(ctx, 3, R.indoor_map_params(), 512, with_grid=True)
    maps.upload(1, om.cells(), om.grid())
    c, g = maps.download(1)
    assert cells_equal(c, om.cells()) and np.array_equal(g, om.grid())
    maps.copy_from(maps, dst_first=2, src_first=1, count=1)
    c2, g2 = maps.download(2)
    assert cells_equal(c2, om.cells()) and np.array_equal(g2, om.grid())
    maps.transform(2, [[np.cos(0.3), np.sin(0.3), 1.0, -2.0]])
    om.transform([np.cos(0.3), np.sin(0.3), 1.0, -2.0])
    c3, g3 = maps.download(2)
    assert cells_equal(c3, om.cells()) and np.array_equal(g3, g2)   # transformMap leaves the grid stale
    maps.clear(1, 2)
    assert maps.counts().tolist() == [0, 0, 0] and (maps.download(2)[1] == -1).all()


def test_no_residuals_keeps_pose(env):
    torch, dev, ctx = env
    mapp = R.indoor_map_params()
    sub = R.Maps(ctx, 1, mapp, 64, with_grid=True)             # empty submap
    scans = R.Maps(ctx, 1, mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(_scan(1005)[None]).to(dev), R.indoor_cluster_params(), scans)
    g = np.array([0.9, 0.1, 1.0, 2.0])
    p, r = host.register_pair(ctx, sub, 0, scans, 0, R.default_matcher_params(), g)
    assert np.array_equal(p, g) and r["n_residuals"] == 0 and r["status"] == 1 and r["cost"] == 0.0


def test_run_to_run_bitwise_determinism(env):
    torch, dev, ctx = env
    prob = synth.make_batch_problem(n_submaps=1, scans_per_submap=6, n_keyframes=6)
    mapp, clu, mp = R.indoor_map_params(), R.indoor_cluster_params(), R.default_matcher_params(parameterization=R.PARAM_MANIFOLD)
    sub = R.Maps(ctx, 1, mapp, 10000, with_grid=True)
    kf = torch.from_numpy(np.stack(prob["submaps"][0]["kf_scans"])).to(dev)
    tmp = R.Maps(ctx, kf.shape[0], mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, kf, clu, tmp)
    sub.merge(0, tmp, 0, synth.pose3_to_pose4(prob["submaps"][0]["kf_rel"]))
    pts = torch.from_numpy(prob["scans"]).to(dev)
    fidx = torch.zeros(6, dtype=torch.int32, device=dev)
    ws = R.Maps(ctx, 6, mapp, 512, with_grid=False)
    outs = []
    for _ in range(3):
        pose = torch.from_numpy(synth.pose3_to_pose4(prob["guess"])).to(dev)
        res = torch.zeros((6, 64), dtype=torch.uint8, device=dev)
        R.scan_register_batch(ctx, pts, clu, sub, fidx, ws, mp, pose, res)
        ctx.synchronize()
        outs.append((pose.cpu().numpy().copy(), res.cpu().numpy().copy()))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])   # fixed reduction tree


# ---------------------------------------------------------------------------------------------------
# paths added with the round-1 kernel rewrites
def _build_both(env, pts, mapp_args, clu_args, cap, ioff=3):
    torch, dev, ctx = env
    mapp, clu = R.MapParams(*mapp_args), R.ClusterParams(*clu_args)
    maps = R.Maps(ctx, 1, mapp, cap, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts[None]).to(dev), clu, maps)
    om = po.Map(mapp_args[0], mapp_args[1], mapp_args[2], (mapp_args[3], mapp_args[4]), mapp_args[5], mapp_args[6], cap)
    om.build(pts, clu.n_clusters, clu.max_range, ioff=ioff)
    cells, grid = maps.download(0)
    return cells, grid, om, maps


@pytest.mark.parametrize("n_points", [1984, 2000, 2048])
@pytest.mark.parametrize("min_points", [0, 1, 2, 3])
def test_build_lds_budget_for_every_min_points(env, n_points, min_points):
    """The build kernel's LDS holds one list entry per cluster that can become a cell (n / (min_points + 1) of them): small
    `min_points_per_cell` values at the largest register-resident scan sizes sit right at the five-workgroups-per-CU budget (a
    randomised soak found the launcher refusing 2048 points at min_points = 2).  Same cells as the oracle for all of them."""
    rng = np.random.default_rng(100 * n_points + min_points)
    pts = np.zeros((n_points, 4), dtype=F)
    pts[:, :2] = (rng.integers(-20, 20, (n_points, 2)) * 0.5 + 0.25 + rng.normal(0, 0.05, (n_points, 2))).astype(F)
    pts[:, 3] = rng.uniform(10, 90, n_points)
    cells, grid, om, _ = _build_both(env, pts, (100, 100, 0.5, 0.0, 0.0, 4.0, min_points, 0), (2304, 24.0), 2048)
    assert om.n_cells > 50
    assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())


def test_cluster_means_outside_the_map_shift_the_cell_indices(env):
    """clusters whose mean leaves a SMALL map are dropped (vector::at in the reference): every later cell's
    compact index shifts, which the build kernel handles by redoing the statistics in strict cluster order."""
    rng = np.random.default_rng(3)
    pts = np.zeros((2000, 4), dtype=F)
    pts[:, :2] = rng.uniform(-20, 20, (2000, 2))                # map below is only +-6 m
    pts[:, 3] = rng.uniform(10, 90, 2000)
    # dense blobs so that clusters pass min_points both inside and outside the map
    pts[:900, :2] = rng.normal(0, 0.08, (900, 2)) + rng.integers(-18, 18, (900, 1)).astype(F) * np.array([[1.0, 0.7]], dtype=F)
    cells, grid, om, _ = _build_both(env, pts, (24, 24, 0.5, 0.0, 0.0, 4.0, 3, 0), (6400, 20.0), 512)
    assert om.n_cells > 5
    assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())


def test_cell_capacity_overflow_keeps_the_first_cells(env):
    rng = np.random.default_rng(4)
    pts = np.zeros((4000, 4), dtype=F)
    pts[:, :2] = (rng.integers(-30, 30, (4000, 2)) * 0.5 + 0.25 + rng.normal(0, 0.03, (4000, 2))).astype(F)
    pts[:, 3] = rng.uniform(10, 90, 4000)
    cells, grid, om, maps = _build_both(env, pts, (100, 100, 0.5, 0.0, 0.0, 4.0, 0, 0), (2304, 24.0), 64)
    assert len(cells) == 64 == om.n_cells                       # more clusters than capacity
    assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())


def test_many_small_clusters_take_the_unranked_order(env):
    """> 256 clusters: the size-class hand-out still applies (any order gives the same cells)."""
    rng = np.random.default_rng(5)
    pts = np.zeros((7000, 4), dtype=F)
    pts[:, :2] = rng.uniform(-23.9, 23.9, (7000, 2))
    pts[:, 3] = rng.uniform(10, 90, 7000)
    cells, grid, om, _ = _build_both(env, pts, (100, 100, 0.5, 0.0, 0.0, 4.0, 1, 0), (2304, 24.0), 4096)
    assert om.n_cells > 256
    assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid())


def _assoc_both(env, fixed_pts, moving_pts, mapp_args, clu_args, k, mahal=1, intensity=1, guess=(1.0, 0.0, 0.0, 0.0)):
    torch, dev, ctx = env
    mapp, clu = R.MapParams(*mapp_args), R.ClusterParams(*clu_args)
    fmap = R.Maps(ctx, 1, mapp, 4096, with_grid=True)
    mmap = R.Maps(ctx, 1, mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(fixed_pts[None]).to(dev), clu, fmap)
    R.ndt_build_batch(ctx, torch.from_numpy(moving_pts[None]).to(dev), clu, mmap)
    mp = R.default_matcher_params(n_neighbours=k, lookup_mahalanobis=mahal, use_intensity=intensity)
    g = torch.tensor([guess], dtype=torch.float64, device=dev)
    corr = torch.full((1, 512, k), -7, dtype=torch.int32, device=dev)
    R.associate_batch(ctx, fmap, torch.zeros(1, dtype=torch.int32, device=dev), mmap, 0, 1, g, mp, corr)
    ctx.synchronize()

    def omap(cap):
        return po.Map(mapp_args[0], mapp_args[1], mapp_args[2], (mapp_args[3], mapp_args[4]), mapp_args[5], mapp_args[6], cap)

    of, om = omap(4096), omap(512)
    of.build(fixed_pts, clu.n_clusters, clu.max_range)
    om.build(moving_pts, clu.n_clusters, clu.max_range)
    oc, _ = po.associate(of, om, np.array(guess), k, mahal, intensity)
    return corr.cpu().numpy()[0, : om.n_cells], oc, om.n_cells, of.n_cells


def _blobs(seed, n_blobs, extent, n=2000, sigma=0.07):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n_blobs, 2))
    pts = np.zeros((n, 4), dtype=F)
    pts[:, :2] = (c[rng.integers(0, n_blobs, n)] + rng.normal(0, sigma, (n, 2))).astype(F)
    pts[:, 3] = rng.uniform(10, 90, n)
    return pts


@pytest.mark.parametrize("max_dist,k", [(0.5, 1), (1.0, 3), (2.0, 8), (4.0, 8), (4.0, 1)])
@pytest.mark.parametrize("n_fixed_blobs", [6, 60])
def test_association_termination_radius_variants(env, max_dist, k, n_fixed_blobs):
    """sparse and dense fixed maps, every window limit rmax = max_dist / 0.5 in {1, 2, 4, 8}: the search stops at
    different radii (enough targets, or the last radius), which the kernel decides with one ballot."""
    mapp = (100, 100, 0.5, 0.0, 0.0, max_dist, 3, 0)
    got, want, nm, nf = _assoc_both(env, _blobs(11, n_fixed_blobs, 20.0), _blobs(12, 40, 20.0), mapp, (2304, 24.0), k,
                                    guess=(np.cos(0.05), np.sin(0.05), 0.3, -0.2))
    assert nm > 5 and nf > 3
    assert np.array_equal(got, want)


@pytest.mark.parametrize("res,size,max_dist,k", [(0.25, 200, 4.0, 4), (0.25, 200, 4.0, 8), (0.25, 200, 3.0, 12), (0.5, 100, 4.0, 12),
                                                 (0.5, 100, 4.0, 16), (0.25, 200, 4.0, 16), (0.2, 250, 3.0, 5)])
@pytest.mark.parametrize("n_fixed_blobs", [6, 60])
def test_association_beyond_the_shipped_window_and_neighbour_counts(env, res, size, max_dist, k, n_fixed_blobs):
    """Configurations the reference takes and no shipped YAML uses (round-3 verdict, missing 3 / item 6):
    rmax = int(max_neighbour_manhattan_distance / resolution) up to 16 (ndt_map.cpp:117: a 0.25 m map with the indoor 4 m
    window searches 31 x 31 slots) and n_results_kd_lookup up to 16 (ndt_matcher.cpp:210) -- the association's WIDE
    instantiation (ring-by-ring outer window, 144 candidates, 16-entry top-k).  Identical tables to the oracle."""
    mapp = (size, size, res, 0.0, 0.0, max_dist, 3, 0)
    got, want, nm, nf = _assoc_both(env, _blobs(31, n_fixed_blobs, 20.0), _blobs(32, 40, 20.0), mapp, (2304, 24.0), k,
                                    guess=(np.cos(-0.04), np.sin(-0.04), 0.2, 0.25))
    assert nm > 5 and nf > 3 and int(max_dist / res) <= 16
    assert np.array_equal(got, want)
    assert (want >= 0).sum() >= (nm if n_fixed_blobs > 6 else 4)  # neighbours were really found (sparse map: few, at the outer radii)


def test_whole_registration_with_a_wide_window_and_twelve_neighbours(env):
    """estimateLoopConstraint end to end on such a configuration: 0.25 m cells, 4 m window, k = 12 -- build, WIDE association,
    solve (more than 1024 residual slots: the raw-slot walk of the pass) against the oracle."""
    torch, dev, ctx = env
    mapp_args = (200, 200, 0.25, 0.0, 0.0, 4.0, 3, 0)
    mapp, clu = R.MapParams(*mapp_args), R.ClusterParams(9216, 24.0)
    world = synth.make_world()
    tr = synth.make_trajectory(3100, 2)
    fixed_pts, moving_pts = synth.make_scan(world, tr[0], 41), synth.make_scan(world, tr[0], 42)
    fmap = R.Maps(ctx, 1, mapp, 4096, with_grid=True)
    mmap = R.Maps(ctx, 1, mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(fixed_pts[None]).to(dev), clu, fmap)
    R.ndt_build_batch(ctx, torch.from_numpy(moving_pts[None]).to(dev), clu, mmap)
    mp = R.default_matcher_params(n_neighbours=12)
    g4 = np.array([[np.cos(0.03), np.sin(0.03), 0.15, -0.1]])
    pose = torch.from_numpy(g4.copy()).to(dev)
    res = torch.zeros((1, 64), dtype=torch.uint8, device=dev)
    R.register_batch(ctx, fmap, torch.zeros(1, dtype=torch.int32, device=dev), mmap, 0, 1, mp, pose, res)
    ctx.synchronize()
    of = po.Map(*mapp_args[:3], (0.0, 0.0), mapp_args[5], mapp_args[6], 4096)
    om = po.Map(*mapp_args[:3], (0.0, 0.0), mapp_args[5], mapp_args[6], 512)
    of.build(fixed_pts, clu.n_clusters, clu.max_range)
    om.build(moving_pts, clu.n_clusters, clu.max_range)
    op = po.default_params()
    for name, _ in R.MatcherParams._fields_:
        if name != "reserved":
            setattr(op, name, getattr(mp, name))
    rc, p4, cost, st = po.register_pair(of, om, op, g4[0])
    r = res.cpu().numpy().view(R.RESULT_DTYPE)[0]
    assert rc == 0 and r["status"] == 0 and r["n_residuals"] == st["n_residuals"] > 200
    assert np.abs(pose.cpu().numpy()[0] - p4).max() <= 1e-7


def test_association_on_a_map_narrower_than_the_window(env):
    """size_x <= 2 (rmax - 1): the search window wraps onto itself and the reference removes repeated
    entries (std::find) -- the radius-by-radius path of the kernel."""
    mapp = (12, 40, 0.5, 0.0, 0.0, 4.0, 3, 0)
    fixed = _blobs(21, 30, 2.5)
    fixed[:, 1] *= 3.5
    moving = _blobs(22, 20, 2.5)
    moving[:, 1] *= 3.5
    got, want, nm, nf = _assoc_both(env, fixed, moving, mapp, (2304, 24.0), 5)
    assert nm > 3 and nf > 3
    assert np.array_equal(got, want)


@pytest.mark.parametrize("sx,sy,res,max_dist,k", [(15, 15, 0.5, 4.0, 4), (16, 14, 0.5, 4.0, 3), (10, 20, 0.5, 4.0, 5), (12, 12, 0.5, 2.0, 2),
                                                  (8, 8, 1.0, 4.0, 4), (6, 30, 0.5, 1.5, 3), (20, 11, 0.4, 1.0, 8)])
def test_association_on_maps_smaller_than_one_window(env, sx, sy, res, max_dist, k):
    """Maps with no more slots than one search window (<= 225): the reference's loop then also ends on `adjacent.size() <
    n_cells_` (every slot seen, ndt_map.cpp:119), and windows wrap in both directions.  Refused until round 4 -- untested, not
    unsupported: the popcount / de-duplication paths give the oracle's tables."""
    ext = 0.45 * min(sx, sy) * res
    mapp = (sx, sy, res, 0.0, 0.0, max_dist, 3, 0)
    got, want, nm, nf = _assoc_both(env, _blobs(51, 14, ext, n=800), _blobs(52, 10, ext, n=600), mapp, (2304, 24.0), k,
                                    guess=(np.cos(0.1), np.sin(0.1), 0.1, -0.05))
    assert nm > 2 and nf > 2 and sx * sy <= 225
    assert np.array_equal(got, want)


def test_error_convention_status_codes_not_exceptions(env):
    """the ABI never throws and never falls back: bad arguments -> RANDT_ERR_INVALID, sizes beyond the kernels ->
    RANDT_ERR_UNSUPPORTED with a message, outputs untouched (SURVEY 8(b) error convention)."""
    import ctypes as C

    torch, dev, ctx = env
    lib = R._capi.load()
    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    maps = R.Maps(ctx, 2, mapp, 512, with_grid=True)
    pts = torch.from_numpy(_scan(1001)[None]).to(dev)
    # null / out-of-range arguments
    assert lib.randt_ndt_build_batch_dev(ctx._h, None, 1, 2000, None, 4, 3, C.byref(clu), maps._h, 0) == 1
    assert lib.randt_ndt_build_batch_dev(ctx._h, pts.data_ptr(), 3, 2000, None, 4, 3, C.byref(clu), maps._h, 0) == 1   # 3 scans into 2 maps
    assert lib.randt_ndt_build_batch_dev(ctx._h, pts.data_ptr(), 1, 2000, None, 4, 7, C.byref(clu), maps._h, 0) == 1   # intensity index >= stride
    # association limits: k > 16 and a window radius beyond 15 (up to there: the WIDE instantiation, tested above)
    R.ndt_build_batch(ctx, pts, clu, maps)
    mp = R.default_matcher_params(n_neighbours=17)
    guess = torch.tensor([[1.0, 0, 0, 0]], dtype=torch.float64, device=dev)
    corr = torch.full((1, 512, 17), -5, dtype=torch.int32, device=dev)
    fidx = torch.zeros(1, dtype=torch.int32, device=dev)
    with pytest.raises(R.RandtError) as e:
        R.associate_batch(ctx, maps, fidx, maps, 1, 1, guess, mp, corr)
    assert e.value.status == 3 and "n_neighbours" in str(e.value)
    ctx.synchronize()
    assert int((corr.cpu() != -5).sum()) == 0                                      # nothing was written
    far = R.Maps(ctx, 1, R.MapParams(100, 100, 0.5, 0.0, 0.0, 8.6, 5, 0), 512, with_grid=True)   # rmax = 17
    with pytest.raises(R.RandtError) as e:
        R.associate_batch(ctx, far, fidx, maps, 1, 1, guess, R.default_matcher_params(), corr[:, :, :4].contiguous())
    assert e.value.status == 3
    # a map batch without an index grid cannot be the fixed side
    nogrid = R.Maps(ctx, 1, mapp, 512, with_grid=False)
    with pytest.raises(R.RandtError) as e:
        R.associate_batch(ctx, nogrid, fidx, maps, 1, 1, guess, R.default_matcher_params(), corr[:, :, :4].contiguous())
    assert e.value.status == 1 and "index grid" in str(e.value)
    # the context stays usable after errors
    R.associate_batch(ctx, maps, fidx, maps, 1, 1, guess, R.default_matcher_params(), corr[:, :, :4].contiguous())
    ctx.synchronize()


def test_solve_workgroup_grouping_does_not_change_results(env, monkeypatch):
    """RANDT_SOLVE_RPB: 1 / 2 / 4 / 8 registrations (one wavefront each) per solve workgroup is a placement choice only --
    poses and result records must be bit-identical, also when the batch does not fill the last workgroup."""
    torch, dev, _ = env
    from util import GpuRig, problem

    prob = problem()
    mp = R.default_matcher_params(gnc_steps=2)
    outs = {}
    for rpb in ("1", "2", "4", "8"):
        monkeypatch.setenv("RANDT_SOLVE_RPB", rpb)
        rig = GpuRig(prob)                                   # the knob is read when the context is created
        rig.ctx.set_solve_mode(R._capi.SOLVE_THROUGHPUT)     # one wavefront per registration (AUTO would split so small a batch)
        rig.build_submaps()
        rig.build_scans()
        for n_pairs in (1, 7, rig.B):
            pose = torch.from_numpy(np.stack([synth.pose3_to_pose4(g) for g in prob["guess"][:n_pairs]])).to(dev)
            corr = torch.full((n_pairs, rig.scan_cap, mp.n_neighbours), -1, dtype=torch.int32, device=dev)
            res = torch.zeros((n_pairs, 64), dtype=torch.uint8, device=dev)
            R.associate_batch(rig.ctx, rig.submaps, rig.fixed_idx, rig.scan_maps, 0, n_pairs, pose, mp, corr)
            R.solve_batch(rig.ctx, rig.submaps, rig.fixed_idx, rig.scan_maps, 0, n_pairs, corr, mp, pose, res)
            rig.ctx.synchronize()
            outs[(rpb, n_pairs)] = (pose.cpu().numpy().copy(), res.cpu().numpy().copy())
    for n_pairs in (1, 7, len(prob["scans"])):
        p1, r1 = outs[("1", n_pairs)]
        for rpb in ("2", "4", "8"):
            p, r = outs[(rpb, n_pairs)]
            assert np.array_equal(p, p1) and np.array_equal(r, r1), (rpb, n_pairs)


def test_parameter_sets_that_would_hang_the_device_are_rejected(env):
    """The GNC / trust-region loops run on the device: a divisor <= 1, non-positive scales or non-finite tolerances are
    refused with RANDT_ERR_INVALID before anything is enqueued, and the context stays usable."""
    torch, dev, ctx = env
    mapp = R.indoor_map_params()
    maps = R.Maps(ctx, 2, mapp, 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(np.stack([_scan(1200), _scan(1201)])).to(dev), R.indoor_cluster_params(), maps)
    pose = torch.from_numpy(np.array([[1.0, 0.0, 0.0, 0.0]])).to(dev)
    fidx = torch.zeros(1, dtype=torch.int32, device=dev)
    res = torch.zeros((1, 64), dtype=torch.uint8, device=dev)
    bad_sets = [dict(gnc_divisor=1.0), dict(gnc_divisor=0.5), dict(gnc_divisor=float("nan")), dict(gnc_steps=0), dict(max_iterations=-1),
                dict(loss_scale=0.0), dict(mu_scale=0.0), dict(mu_scale=float("inf")), dict(loss_weight=-1.0), dict(function_tolerance=float("nan")),
                dict(initial_radius=0.0), dict(loss_alpha=float("nan")), dict(parameterization=7), dict(max_consecutive_invalid_steps=0)]
    for over in bad_sets:
        mp = R.default_matcher_params(**over)
        with pytest.raises(R.RandtError) as e:
            R.register_batch(ctx, maps, fidx, maps, 1, 1, mp, pose, res)
        assert e.value.status == 1, over
    # window entry point: same gate
    st = np.array([R.make_state([1, 0, 0, 0]), R.make_state([1, 0, 0.1, 0], stamp=0.25)], dtype=R.STATE_DTYPE)
    with pytest.raises(R.RandtError) as e:
        R.register_window(ctx, maps, [0], maps, [1], st, R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_divisor=0.9),
                          R.window_params(), [1, 0, 0, 0])
    assert e.value.status == 1 and "gnc_divisor" in str(e.value)
    # correlative search: a zero / negative step would never terminate on the host
    for over in (dict(linear_step=0.0), dict(linear_step=-0.4), dict(n_iter=0), dict(max_px_accurate_range=0.0), dict(linear_step=float("nan"))):
        with pytest.raises(R.RandtError) as e:
            host.search_global(ctx, maps, 0, maps, 1, R.default_matcher_params(), host.bnb_params(**over), [1, 0, 0, 0])
        assert e.value.status == 1, over
    # still alive
    R.register_batch(ctx, maps, fidx, maps, 1, 1, R.default_matcher_params(), pose, res)
    ctx.synchronize()
    assert np.isfinite(pose.cpu().numpy()).all()


def test_slot_pool_exhaustion_is_a_descriptive_error(env):
    from randt_slam_amd import odometry

    torch, dev, ctx = env
    be = odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params(), scan_slots=2, submap_slots=1)
    s = _scan(1300)
    be.build_scan(s)
    be.build_scan(s)
    with pytest.raises(R.RandtError) as e:
        be.build_scan(s)
    assert "scan slot pool exhausted" in str(e.value) and "scan_slots" in str(e.value)
