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


def test_storage_pool_reuses_blocks_and_clones_are_exact(built):
    """Round-4 verdict, item 1(a): destroyed batches park their storage in the context's pool (no hipFree, no synchronisation),
    created / cloned batches take it from there (no hipMalloc); a recycled block starts as clean as a fresh one; randt_maps_clone
    = create + copy in one launch; the pool can be trimmed and switched off (RANDT_POOL_MAX_BYTES=0 is exercised by the bench)."""
    import ctypes as C

    import torch

    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    d_pts = torch.from_numpy(_scan(1200)[None]).cuda()

    a = R.Maps(ctx, 1, mapp, 600, with_grid=True)
    R.ndt_build_batch(ctx, d_pts, clu, a)
    cells_a, grid_a = a.download(0)
    assert len(cells_a) > 20
    s0 = ctx.pool_stats()
    b = a.clone()                      # a pool miss: nothing is parked yet
    s1 = ctx.pool_stats()
    assert s1["device_allocs"] == s0["device_allocs"] + 1 and s1["pool_hits"] == s0["pool_hits"] and s1["stream_syncs"] == s0["stream_syncs"]
    cells_b, grid_b = b.download(0)
    assert cells_b.tobytes() == cells_a.tobytes() and np.array_equal(grid_a, grid_b)
    s1 = ctx.pool_stats()
    b.close()                          # parks the block: no free, no synchronisation
    s2 = ctx.pool_stats()
    assert s2["device_frees"] == s1["device_frees"] and s2["stream_syncs"] == s1["stream_syncs"]
    assert s2["pool_blocks"] == 1 and s2["pool_bytes"] >= 600 * 48 + 4 * 10000
    c = R.Maps(ctx, 1, mapp, 600, with_grid=True)       # same size: served from the pool, and CLEAN although the block held a map
    s3 = ctx.pool_stats()
    assert s3["device_allocs"] == s2["device_allocs"] and s3["pool_hits"] == s2["pool_hits"] + 1 and s3["pool_blocks"] == 0
    cells_c, grid_c = c.download(0)
    assert len(cells_c) == 0 and (grid_c == -1).all()
    # ... over its whole capacity, like a fresh one (the cells behind `count` are what randt_maps_copy / RCCL broadcasts move)
    d_cells = C.c_void_p()
    ctx._check(ctx._lib.randt_maps_device_ptrs(c._h, C.byref(d_cells), None, None), "randt_maps_device_ptrs")
    whole = torch.ones(600 * 48, dtype=torch.uint8, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.c_void_p(whole.data_ptr()), d_cells, 600 * 48, 3) == 0
    assert int(whole.cpu().numpy().astype(np.int64).sum()) == 0
    # a much smaller request does not take a big block (<= 2x rule): the 70 KB class is not handed to a 3 KB scan map
    c.close()
    small = R.Maps(ctx, 1, mapp, 64, with_grid=False)
    s4 = ctx.pool_stats()
    assert s4["pool_blocks"] == 1 and s4["device_allocs"] == s3["device_allocs"] + 1
    small.close()
    ctx.pool_trim()
    s5 = ctx.pool_stats()
    assert s5["pool_blocks"] == 0 and s5["pool_bytes"] == 0 and s5["device_frees"] == s4["device_frees"] + 2
    # merges / transforms take their poses through the pinned ring: no synchronisation, same result as the oracle-checked paths
    m = a.clone()
    sub = R.Maps(ctx, 1, mapp, mapp.size_x * mapp.size_y, with_grid=True)
    before = ctx.pool_stats()["stream_syncs"]
    m.transform(0, np.array([[np.cos(0.3), np.sin(0.3), 1.0, -2.0]]))
    sub.merge(0, a, 0, np.array([[1.0, 0.0, 0.0, 0.0]]))
    assert ctx.pool_stats()["stream_syncs"] == before
    got, _ = sub.download(0)
    assert 20 < len(got) <= len(cells_a)      # (cells that share a slot are merged, all others inserted)
    moved, _ = m.download(0)
    c3, s3_ = np.cos(0.3), np.sin(0.3)
    assert np.allclose(moved["mean"][:, 0], c3 * cells_a["mean"][:, 0] - s3_ * cells_a["mean"][:, 1] + 1.0, atol=1e-5)


def test_storage_pool_can_be_switched_off(built):
    """RANDT_POOL_MAX_BYTES=0 (read at context creation): nothing is parked, every destroy waits for the stream and frees -- the
    behaviour before round 5, kept as the comparison leg of bench.py's cpp_local_fuser_drive."""
    import os

    import torch

    old = os.environ.get("RANDT_POOL_MAX_BYTES")
    os.environ["RANDT_POOL_MAX_BYTES"] = "0"
    try:
        ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    finally:
        if old is None:
            del os.environ["RANDT_POOL_MAX_BYTES"]
        else:
            os.environ["RANDT_POOL_MAX_BYTES"] = old
    mapp = R.indoor_map_params()
    s0 = ctx.pool_stats()
    for _ in range(3):
        m = R.Maps(ctx, 1, mapp, 256, with_grid=True)
        m.close()
    s1 = ctx.pool_stats()
    assert s1["device_allocs"] == s0["device_allocs"] + 3 and s1["device_frees"] == s0["device_frees"] + 3
    assert s1["stream_syncs"] == s0["stream_syncs"] + 3 and s1["pool_hits"] == s0["pool_hits"] and s1["pool_blocks"] == 0


def test_cluster_list_in_one_launch_equals_the_batched_build_and_the_loop(env):
    """randt_maps_insert_clusters = HierarchicalMap::addClusters (ndt_hierarchical_map.cpp:28-33) on a whole cluster list in ONE
    launch: the very map -- cells, order, index grid, bit for bit -- that one insertCluster call per cluster gives and that the
    one-launch build of the scan gives; appending to a map that already holds cells; clusters that cannot be placed."""
    torch, dev, ctx = env
    pts = _scan(1300)
    labels = po.grid_labels(pts, IP["n_clusters"], IP["max_range"])
    order = np.argsort(labels, kind="stable")                       # labelClouds: ascending label, cloud order inside a cluster
    sorted_pts = pts[order]
    uniq, first = np.unique(labels[order], return_index=True)
    offsets = np.concatenate([first, [len(pts)]]).astype(np.int32)
    whole = R.Maps(ctx, 1, R.indoor_map_params(), 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(pts[None]).to(dev), R.indoor_cluster_params(), whole)
    lst = R.Maps(ctx, 1, R.indoor_map_params(), 512, with_grid=True)
    n_acc = lst.insert_clusters(0, sorted_pts, offsets)
    a, ga = whole.download(0)
    b, gb = lst.download(0)
    assert n_acc == len(a) > 20 and cells_equal(a, b) and np.array_equal(ga, gb)
    # asynchronous form, in two halves onto a map that is not empty any more: the same map again
    half = len(uniq) // 2
    two = R.Maps(ctx, 1, R.indoor_map_params(), 512, with_grid=True)
    before = ctx.pool_stats()["stream_syncs"]
    assert two.insert_clusters(0, sorted_pts[:offsets[half]], offsets[:half + 1], wait=False) is None
    assert two.insert_clusters(0, sorted_pts[offsets[half]:], offsets[half:] - offsets[half], wait=False) is None
    assert ctx.pool_stats()["stream_syncs"] == before
    c, gc = two.download(0)
    assert cells_equal(a, c) and np.array_equal(ga, gc)
    # a cluster outside the index grid and a capacity that runs out: reported, the rest placed
    far = sorted_pts.copy()
    k0 = int(np.argmax(np.diff(offsets) > IP["min_points_per_cell"]))
    far[offsets[k0]:offsets[k0 + 1], :2] += 500.0                     # (x alone would stay "inside": the reference only checks the flat index)
    out = R.Maps(ctx, 1, R.indoor_map_params(), 512, with_grid=True)
    with pytest.raises(R.RandtError) as e:
        out.insert_clusters(0, far, offsets)
    assert e.value.status == R._capi.ERR_INVALID and len(out.download(0)[0]) == len(a) - 1
    tiny = R.Maps(ctx, 1, R.indoor_map_params(), 10, with_grid=True)
    with pytest.raises(R.RandtError) as e:
        tiny.insert_clusters(0, sorted_pts, offsets)
    assert e.value.status == R._capi.ERR_UNSUPPORTED
    t, _ = tiny.download(0)
    assert len(t) == 10 and cells_equal(t, a[:10])
    tiny2 = R.Maps(ctx, 1, R.indoor_map_params(), 10, with_grid=True)
    tiny2.insert_clusters(0, sorted_pts, offsets, wait=False)          # asynchronous: the status arrives with the next read, once
    with pytest.raises(R.RandtError) as e:
        tiny2.counts()
    assert e.value.status == R._capi.ERR_UNSUPPORTED
    assert tiny2.counts()[0] == 10
    # the same deferred report through the DOWNLOAD: the status comes back once, behind completed outputs -- the
    # count, the cells that were placed and the index grid -- not instead of them
    far2 = R.Maps(ctx, 1, R.indoor_map_params(), 512, with_grid=True)
    far2.insert_clusters(0, far, offsets, wait=False)
    rc, cells_f, grid_f = far2.download_with_status(0)
    ref_f, ref_g = out.download(0)                                     # the synchronous insert of the same list, checked above
    assert rc == R._capi.ERR_INVALID and len(cells_f) == len(a) - 1 and cells_equal(cells_f, ref_f) and np.array_equal(grid_f, ref_g)
    rc, cells_f2, grid_f2 = far2.download_with_status(0)
    assert rc == 0 and cells_equal(cells_f2, ref_f) and np.array_equal(grid_f2, ref_g)
    tiny3 = R.Maps(ctx, 1, R.indoor_map_params(), 10, with_grid=True)
    tiny3.insert_clusters(0, sorted_pts, offsets, wait=False)
    rc, cells_t, _ = tiny3.download_with_status(0)
    assert rc == R._capi.ERR_UNSUPPORTED and len(cells_t) == 10 and cells_equal(cells_t, a[:10])


def test_destroying_a_batch_another_contexts_stream_still_reads(built):
    """ADVICE r5 #3.  randt_maps_destroy parks the block without a host synchronisation, and the block's next owner is
    served by the OWNER's stream -- so a reader on another context's stream used to race with it silently.  The batch now
    remembers the contexts it was handed to, and its destruction makes the owner's stream wait (on the device) for a marker
    on each of their streams: context B keeps six registration batches in flight against a batch context A owns, A destroys
    it and immediately takes the same block for a new, cleared batch -- B's results are those of the undisturbed run, A made
    no host wait, and `foreign_waits` counted the one device-side wait."""
    import torch
    from util import GpuRig

    rig = GpuRig(problem(n_submaps=2, scans_per_submap=32, n_keyframes=12))
    rig.build_submaps()
    rig.build_scans()
    a = rig.ctx
    side = torch.cuda.Stream()
    b = R.Context(0, side.cuda_stream)
    mp = R.default_matcher_params()
    g4 = synth.pose3_to_pose4(rig.prob["guess"])
    n = rig.B

    def run(fixed, reps):
        poses = [torch.from_numpy(g4.copy()).to(rig.dev) for _ in range(reps)]
        res = [torch.zeros((n, 64), dtype=torch.uint8, device=rig.dev) for _ in range(reps)]
        torch.cuda.synchronize()
        for r in range(reps):
            R.register_batch(b, fixed, rig.fixed_idx, rig.scan_maps, 0, n, mp, poses[r], res[r])
        return poses, res

    ref, _ = run(rig.submaps, 1)
    b.synchronize()
    ref = ref[0].cpu().numpy()
    assert np.abs(ref - g4).max() > 1e-3                      # the registrations do move their poses

    shared = rig.submaps.clone()                              # owned by A
    a.synchronize()
    s0 = a.pool_stats()
    poses, _ = run(shared, 6)                                 # B: ~1 ms of kernels in flight on its own stream
    shared.close()                                            # A: parks the block ...
    fresh = R.Maps(a, rig.n_sub, rig.mapp, rig.mapp.size_x * rig.mapp.size_y, with_grid=True)   # ... and clears it for its next owner
    s1 = a.pool_stats()
    assert s1["pool_hits"] == s0["pool_hits"] + 1             # the very block
    assert s1["foreign_waits"] == s0["foreign_waits"] + 1 and s1["stream_syncs"] == s0["stream_syncs"]
    b.synchronize()
    a.synchronize()
    for p in poses:
        assert np.array_equal(p.cpu().numpy(), ref)
    assert fresh.counts().sum() == 0
    # a batch nobody else has used costs no wait
    own = rig.submaps.clone()
    own.close()
    assert a.pool_stats()["foreign_waits"] == s1["foreign_waits"]
