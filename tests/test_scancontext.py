"""Row f-4 (loop-closure candidate search): Scan Context descriptor, keys and detectLoopClosureID.
CPU part: the oracle restatement against hand-checkable cases; -m gpu part: the HIP kernels against the oracle
through the C ABI."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import host, synth

F = np.float32


def osp(**kw):
    d = dict(num_ring=20, num_sector=45, max_radius=15.0, num_exclude_recent=15, num_candidates=10, search_ratio=0.3, dist_thresh=0.6,
             assumed_drift=0.05, odom_eps=1.2, odom_weight=0.2, intensity_factor=0.04)
    d.update(kw)
    return po.ScParams(*[d[k] for k in ("num_ring", "num_sector", "max_radius", "num_exclude_recent", "num_candidates", "search_ratio",
                                        "dist_thresh", "assumed_drift", "odom_eps", "odom_weight", "intensity_factor")])


def _pts(xy, inten):
    p = np.zeros((len(xy), 4), dtype=F)
    p[:, :2] = xy
    p[:, 3] = inten
    return p


def test_descriptor_bins_and_no_point_quirk():
    sp = osp()
    # one point at range 7.4 m, azimuth 10 deg -> ring ceil(7.4 / 15 * 20) = 10, sector ceil(10 / 360 * 45) = 2
    a = np.deg2rad(10.0)
    desc, rk, sk = po.sc_make(_pts([[7.4 * np.cos(a), 7.4 * np.sin(a)]] * 3, [10.0, 20.0, 30.0]), sp)
    want = -1000.0 + float(F(10 * 0.04)) + float(F(20 * 0.04)) + float(F(30 * 0.04))   # touched bins start at NO_POINT (:161,187)
    assert desc.shape == (45, 20) and desc[1, 9] == want and np.count_nonzero(desc) == 1
    assert rk[9] == want / 45 and sk[1] == want / 20
    # beyond max_radius: dropped; exactly on the rim: last ring; origin: ring 1 (max(.., 1)), sector 1
    desc, _, _ = po.sc_make(_pts([[15.1, 0.0], [0.0, 15.0], [0.0, 0.0]], [50, 50, 50]), sp)
    assert np.count_nonzero(desc) == 2 and desc[11, 19] != 0 and desc[0, 0] != 0
    # quadrants of xy2theta
    for ang in (30, 120, 200, 300, 359.9):
        x, y = 5 * np.cos(np.deg2rad(ang)), 5 * np.sin(np.deg2rad(ang))
        desc, _, _ = po.sc_make(_pts([[x, y]], [50]), sp)
        s = int(np.nonzero(desc.sum(axis=1))[0][0])
        assert s == int(np.ceil(ang / 360 * 45)) - 1


def _scene_scan(seed, pose, n=1500):
    world = synth.make_world()
    return synth.make_scan(world, pose, seed)


def test_distance_recovers_the_yaw_shift():
    """the same place seen with the sensor rotated by 5 sectors: distance ~ 0 at the matching column shift"""
    sp = osp(max_radius=30.0)
    rng = np.random.default_rng(0)
    rad = rng.uniform(1, 28, 1500)
    ang = rng.uniform(0, 2 * np.pi, 1500)
    inten = rng.uniform(10, 90, 1500)
    a = _pts(np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1), inten)
    rot = 5 * (2 * np.pi / 45)
    b = _pts(np.stack([rad * np.cos(ang + rot), rad * np.sin(ang + rot)], 1), inten)
    da, _, _ = po.sc_make(a, sp)
    db, _, _ = po.sc_make(b, sp)
    d, sh = po.sc_distance(sp, db, da, (0, 0), (0.2, 0.1), 100.0, 10.0)
    assert sh == 5 and d < 0.05
    d0, sh0 = po.sc_distance(sp, da, da, (0, 0), (0, 0), 100.0, 10.0)
    assert sh0 == 0 and abs(d0) < 1e-12
    # odometry term: the same descriptors far apart in odometry are penalised (:148-150)
    d_far, _ = po.sc_distance(sp, da, da, (0, 0), (30.0, 0), 100.0, 10.0)
    assert d_far > d0 + 0.5


def _database(n_db=60, revisit=(50, 5), seed=3):
    """a drive whose node `revisit[0]` comes back to the place of node `revisit[1]` (rotated)"""
    world = synth.make_world()
    traj = synth.make_trajectory(4100, n_db, step=0.8)
    traj[revisit[0]] = traj[revisit[1]] + np.array([0.15, -0.1, 2 * (2 * np.pi / 45)])
    scans = np.stack([synth.make_scan(world, traj[i], 30000 + i) for i in range(n_db)])
    dist = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(traj[:, :2], axis=0), axis=1))])
    return scans, traj[:, :2].copy(), dist


def test_detect_finds_the_revisit_and_respects_the_exclusion_window():
    sp = osp(max_radius=20.0, dist_thresh=0.5, num_exclude_recent=15)
    scans, pos, dist = _database()
    descs, rks = [], []
    for s in scans:
        d, rk, _ = po.sc_make(s, sp)
        descs.append(d)
        rks.append(rk)
    descs, rks = np.stack(descs), np.stack(rks)
    lid, yaw, md = po.sc_detect(sp, descs, rks, pos, dist, 50)
    assert lid == 5 and md < 0.5
    assert abs(abs(yaw) - 2 * (2 * np.pi / 45)) < 1.5 * (2 * np.pi / 45) or abs(abs(yaw) - (2 * np.pi - 2 * (2 * np.pi / 45))) < 1.5 * (2 * np.pi / 45)
    assert po.sc_detect(sp, descs, rks, pos, dist, 10)[0] == -1          # node_id < NUM_EXCLUDE_RECENT + 1
    lid2, _, md2 = po.sc_detect(sp, descs, rks, pos, dist, 30)            # no revisit: best distance above the threshold
    assert lid2 == -1 and md2 >= 0.5


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_descriptors_and_detection_match_oracle(built):
    import torch

    dev = torch.device("cuda:0")
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    sp_o = osp(max_radius=20.0, dist_thresh=0.5)
    sp = host.sc_params(max_radius=20.0, dist_thresh=0.5)
    scans, pos, dist = _database()
    n_db = scans.shape[0]
    # ragged batch: the second half carries fewer valid points
    n_pts = np.full(n_db, scans.shape[1], dtype=np.int32)
    n_pts[n_db // 2:] = 1500
    d_desc = torch.zeros((n_db, 45, 20), dtype=torch.float64, device=dev)
    d_rk = torch.zeros((n_db, 20), dtype=torch.float64, device=dev)
    d_sk = torch.zeros((n_db, 45), dtype=torch.float64, device=dev)
    host.sc_make_batch(ctx, torch.from_numpy(scans).to(dev), sp, d_desc, d_rk, d_sk, n_points=torch.from_numpy(n_pts).to(dev))
    ctx.synchronize()
    descs, rks, sks = [], [], []
    for s, n in zip(scans, n_pts):
        d, rk, sk = po.sc_make(s[:n], sp_o)
        descs.append(d)
        rks.append(rk)
        sks.append(sk)
    descs, rks, sks = np.stack(descs), np.stack(rks), np.stack(sks)
    # sequential per-bin sums in input order on both sides: identical bits
    assert np.array_equal(d_desc.cpu().numpy(), descs)
    assert np.array_equal(d_rk.cpu().numpy(), rks) and np.array_equal(d_sk.cpu().numpy(), sks)

    q = np.arange(n_db, dtype=np.int32)
    loop = torch.zeros(n_db, dtype=torch.int32, device=dev)
    yaw = torch.zeros(n_db, dtype=torch.float32, device=dev)
    md = torch.zeros(n_db, dtype=torch.float64, device=dev)
    host.sc_detect_batch(ctx, sp, d_desc, d_rk, torch.from_numpy(pos).to(dev), torch.from_numpy(dist).to(dev), torch.from_numpy(q).to(dev),
                         loop, yaw, md)
    ctx.synchronize()
    loop, yaw, md = loop.cpu().numpy(), yaw.cpu().numpy(), md.cpu().numpy()
    for i in range(n_db):
        lid, y, m = po.sc_detect(sp_o, descs, rks, pos, dist, i)
        assert loop[i] == lid, i
        assert yaw[i] == np.float32(y), i
        assert abs(md[i] - m) <= 1e-12 * max(1.0, abs(m)), i       # exp() of the odometry term may differ in the last bit
    assert loop[50] in (4, 5, 6) and (loop >= 0).sum() >= 1   # the revisit (truncated scans may prefer a neighbouring node)


@pytest.mark.gpu
def test_hip_scan_context_other_shapes(built):
    """outdoor-like shape (40 x 60), fewer candidates than the database, PCL point stride"""
    import torch

    dev = torch.device("cuda:0")
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    kw = dict(num_ring=40, num_sector=60, max_radius=25.0, num_exclude_recent=3, num_candidates=4, search_ratio=0.1, dist_thresh=0.7,
              intensity_factor=0.05, odom_weight=0.25)
    sp_o, sp = osp(**kw), host.sc_params(**kw)
    scans, pos, dist = _database(n_db=24, revisit=(20, 2))
    pcl = np.zeros((scans.shape[0], scans.shape[1], 8), dtype=F)
    pcl[:, :, :3] = scans[:, :, :3]
    pcl[:, :, 4] = scans[:, :, 3]
    n_db = pcl.shape[0]
    d_desc = torch.zeros((n_db, 60, 40), dtype=torch.float64, device=dev)
    d_rk = torch.zeros((n_db, 40), dtype=torch.float64, device=dev)
    d_sk = torch.zeros((n_db, 60), dtype=torch.float64, device=dev)
    host.sc_make_batch(ctx, torch.from_numpy(pcl).to(dev), sp, d_desc, d_rk, d_sk)
    ctx.synchronize()
    descs, rks = [], []
    for s in pcl:
        d, rk, _ = po.sc_make(s, sp_o)
        descs.append(d)
        rks.append(rk)
    descs, rks = np.stack(descs), np.stack(rks)
    assert np.array_equal(d_desc.cpu().numpy(), descs)
    q = np.array([20, 23, 2, 7], dtype=np.int32)
    loop = torch.zeros(4, dtype=torch.int32, device=dev)
    yaw = torch.zeros(4, dtype=torch.float32, device=dev)
    host.sc_detect_batch(ctx, sp, d_desc, d_rk, torch.from_numpy(pos).to(dev), torch.from_numpy(dist).to(dev), torch.from_numpy(q).to(dev),
                         loop, yaw)
    ctx.synchronize()
    for j, i in enumerate(q):
        lid, y, _ = po.sc_detect(sp_o, descs, rks, pos, dist, int(i))
        assert loop.cpu().numpy()[j] == lid and yaw.cpu().numpy()[j] == np.float32(y)


@pytest.mark.gpu
def test_hip_scan_context_database_object(built):
    """randt_sc_db_*: the SCManager call pattern (append keyframes one by one, query a node) grows its device
    arrays and answers like the batch entry points / the oracle."""
    import torch

    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    sp_o = osp(max_radius=20.0, dist_thresh=0.5)
    sp = host.sc_params(max_radius=20.0, dist_thresh=0.5)
    scans, pos, dist = _database(n_db=70, revisit=(66, 9))
    db = host.ScDatabase(ctx, sp, capacity=8)               # forces several re-allocations
    descs, rks = [], []
    for i, s in enumerate(scans):
        n = len(s) if i % 3 else 1200                        # ragged scans
        assert db.append(s[:n], pos[i], dist[i]) == i
        d, rk, _ = po.sc_make(s[:n], sp_o)
        descs.append(d)
        rks.append(rk)
    assert len(db) == 70
    descs, rks = np.stack(descs), np.stack(rks)
    for i in (0, 7, 33, 69):
        d, rk, sk = db.download(i)
        assert np.array_equal(d, descs[i]) and np.array_equal(rk, rks[i])
    hits = 0
    for i in (5, 16, 40, 66, 69):
        lid, yaw, md = db.detect(i)
        olid, oyaw, omd = po.sc_detect(sp_o, descs, rks, pos, dist, i)
        assert lid == olid and yaw == np.float32(oyaw) and abs(md - omd) <= 1e-12 * max(1.0, abs(omd))
        hits += lid >= 0
    assert hits >= 1
    db.close()
