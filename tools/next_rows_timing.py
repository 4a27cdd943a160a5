"""Wall / HIP-event times of the f-4 device stages that have no line of their own in bench.py: Scan Context descriptors and
detection, pose-graph optimisation, map transform + merge.  Usage (GPU box): python tools/next_rows_timing.py"""
import os
import sys
import time

import numpy as np
import torch

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [_root, os.path.join(_root, "tests"), os.path.join(_root, "oracle")]   # (tests/test_posegraph.py: the graph generator; it imports the oracle module)
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import host, synth  # noqa: E402

dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
ctx = R.Context(0, st.cuda_stream)


def ev_time(fn, reps=20, warm=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = []
    for i in range(reps + warm):
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        if i >= warm:
            t.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(t))


w = synth.make_world()
tr = synth.make_trajectory(3000, 64)
scans = np.stack([synth.make_scan(w, tr[i], 100 + i) for i in range(64)])
for n_scans in (1, 64, 512):
    pts = torch.from_numpy(np.tile(scans, (max(1, n_scans // 64), 1, 1))[:n_scans]).to(dev)
    sp = host.sc_params(max_radius=20.0, dist_thresh=0.5)
    desc = torch.zeros((n_scans, 45, 20), dtype=torch.float64, device=dev)
    rk = torch.zeros((n_scans, 20), dtype=torch.float64, device=dev)
    sk = torch.zeros((n_scans, 45), dtype=torch.float64, device=dev)
    print("sc_make_batch   %4d scans x %d points: %8.1f us" % (n_scans, pts.shape[1], ev_time(lambda: host.sc_make_batch(ctx, pts, sp, desc, rk, sk))))
# detection: database of 2048 nodes
n_db = 2048
desc = torch.rand((n_db, 45, 20), dtype=torch.float64, device=dev)
rk = desc.mean(dim=1).contiguous()
pos = torch.from_numpy(np.cumsum(np.random.default_rng(0).normal(0, 0.3, (n_db, 2)), 0)).to(dev)
dist = torch.from_numpy(np.arange(n_db) * 0.3).to(dev)
for nq in (1, 64):
    q = torch.arange(n_db - nq, n_db, dtype=torch.int32, device=dev)
    loop = torch.zeros(nq, dtype=torch.int32, device=dev)
    yaw = torch.zeros(nq, dtype=torch.float32, device=dev)
    md = torch.zeros(nq, dtype=torch.float64, device=dev)
    print("sc_detect_batch %4d queries, %d-node database: %8.1f us" % (nq, n_db, ev_time(lambda: host.sc_detect_batch(ctx, sp, desc, rk, pos, dist, q, loop, yaw, md))))
# pose graph
from test_posegraph import make_graph, compose  # noqa: E402
for n, nl in ((300, 10), (2200, 60)):
    rng = np.random.default_rng(21)
    loops = [(int(a), int(a) + n // 2 + int(o)) for a, o in zip(rng.integers(0, n // 2 - 50, nl), rng.integers(-20, 20, nl))]
    truth, x0, ia, ib, meas, sq = make_graph(n, loops, seed=23, laps=2.0, radius=60.0)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        x1, r1 = host.pose_graph_optimize(ctx, x0, ia, ib, meas, sq, n)
        t.append(time.perf_counter() - t0)
    print("pose_graph_optimize %5d poses, %3d loop closures: %8.2f ms per call, %d iterations, %d separators" % (n, nl, np.median(t) * 1e3, r1["iterations"], r1["n_separator_poses"]))
# transform + merge of 64 keyframe maps into one submap
mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
tmp = R.Maps(ctx, 64, mapp, 512, with_grid=False)
R.ndt_build_batch(ctx, torch.from_numpy(scans).to(dev), clu, tmp)
origin_inv = synth.se2_inv3(tr[0])
rel = np.array([synth.se2_mul3(origin_inv, p) for p in tr])
sub = R.Maps(ctx, 1, mapp, 10000, with_grid=True)


def merge_all():
    sub.clear()
    sub.merge(0, tmp, 0, synth.pose3_to_pose4(rel))


t = []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    merge_all()
    ctx.synchronize()
    t.append(time.perf_counter() - t0)
print("transformMap + mergeMapCell of 64 scan maps into a submap: %8.1f us per call (%d cells)" % (np.median(t[2:]) * 1e6, int(sub.counts()[0])))
