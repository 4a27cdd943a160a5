"""Time randt_pose_graph_optimize on Oxford-sized graphs (and the dense oracle on a small one)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import pyoracle as po  # noqa: E402
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import host  # noqa: E402
from test_posegraph import make_graph  # noqa: E402

ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(21)
for n, n_loops in ((400, 10), (2200, 60), (2200, 300), (8800, 300)):
    loops = [(int(a), int(a) + n // 2 + int(o)) for a, o in zip(rng.integers(0, n // 2 - 50, n_loops), rng.integers(-40, 40, n_loops))]
    truth, x0, ia, ib, meas, sq = make_graph(n, loops, seed=23, laps=2.0, radius=60.0)
    host.pose_graph_optimize(ctx, x0, ia, ib, meas, sq, n)
    t0 = time.perf_counter()
    x, r = host.pose_graph_optimize(ctx, x0, ia, ib, meas, sq, n)
    dt = time.perf_counter() - t0
    line = f"n={n} loops={n_loops} sep={r['n_separator_poses']} it={r['iterations']} term={r['termination']} gpu {dt * 1e3:.1f} ms ({dt * 1e3 / r['iterations']:.2f} ms/it)"
    if n <= 400:
        t0 = time.perf_counter()
        xo, ro = po.pose_graph_optimize(x0, ia, ib, meas, sq, n)
        line += f" | dense oracle {1e3 * (time.perf_counter() - t0):.1f} ms, max diff {np.abs(x - xo).max():.2e}"
    print(line, flush=True)
