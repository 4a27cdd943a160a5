import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import host
from test_posegraph import make_graph
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
os.environ["RANDT_PG_SEGMENT"] = "100000"
for n, loops in ((100, [(10, 50), (11, 51), (12, 52)]), (100, [(10, 50), (11, 51), (12, 52), (13, 53), (14, 54), (15, 55)]), (100, [(10, 50), (30, 70)]), (100, [(10, 50), (30, 70), (31, 71)]), (300, [(4, 290), (150, 20)]), (300, [(4, 290), (150, 20), (60, 200)]), (300, [(4, 290), (150, 20), (60, 200), (100, 250)]),
                 (300, [(10 + 12 * k, 160 + 11 * k) for k in range(10)]), (100, [(10, 50)]), (100, [(10, 50), (30, 70), (20, 90)])):
    truth, x0, ia, ib, meas, sq = make_graph(n, loops, seed=12, laps=1.0, radius=25.0)
    xo, ro = po.pose_graph_optimize(x0, ia, ib, meas, sq, n, po.pg_params(max_iterations=1))
    xg, rg = host.pose_graph_optimize(ctx, x0, ia, ib, meas, sq, n, host.pg_params(max_iterations=1))
    print(n, len(loops), rg["n_separator_poses"], "max diff after one step", np.abs(xg - xo).max(), flush=True)
