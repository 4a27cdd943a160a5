"""cProfile of the config-5 loop (host side) -- where the wall time of a polar scan goes."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import randt_slam_amd as R
from randt_slam_amd import host, odometry, synth

ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
world = synth.make_world()
n, dt = 60, 0.25
traj = synth.make_trajectory(3300, n, step=0.25)
dev = torch.device("cuda:0")
raw = [torch.from_numpy(synth.make_polar_scan(world, traj[i], 61000 + i)).to(dev) for i in range(n)]
mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
wp, fp = R.window_params(), host.filter_params()
for rep in range(4):
    odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if rep == 3:
        pr.enable()
    for i in range(n):
        odo.process_scan(raw[i], i * dt, polar_filter=fp)
    torch.cuda.synchronize()
    if rep == 3:
        pr.disable()
    print("ms/scan", (time.perf_counter() - t0) / n * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
