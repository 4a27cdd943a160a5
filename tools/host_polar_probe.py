"""What the PCIe hand-over costs when the boundary is given HOST buffers (config 5's 19.2 MB raw polar scans): randt_filter_build
(upload + filter + clustering + NDT) per scan from pageable and from pinned host memory, beside the device-resident chain."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import host, synth  # noqa: E402

ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
world = synth.make_world()
tr = synth.make_trajectory(3400, 4)
raws = [synth.make_polar_scan(world, tr[i], 70 + i) for i in range(4)]
fp, clu = host.filter_params(), R.indoor_cluster_params()
maps = R.Maps(ctx, 1, R.indoor_map_params(), 1024, with_grid=True)
nbytes = raws[0].nbytes


def run(bufs, reps=40):
    for b in bufs:
        host.filter_build(ctx, b, fp, clu, maps, 0, wait=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        host.filter_build(ctx, bufs[i % len(bufs)], fp, clu, maps, 0, wait=False)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps


t_page = run(raws)
pinned = [torch.from_numpy(r).pin_memory() for r in raws]
t_pin = run([p.numpy() for p in pinned])
dev = [torch.from_numpy(r).cuda() for r in raws]
out = torch.zeros((1, 6144, 4), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
st = torch.zeros(1, dtype=torch.int32, device="cuda")
for d in dev:
    host.filter_scan_batch(ctx, d[None], fp, out, cnt, st)
ctx.synchronize()
t0 = time.perf_counter()
for i in range(40):
    host.filter_scan_batch(ctx, dev[i % 4][None], fp, out, cnt, st)
    host.ndt_build_batch(ctx, out, clu, maps, n_points=cnt)
ctx.synchronize()
t_dev = (time.perf_counter() - t0) / 40
print("raw scan %.1f MB: filter + build per scan: device-resident %.1f us | pinned host buffer %.1f us (%.1f GB/s incl. the upload) | pageable host buffer %.1f us (%.1f GB/s)"
      % (nbytes / 1e6, t_dev * 1e6, t_pin * 1e6, nbytes / t_pin / 1e9, t_page * 1e6, nbytes / t_page / 1e9))
