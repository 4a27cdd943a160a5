"""Phase timing of k_ndt_build on filtered polar scans (config 5 shape).  Needs `make TIMING=1`."""
import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import randt_slam_amd as R
from randt_slam_amd import synth, host
lib = R._capi.load()
dev = torch.device("cuda:0")
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
world = synth.make_world(); tr = synth.make_trajectory(3400, 4)
raw = torch.stack([torch.from_numpy(synth.make_polar_scan(world, tr[i % 4], 70 + i % 4)).to(dev) for i in range(16)]).contiguous()
pitch = 6144
out = torch.zeros((16, pitch, 4), dtype=torch.float32, device=dev)
counts = torch.zeros(16, dtype=torch.int32, device=dev); status = torch.zeros(16, dtype=torch.int32, device=dev)
fp = host.filter_params()
maps = R.Maps(ctx, 16, R.indoor_map_params(), 1024, with_grid=False)
for _ in range(3):
    host.filter_scan_batch(ctx, raw, fp, out, counts, status)
    R.ndt_build_batch(ctx, out, R.indoor_cluster_params(), maps, n_points=counts)
ctx.synchronize()
o = (C.c_longlong * 32)()
lib.randt_debug_timing(o)
t = np.array(o[:10], dtype=np.float64)
print("counts", counts.cpu().numpy()[:4], "cells", maps.counts()[:4])
print("phase us:", np.diff(t) * 0.01)
r = np.array([o[7], o[10], o[11], o[12], o[13], o[8]], dtype=np.float64)
print("rounds: pass1, pass2, finish(r0), round1, rest us:", np.diff(r) * 0.01)
