import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
import randt_slam_amd as R
from randt_slam_amd import synth
lib = R._capi.load()
dev = torch.device("cuda:0")
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
w = synth.make_world(); tr = synth.make_trajectory(3000, 2)
pts = np.stack([synth.make_scan(w, tr[0], 1000+i) for i in range(8)])
pts = torch.from_numpy(np.tile(pts, (64,1,1))).to(dev)
maps = R.Maps(ctx, pts.shape[0], R.indoor_map_params(), 512, with_grid=False)
for _ in range(3):
    R.ndt_build_batch(ctx, pts, R.indoor_cluster_params(), maps)
ctx.synchronize()
out = (C.c_longlong*32)()
lib.randt_debug_timing(out)
t = np.array(out[:10], dtype=np.float64)
print("phase us:", np.diff(t)*0.01)
r = np.array([out[7], out[10], out[11], out[12], out[13], out[8]], dtype=np.float64)
print("rounds: pass1, pass2, finish(r0), round1, rest us:", np.diff(r)*0.01)
