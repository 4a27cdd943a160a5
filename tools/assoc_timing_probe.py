"""Phase timing of k_associate (first chunk of workgroup 0).  Needs `make TIMING=1`."""
import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import randt_slam_amd as R
from randt_slam_amd import synth
lib = R._capi.load()
dev = torch.device("cuda:0")
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
prob = synth.make_batch_problem(2, 32, 34)
mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
sub = R.Maps(ctx, 2, mapp, 10000, with_grid=True)
for j, sm in enumerate(prob["submaps"]):
    kf = torch.from_numpy(np.stack(sm["kf_scans"])).to(dev)
    tmp = R.Maps(ctx, kf.shape[0], mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, kf, clu, tmp)
    sub.merge(j, tmp, 0, synth.pose3_to_pose4(sm["kf_rel"]))
B = prob["scans"].shape[0]
scans = R.Maps(ctx, B, mapp, 512, with_grid=False)
R.ndt_build_batch(ctx, torch.from_numpy(prob["scans"]).to(dev), clu, scans)
mp = R.default_matcher_params()
corr = torch.full((B, 512, mp.n_neighbours), -1, dtype=torch.int32, device=dev)
guess = torch.from_numpy(synth.pose3_to_pose4(prob["guess"])).to(dev)
fidx = torch.from_numpy(prob["submap_of"]).to(dev)
for _ in range(3):
    R.associate_batch(ctx, sub, fidx, scans, 0, B, guess, mp, corr)
ctx.synchronize()
o = (C.c_longlong * 16)()
lib.randt_debug_assoc_timing(o)
t = np.array(o[:6], dtype=np.float64)
print("cells", scans.counts()[:4])
print("chunk 0: P1 transform, P2 window, prefix, P3 distances, P4 top-k [us]:", np.diff(t) * 0.01)
