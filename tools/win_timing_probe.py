import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import randt_slam_amd as R
from randt_slam_amd import synth, odometry
lib = R._capi.load()
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
world = synth.make_world(); dt = 0.25
traj = synth.make_trajectory(3300, 40, step=0.25)
scans = np.stack([synth.make_scan(world, traj[i], 20000 + i) for i in range(40)])
d = torch.from_numpy(scans).cuda()
mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3); wp = R.window_params()
odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp)
names = ["setup+raw", "ndt_pass(+factors)", "factors_weight", "assemble", "scale/grad", "LM rest+barrier", "decide", "tail", "LM:buildA", "LM:eliminate", "LM:mcc", "LM:plus", "LM:ambient"]
for i in range(40):
    odo.process_scan(d[i], i * dt)
    if i in (20, 38, 39):
        out = (C.c_longlong * 16)()
        lib.randt_debug_win_timing(out)
        t = np.array(out[:13]) * 0.01
        print("   cumulative over all scans so far (take deltas): factor wavefront %.1f us (of it exp..log chain %.1f us), NDT wavefront 0 %.1f us" % (out[13] * 0.01, out[15] * 0.01, out[14] * 0.01))
        print("scan", i, "iters", int(odo.last_result["iterations"]), "total %.1f us" % t.sum(), {n: round(v, 1) for n, v in zip(names, t)})
