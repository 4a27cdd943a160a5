import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import randt_slam_amd as R
from randt_slam_amd import odometry, synth, host
world = synth.make_world(); n=400
th = 2*np.pi*np.arange(n)/160
truth = np.stack([5.0*np.cos(th), 5.0*np.sin(th), th+np.pi/2],1)
scans = np.stack([synth.make_scan(world, truth[i], 71000+i) for i in range(n)])
d = torch.from_numpy(scans).cuda()
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3); wp = R.window_params()
tc = [0.0]
orig = ctx._lib.randt_register_window
class W:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); tc[0] += time.perf_counter() - t0; return r
ctx._lib.randt_register_window = W()
tb = [0.0]
ob = host.ndt_build_batch
def nb(*a, **k):
    t0 = time.perf_counter(); r = ob(*a, **k); tb[0] += time.perf_counter() - t0; return r
host.ndt_build_batch = nb
for rep in range(2):
    odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params(), scan_slots=150, submap_slots=16), mp, wp)
    tc[0] = tb[0] = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): odo.process_scan(d[i], i*0.25)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("per scan: total %.1f us, inside C register_window %.1f us, build call %.1f us, other python %.1f us" % (el/n*1e6, tc[0]/n*1e6, tb[0]/n*1e6, (el - tc[0] - tb[0])/n*1e6))
