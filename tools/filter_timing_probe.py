"""Times the polar filter alone (k_filter_rows + k_filter_emit, 16 scans per launch = 307 MB) with HIP events on the launch
stream; RANDT_LIB selects an A/B build (tools/ab_build.sh).  Prints us per launch and the fraction of 8 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import randt_slam_amd as R
from randt_slam_amd import host, synth

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
world = synth.make_world()
tr = synth.make_trajectory(3400, 4)
base = [torch.from_numpy(synth.make_polar_scan(world, tr[i], 70 + i)).to(dev) for i in range(4)]
raw = torch.stack([base[i % 4] for i in range(n_scans)]).contiguous()
fill = os.environ.get("FILL")
if fill == "const":
    raw.fill_(0.0115)
elif fill == "rand":
    raw.copy_(torch.rand_like(raw) * 0.5 + 0.5)
elif fill == "fresh":          # the same values in a freshly allocated buffer (not the torch.stack result)
    raw = raw.clone()
out = torch.zeros((n_scans, 6144, 4), dtype=torch.float32, device=dev)
counts = torch.zeros(n_scans, dtype=torch.int32, device=dev)
status = torch.zeros(n_scans, dtype=torch.int32, device=dev)
fp = host.filter_params()
st = torch.cuda.current_stream()
for _ in range(3):
    host.filter_scan_batch(ctx, raw, fp, out, counts, status)
torch.cuda.synchronize()
reps, t = 30, []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    host.filter_scan_batch(ctx, raw, fp, out, counts, status)
    e1.record(st)
    torch.cuda.synchronize()
    t.append(e0.elapsed_time(e1) * 1e3)
t.sort()
nbuf = int(os.environ.get("NBUF", "1"))
raws = [raw] + [raw.clone() for _ in range(nbuf - 1)]
# the same launches back to back (no host synchronisation between them): one event pair around REPS launches
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for i in range(reps):
    host.filter_scan_batch(ctx, raws[i % nbuf], fp, out, counts, status)
e1.record(st)
torch.cuda.synchronize()
b2b = e0.elapsed_time(e1) * 1e3 / reps
nbytes = raw.numel() * 4
med = t[len(t) // 2]
print((fill or "real") + " nbuf=%d" % nbuf + " %s b2b_us %.2f us_median %.2f us_min %.2f TB/s %.3f frac %.3f points %.1f status_ok %s" % (os.environ.get("RANDT_LIB", "main"), b2b, med, t[0], nbytes / med * 1e-6,
      nbytes / med * 1e-6 / 8.0, float(counts.float().mean().item()), bool((status == 0).all().item())))
