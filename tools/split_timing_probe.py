"""Where wavefront 0 spends a split-mode solve (developer probe; needs a library built with -DRANDT_SPLIT_TIMING:
tools/ab_build.sh stim "-DRANDT_SPLIT_TIMING" solve; RANDT_LIB=build/ab/stim/librandt_hip.so python tools/split_timing_probe.py)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import synth, _capi  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    prob = synth.make_batch_problem(8, 64, 34)
    mapp, clu, mp = R.indoor_map_params(), R.indoor_cluster_params(), R.default_matcher_params()
    st = torch.cuda.current_stream()
    ctx = R.Context(0, st.cuda_stream)
    n_slots = mapp.size_x * mapp.size_y
    sub = R.Maps(ctx, 8, mapp, n_slots, with_grid=True)
    for j, sm in enumerate(prob["submaps"]):
        kf = torch.from_numpy(np.stack(sm["kf_scans"])).to(dev)
        tmp = R.Maps(ctx, kf.shape[0], mapp, 512, with_grid=False)
        R.ndt_build_batch(ctx, kf, clu, tmp)
        sub.merge(j, tmp, 0, synth.pose3_to_pose4(sm["kf_rel"]))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    pts = torch.from_numpy(prob["scans"][:B]).to(dev)
    fidx = torch.from_numpy(prob["submap_of"][:B]).to(dev)
    g4 = torch.from_numpy(synth.pose3_to_pose4(prob["guess"])[:B]).to(dev)
    ws = R.Maps(ctx, B, mapp, 512, with_grid=False)
    corr = torch.full((B, 512, mp.n_neighbours), -1, dtype=torch.int32, device=dev)
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    R.ndt_build_batch(ctx, pts, clu, ws)
    R.associate_batch(ctx, sub, fidx, ws, 0, B, g4, mp, corr)
    for _ in range(3):
        p = g4.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        R.solve_batch(ctx, sub, fidx, ws, 0, B, corr, mp, p, res)
        e1.record(st)
        torch.cuda.synchronize()
    out = (C.c_longlong * 10)()
    lib = _capi.load()
    lib.randt_debug_split_timing.restype = C.c_int
    assert lib.randt_debug_split_timing(out) == 0
    t = np.array(list(out)[:5], dtype=float)
    n = out[8]
    r = res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)[0]
    names = ["barrier A (request published)", "own trip 0", "barrier B (helpers done)", "combine + reduction", "solver algebra between passes"]
    print(f"solve launch {e0.elapsed_time(e1) * 1e3:.1f} us; registration 0: {n} passes, {int(r['n_residuals'])} residuals, {int(r['iterations'])} iterations; clock64 ticks:")
    for nm, v in zip(names, t):
        print(f"  {nm:34s} {v:10.0f} ticks  {v / max(n, 1):8.0f} per pass  {100 * v / t.sum():5.1f} %")
    print(f"  total {t.sum():.0f} ticks = {t.sum() / (e0.elapsed_time(e1) * 1e3):.0f} ticks per us of the launch")


if __name__ == "__main__":
    main()
