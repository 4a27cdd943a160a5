"""Pair-solve geometry probe (runs on the GPU box): for batches of 512 / 256 / 64 registrations, the solve kernel alone and
the whole build -> associate -> solve batch, one batch at a time on one stream, for every wavefronts-per-registration
setting (RANDT_SOLVE_SPLIT = 0 -> the one-wavefront kernel, 2..8 -> split mode, -1 -> the library's own choice); results
must be bit-identical across all of them."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    prob = synth.make_batch_problem(8, 64, 34)
    mapp, clu, mp = R.indoor_map_params(), R.indoor_cluster_params(), R.default_matcher_params()
    st = torch.cuda.current_stream()
    ctx0 = R.Context(0, st.cuda_stream)
    n_slots = mapp.size_x * mapp.size_y
    sub = R.Maps(ctx0, 8, mapp, n_slots, with_grid=True)
    for j, sm in enumerate(prob["submaps"]):
        kf = torch.from_numpy(np.stack(sm["kf_scans"])).to(dev)
        tmp = R.Maps(ctx0, kf.shape[0], mapp, 512, with_grid=False)
        R.ndt_build_batch(ctx0, kf, clu, tmp)
        sub.merge(j, tmp, 0, synth.pose3_to_pose4(sm["kf_rel"]))
    ctx0.synchronize()
    pts_all = torch.from_numpy(prob["scans"]).to(dev)
    fidx_all = torch.from_numpy(prob["submap_of"]).to(dev)
    g4_all = torch.from_numpy(synth.pose3_to_pose4(prob["guess"])).to(dev)
    widths = [int(w) for w in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,-1,2,4,5,6,8".split(","))]
    for B in (512, 256, 64):
        pts, fidx, g4 = pts_all[:B].contiguous(), fidx_all[:B].contiguous(), g4_all[:B].contiguous()
        ref = None
        for W in widths:
            if W >= 0:
                os.environ["RANDT_SOLVE_SPLIT"] = str(W)
            else:
                os.environ.pop("RANDT_SOLVE_SPLIT", None)
            ctx = R.Context(0, st.cuda_stream)
            subv = R.Maps(ctx, 8, mapp, n_slots, storage=sub.device_ptrs(), clear=False)
            ws = R.Maps(ctx, B, mapp, 512, with_grid=False)
            corr = torch.full((B, 512, mp.n_neighbours), -1, dtype=torch.int32, device=dev)
            res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
            R.ndt_build_batch(ctx, pts, clu, ws)
            R.associate_batch(ctx, subv, fidx, ws, 0, B, g4, mp, corr)
            reps = 40
            poses = [g4.clone() for _ in range(reps + 3)]
            for i in range(3):
                R.solve_batch(ctx, subv, fidx, ws, 0, B, corr, mp, poses[reps + i], res)
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            for i in range(reps):
                e[i].record(st)
                R.solve_batch(ctx, subv, fidx, ws, 0, B, corr, mp, poses[i], res)
            e[reps].record(st)
            torch.cuda.synchronize()
            solve_us = np.mean([e[i].elapsed_time(e[i + 1]) for i in range(reps)]) * 1e3
            # whole batch
            poses2 = [g4.clone() for _ in range(reps)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(reps):
                R.scan_register_batch(ctx, pts, clu, subv, fidx, ws, mp, poses2[i], res)
            e1.record(st)
            torch.cuda.synchronize()
            batch_us = e0.elapsed_time(e1) * 1e3 / reps
            out = (poses[0].cpu().numpy(), res.cpu().numpy())
            same = True if ref is None else (np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]))
            if ref is None:
                ref = out
            print("B %4d  W %2d  solve %7.1f us  batch %7.1f us  (%.2f M reg/s)  bit-identical %s" % (B, W, solve_us, batch_us, B / batch_us, same), flush=True)


if __name__ == "__main__":
    main()
