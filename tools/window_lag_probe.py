"""Wall time of one randt_register_window call (association + window solve + the two staging copies) by lag and kernel, on
the simulated drive of tests/test_gpu_window.py.  Usage (GPU box): python tools/window_lag_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import synth  # noqa: E402


def make_drive(env=None, n_scans=8):
    """Device side of the drive fixture in tests/test_gpu_window.py: a submap of 8 keyframes, a sparser second one, 8 scans."""
    import torch

    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)   # the environment knobs are read at creation
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    world = synth.make_world()
    dt = 0.25
    traj = synth.make_trajectory(3100, n_scans + 34, step=0.25)
    origin_inv = synth.se2_inv3(traj[0])
    rel = np.array([synth.se2_mul3(origin_inv, p) for p in traj])
    rel[:, 2] = synth.wrap_angle(rel[:, 2])
    kf = [synth.make_scan(world, traj[t], 7000 + t) for t in range(0, 32, 4)]
    kf_rel = rel[0:32:4]
    scans = [synth.make_scan(world, traj[32 + i], 8000 + i) for i in range(n_scans)]
    dev = torch.device("cuda:0")
    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    sub = R.Maps(ctx, 2, mapp, 10000, with_grid=True)
    tmp = R.Maps(ctx, len(kf), mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(np.stack(kf)).to(dev), clu, tmp)
    sub.merge(0, tmp, 0, synth.pose3_to_pose4(kf_rel))
    for i in range(0, len(kf), 2):
        sub.merge(1, tmp, i, synth.pose3_to_pose4(kf_rel[i:i + 1]))
    smaps = R.Maps(ctx, n_scans, mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(np.stack(scans)).to(dev), clu, smaps)
    ctx.synchronize()
    return dict(ctx=ctx, sub=sub, smaps=smaps, truth=rel[32:32 + n_scans], dt=dt, keep=(tmp,))


def run(drive, lag, n_fixed, use_imu, const_vel, reps=5):
    ctx = drive["ctx"]
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    wp = R.window_params(use_imu=use_imu, const_vel=const_vel)
    truth, dt = drive["truth"], drive["dt"]
    s0 = R.make_state(synth.pose3_to_pose4(truth[0]), lin_vel=(0.8, 0.0), rot_vel=0.0, stamp=0.0)
    gs = [s0]
    trans = synth.pose3_to_pose4(truth[0])
    out = []
    for i in range(1, len(truth)):
        gs.append(R.predict_state(gs[-1], i * dt))
        S = min(len(gs) - 1, lag)
        win = list(range(i - S + 1, i + 1))
        imu = np.full(S, 0.002) if use_imu else None
        st_in = np.array(gs[-S - 1:], dtype=R.STATE_DTYPE)
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            g_states, tr, rej, res = R.register_window(ctx, drive["sub"], list(range(n_fixed)), drive["smaps"], win, st_in, mp, wp, trans, imu)
            best = min(best, time.perf_counter() - t0)
        trans = tr
        for j in range(S + 1):
            gs[len(gs) - S - 1 + j] = g_states[j]
        out.append((S, best * 1e6, int(res["iterations"]), int(res["n_evals"]), int(res["n_residuals"])))
    return out


if __name__ == "__main__":
    tuned, general, long_drive = make_drive(), make_drive({"RANDT_WINDOW_GENERAL": "1"}), make_drive(n_scans=14)
    for name, d, lag, kw in [("tuned   lag 3", tuned, 3, dict(n_fixed=1, use_imu=0, const_vel=1)),
                             ("general lag 3", general, 3, dict(n_fixed=1, use_imu=0, const_vel=1)),
                             ("tuned   lag 3 2 fixed imu acc", tuned, 3, dict(n_fixed=2, use_imu=1, const_vel=0)),
                             ("general lag 3 2 fixed imu acc", general, 3, dict(n_fixed=2, use_imu=1, const_vel=0)),
                             ("general lag 5", tuned, 5, dict(n_fixed=1, use_imu=0, const_vel=1)),
                             ("general lag 7", tuned, 7, dict(n_fixed=1, use_imu=0, const_vel=1)),
                             ("general lag 7 2 fixed imu acc", tuned, 7, dict(n_fixed=2, use_imu=1, const_vel=0)),
                             ("big     lag 8", long_drive, 8, dict(n_fixed=1, use_imu=0, const_vel=1)),      # window_gen_big.hip
                             ("big     lag 12", long_drive, 12, dict(n_fixed=1, use_imu=0, const_vel=1)),
                             ("big     lag 12 2 fixed imu acc", long_drive, 12, dict(n_fixed=2, use_imu=1, const_vel=0))]:
        rows = run(d, lag, **kw)
        S, us, it, ev, nr = rows[-1]
        print(f"{name:32s} S={S} {us:8.1f} us/window  iterations={it} passes={ev} residuals={nr}  us/iteration={us / max(it, 1):.2f}")
