#!/usr/bin/env python3
"""Generates tests/golden/odometry_drive.npz: the poses of a 40-scan drive through the LocalFuser::processScan call
pattern (randt-slam_amd/odometry.py) computed by the CPU oracle backend -- a regression pin for the fixed-lag path
(window solve, keyframe merges, one submap roll-over).  The scans are regenerated from seeds (synth.py), only the
outputs are stored.  Run from the repo root:  python tests/golden/make_golden_odometry.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import odometry, synth  # noqa: E402
from oracle_backend import OracleBackend  # noqa: E402

N_SCANS, DT, TRAJ_SEED, SCAN_SEED0 = 40, 0.25, 3700, 12000
SMALL = dict(submap_size_poses=24, submap_overlap=8)


def drive_inputs():
    world = synth.make_world()
    traj = synth.make_trajectory(TRAJ_SEED, N_SCANS, step=0.25)
    return traj, [synth.make_scan(world, traj[i], SCAN_SEED0 + i) for i in range(N_SCANS)]


def main():
    traj, scans = drive_inputs()
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    odo = odometry.Odometry(OracleBackend(), mp, R.window_params(), SMALL)
    poses, iters = [], []
    for i in range(N_SCANS):
        poses.append(odo.process_scan(scans[i], i * DT).copy())
        iters.append(int(odo.last_result["n_iterations"]) if odo.last_result is not None else 0)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "odometry_drive.npz")
    np.savez_compressed(out, poses4=np.array(poses), lm_iterations=np.array(iters, dtype=np.int32),
                        submaps_finished=np.array(odo.n_finished_submaps), registrations=np.array(odo.n_registrations))
    print(out, os.path.getsize(out), "bytes; submaps finished", odo.n_finished_submaps)


if __name__ == "__main__":
    main()
