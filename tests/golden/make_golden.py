#!/usr/bin/env python3
"""Generates tests/golden/pair_indoor.npz.

The reference has no golden vectors and cannot be built or run here (Eigen / Ceres / Sophus / PCL /
ROS are absent), so these vectors come from the CPU oracle (oracle/randt_oracle.c) at the commit that
introduced them: they freeze the oracle's behaviour (regression pin) and give the GPU tests a
fixture that does not need the oracle binary.  Inputs: world seed 1234, trajectory seed 3000, scan
seeds 1000/1001, guess seeds 2000/2001 (randt-slam_amd/synth.py).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import pyoracle as po  # noqa: E402
from randt_slam_amd import synth  # noqa: E402
from util import oracle_scan_map, oracle_submap  # noqa: E402


def main():
    prob = synth.make_batch_problem(n_submaps=1, scans_per_submap=2, n_keyframes=8)
    sub = oracle_submap(prob["submaps"][0])
    out = dict(
        scans=prob["scans"], guess3=prob["guess"], truth3=prob["truth"],
        kf_scans=np.stack(prob["submaps"][0]["kf_scans"]), kf_rel3=prob["submaps"][0]["kf_rel"],
        submap_cells=sub.cells().view(np.uint8).reshape(-1, 48), submap_grid_slots=np.nonzero(sub.grid() >= 0)[0].astype(np.int32),
        submap_grid_vals=sub.grid()[sub.grid() >= 0].astype(np.int32),
    )
    for i in range(2):
        scan = oracle_scan_map(prob["scans"][i])
        out[f"scan{i}_cells"] = scan.cells().view(np.uint8).reshape(-1, 48)
        g4 = synth.pose3_to_pose4(prob["guess"][i])
        corr, _ = po.associate(sub, scan, g4, 4, 1, 1)
        out[f"scan{i}_corr"] = corr
        for name, param in (("ambient4", po.PARAM_AMBIENT4), ("manifold", po.PARAM_MANIFOLD)):
            rc, p4, cost, st = po.register_pair(sub, scan, po.default_params(parameterization=param), g4)
            out[f"scan{i}_{name}_pose4"] = p4
            out[f"scan{i}_{name}_cost"] = np.array([cost])
            out[f"scan{i}_{name}_trace_cost"] = st["trace_cost"]
            out[f"scan{i}_{name}_trace_radius"] = st["trace_radius"]
            out[f"scan{i}_{name}_trace_flag"] = st["trace_flag"]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pair_indoor.npz"), **out)
    print("wrote pair_indoor.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
