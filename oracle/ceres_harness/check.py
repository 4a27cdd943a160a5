#!/usr/bin/env python3
"""Build (if a Ceres Solver installation exists) and run the real-Ceres harness on one seeded registration, and compare
it with the oracle's restatement.  Prints ONE status line; never fails the caller: without Ceres it prints
"real-Ceres harness: skipped (...)".  Test infrastructure (oracle/), called by __graft_entry__.smoke() and runnable by hand:
    python oracle/ceres_harness/check.py
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def build():
    """Returns (path to the harness binary or None, reason)."""
    if shutil.which("cmake") is None:
        return None, "cmake not available"
    bdir = os.path.join(tempfile.gettempdir(), "randt_ceres_harness_build")
    try:
        cfg = subprocess.run(["cmake", "-S", HERE, "-B", bdir, "-DCMAKE_BUILD_TYPE=Release"], capture_output=True, text=True, timeout=300)
    except Exception as e:  # noqa: BLE001
        return None, "cmake configure failed: %s" % e
    if "Ceres not found" in cfg.stdout or cfg.returncode != 0:
        return None, "Ceres Solver is not installed in this image (find_package(Ceres) failed)"
    b = subprocess.run(["cmake", "--build", bdir, "-j", "4"], capture_output=True, text=True, timeout=1200)
    exe = os.path.join(bdir, "ceres_harness")
    if b.returncode != 0 or not os.path.exists(exe):
        return None, "harness did not compile against the installed Ceres: " + b.stderr[-300:]
    return exe, "built"


def main():
    exe, why = build()
    if exe is None:
        print("real-Ceres harness: skipped (%s) -- oracle parity stays UNPINNED by Ceres itself" % why)
        return 0
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle as po
    from randt_slam_amd import synth
    from util import oracle_scan_map, oracle_submap, problem

    prob = problem()
    sub = oracle_submap(prob["submaps"][0])
    worst = 0.0
    for i in range(4):
        for param, manifold in ((po.PARAM_AMBIENT4, 0), (po.PARAM_MANIFOLD, 1)):
            scan = oracle_scan_map(prob["scans"][i])
            g4 = synth.pose3_to_pose4(prob["guess"][i])
            prm = po.default_params(parameterization=param)
            corr, _ = po.associate(sub, scan, g4, prm.n_neighbours, 1, 1)
            rc, p4, st = po.solve_pair(sub, scan, corr, prm, g4)
            fc, mc = sub.cells(), scan.cells()
            pairs = [(mc[m], fc[j]) for m in range(len(mc)) for j in corr[m] if j >= 0]
            with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
                f.write("%.17g %.17g %.17g %.17g %.17g %d %d %d %d %d\n" % (prm.loss_scale, prm.mu_scale, prm.loss_alpha, prm.loss_weight, prm.gnc_divisor,
                                                                         prm.gnc_steps, prm.max_iterations, 1, manifold, len(pairs)))
                f.write("%.17g %.17g %.17g %.17g\n" % tuple(g4))
                for m, c in pairs:
                    f.write(" ".join("%.9g" % v for v in list(m["mean"]) + list(m["cov"]) + list(c["mean"]) + list(c["cov"])) + "\n")
                path = f.name
            out = subprocess.run([exe, path], capture_output=True, text=True, timeout=120).stdout
            os.unlink(path)
            pose = [float(v) for v in re.search(r"pose (.*)", out).group(1).split()]
            its = sum(int(v) for v in re.findall(r"iterations (\d+)", out))
            worst = max(worst, float(np.abs(np.array(pose) - p4).max()))
            if its != st["n_iterations"]:
                print("real-Ceres harness: iteration counts differ on pair %d (Ceres %d, oracle %d)" % (i, its, st["n_iterations"]))
    print("real-Ceres harness: ran on 8 registrations, max |pose(Ceres) - pose(oracle)| = %.3e" % worst)
    return 0


if __name__ == "__main__":
    sys.exit(main())
