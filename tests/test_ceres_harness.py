"""The optional real-Ceres harness (oracle/ceres_harness, SURVEY 8(c) last row) skips cleanly when no Ceres Solver is
installed -- one status line, exit code 0 -- and, where one is, reports the gap between Ceres and the oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_harness_reports_one_status_line(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ceres_harness", "check.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("real-Ceres harness: ")
    if "skipped" not in last:
        gap = float(last.rsplit("=", 1)[1])
        assert gap < 1e-4      # north_star tolerance: the restatement and true Ceres land on the same pose
