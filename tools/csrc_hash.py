#!/usr/bin/env python3
"""Fingerprint of the device code the hot-path counters belong to: SHA-256 over the hot kernels' sources with comments and
blank lines removed (a documentation edit does not stale the counters, a code edit does).  tools/pmc_summary.py stamps
every row of profiles/*_sq_summary.csv with it; bench.py recomputes it and refuses counters taken from other code."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (randt_internal.h is left out on purpose: it also carries the launcher declarations of every other translation unit, which
# change without touching a hot kernel; MapView / SolveParams / randt_ctx live there and change rarely -- re-collect when they do)
HOT_SOURCES = ("solve.hip", "solve_pass.h", "solve_algebra.h", "solve_math.h", "associate.hip", "ndt_build.hip", "cell_math.h")


def _strip(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for line in text.splitlines():
        line = re.sub(r"//.*$", "", line).strip()
        if line:
            out.append(re.sub(r"\s+", " ", line))
    return "\n".join(out)


def csrc_hash(root=ROOT):
    h = hashlib.sha256()
    for name in HOT_SOURCES:
        with open(os.path.join(root, "randt-slam_amd", "csrc", name)) as f:
            h.update(name.encode())
            h.update(_strip(f.read()).encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
