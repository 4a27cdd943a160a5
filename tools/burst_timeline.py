#!/usr/bin/env python3
"""Timeline of the LAST K-step burst of a rocprofv3 --kernel-trace of bench.py (run on the GPU box):
    rocprofv3 --kernel-trace -d DIR -o run --output-format csv -- python bench.py --steps 20 ...
    python tools/burst_timeline.py DIR [K]
prints, for the last 3 K hot-path launches (build / associate / solve of K steps), start and end relative to the first start,
per queue, and the busy-time union per kernel type -- where a burst's time goes (fill, lock-step phases, drain)."""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        kind = "build" if "k_ndt_build" in n else ("assoc" if "k_associate" in n else ("solve" if n.startswith("void randt_solve::k_solve<") or "k_solve<" in n and "window" not in n else None))
        if "k_solve_order" in n:
            kind = None
        if kind is None:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?"), int(r.get("Grid_Size", 0) or 0)))
    rows.sort()
    # the headline's bursts are runs of 3 K launches of grid sizes of a 512 batch; take the last such run that is dense in time
    last = rows[-3 * K:]
    t0 = min(r[0] for r in last)
    print("file", f, "launches", len(last), "span_us %.1f" % ((max(r[1] for r in last) - t0) / 1e3))
    for kind in ("build", "assoc", "solve"):
        iv = sorted((r[0], r[1]) for r in last if r[2] == kind)
        busy, cur_s, cur_e = 0, None, None
        for s, e in iv:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += (cur_e - cur_s) if cur_e is not None else 0
        print("%-6s n %2d first_start %.1f last_end %.1f union_busy_us %.1f mean_dur_us %.1f" % (
            kind, len(iv), (iv[0][0] - t0) / 1e3, (max(e for _, e in iv) - t0) / 1e3, busy / 1e3, sum(e - s for s, e in iv) / len(iv) / 1e3))
    qs = sorted(set(r[3] for r in last))
    for r in last:
        print("q%-3s %-6s %8.1f -> %8.1f  (%6.1f us) grid %d" % (qs.index(r[3]), r[2], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[4]))


if __name__ == "__main__":
    main()
