#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one counter per pass) into one small table:
mean counter value per kernel dispatch shape.  Usage: pmc_summary.py OUT.csv COUNTER=path.csv ..."""
import collections
import csv
import re
import sys

out = sys.argv[1]
table = collections.defaultdict(dict)
for arg in sys.argv[2:]:
    counter, path = arg.split("=", 1)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z_0-9]+(?:<[^>]*>)?)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"].split("(")[0][-48:]
        agg[(name, int(r["Grid_Size"]), int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        table[k][counter] = (len(v), sum(v) / len(v))
counters = [a.split("=", 1)[0] for a in sys.argv[2:]]
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "workgroup_size", "dispatches"] + [c + "_mean_KB" for c in counters])
    for k, d in sorted(table.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
        n = max(x[0] for x in d.values())
        w.writerow([k[0], k[1], k[2], n] + ["%.1f" % d[c][1] if c in d else "" for c in counters])
print(open(out).read())
