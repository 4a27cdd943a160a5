"""Per (kernel, grid size) launch durations from a rocprofv3 --kernel-trace CSV: count, average, min, median in us.
Usage: trace_by_grid.py <dir with *_kernel_trace.csv> [name filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        rows[(short, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("kernel,grid_threads,launches,avg_us,min_us,median_us")
for (k, g), v in sorted(rows.items()):
    v = np.array(v)
    print("%s,%d,%d,%.2f,%.2f,%.2f" % (k, g, len(v), v.mean(), v.min(), np.median(v)))
