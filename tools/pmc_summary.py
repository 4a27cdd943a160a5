#!/usr/bin/env python3
"""Summarise one tools/collect_profiles.sh run (gpurun_out/TAG/) into the small CSVs that get committed under profiles/:

  pmc_summary.py gpurun_out/TAG profiles/rNN_x
    -> profiles/rNN_x_sq_summary.csv         mean counter value per dispatch of every randt kernel (kernel, grid, workgroup),
                                             all counters of all --pmc passes side by side, plus the derived columns
                                             valu_issue_cycles (see VALU_COST) and fp64_flops, plus that kernel's average
                                             duration in the un-profiled single-stream kernel trace of the same run
    -> profiles/rNN_x_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the DEFAULT bench command
    -> profiles/rNN_x_single_stream_kernel_stats.csv   the same with one stream (clean, non-overlapped durations)
    -> profiles/rNN_x_durations_by_grid.csv   per (kernel, grid size) average duration of both traces

FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them; FETCH_SIZE is NOT doubled here (the gfx950 x2 correction of
MI355X_MICROARCH.md is applied where the number is used, bench.py and DESIGN.md, so that this file stays raw).
"""
import collections
import csv
import glob
import os
import re
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash  # noqa: E402

# SIMD-cycles one wave64 instruction occupies its SIMD's VALU issue port, from the chip's SPEC rates (MI355X_MICROARCH.md:
# SIMD-32, fp32 FMA 157.3 TFLOP/s = 2 cycles per wave instruction; vector fp64 78.6 TFLOP/s = half rate = 4 cycles;
# transcendentals quarter rate: 8 cycles fp32, 16 cycles fp64), confirmed by tools/clock_probe.hip
# (profiles/r03_clock_probe.csv: at 8 wavefronts per SIMD the event-timed rates reach 0.88-0.97 of these, at a shader clock
# of 2.1-2.4 GHz measured as d(s_memtime) / d(s_memrealtime)).  Round 2 priced fp64 at 3.33 "ticks" from
# tools/valu_rate_probe.hip: that probe divided ONE wavefront's tick count by the instructions of the four wavefronts it
# assumed co-resident for the whole loop, but a wavefront's loop lasts only ~75 % of the launch (dispatch ramp), so the
# figure was 4.0 x 0.83 -- an artefact, as was the "1.56-1.85 GHz under load" clock derived the same way.
# "other" = SQ_INSTS_VALU minus every counted class (moves, compares, selects, DPP, readlane): priced at the cheapest
# class, 2 (a LOWER bound of the issue cycles).
VALU_COST = {"F64": 4.0, "TRANS_F64": 16.0, "INT64": 4.0, "F32": 2.0, "TRANS_F32": 8.0, "INT32": 2.0, "CVT": 2.0, "OTHER": 2.0}


def valu_issue_cycles(c):
    """Counter dict of one dispatch -> (issue cycles, fp64 flops).  Needs the sq_a and sq_c counter sets."""
    g = lambda k: c.get(k, 0.0)
    f64 = g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64")
    t64 = g("SQ_INSTS_VALU_TRANS_F64")
    f32 = g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_ADD_F32")
    t32 = g("SQ_INSTS_VALU_TRANS_F32")
    i32, i64, cvt = g("SQ_INSTS_VALU_INT32"), g("SQ_INSTS_VALU_INT64"), g("SQ_INSTS_VALU_CVT")
    other = max(0.0, g("SQ_INSTS_VALU") - f64 - t64 - f32 - t32 - i32 - i64 - cvt)
    cyc = (f64 * VALU_COST["F64"] + t64 * VALU_COST["TRANS_F64"] + i64 * VALU_COST["INT64"] + f32 * VALU_COST["F32"] +
           t32 * VALU_COST["TRANS_F32"] + i32 * VALU_COST["INT32"] + cvt * VALU_COST["CVT"] + other * VALU_COST["OTHER"])
    flops = 64.0 * (2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") + t64)
    return cyc, flops


def kname(full):
    m = re.search(r"(k_[a-z_0-9]+(?:<[^>]*>)?)", full)
    return m.group(1).replace(" ", "") if m else None


def trace_durations(path):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        n = kname(r["Kernel_Name"])
        if n:
            # (the counter CSV's Grid_Size is the whole grid: multiply the trace's three dimensions out for 2-D launches)
            grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)) * max(1, int(r.get("Grid_Size_Y", 1))) * max(1, int(r.get("Grid_Size_Z", 1)))
            agg[(n, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg


def main():
    src, dst = sys.argv[1], sys.argv[2]
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    order = []
    for path in sorted(glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            n = kname(r["Kernel_Name"])
            if not n:
                continue
            key = (n, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            table[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] not in order:
                order.append(r["Counter_Name"])
            table[key]["__vgpr"] = [int(r["VGPR_Count"])]
            table[key]["__lds"] = [int(r["LDS_Block_Size"])]
    single = trace_durations(os.path.join(src, "single", "run_kernel_trace.csv"))
    default = trace_durations(os.path.join(src, "stats", "run_kernel_trace.csv"))
    with open(dst + "_sq_summary.csv", "w", newline="") as f:
        w = csv.writer(f)
        # the code these counters were taken from (bench.py refuses them for other code): recorded on the GPU box by
        # collect_profiles.sh at collection time
        hp = os.path.join(src, "csrc_hash.txt")
        stamp = open(hp).read().strip() if os.path.exists(hp) else csrc_hash()
        w.writerow(["kernel", "grid_size", "workgroup_size", "dispatches", "rocprof_vgpr_count", "lds_bytes"] + order +
                   ["valu_issue_cycles", "fp64_flops", "single_stream_avg_us", "single_stream_launches", "csrc_hash"])
        for key, d in sorted(table.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", [0]))):
            mean = {c: sum(v) / len(v) for c, v in d.items() if not c.startswith("__")}
            cyc, fl = valu_issue_cycles(mean)
            dur = single.get((key[0], key[1]), [])
            w.writerow([key[0], key[1], key[2], max(len(v) for c, v in d.items() if not c.startswith("__")), d["__vgpr"][0], d["__lds"][0]] +
                       ["%.1f" % mean[c] if c in mean else "" for c in order] +
                       ["%.0f" % cyc, "%.0f" % fl, "%.2f" % (sum(dur) / len(dur) / 1e3) if dur else "", len(dur), stamp])
    with open(dst + "_durations_by_grid.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["trace", "kernel", "grid_size", "launches", "avg_us", "min_us", "max_us"])
        for label, agg in (("default_16_streams", default), ("single_stream", single)):
            for (n, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([label, n, grid, len(v), "%.2f" % (sum(v) / len(v) / 1e3), "%.2f" % (min(v) / 1e3), "%.2f" % (max(v) / 1e3)])
    for a, b in (("stats/run_kernel_stats.csv", "_kernel_stats.csv"), ("single/run_kernel_stats.csv", "_single_stream_kernel_stats.csv"),
                 ("stats_bench.json", "_default_bench_profiled.json"), ("bench.json", "_default_bench.json"),
                 ("single_bench.json", "_single_stream_bench.json"), ("valu_rate_probe.csv", "_valu_rate_probe.csv"),
                 ("clock_probe.csv", "_clock_probe.csv")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), dst + b)
    print(open(dst + "_sq_summary.csv").read()[:6000])


if __name__ == "__main__":
    main()
