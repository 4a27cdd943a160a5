#!/bin/bash
# Developer A/B hook: SQ instruction counters of k_solve for library variants built by tools/ab_build.sh.
#   tools/ab_pmc.sh VARIANT...   ("main" = the in-tree build)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
for v in "$@"; do
  lib=""; [ "$v" != main ] && lib=$PWD/build/ab/$v/librandt_hip.so
  out=gpurun_out/abpmc_$v
  rm -rf "$out"; mkdir -p "$out"
  RANDT_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES \
    -d "$out" -o run --output-format csv -- python bench.py --streams 1 --solve-mode throughput --steps 24 --warmup 2 --min-seconds 0 \
    --odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-roofline-sections > "$out/bench.json" 2> "$out/err.txt"
  python - "$v" "$out" <<'PY'
import csv, glob, sys, collections
v, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "k_solve" not in n: continue
        acc[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(v, n[-40:], {k: round(sum(x) / len(x) / 1e6, 3) for k, x in c.items()}, "dispatches", len(next(iter(c.values()))))
PY
done
