# quick A/B of throughput AND the single-batch latencies: tools/ab_latency.sh [variant names under build/ab/ ...]   ("main" = the in-tree library)
for v in "$@"; do
  lib=""; [ "$v" != main ] && lib=build/ab/$v/librandt_hip.so
  RANDT_LIB=$lib python bench.py --odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_$v.json
  python - "$v" <<PY
import json,sys
d=json.load(open("gpurun_out/ab_%s.json"%sys.argv[1]))
sb=d.get("single_batch") or {}
rf=d.get("roofline") or {}
print("%-6s %.3f M/s  %.2f us/step | single512 %.1f us (solve %.1f) | b64 %.1f us (solve %.1f) | sat launch %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"]*1e3, sb.get("batch_latency_us",0), sb.get("kernel_us",{}).get("k_solve",0), sb.get("batch_of_64",{}).get("batch_latency_us",0), sb.get("batch_of_64",{}).get("kernel_us",{}).get("k_solve",0), rf.get("launch_us")))
PY
done
