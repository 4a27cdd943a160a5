#!/bin/bash
# polar filter A/B (runs on the GPU box): tools/ab_polar.sh base wpe6 ...   (variants built with tools/ab_build.sh NAME FLAGS filter)
export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = base ]; then unset RANDT_LIB; else export RANDT_LIB=$PWD/build/ab/$v/librandt_hip.so; fi
  rm -rf /tmp/pp_$v
  rocprofv3 --kernel-trace -d /tmp/pp_$v -o run --output-format csv -- python tools/polar_filter_probe.py > /tmp/pp_$v.json 2>/dev/null
  echo "== $v"; python - <<PY
import json
d=json.load(open("/tmp/pp_$v.json"))["roofline"]
print("alone16 %.1f us  b2b %.1f us  single %.1f us chain %.1f us  64: %.1f us" % (307.2e6/d["achieved"]/1e3, d["back_to_back"]["ms"]*1e3, d["single_scan"]["us"], d["single_scan"]["chain_us_per_launch"], d["long_launch"]["ms"]*1e3))
PY
  python tools/trace_by_grid.py /tmp/pp_$v k_filter
done
