#!/bin/bash
# Developer A/B hook: build a VARIANT of librandt_hip.so into build/ab/NAME/ (git-ignored, travels with gpurun) with extra
# compiler flags, to be selected at run time with RANDT_LIB=build/ab/NAME/librandt_hip.so.
#   tools/ab_build.sh NAME "-DSOME_KNOB=3" [unit ...]      (units: api ndt_build associate solve ...; default: all.  Units
#   that are not listed are taken from the main build's objects in randt-slam_amd/csrc/)
set -e
NAME=$1; EXTRA=$2; shift 2 || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/randt-slam_amd/csrc
OUT=$ROOT/build/ab/$NAME
mkdir -p "$OUT"
ALL="api ndt_build associate solve window window_gen window_gen_big filter csdiv scancontext posegraph cellops ndt_build_big group mapops"
UNITS=${*:-$ALL}
COMMON="$EXTRA -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$SRC -Wall -Wno-unused-function -Wno-pass-failed"
pids=()
for f in $ALL; do
  if [[ " $UNITS " == *" $f "* ]]; then
    exact=""
    case $f in ndt_build|associate|filter|csdiv|scancontext|cellops|ndt_build_big|mapops) exact="-ffp-contract=off";; esac
    /opt/rocm/bin/hipcc $COMMON $exact -c "$SRC/$f.hip" -o "$OUT/$f.o" &
    pids+=($!)
  else
    cp "$SRC/$f.o" "$OUT/$f.o"
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/librandt_hip.so" "$OUT"/*.o -ldl
echo "$OUT/librandt_hip.so"
