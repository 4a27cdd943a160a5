#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the evidence behind bench.py's `roofline` line.
#   tools/collect_profiles.sh TAG [extra bench args]
# writes under gpurun_out/TAG/:
#   bench.json                 the default bench line (un-profiled)
#   stats/                     rocprofv3 --kernel-trace --stats of the same default command
#   single/                    the same, one stream (clean per-kernel durations)
#   pmc_<set>/                 one rocprofv3 --pmc pass per counter set (kernel-trace only; never with sys/hip traces)
# Summaries to commit are produced afterwards (on either side) by tools/pmc_summary.py.
set -u
TAG=${1:-r02}
shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# (40 scans of the fixed-lag odometry stay in: k_solve_window gets counter rows; the polar filter of config 5 stays in: its two kernels are the one HBM-streaming stage and get their FETCH_SIZE / WRITE_SIZE rows too)
HEAD_ONLY="--odometry-scans 40 --polar-scans 16 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-config2 --cpp-drive-scans 0 --replica-steps 0 --distinct-inputs 0 --no-auto-region"

python tools/csrc_hash.py > "$OUT/csrc_hash.txt"   # fingerprint of the kernels these counters belong to
# the cost table's evidence: issue rates and the shader clock, three time bases (tools/clock_probe.hip)
hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe > "$OUT/clock_probe.csv" 2>&1

python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"; echo

rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python bench.py "$@" > "$OUT/stats_bench.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --stats -d "$OUT/single" -o run --output-format csv -- python bench.py --streams 1 --solve-mode throughput --steps 300 --min-seconds 0 $HEAD_ONLY "$@" > "$OUT/single_bench.json" 2> "$OUT/single.err"

# SQ: 8 slots per pass; TCC: FETCH_SIZE and WRITE_SIZE cannot share a pass (MI355X_MICROARCH.md, rocprofv3 PMC slots)
declare -A SETS
SETS[sq_a]="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
SETS[sq_b]="SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
SETS[sq_c]="SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CYCLES"
SETS[fetch]="FETCH_SIZE GRBM_GUI_ACTIVE"
SETS[write]="WRITE_SIZE"
for s in sq_a sq_b sq_c fetch write; do
  rocprofv3 --kernel-trace --pmc ${SETS[$s]} -d "$OUT/pmc_$s" -o run --output-format csv -- \
    python bench.py --streams 1 --solve-mode throughput --steps 24 --warmup 2 --min-seconds 0 $HEAD_ONLY "$@" > "$OUT/pmc_$s.json" 2> "$OUT/pmc_$s.err"
  ls "$OUT/pmc_$s" | head -3
done
find "$OUT" -name '*agent_info.csv' -delete
du -sh "$OUT"
# summarise ON THE BOX (the raw traces of the default command exceed what gpurun copies back) and keep only the summaries,
# the bench lines and the error logs
mkdir -p "$OUT/summary"
python tools/pmc_summary.py "$OUT" "$OUT/summary/$TAG" > "$OUT/summary/pmc_summary.log" 2>&1
rm -rf "$OUT"/pmc_*/ "$OUT/stats" "$OUT/single"
du -sh "$OUT"
