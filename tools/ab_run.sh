# quick A/B of the headline path: tools/ab_run.sh [variant names under build/ab/ ...]   ("base" = the in-tree library)
H="--odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-roofline-sections"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]/1e6,3),"M/s", round(d["ms_per_step"]*1e3/d["config"]["batch_per_gpu"]*512,1),"us/512", "host", round(d["host_enqueue_ms_per_step"]*1e3,1), {k:(round(v*1e3,1) if not isinstance(v,str) else "") for k,v in d["stage_ms"].items()})'
for v in "$@"; do
  if [ $v = base ]; then unset RANDT_LIB; else export RANDT_LIB=$PWD/build/ab/$v/librandt_hip.so; fi
  python bench.py $H --steps 2000 | python -c "$P" "$v full"
  for only in build associate solve; do
    python bench.py $H --only $only --batch-scale 8 --steps 300 | python -c "$P" "$v only-$only x8"
  done
  python bench.py $H --streams 1 --steps 500 | python -c "$P" "$v single-stream"
done
