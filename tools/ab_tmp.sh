H="--odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-roofline-sections"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]/1e6,3),"M/s", round(d["ms_per_step"]*1e3/d["config"]["batch_per_gpu"]*512,1),"us/512", {k:(round(v*1e3,1) if not isinstance(v,str) else "") for k,v in d["stage_ms"].items()})'
for rep in 1 2; do
for v in base nofence; do
  if [ $v = base ]; then unset RANDT_LIB; else export RANDT_LIB=$PWD/build/ab/$v/librandt_hip.so; fi
  python bench.py $H --steps 3000 2>/dev/null | python -c "$P" "$v"
  python bench.py $H --streams 1 --steps 500 2>/dev/null | python -c "$P" "$v single"
  python bench.py $H --only build --batch-scale 8 --steps 300 2>/dev/null| python -c "$P" "$v only-build x8"
done; done
