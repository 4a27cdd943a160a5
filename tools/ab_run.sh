H="--odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline"
for v in base wpe3 wpe4; do
  if [ $v = base ]; then unset RANDT_LIB; else export RANDT_LIB=$PWD/build/ab/$v/librandt_hip.so; fi
  for only in "" "--only solve"; do
    python bench.py $H $only --steps 2000 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v','$only',round(d['value']/1e6,3),'M/s', round(d['ms_per_step']*1e3,1),'us/step', d['stage_ms'])"
  done
  python bench.py $H --streams 1 --steps 500 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v single-stream',round(d['value']/1e6,3),'M/s', d['stage_ms'])"
done
