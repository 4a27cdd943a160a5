# A/B of the window solve through config 3 (same box, same run): tools/ab_win.sh [variants under build/ab/ ... | base]
H="--polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-roofline-sections --steps 200"
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset RANDT_LIB; else export RANDT_LIB=$PWD/build/ab/$v/librandt_hip.so; fi
  python bench.py $H | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config3_streaming_odometry']; print('$v', round(c['scans_per_sec'],1), round(c['ms_per_scan']*1e3,1), c['end_pose_error_vs_truth_m'])"
done; done
