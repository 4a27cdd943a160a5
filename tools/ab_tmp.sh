H="--odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-roofline-sections"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]/1e6,3),"M/s", round(d["ms_per_step"]*1e3,1),"us/step")'
timeout 600 python -m pytest tests -m gpu -x -q -k "parity or config4 or independent" 2>&1 | tail -1
for rep in 1 2; do
for v in base s_pro3 s_alg1 s_alg2pro3 lat2 lat1; do
  if [ $v = base ]; then unset RANDT_LIB; else export RANDT_LIB=$PWD/build/ab/$v/librandt_hip.so; fi
  python bench.py $H --steps 3000 2>/dev/null | python -c "$P" "$v"
done; done
unset RANDT_LIB
for ch in 32 48; do RANDT_ASSOC_TP_CH=$ch python bench.py $H --steps 3000 2>/dev/null | python -c "$P" "base tp_ch=$ch"; done
for only in build associate solve; do python bench.py $H --only $only --batch-scale 8 --steps 300 | python -c "$P" "base only-$only x8"; done
