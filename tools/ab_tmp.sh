H="--odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-roofline-sections"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]/1e6,3),"M/s", round(d["ms_per_step"]*1e3/d["config"]["batch_per_gpu"]*512,1),"us/512", d.get("pose_err_vs_oracle",{}).get("max_abs_translation_m"))'
python -c "import torch; print(torch.cuda.get_device_properties(0).name)"; python - <<'PY'
import ctypes
h=ctypes.CDLL('/opt/rocm/lib/libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); print('prio range', h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo),ctypes.byref(hi)), lo.value, hi.value)
PY
for rep in 1 2; do
python bench.py $H --steps 3000 2>/dev/null | python -c "$P" "base"
for pr in -1 0 1; do RANDT_SOLVE_STREAM_PRIO=$pr python bench.py $H --steps 3000 2>/dev/null | python -c "$P" "solve_stream_prio=$pr"; done
for pr in -1 0; do GPU_MAX_HW_QUEUES=32 RANDT_SOLVE_STREAM_PRIO=$pr python bench.py $H --steps 3000 2>/dev/null | python -c "$P" "q32 solve_stream_prio=$pr"; done
done
