mkdir -p gpurun_out/rccl
for w in 2 4 8; do timeout 300 python tools/rccl_two_ranks_probe.py $w > gpurun_out/rccl/probe_world$w.log 2>&1; tail -1 gpurun_out/rccl/probe_world$w.log; done
python - <<'P'
import os, subprocess, sys, json
sys.path.insert(0, os.getcwd())
from randt_slam_amd import shard
cmd=[sys.executable,"bench.py","--gpus","2","--steps","200","--warmup","5","--no-cpu-baseline","--no-config2","--no-roofline-sections","--odometry-scans","0","--polar-scans","0","--slam-scans","0","--polar-odometry-scans","0","--cpp-drive-scans","0","--replica-steps","0","--distinct-inputs","0","--no-auto-region"]
ps=[subprocess.Popen(cmd, env=shard.shared_gpu_rank_env(r,2,29577), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
outs=[p.communicate(timeout=600) for p in ps]
open("gpurun_out/rccl/bench_gpus2_shared_gpu.json","w").write(outs[0][0])
open("gpurun_out/rccl/bench_gpus2_shared_gpu.err","w").write(outs[0][1][-4000:]+"\n----rank1\n"+outs[1][1][-4000:])
d=json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
print({k:d.get(k) for k in ("n_gpus","value","ranks_share_one_gpu","group_fallback","group_transport")})
ss=d["strong_scaling"]; print({k:ss.get(k) for k in ("value","ms_per_step","kernel_us_per_step","gather_us_per_step","poses_bit_identical_to_unsharded","submap_broadcast_ms")}, ss["pipelined"]["value"])
P
