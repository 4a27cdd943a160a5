#!/usr/bin/env python3
"""bench.py --gpus N AS THE DRIVER LAUNCHES IT (one process per rank, torch.distributed "nccl" + the C ABI's RCCL group), with all
N ranks on the ONE GPU of the box (shard.shared_gpu_rank_env: an NCCL_HOSTID per rank, RCCL's socket transport).  Exercises BASELINE
config 4's sharded path -- weak region, the 512-registration batch split over N ranks with one ncclAllGather per step, the pipelined
strong region -- and checks the line; the rates are N processes time-sharing one GPU, NOT scaling figures.
    python tools/rccl_bench_shared_gpu.py N [out.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from randt_slam_amd import shard  # noqa: E402


def main():
    n = int(sys.argv[1])
    cmd = [sys.executable, "bench.py", "--gpus", str(n), "--steps", "40", "--warmup", "4", "--repeats", "3", "--min-seconds", "0.05", "--streams", "4",
           "--no-cpu-baseline", "--no-config2", "--no-roofline-sections", "--odometry-scans", "0", "--polar-scans", "0", "--slam-scans", "0",
           "--polar-odometry-scans", "0", "--cpp-drive-scans", "0", "--replica-steps", "0", "--distinct-inputs", "0", "--no-auto-region"]
    port = 29600 + n
    env0 = dict(os.environ, RANDT_BENCH_STRONG_GROUPS="2")
    ps = [subprocess.Popen(cmd, cwd=ROOT, env=shard.shared_gpu_rank_env(r, n, port, base=env0), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(n)]
    outs = []
    for p in ps:
        try:
            outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate())
    rcs = [p.returncode for p in ps]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    if any(rcs) or len(lines) != 1:
        print("FAILED rcs", rcs)
        print(outs[0][1][-3000:])
        return 1
    d = json.loads(lines[0])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(lines[0] + "\n")
    ss = d["strong_scaling"]
    summary = {"n_gpus": d["n_gpus"], "ranks_share_one_gpu": d.get("ranks_share_one_gpu"), "group_fallback": d["group_fallback"], "group_transport": d["group_transport"],
               "weak_value": d["value"], "strong_registrations_per_rank": ss["registrations_per_gpu_per_step"], "strong_bit_identical": ss["poses_bit_identical_to_unsharded"],
               "strong_ms_per_step": ss["ms_per_step"], "kernel_us_per_step": ss.get("kernel_us_per_step"), "gather_us_per_step": ss.get("gather_us_per_step"),
               "pipelined_groups": ss.get("pipelined", {}).get("groups_in_flight"), "submap_broadcast_ms": ss["submap_broadcast_ms"]}
    print("RCCL_BENCH " + json.dumps(summary))
    ok = d["n_gpus"] == n and d.get("ranks_share_one_gpu") is True and d["group_fallback"] is False and ss["poses_bit_identical_to_unsharded"] is True
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
