"""-m gpu: bench.py contract -- one JSON line with the required keys; the multi-rank control flow
(barriers, submap broadcast, MAX over ranks, rank-0 print) exercised with two ranks on one GPU
through gloo (RCCL itself needs >= 2 GPUs, which the driver provides at round end)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--odometry-scans", "8", "--polar-scans", "2",
                        "--cpu-seconds", "0.5", "--min-seconds", "0.1", "--slam-scans", "0", "--polar-odometry-scans", "0",
                        "--replica-steps", "10", "--cpp-drive-scans", "24", "--distinct-inputs", "3"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    d = _last_json(r.stdout)
    for k in REQUIRED + ["cpu_baseline"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps_requested"] == 4 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["steps"] == 4 and d["repeats"] == len(d["region_ms"]) >= 1             # EXACTLY what was asked for
    assert d["sustained"]["timed_region_s"] >= 0.1 and d["sustained"]["value"] > 1e5  # --min-seconds 0.1 below: the settle region
    c2 = d["config2_single_pair"]
    assert c2["randt_register_pair"]["median_us"] > 10 and c2["randt_scan_register_batch_dev_B1"]["kernel_us_hip_events"] > 10
    assert c2["cpu_oracle"]["one_thread"]["cores"] == 1 and c2["cpu_oracle"]["residual_parallel"]["cores"] >= 1
    assert c2["cpu_oracle"]["pose_vs_gpu_max_abs"] <= 1e-6
    assert d["config3_streaming_odometry"]["cpu_oracle_residual_parallel"]["scans_per_sec"] > 0
    assert abs(d["ms_per_step"] * d["steps"] * 1e-3 - d["timed_region_s"]) < 1e-9
    assert d["value"] > 1e5 and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "valu_issue" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["traffic"] > 0
    assert 0.0 < rf["frac"] < 1.0 and 0.0 < rf["path"]["frac"] < 1.0
    # the line's frac follows from the committed counter summary: cycles per launch / measured launch duration / peak
    assert abs(rf["valu_issue_cycles_per_launch"] / rf["avg_launch_us"] * 1e-3 / rf["peak"] - rf["frac"]) < 1e-12
    assert d["single_batch"]["value"] > 1e5 and d["single_batch"]["batch_latency_us"] > 10
    lg = d["loop_gate_and_search"]                                              # f-2 / f-3 have measurements of their own
    assert lg["f2_cs_divergence"]["pairs_per_sec"] > 1e4 and lg["f2_cs_divergence"]["cpu_oracle"]["max_abs_difference_vs_gpu"] < 1e-9
    assert lg["f3_global_search"]["cost_batch"]["pose_evaluations_per_sec"] > 1e5 and lg["f3_global_search"]["cpu_oracle"]["same_result"] is True
    pf = d["config5_polar_filter"]                                              # f-1: the one HBM-streaming stage has its own roofline object
    assert pf["status_ok"] and pf["roofline"]["bound"] == "hbm" and 0.0 < pf["roofline"]["frac"] < 1.0 and pf["roofline"]["algorithmic_bytes"] == 2 * 19200000
    # round-4 verdict, item 3: what north_star asks for is IN the line -- HBM fraction from counters at top level (headline and
    # sustained step), the path's issue fraction on the headline's own step, the headline over inputs at distinct addresses,
    # config 3's GB/s, the replicas of config 3, the C++ drop-in drive, the polar filter at one scan per launch
    hb = d["hbm_achieved"]
    assert hb["bound"] == "hbm" and 0.0 < hb["frac"] < 0.5 and hb["bytes_per_step"] > 1e6 and 0.0 < hb["sustained"]["frac"] < 0.5
    assert abs(hb["achieved"] - hb["bytes_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * hb["achieved"]
    assert abs(rf["path"]["ms_per_step"] - d["ms_per_step"]) < 1e-9 and 0.0 < rf["path"]["sustained"]["frac"] < 1.0
    di = d["distinct_inputs"]
    assert di["copies"] == 3 and di["value"] > 1e5 and di["sustained"]["value"] > 1e5 and di["poses_equal_headline"] is True
    c3 = d["config3_streaming_odometry"]["hbm_achieved"]
    assert 0.0 < c3["frac"] < 0.01 and c3["bytes_per_scan"] > 1e5
    rp = d["config3_replicas"]
    assert rp["R64"]["scans_per_sec"] > 2e4 and rp["R256"]["scans_per_sec"] > rp["R64"]["scans_per_sec"] and 0.0 < rp["R256"]["window_kernel_issue_frac_of_chip"] < 1.0
    sa = d["solve_auto_detection"]                                               # verdict item 6: AUTO = the explicit mode's rate (4-step bursts are noisy: a wide band)
    assert sa["poses_equal_headline"] is True and 0.6 < sa["sustained"]["vs_explicit_throughput_mode"] < 1.6
    cd = d["cpp_local_fuser_drive"]
    assert cd["add_scan_pointxyzi"]["ms_per_scan"] > 0.05 and cd["insert_cluster_loop_pointxyzi"]["ms_per_scan"] > cd["add_clusters_pointxyzi"]["ms_per_scan"]
    assert cd["poses_equal_across_legs"]["packed_vs_pointxyzi_max_abs"] == 0.0 and cd["poses_equal_across_legs"]["add_scan_vs_add_clusters_max_abs"] == 0.0 and cd["poses_equal_across_legs"]["add_scan_vs_insert_cluster_loop_max_abs"] == 0.0
    ss1 = pf["roofline"]["single_scan"]
    assert ss1["status_ok"] and ss1["same_count_as_batched"] and 0.0 < ss1["frac"] < 1.0 and ss1["bytes"] == 19200000
    assert d["cpu_baseline"]["single_thread"]["cores"] == 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["pose_err_vs_oracle"]["max_abs_translation_m"] <= 1e-4 and d["pose_err_vs_oracle"]["max_abs_rotation_rad"] <= 1e-4


def test_bench_two_ranks_control_flow():
    env = dict(os.environ, RANDT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
           "--odometry-scans", "0", "--polar-scans", "0", "--streams", "2", "--min-seconds", "0.05"]   # two processes share ONE GPU here
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 1e5 and d["scaling"] == "weak"
    # config 4 as BASELINE states it: ONE 512 batch split over the ranks, results gathered, identical to the unsharded run
    ss = d["strong_scaling"]
    assert ss["registrations_per_gpu_per_step"] == 256 and ss["value"] > 1e5 and ss["poses_bit_identical_to_unsharded"] is True
    assert d["group_fallback"] is False and "gloo" in d["group_transport"]      # (a control-flow test, not a fallback)


def test_bench_group_path_on_one_rank():
    """The multi-GPU code of bench.py -- randt_group over RCCL created per stream (unique id, ncclCommInitRank, agreement),
    the strong region through randt_group_scan_register_batch_dev with the ONE-exchange gather (ncclAllGather of packed rows),
    the same region without the exchange, several groups in flight, and the loud fallback keys -- with a single rank on this
    box's one GPU (--force-group-path): everything except more than one peer."""
    r = subprocess.run([sys.executable, "bench.py", "--force-group-path", "--steps", "6", "--warmup", "1", "--repeats", "3", "--min-seconds", "0.05",
                        "--no-cpu-baseline", "--no-config2", "--no-roofline-sections", "--odometry-scans", "0", "--polar-scans", "0",
                        "--slam-scans", "0", "--polar-odometry-scans", "0", "--streams", "4"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MASTER_PORT="29535"))
    d = _last_json(r.stdout)
    assert d["group_fallback"] is False and "RCCL" in d["group_transport"] and "group_error" not in d
    ss = d["strong_scaling"]
    assert ss["poses_bit_identical_to_unsharded"] is True and ss["steps"] == 6 and "ncclAllGather" in ss["entry"]
    assert ss["kernel_us_per_step"] > 10 and abs(ss["gather_us_per_step"]) < ss["kernel_us_per_step"]
    assert ss["pipelined"]["groups_in_flight"] == 4 and ss["pipelined"]["value"] > 1e5
    assert "expected_ceiling" in ss


def test_bench_two_ranks_over_rccl_sharing_the_one_gpu():
    """bench.py --gpus 2 AS THE DRIVER LAUNCHES IT on an 8-GPU node -- one process per rank, torch.distributed "nccl" (= RCCL) for the
    launch plumbing, randt_group_create_rank + ncclBroadcast of the submap tables + the strong region's ncclAllGather per step --
    with both ranks on the one GPU of this box (shard.shared_gpu_rank_env: an NCCL_HOSTID per rank, socket transport).  The
    line must come from the RCCL group path (no fallback), the strong split must equal the unsharded batch bit for bit, and the
    line says that its ranks shared a GPU."""
    from randt_slam_amd import shard

    port = 29540
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "2", "--repeats", "3", "--min-seconds", "0.05",
           "--no-cpu-baseline", "--no-config2", "--no-roofline-sections", "--odometry-scans", "0", "--polar-scans", "0", "--slam-scans", "0",
           "--polar-odometry-scans", "0", "--streams", "4"]
    procs = [subprocess.Popen(cmd, cwd=ROOT, env=shard.shared_gpu_rank_env(r, 2, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))   # (~25 s normally)
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate())
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0][-1500:] + o[1][-1500:] for o in outs)
    d = _last_json(outs[0][0])
    assert d["n_gpus"] == 2 and d["ranks_share_one_gpu"] is True and d["scaling"] == "weak"
    assert d["group_fallback"] is False and "RCCL" in d["group_transport"] and "group_error" not in d
    ss = d["strong_scaling"]
    assert ss["poses_bit_identical_to_unsharded"] is True and "ncclAllGather" in ss["entry"]
    assert ss["pipelined"]["value"] > 1e4 and d["value"] > 1e5


def test_bench_two_ranks_rccl():
    """The same over RCCL (backend "nccl") when the box has two GPUs (the driver's multi-GPU node; skipped on a 1-GPU box)."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", "bench.py", "--gpus", "2", "--steps", "50", "--warmup", "2", "--min-seconds", "0.05"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["strong_scaling"]["poses_bit_identical_to_unsharded"] is True
    assert d["group_fallback"] is False and d["strong_scaling"]["pipelined"]["value"] > 1e5
