#!/usr/bin/env python3
"""Can RCCL run a communicator of MORE THAN ONE rank on a one-GPU box?

The builder's GPU leases have one MI355X, and RCCL refuses two ranks on one device ("Duplicate GPU detected": ranks with
the same host hash and bus id).  The host hash comes from the host name -- or from NCCL_HOSTID when that is set.  Two
processes that set DIFFERENT NCCL_HOSTIDs therefore look like two one-GPU nodes to RCCL; it connects them with its
built-in socket transport (over the loopback interface here) instead of xGMI / shared memory.  The bytes take another
road, everything else is the real thing: ncclGetUniqueId, ncclCommInitRank with world > 1, the bootstrap, ncclBroadcast /
ncclAllGather between two ranks, stream ordering against the kernels.

    python tools/rccl_two_ranks_probe.py [world]        (parent: spawns the ranks, prints one JSON line per rank)

Each rank: torch.distributed (backend "nccl" = RCCL) all-reduce, then the C ABI's multi-GPU group (randt_group_create_rank,
csrc/group.hip -- RCCL opened by dlopen), randt_group_broadcast_maps from rank 0, randt_group_scan_register_batch_dev with
the all-gather, compared bit for bit with one context running the whole batch."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def shared_gpu_env(rank, world, port):
    from randt_slam_amd import shard

    return shard.shared_gpu_rank_env(rank, world, port)


def child():
    import numpy as np
    import torch
    import torch.distributed as dist

    import randt_slam_amd as R
    from randt_slam_amd import synth

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = {"rank": rank, "world": world}
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    dist.init_process_group(backend="nccl", device_id=dev)
    x = torch.full((1024,), float(rank + 1), device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    out["torch_all_reduce_ok"] = bool((x == world * (world + 1) / 2).all().item())
    out["torch_init_s"] = round(time.time() - t0, 2)

    # the C ABI's group over the same library
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.from_numpy(R.group_unique_id()))
    dist.broadcast(uid, src=0)
    t0 = time.time()
    grp = R.Group(device=0, rank=rank, world=world, unique_id=uid.cpu().numpy())
    out["group_create_s"] = round(time.time() - t0, 2)
    out["group"] = dict(world=grp.world, n_local=grp.n_local, first_rank=grp.first_rank, transport=grp.transport)
    ctx = grp.ctxs[0]
    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    prob = synth.make_batch_problem(2, 17, 10)           # 34 registrations: uneven shards at 4 ranks
    n_sub, B = len(prob["submaps"]), len(prob["scans"])
    n_slots = mapp.size_x * mapp.size_y
    mp = R.default_matcher_params()
    g4 = synth.pose3_to_pose4(prob["guess"])
    pts = torch.from_numpy(prob["scans"]).to(dev)
    fidx = torch.from_numpy(prob["submap_of"]).to(dev)

    # reference: one context, the whole batch, submaps built locally
    ref_ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    ref_sub = R.Maps(ref_ctx, n_sub, mapp, n_slots, with_grid=True)
    for j, sm in enumerate(prob["submaps"]):
        kf = torch.from_numpy(np.stack(sm["kf_scans"])).to(dev)
        tmp = R.Maps(ref_ctx, kf.shape[0], mapp, 512, with_grid=False)
        R.ndt_build_batch(ref_ctx, kf, clu, tmp)
        ref_sub.merge(j, tmp, 0, synth.pose3_to_pose4(sm["kf_rel"]))
        tmp.close()
    ref_pose = torch.from_numpy(g4.copy()).to(dev)
    ref_res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    ref_ws = R.Maps(ref_ctx, B, mapp, 512, with_grid=False)
    R.scan_register_batch(ref_ctx, pts, clu, ref_sub, fidx, ref_ws, mp, ref_pose, ref_res)
    ref_ctx.synchronize()

    # the group: only rank 0 holds the submap tables before the broadcast
    subs = R.Maps(ctx, n_sub, mapp, n_slots, with_grid=True)
    if rank == 0:
        subs.copy_from(ref_sub)
    torch.cuda.synchronize()
    t0 = time.time()
    grp.broadcast_maps([subs], root=0)
    grp.synchronize()
    out["broadcast_s"] = round(time.time() - t0, 3)
    same = True
    for j in range(n_sub):
        c0, g0 = ref_sub.download(j)
        c1, g1 = subs.download(j)
        same = same and np.array_equal(c0.view(np.uint8), c1.view(np.uint8)) and np.array_equal(g0, g1)
    out["broadcast_tables_equal"] = bool(same)
    pose = torch.from_numpy(g4.copy()).to(dev)
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    ws = R.Maps(ctx, B, mapp, 512, with_grid=False)
    t0 = time.time()
    grp.scan_register_batch([pts], clu, [subs], [fidx], [ws], mp, [pose], [res], gather=True)
    grp.synchronize()
    out["register_gather_s"] = round(time.time() - t0, 3)
    out["gathered_poses_bit_identical"] = bool(np.array_equal(pose.cpu().numpy(), ref_pose.cpu().numpy()))
    out["gathered_records_bit_identical"] = bool(np.array_equal(res.cpu().numpy(), ref_res.cpu().numpy()))
    # without the gather: exactly the shard's rows
    pose2 = torch.from_numpy(g4.copy()).to(dev)
    grp.scan_register_batch([pts], clu, [subs], [fidx], [ws], mp, [pose2], [res], gather=False)
    grp.synchronize()
    lo, hi = R.shard_range(B, world, rank)
    got, refp = pose2.cpu().numpy(), ref_pose.cpu().numpy()
    mask = np.ones(B, bool)
    mask[lo:hi] = False
    out["shard"] = [int(lo), int(hi)]
    out["ungathered_rows_ok"] = bool(np.array_equal(got[lo:hi], refp[lo:hi]) and np.array_equal(got[mask], g4[mask]))
    dist.barrier()
    grp.close()
    dist.destroy_process_group()
    print("RCCL_PROBE " + json.dumps(out), flush=True)


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
    port = 29570 + (os.getpid() % 50)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"], env=shared_gpu_env(r, world, port), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    ok = True
    deadline = time.time() + float(os.environ.get("RCCL_PROBE_TIMEOUT", "240"))
    for r, p in enumerate(procs):
        try:
            text, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            text, _ = p.communicate()
            text += "\n[rank %d killed at the probe's time limit]" % r
        lines = [ln for ln in text.splitlines() if ln.startswith("RCCL_PROBE ")]
        if p.returncode != 0 or not lines:
            ok = False
            print("rank %d failed (rc %s):\n%s" % (r, p.returncode, text[-3000:]))
        else:
            print(lines[-1])
            d = json.loads(lines[-1][len("RCCL_PROBE "):])
            ok = ok and all(d[k] for k in ("torch_all_reduce_ok", "broadcast_tables_equal", "gathered_poses_bit_identical",
                                           "gathered_records_bit_identical", "ungathered_rows_ok")) and d["group"]["world"] == world
    print("RCCL_PROBE_SUMMARY " + json.dumps({"world": world, "ok": ok}))
    return 0 if ok else 1


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        sys.exit(main())
