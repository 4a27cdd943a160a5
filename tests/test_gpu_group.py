"""-m gpu: the multi-GPU group of the C ABI (randt_group_*, csrc/group.hip) -- sharding, map broadcast, result gather.

On a one-GPU box the members are "virtual ranks" (several contexts + streams on device 0, peer-copy transport): the
control flow, the sharding and the exchanges are the real ones, and the results must be bit-identical to one context
running the whole batch.  RCCL itself is exercised with a one-rank communicator (dlopen, ncclCommInitRank,
ncclBroadcast in a group call) and, where the box has >= 2 GPUs, with a real two-device group."""
import numpy as np
import pytest

import randt_slam_amd as R
from randt_slam_amd import _capi, synth
from util import GpuRig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def base(built):
    import torch

    prob = synth.make_batch_problem(4, 33, 20)          # 132 registrations: uneven shards at G = 8
    rig = GpuRig(prob)
    rig.build_submaps()
    mp = R.default_matcher_params()
    g4 = synth.pose3_to_pose4(prob["guess"])
    pose = torch.from_numpy(g4.copy()).to(rig.dev)
    res = torch.zeros((rig.B, 64), dtype=torch.uint8, device=rig.dev)
    ws = R.Maps(rig.ctx, rig.B, rig.mapp, rig.scan_cap, with_grid=False)
    R.scan_register_batch(rig.ctx, rig.points, rig.clu, rig.submaps, rig.fixed_idx, ws, mp, pose, res)
    rig.ctx.synchronize()
    return prob, rig, mp, g4, pose.cpu().numpy(), res.cpu().numpy()


def _replicas(grp, rig, torch, g4, root_maps=None):
    """Per-member device state of a sharded batch: submap batch (empty unless this member is the root), full point / index /
    guess arrays, workspace scan maps, outputs."""
    n_slots = rig.mapp.size_x * rig.mapp.size_y
    devs = [torch.device("cuda", c.device) for c in grp.ctxs]
    subs = [R.Maps(c, rig.n_sub, rig.mapp, n_slots, with_grid=True) for c in grp.ctxs]
    pts = [rig.points.to(d).clone() for d in devs]
    fidx = [rig.fixed_idx.to(d).clone() for d in devs]
    pose = [torch.from_numpy(g4.copy()).to(d) for d in devs]
    res = [torch.zeros((rig.B, 64), dtype=torch.uint8, device=d) for d in devs]
    ws = [R.Maps(c, rig.B, rig.mapp, rig.scan_cap, with_grid=False) for c in grp.ctxs]
    return subs, pts, fidx, pose, res, ws


@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_virtual_rank_group_bit_identical_to_one_context(base, G):
    import torch

    prob, rig, mp, g4, ref_pose, ref_res = base
    grp = R.Group(devices=[0] * G)
    assert (grp.world, grp.n_local, grp.first_rank, grp.transport) == (G, G, 0, _capi.TRANSPORT_PEER)
    subs, pts, fidx, pose, res, ws = _replicas(grp, rig, torch, g4)
    # the root (member G - 1, to make it interesting) owns the submap epoch: it gets the tables, everyone else has empty maps
    root = G - 1
    subs[root].copy_from(rig.submaps)
    torch.cuda.synchronize()
    grp.broadcast_maps(subs, root=root)
    grp.synchronize()
    for m in subs:
        for j in range(rig.n_sub):
            c0, g0 = rig.submaps.download(j)
            c1, g1 = m.download(j)
            assert np.array_equal(c0.view(np.uint8), c1.view(np.uint8)) and np.array_equal(g0, g1)
    grp.scan_register_batch(pts, rig.clu, subs, fidx, ws, mp, pose, res, gather=True)
    grp.synchronize()
    for i in range(G):      # every member holds the whole batch's outputs
        assert np.array_equal(pose[i].cpu().numpy(), ref_pose), i
        assert np.array_equal(res[i].cpu().numpy(), ref_res), i
    # without the gather a member holds exactly its own rows
    pose2 = [torch.from_numpy(g4.copy()).to(rig.dev) for _ in range(G)]
    grp.scan_register_batch(pts, rig.clu, subs, fidx, ws, mp, pose2, res, gather=False)
    grp.synchronize()
    for i in range(G):
        lo, hi = R.shard_range(rig.B, G, i)
        got = pose2[i].cpu().numpy()
        assert np.array_equal(got[lo:hi], ref_pose[lo:hi])
        mask = np.ones(rig.B, bool)
        mask[lo:hi] = False
        assert np.array_equal(got[mask], g4[mask])
    grp.close()


def test_group_register_pairs_host_entry(base):
    """randt_group_register_pairs: host poses in / out (what LocalFuser::detectLoopClosures holds), pre-built moving maps."""
    import torch

    prob, rig, mp, g4, ref_pose, ref_res = base
    G = 3
    grp = R.Group(devices=[0] * G)
    subs, pts, fidx, pose, res, ws = _replicas(grp, rig, torch, g4)
    subs[0].copy_from(rig.submaps)
    # moving maps: built once on member 0, broadcast like any other map batch
    R.ndt_build_batch(grp.ctxs[0], pts[0], rig.clu, ws[0])
    grp.broadcast_maps(subs, root=0)
    grp.broadcast_maps(ws, root=0)
    p, r = grp.register_pairs(subs, prob["submap_of"], ws, mp, g4)
    assert np.array_equal(p, ref_pose)
    assert np.array_equal(r.view(np.uint8).reshape(rig.B, 64), ref_res)
    grp.close()


def test_rccl_transport_one_rank(base):
    """The RCCL code path on a single GPU: unique id, ncclCommInitRank (world 1), in-place broadcasts inside a group call."""
    import torch

    prob, rig, mp, g4, ref_pose, ref_res = base
    uid = R.group_unique_id()
    assert uid.shape == (128,) and uid.any()
    grp = R.Group(device=0, rank=0, world=1, unique_id=uid)
    assert grp.transport == _capi.TRANSPORT_RCCL and grp.n_local == 1
    subs, pts, fidx, pose, res, ws = _replicas(grp, rig, torch, g4)
    subs[0].copy_from(rig.submaps)
    torch.cuda.synchronize()
    grp.broadcast_maps(subs, root=0)
    grp.scan_register_batch(pts, rig.clu, subs, fidx, ws, mp, pose, res, gather=True)
    grp.synchronize()
    assert np.array_equal(pose[0].cpu().numpy(), ref_pose) and np.array_equal(res[0].cpu().numpy(), ref_res)
    grp.close()


def test_rccl_two_devices_one_process(base):
    """ONE process driving two GPUs over RCCL (ncclCommInitAll): the reference's single-process node.  Needs >= 2 GPUs."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    prob, rig, mp, g4, ref_pose, ref_res = base
    for transport in (_capi.TRANSPORT_RCCL, _capi.TRANSPORT_PEER):
        grp = R.Group(devices=[0, 1], transport=transport)
        subs, pts, fidx, pose, res, ws = _replicas(grp, rig, torch, g4)
        subs[0].copy_from(rig.submaps)
        torch.cuda.synchronize()
        grp.broadcast_maps(subs, root=0)
        grp.scan_register_batch(pts, rig.clu, subs, fidx, ws, mp, pose, res, gather=True)
        grp.synchronize()
        for i in range(2):
            assert np.array_equal(pose[i].cpu().numpy(), ref_pose) and np.array_equal(res[i].cpu().numpy(), ref_res)
        grp.close()


def test_group_argument_checks(base):
    prob, rig, mp, g4, _, _ = base
    with pytest.raises(R.RandtError) as e1:
        R.Group(devices=[0, 0], transport=_capi.TRANSPORT_RCCL)      # RCCL refuses two ranks on one device
    assert "distinct devices" in str(e1.value)                           # the text survives the (never created) group object
    with pytest.raises(R.RandtError) as e2:
        R.Group(devices=[99])
    assert "out of range" in str(e2.value)
    grp = R.Group(devices=[0, 0])
    foreign = R.Maps(rig.ctx, 1, rig.mapp, 16, with_grid=True)           # not created on the members' contexts
    with pytest.raises(R.RandtError):
        grp.broadcast_maps([foreign, foreign])
    grp.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_several_ranks_sharing_the_one_gpu(built, world):
    """RCCL with MORE THAN ONE RANK on a one-GPU box (round 6).  One process per rank, every rank on GPU 0, each with an
    NCCL_HOSTID of its own (randt_slam_amd.shard.shared_gpu_rank_env) so that RCCL sees one-GPU nodes instead of a duplicate
    device and connects them over its socket transport: ncclCommInitRank with world > 1, randt_group_broadcast_maps from
    rank 0 (the other ranks start with EMPTY tables), randt_group_scan_register_batch_dev with the ncclAllGather of packed
    rows -- every rank ends with the whole batch's poses and records, bit-identical to one context running it alone; without
    the gather a rank holds exactly its shard (uneven at 4 and 8 ranks: 34 registrations).  tools/rccl_two_ranks_probe.py is
    the per-rank program."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_two_ranks_probe.py"), str(world)], capture_output=True, text=True, timeout=200,
                       env=dict(os.environ, RCCL_PROBE_TIMEOUT="150"))   # (a run takes ~10 s; the driver caps the whole GPU suite at 1200 s)
    text = r.stdout + r.stderr
    ranks = [json.loads(ln[len("RCCL_PROBE "):]) for ln in r.stdout.splitlines() if ln.startswith("RCCL_PROBE ")]
    assert r.returncode == 0 and len(ranks) == world, text[-3000:]
    assert sorted(d["rank"] for d in ranks) == list(range(world))
    covered = []
    for d in ranks:
        assert d["group"] == dict(world=world, n_local=1, first_rank=d["rank"], transport=_capi.TRANSPORT_RCCL), d
        assert d["torch_all_reduce_ok"] and d["broadcast_tables_equal"], d
        assert d["gathered_poses_bit_identical"] and d["gathered_records_bit_identical"] and d["ungathered_rows_ok"], d
        covered += list(range(*d["shard"]))
    assert sorted(covered) == list(range(34))
