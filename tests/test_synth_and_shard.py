"""Host logic on CPU: synthetic-scene generator, sharding helpers, and the N>1 path under gloo
(world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from randt_slam_amd import shard, synth


def test_scan_shape_and_determinism():
    w = synth.make_world()
    tr = synth.make_trajectory(3000, 4)
    a = synth.make_scan(w, tr[0], 1000)
    b = synth.make_scan(w, tr[0], 1000)
    assert a.shape == (2000, 4) and a.dtype == np.float32 and np.array_equal(a, b)
    r = np.hypot(a[:, 0], a[:, 1])
    assert r.min() > synth.MIN_RANGE - 0.1 and r.max() < synth.MAX_RANGE + 0.1
    az = np.unwrap(np.arctan2(a[2::5, 1], a[2::5, 0]))       # centre bin of each azimuth
    assert (np.diff(az) > 0).mean() > 0.7                   # azimuth ordered (up to position noise / re-drawn rays)
    assert not np.array_equal(a, synth.make_scan(w, tr[0], 1001))


def test_pose_helpers_roundtrip():
    p = np.array([1.0, -2.0, 0.7])
    q = synth.se2_mul3(p, synth.se2_inv3(p))
    assert np.allclose(q, 0, atol=1e-12)
    assert np.allclose(synth.pose4_to_pose3(synth.pose3_to_pose4(p)), p)


def test_shard_range_covers_everything():
    for n in (512, 513, 7, 0):
        for world in (1, 2, 4, 8):
            spans = [shard.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (a) submap tables: owner rank 0 -> everyone
    cells = torch.zeros(48 * 16, dtype=torch.uint8)
    counts = torch.zeros(2, dtype=torch.int32)
    grid = torch.full((200,), -1, dtype=torch.int32)
    if rank == 0:
        cells[:] = torch.arange(48 * 16) % 251
        counts[:] = torch.tensor([7, 9])
        grid[::3] = 5
    shard.broadcast_submap_tables((cells, counts, grid), src=0)
    ok = bool((cells == (torch.arange(48 * 16) % 251).to(torch.uint8)).all() and counts.tolist() == [7, 9] and int((grid == 5).sum()) == 67)
    # (b) contiguous shards, results gathered in rank order (uneven on purpose)
    lo, hi = shard.shard_range(13, world, rank)
    local = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1).repeat(1, 4)
    allp = shard.gather_results(local)
    ok = ok and allp.shape == (13, 4) and bool((allp[:, 0] == torch.arange(13, dtype=torch.float64)).all())
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
