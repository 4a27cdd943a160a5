"""Multi-GPU sharding of independent registrations.

Pair registrations share nothing but the read-only submap tables, so a batch splits contiguously
across ranks with no data-path collective.  The only exchanges are (a) a broadcast of the submap
cell tables + index grids from the owner rank, once per submap epoch, and (b) an all-gather of the
small per-registration results.  The product path for both is the C ABI's randt_group_* (csrc/group.hip: RCCL opened
by the library itself, or peer copies inside one process; host.Group binds it).  The torch.distributed helpers below
remain for launch-time plumbing (shipping the RCCL unique id) and for the gloo control-flow tests on boxes with fewer
GPUs than ranks.
"""
import os

import torch
import torch.distributed as dist


def shared_gpu_rank_env(rank, world, port, base=None):
    """TEST / PROBE PLUMBING: the environment of rank `rank` of a `world`-rank job whose ranks ALL sit on GPU 0.

    RCCL refuses two ranks on one device ("Duplicate GPU detected": equal host hash and bus id).  The host hash is taken from
    NCCL_HOSTID when it is set, so ranks that set DIFFERENT ids look like one-GPU nodes of their own: RCCL connects them with
    its built-in socket transport (loopback) instead of xGMI / shared memory.  The bytes take another road; everything else
    is the real thing -- ncclGetUniqueId, ncclCommInitRank with world > 1, the bootstrap, ncclBroadcast / ncclAllGather
    between ranks, stream ordering against the kernels.  This is how the multi-rank RCCL path of csrc/group.hip and of
    bench.py --gpus N is exercised on the builder's one-GPU leases (tools/rccl_two_ranks_probe.py, tests/test_gpu_group.py,
    tests/test_gpu_bench.py); never used for a reported scaling number (bench.py marks such a line `ranks_share_one_gpu`)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               NCCL_HOSTID="randt-fake-node-%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket",
               NCCL_DEBUG=env.get("NCCL_DEBUG", "WARN"), HSA_ENABLE_IPC_MODE_LEGACY="0", RANDT_RANKS_SHARE_GPU="1")
    return env


def shard_range(n_items, world_size, rank):
    """Contiguous split: rank r gets [r*B/G, (r+1)*B/G) (uneven remainders go to the low ranks) -- randt_shard_range of the
    C ABI, the split randt_group_* uses."""
    from .host import shard_range as _abi

    return _abi(n_items, world_size, rank)


def broadcast_submap_tables(tables, src=0, group=None):
    """tables: iterable of tensors backing a randt_maps batch (cells as uint8, counts / grid as int32).
    520 KB per indoor submap; all maps of a batch travel in one call per tensor."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in tables:
        dist.broadcast(t, src=src, group=group)


def gather_results(local, group=None):
    """All-gather per-registration outputs (poses as float64 (n,4), results as uint8 (n,64));
    ranks may hold different counts.  Returns the concatenation in rank order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    pad = max(counts)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return torch.cat([o[:c] for o, c in zip(outs, counts)], dim=0)
