#!/usr/bin/env python3
"""What bounds a K-step BURST of the headline path (the driver's `--steps 20`): the host's enqueue rate or the GPU?

The K steps (512 registrations each: build -> associate -> solve on stream s % 16) are enqueued (a) as bench.py does, timed from
the first enqueue to the last kernel's end, and (b) behind a GATE -- every stream first waits for an event behind a ~3 ms sleep
kernel, so that all 3 K launches are queued before the GPU may start any: the GPU-limited duration of the same burst (HIP
events from the gate's release to the last kernel's end).  (b) - is what an infinitely fast host would get.
    python tools/burst_gate_probe.py [K ...]"""
import os
import sys
import time

# bench.py's set-up is the default stream + 15 created ones on 16 hardware queues.  PROBE_PRIO creates all sixteen (priorities are a
# creation-time property) and the default stream keeps a queue of its own: 17 queues, or two streams share one (-27 % at K = 256)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "17" if os.environ.get("PROBE_PRIO") else "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import synth  # noqa: E402


def main():
    Ks = [int(a) for a in sys.argv[1:]] or [20]
    dev = torch.device("cuda", 0)
    n_streams = int(os.environ.get("PROBE_STREAMS", "16"))
    # PROBE_PRIO=first4 / last4 / alt: HIP stream priorities (-1 = high) for some of the streams -- does the queue scheduler's
    # preference shorten the burst's drain?
    prio = os.environ.get("PROBE_PRIO", "")
    def pr(j):
        if prio == "first4":
            return -1 if j < 4 else 0
        if prio == "last4":
            return -1 if j >= n_streams - 4 else 0
        if prio == "alt":
            return -1 if j % 2 == 0 else 0
        return 0
    if prio:
        streams = [torch.cuda.Stream(device=dev, priority=pr(j)) for j in range(n_streams)]
        torch.cuda.set_stream(streams[0])
    else:   # bench.py's own set-up
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    ctxs = [R.Context(0, st.cuda_stream) for st in streams]
    for c in ctxs:
        c.set_solve_mode(R._capi.SOLVE_THROUGHPUT)
    mapp, clu, mp = R.indoor_map_params(), R.indoor_cluster_params(), R.default_matcher_params()
    base = synth.make_batch_problem(bench.N_SUBMAPS, bench.SCANS_PER_SUBMAP, bench.N_KEYFRAMES)
    n_slots = mapp.size_x * mapp.size_y
    submaps = R.Maps(ctxs[0], bench.N_SUBMAPS, mapp, n_slots, with_grid=True)
    for j, sm in enumerate(base["submaps"]):
        kf = torch.from_numpy(np.stack(sm["kf_scans"])).to(dev)
        tmp = R.Maps(ctxs[0], kf.shape[0], mapp, 512, with_grid=False)
        R.ndt_build_batch(ctxs[0], kf, clu, tmp)
        submaps.merge(j, tmp, 0, synth.pose3_to_pose4(sm["kf_rel"]))
        tmp.close()
    ctxs[0].synchronize()
    views = [submaps] + [R.Maps(ctxs[i], bench.N_SUBMAPS, mapp, n_slots, storage=submaps.device_ptrs(), clear=False) for i in range(1, n_streams)]
    pts = torch.from_numpy(base["scans"]).to(dev)
    fidx = torch.from_numpy(base["submap_of"]).to(dev)
    g4 = torch.from_numpy(synth.pose3_to_pose4(base["guess"])).to(dev)
    batch = bench.Batch(R, torch, ctxs, views, mapp, clu, mp, pts, fidx, g4)
    bench.warm_up(torch, batch, streams, 4 * n_streams)

    def burst(K, gated):
        poses = [g4.clone() for _ in range(K)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        gate = torch.cuda.Stream(device=dev)
        if gated:
            with torch.cuda.stream(gate):
                torch.cuda._sleep(int(2.0e9 * 0.004))
        e0.record(gate)
        for st in streams:
            st.wait_event(e0)
        t0 = time.perf_counter()
        tail = int(os.environ.get("PROBE_TAIL_LATENCY", "0"))   # the last `tail` steps of the burst declare the latency placement
        for s in range(K):
            if tail and s >= K - tail:
                ctxs[s % n_streams].set_solve_mode(R._capi.SOLVE_LATENCY)
            batch.step(s % n_streams, streams[s % n_streams], poses[s])
        t_enq = time.perf_counter() - t0
        if tail:
            for c in ctxs:
                c.set_solve_mode(R._capi.SOLVE_THROUGHPUT)
        for st in streams[1:]:
            streams[0].wait_stream(st)
        e1.record(streams[0])
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        return e0.elapsed_time(e1) * 1e3, t_enq * 1e6, wall * 1e6, poses[0]

    ref = None
    for K in Ks:
        rows = {}
        for gated in (False, True):
            runs = [burst(K, gated) for _ in range(9)]
            runs.sort(key=lambda r: r[0])
            m = runs[len(runs) // 2]
            rows[gated] = m
            if ref is None:
                ref = m[3].cpu()
            assert torch.equal(m[3].cpu(), ref)
        a, b = rows[False], rows[True]
        print("K %3d streams %d | as enqueued: gpu %.0f us (%.2f M registrations/s), host enqueue %.0f us | gated: gpu %.0f us (%.2f M/s) [enqueue %.0f us hidden behind the gate]"
              % (K, n_streams, a[0], 512 * K / a[0], a[1], b[0], 512 * K / b[0], b[1]))


if __name__ == "__main__":
    main()
