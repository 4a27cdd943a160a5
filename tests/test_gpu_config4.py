"""-m gpu: the BENCHMARKED code paths at their stated sizes, as assertions (round-2 verdict, item 1).

(a) BASELINE config 4 in full -- 512 registrations (8 submaps x 64 scans, seeds 1000..1511) through the default entry
    (`randt_scan_register_batch_dev`, ambient-4 parameterisation, closed-form loss, FOUR registrations per solve
    workgroup = the `k_solve<3,1,64,true,4>` instantiation bench.py times) against the CPU oracle on every registration:
    pose, residual count, GNC solves, LM iterations, passes, termination type, cost; plus the per-iteration trace of the
    same instantiation against the oracle's on a sample, and bit-identity with the one-registration-per-workgroup kernel.
(c) the strong split of that batch over G = 2 / 4 / 8 "virtual ranks" on one GPU (contiguous `shard_range` pieces, one
    context + stream per piece) is bit-identical to the unsharded batch.
(d) the solve geometries -- split mode at every width, four-per-workgroup, one-per-workgroup -- are bit-identical.
"""
import os

import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import shard, synth
from util import IP, GpuRig, oracle_scan_map, oracle_submap, to_oracle_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg4(built):
    prob = synth.make_batch_problem(8, 64, 34)          # bench.py's base problem
    rig = GpuRig(prob)
    rig.build_submaps()
    rig.ctx.synchronize()
    osub = [oracle_submap(sm) for sm in prob["submaps"]]
    return prob, rig, osub


def _run(rig, ctx, mp, points, fidx, g4, trace_len=0):
    torch = rig.torch
    B = points.shape[0]
    pose = torch.from_numpy(np.ascontiguousarray(g4)).to(rig.dev)
    res = torch.zeros((B, 64), dtype=torch.uint8, device=rig.dev)
    ws = R.Maps(ctx, B, rig.mapp, rig.scan_cap, with_grid=False)
    trace = torch.zeros((B, max(trace_len, 1)), dtype=torch.float64, device=rig.dev)
    if trace_len:
        ctx.set_trace(trace, trace_len)
    sub = rig.submaps if ctx is rig.ctx else R.Maps(ctx, rig.n_sub, rig.mapp, rig.mapp.size_x * rig.mapp.size_y,
                                                    storage=rig.submaps.device_ptrs(), clear=False)
    R.scan_register_batch(ctx, points, rig.clu, sub, fidx, ws, mp, pose, res)
    ctx.synchronize()
    if trace_len:
        ctx.set_trace(None, 0)
    return pose.cpu().numpy(), res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1), trace.cpu().numpy()


def test_config4_full_batch_matches_oracle_on_every_registration(cfg4):
    prob, rig, osub = cfg4
    mp = R.default_matcher_params()                      # what bench.py runs
    g4 = synth.pose3_to_pose4(prob["guess"])
    TL = 3 * 400 + 1
    pose, res, trace = _run(rig, rig.ctx, mp, rig.points, rig.fixed_idx, g4, TL)
    fail, op, ocost, oit, ost = po.register_batch(prob["scans"], osub, prob["submap_of"], to_oracle_params(mp), g4, IP["n_clusters"],
                                                  IP["max_range"], want_stats=True)
    assert fail == 0 and (res["status"] == 0).all()
    # north_star tolerance ...
    assert np.abs(pose[:, 2:] - op[:, 2:]).max() <= 1e-4
    dth = np.arctan2(pose[:, 1], pose[:, 0]) - np.arctan2(op[:, 1], op[:, 0])
    assert np.abs((dth + np.pi) % (2 * np.pi) - np.pi).max() <= 1e-4
    # ... and what two fp64 implementations of the same control flow really give
    assert np.abs(pose - op).max() <= 1e-7, np.abs(pose - op).max()
    assert np.array_equal(res["n_residuals"], ost[:, 0])
    assert np.array_equal(res["gnc_solves"], ost[:, 1])
    assert np.array_equal(res["termination"], ost[:, 2])
    assert np.array_equal(res["iterations"], oit)
    # passes over the correspondence set: one per minimizer iteration (the candidate is evaluated WITH its Jacobian, where the
    # oracle, like Ceres, takes a cost-only evaluation and a Jacobian evaluation behind an accepted step) + the raw-residual
    # pass in front of the GNC loop (ndt_matcher.cpp:466-474)
    assert np.array_equal(res["n_evals"], oit + 1) and (ost[:, 3] >= oit).all()
    assert np.allclose(res["cost"], ocost, rtol=1e-8, atol=0)
    # per-iteration traces of THIS instantiation (four registrations per workgroup) on a sample of registrations
    for i in range(0, rig.B, 16):
        om = oracle_scan_map(prob["scans"][i])
        rc, p4, cost, st = po.register_pair(osub[prob["submap_of"][i]], om, to_oracle_params(mp), g4[i])
        n = int(trace[i, 0])
        assert n == len(st["trace_cost"]), i
        t = trace[i, 1: 1 + 3 * n].reshape(n, 3)
        assert np.allclose(t[:, 0], st["trace_cost"], rtol=1e-8) and np.allclose(t[:, 1], st["trace_radius"], rtol=1e-8)
        assert np.array_equal(t[:, 2].astype(int), st["trace_flag"])


def _ctx_with(torch, env=None, mode=None):
    """A context created under the given environment knobs (they are read at creation) / solve mode."""
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    if mode is not None:
        ctx.set_solve_mode(mode)
    return ctx


def test_config4_solve_geometries_bit_identical_including_traces(cfg4):
    """The solve geometry is a placement decision, not an arithmetic one.  The library's own choice for a lone
    512-registration batch (RANDT_SOLVE_AUTO: split mode, eight wavefronts per registration), the throughput kernel bench.py's
    16-stream region runs (one wavefront per registration, four registrations per workgroup) and the one-registration-per-
    workgroup kernel give the same poses, result records AND per-iteration traces bit for bit."""
    prob, rig, _ = cfg4
    torch = rig.torch
    mp = R.default_matcher_params()
    g4 = synth.pose3_to_pose4(prob["guess"])
    TL = 3 * 400 + 1
    p_auto, r_auto, t_auto = _run(rig, rig.ctx, mp, rig.points, rig.fixed_idx, g4, TL)                   # AUTO -> split, W = 8
    ctx_tp = _ctx_with(torch, mode=R._capi.SOLVE_THROUGHPUT)                                            # k_solve<..., RPB = 4>
    p4w, r4w, t4w = _run(rig, ctx_tp, mp, rig.points, rig.fixed_idx, g4, TL)
    ctx1 = _ctx_with(torch, env={"RANDT_SOLVE_RPB": "1"}, mode=R._capi.SOLVE_THROUGHPUT)                 # k_solve<..., RPB = 1>
    p1, r1, t1 = _run(rig, ctx1, mp, rig.points, rig.fixed_idx, g4, TL)
    assert np.array_equal(p4w, p1) and np.array_equal(r4w, r1) and np.array_equal(t4w, t1)
    assert np.array_equal(p_auto, p1) and np.array_equal(r_auto, r1) and np.array_equal(t_auto, t1)


@pytest.mark.parametrize("param", [R.PARAM_AMBIENT4, R.PARAM_MANIFOLD, R.PARAM_VECTOR, R.PARAM_ANALYTIC])
@pytest.mark.parametrize("intensity", [1, 0])
def test_split_mode_bit_identical_for_every_instantiation(cfg4, param, intensity):
    """Split mode (several wavefronts per registration) against the one-wavefront kernel for every parameterisation and
    both residual dimensions, at the widths the library picks (8, 4) and at widths that leave trips to wavefront 0
    (2, 3: n_res > 64 W for most registrations here) or do not divide the trips evenly (5, 7); batch sizes with a ragged end."""
    prob, rig, _ = cfg4
    torch = rig.torch
    mp = R.default_matcher_params(parameterization=param, use_intensity=intensity)
    g4 = synth.pose3_to_pose4(prob["guess"])
    n = 70
    pts, fidx = rig.points[:n].contiguous(), rig.fixed_idx[:n].contiguous()
    ref = _run(rig, _ctx_with(torch, mode=R._capi.SOLVE_THROUGHPUT), mp, pts, fidx, g4[:n], 3 * 400 + 1)
    assert (ref[1]["status"] == 0).all() and ref[1]["n_residuals"].max() > 64 * 5
    for w in ("2", "3", "4", "5", "7", "8"):
        got = _run(rig, _ctx_with(torch, env={"RANDT_SOLVE_SPLIT": w}), mp, pts, fidx, g4[:n], 3 * 400 + 1)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), w
    auto = _run(rig, rig.ctx, mp, pts, fidx, g4[:n], 3 * 400 + 1)          # the library's choice
    for a, b in zip(auto, ref):
        assert np.array_equal(a, b)


def test_split_mode_general_loss_shapes_fall_back_to_one_wavefront(cfg4):
    """Only the closed-form (alpha = -2) kernels have a split instantiation; other Barron shapes run one wavefront per
    registration whatever the batch size -- same results as in throughput mode."""
    prob, rig, _ = cfg4
    torch = rig.torch
    mp = R.default_matcher_params(loss_alpha=-1.0)
    g4 = synth.pose3_to_pose4(prob["guess"])
    n = 24
    pts, fidx = rig.points[:n].contiguous(), rig.fixed_idx[:n].contiguous()
    a = _run(rig, rig.ctx, mp, pts, fidx, g4[:n])
    b = _run(rig, _ctx_with(torch, mode=R._capi.SOLVE_THROUGHPUT), mp, pts, fidx, g4[:n])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("G", [2, 4, 8])
def test_config4_strong_split_virtual_ranks_bit_identical(cfg4, G):
    prob, rig, _ = cfg4
    torch = rig.torch
    mp = R.default_matcher_params()
    g4 = synth.pose3_to_pose4(prob["guess"])
    ref_p, ref_r, _ = _run(rig, rig.ctx, mp, rig.points, rig.fixed_idx, g4)
    streams = [torch.cuda.Stream(device=rig.dev) for _ in range(G)]
    ctxs = [R.Context(0, s.cuda_stream) for s in streams]
    torch.cuda.synchronize()
    parts_p, parts_r = [], []
    for r in range(G):
        lo, hi = shard.shard_range(rig.B, G, r)
        p, rr, _ = _run(rig, ctxs[r], mp, rig.points[lo:hi].contiguous(), rig.fixed_idx[lo:hi].contiguous(), g4[lo:hi])
        parts_p.append(p)
        parts_r.append(rr)
    assert np.array_equal(np.concatenate(parts_p), ref_p)
    assert np.array_equal(np.concatenate(parts_r), ref_r)


def test_auto_solve_mode_sees_other_contexts_in_flight(cfg4):
    """ADVICE r3 #1 / round-4 verdict item 6: RANDT_SOLVE_AUTO no longer ASSUMES an idle device.  A 64-registration batch on a
    context of its own takes the latency placement (eight wavefronts per registration) when nothing else of this process is in
    flight on the GPU, and the throughput placement (one wavefront each) while another context has work in flight -- detected by
    the library (enqueue stamps, hipStreamQuery for older ones), not declared by the caller; RANDT_SOLVE_LATENCY / _THROUGHPUT
    override in both directions.  The placement never changes a result."""
    import time

    prob, rig, _ = cfg4
    torch, lib = rig.torch, R._capi.load()
    mp = R.default_matcher_params()
    g4 = synth.pose3_to_pose4(prob["guess"])
    sa, sb = torch.cuda.Stream(device=rig.dev), torch.cuda.Stream(device=rig.dev)
    ca, cb = R.Context(0, sa.cuda_stream), R.Context(0, sb.cuda_stream)
    torch.cuda.synchronize()

    def views(ctx):
        return R.Maps(ctx, rig.n_sub, rig.mapp, rig.mapp.size_x * rig.mapp.size_y, storage=rig.submaps.device_ptrs(), clear=False)

    sub_a, sub_b = views(ca), views(cb)
    nb = 64
    pts_b, fidx_b = rig.points[:nb].contiguous(), rig.fixed_idx[:nb].contiguous()
    ws_b = R.Maps(cb, nb, rig.mapp, rig.scan_cap, with_grid=False)
    res_b = torch.zeros((nb, 64), dtype=torch.uint8, device=rig.dev)
    big = rig.points.repeat(8, 1, 1).contiguous()                       # 4096 registrations: a few hundred microseconds per launch
    fidx_big = rig.fixed_idx.repeat(8).contiguous()
    ws_a = R.Maps(ca, big.shape[0], rig.mapp, rig.scan_cap, with_grid=False)
    res_a = torch.zeros((big.shape[0], 64), dtype=torch.uint8, device=rig.dev)
    pose_a0 = torch.from_numpy(np.ascontiguousarray(np.tile(g4, (8, 1)))).to(rig.dev)

    def run_b():
        pose = torch.from_numpy(np.ascontiguousarray(g4[:nb])).to(rig.dev)
        with torch.cuda.stream(sb):
            R.scan_register_batch(cb, pts_b, rig.clu, sub_b, fidx_b, ws_b, mp, pose, res_b)
        placement = lib.randt_debug_last_solve_placement(cb._h)
        a_busy = not sa.query()              # still busy NOW => it was busy when the library decided (the test's premise, below)
        cb.synchronize()
        return placement, pose.cpu().numpy().copy(), res_b.cpu().numpy().copy(), a_busy

    def occupy_a(n=6):
        poses = [pose_a0.clone() for _ in range(n)]
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            for p in poses:
                R.scan_register_batch(ca, big, rig.clu, sub_a, fidx_big, ws_a, mp, p, res_a)
        return poses

    torch.cuda.synchronize()
    alone = run_b()
    assert alone[0] == 8                                                  # nothing else in flight: the split geometry
    def run_b_beside_a():
        # ca's launches must still be running when cb enqueues (6 x ~0.4 ms normally does; on a loaded host the few host calls
        # in between can outlast them, then the premise -- not the library -- failed: more work, once more)
        for n in (6, 24, 96):
            keep = occupy_a(n)
            out = run_b()
            ca.synchronize()
            del keep
            if out[3]:
                return out
        pytest.skip("the host could not keep context A busy while context B enqueued")

    busy = run_b_beside_a()
    assert busy[0] == 0, busy[0]
    assert np.array_equal(alone[1], busy[1]) and np.array_equal(alone[2], busy[2])   # a placement, not a different answer
    ca.synchronize()                                                      # the library's own synchronisation resets ca's stamp
    assert run_b()[0] == 8
    keep = occupy_a(2)
    torch.cuda.synchronize()                                              # a synchronisation the library does not see ...
    time.sleep(0.002)                                                     # ... and an old stamp: the stream is asked
    assert run_b()[0] == 8
    keep = occupy_a()
    cb.set_solve_mode(R._capi.SOLVE_LATENCY)                              # the caller knows better
    assert run_b()[0] == 8
    torch.cuda.synchronize()
    ca.synchronize()
    cb.set_solve_mode(R._capi.SOLVE_THROUGHPUT)
    tp = run_b()
    assert tp[0] == 0 and np.array_equal(alone[1], tp[1])
    del keep
