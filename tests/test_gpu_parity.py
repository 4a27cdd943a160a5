"""-m gpu parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bars: bit-exact for the fp32 cell statistics, index grids, merge results and the integer
correspondence table; fp64 solve: pose within 1e-4 m / 1e-4 rad as BASELINE's north_star states
(observed ~1e-10), per-iteration cost trace within 1e-8 relative.
"""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import synth
from util import IP, GpuRig, cells_equal, oracle_scan_map, oracle_submap, problem, to_oracle_params

pytestmark = pytest.mark.gpu

POSE_TOL_T = 1e-4   # metres   (north_star)
POSE_TOL_R = 1e-4   # radians  (north_star)


@pytest.fixture(scope="module")
def rig(built):
    r = GpuRig(problem())
    r.build_submaps()
    r.build_scans()
    return r


@pytest.fixture(scope="module")
def osub(built):
    return [oracle_submap(sm) for sm in problem()["submaps"]]


def test_ndt_build_bit_exact(rig):
    counts = rig.scan_maps.counts()
    for i in range(rig.B):
        om = oracle_scan_map(rig.prob["scans"][i])
        cells, grid = rig.scan_maps.download(i)
        assert counts[i] == om.n_cells
        assert cells_equal(cells, om.cells()), f"scan {i}: cell statistics differ"
        assert np.array_equal(grid, om.grid())


def test_submap_merge_bit_exact(rig, osub):
    for j in range(rig.n_sub):
        cells, grid = rig.submaps.download(j)
        assert len(cells) == osub[j].n_cells
        assert cells_equal(cells, osub[j].cells()), f"submap {j}: merged cells differ"
        assert np.array_equal(grid, osub[j].grid())


@pytest.mark.parametrize("mahal,intensity", [(1, 1), (0, 1), (1, 0)])
def test_association_identical(rig, osub, mahal, intensity):
    torch = rig.torch
    mp = R.default_matcher_params(lookup_mahalanobis=mahal, use_intensity=intensity)
    k = mp.n_neighbours
    guess = torch.from_numpy(synth.pose3_to_pose4(rig.prob["guess"])).to(rig.dev)
    corr = torch.full((rig.B, rig.scan_cap, k), -7, dtype=torch.int32, device=rig.dev)
    R.associate_batch(rig.ctx, rig.submaps, rig.fixed_idx, rig.scan_maps, 0, rig.B, guess, mp, corr)
    rig.ctx.synchronize()
    corr = corr.cpu().numpy()
    for i in range(rig.B):
        om = oracle_scan_map(rig.prob["scans"][i])
        oc, _ = po.associate(osub[rig.prob["submap_of"][i]], om, synth.pose3_to_pose4(rig.prob["guess"][i]), k, mahal, intensity)
        assert np.array_equal(corr[i, : om.n_cells], oc), f"pair {i}"


def test_throughput_placement_instantiations_bit_exact(rig, osub):
    """RANDT_SOLVE_THROUGHPUT (the caller keeps several batches in flight) selects the register-capped instantiations of the build
    (64 registers, fenced chain reads) and of the association (128 registers, four-pair walk): same cells, index grids and
    correspondence tables, bit for bit, as the oracle -- and the same poses as the lone-batch geometry."""
    torch = rig.torch
    ctx = rig.ctx
    ctx.set_solve_mode(R._capi.SOLVE_THROUGHPUT)
    maps = R.Maps(ctx, rig.B, rig.mapp, rig.scan_cap, with_grid=True)
    R.ndt_build_batch(ctx, rig.points, rig.clu, maps)
    ctx.synchronize()
    counts = maps.counts()
    oms = [oracle_scan_map(rig.prob["scans"][i]) for i in range(rig.B)]
    for i in range(rig.B):
        cells, grid = maps.download(i)
        assert counts[i] == oms[i].n_cells
        assert cells_equal(cells, oms[i].cells()), f"scan {i}: cell statistics differ"
        assert np.array_equal(grid, oms[i].grid())
    mp = R.default_matcher_params()
    k = mp.n_neighbours
    g4 = synth.pose3_to_pose4(rig.prob["guess"])
    guess = torch.from_numpy(g4).to(rig.dev)
    corr = torch.full((rig.B, rig.scan_cap, k), -7, dtype=torch.int32, device=rig.dev)
    R.associate_batch(ctx, rig.submaps, rig.fixed_idx, maps, 0, rig.B, guess, mp, corr)
    ctx.synchronize()
    c = corr.cpu().numpy()
    for i in range(rig.B):
        oc, _ = po.associate(osub[rig.prob["submap_of"][i]], oms[i], g4[i], k, 1, 1)
        assert np.array_equal(c[i, : oms[i].n_cells], oc), f"pair {i}"
    # whole path in both placements: identical poses and result records
    out = []
    for mode in (R._capi.SOLVE_THROUGHPUT, R._capi.SOLVE_AUTO):  # (ends in SOLVE_AUTO, the module's default)
        ctx.set_solve_mode(mode)
        pose = torch.from_numpy(g4.copy()).to(rig.dev)
        res = torch.zeros((rig.B, 64), dtype=torch.uint8, device=rig.dev)
        R.scan_register_batch(ctx, rig.points, rig.clu, rig.submaps, rig.fixed_idx, maps, mp, pose, res)
        ctx.synchronize()
        out.append((pose.cpu().numpy(), res.cpu().numpy()))
    assert ctx._lib is not None
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    maps.close()


def _solve_both(rig, osub, mp, trace_len=3 * 512 + 1):
    torch = rig.torch
    k = mp.n_neighbours
    g4 = synth.pose3_to_pose4(rig.prob["guess"])
    pose = torch.from_numpy(g4.copy()).to(rig.dev)
    corr = torch.full((rig.B, rig.scan_cap, k), -1, dtype=torch.int32, device=rig.dev)
    res = torch.zeros((rig.B, 64), dtype=torch.uint8, device=rig.dev)
    trace = torch.zeros((rig.B, trace_len), dtype=torch.float64, device=rig.dev)
    R.associate_batch(rig.ctx, rig.submaps, rig.fixed_idx, rig.scan_maps, 0, rig.B, pose, mp, corr)
    rig.ctx.set_trace(trace, trace_len)
    R.solve_batch(rig.ctx, rig.submaps, rig.fixed_idx, rig.scan_maps, 0, rig.B, corr, mp, pose, res)
    rig.ctx.synchronize()
    rig.ctx.set_trace(None, 0)
    pose = pose.cpu().numpy()
    res = res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)
    trace = trace.cpu().numpy()
    op = to_oracle_params(mp)
    out = []
    for i in range(rig.B):
        om = oracle_scan_map(rig.prob["scans"][i])
        rc, p4, cost, st = po.register_pair(osub[rig.prob["submap_of"][i]], om, op, g4[i])
        out.append((p4, cost, st))
    return pose, res, trace, out


@pytest.mark.parametrize("param", [R.PARAM_AMBIENT4, R.PARAM_MANIFOLD, R.PARAM_VECTOR, R.PARAM_ANALYTIC])
@pytest.mark.parametrize("intensity", [1, 0])
def test_solve_matches_oracle(rig, osub, param, intensity):
    mp = R.default_matcher_params(parameterization=param, use_intensity=intensity)
    pose, res, trace, ref = _solve_both(rig, osub, mp)
    for i in range(rig.B):
        p4, cost, st = ref[i]
        # pose: SE(2) tolerance of north_star (translation components, rotation angle)
        assert abs(pose[i, 2] - p4[2]) <= POSE_TOL_T and abs(pose[i, 3] - p4[3]) <= POSE_TOL_T
        dth = np.arctan2(pose[i, 1], pose[i, 0]) - np.arctan2(p4[1], p4[0])
        assert abs((dth + np.pi) % (2 * np.pi) - np.pi) <= POSE_TOL_R
        # in practice the fp64 paths agree far tighter
        assert np.allclose(pose[i], p4, rtol=0, atol=1e-7), (pose[i], p4)
        assert res["n_residuals"][i] == st["n_residuals"]
        assert res["gnc_solves"][i] == st["n_solves"]
        assert res["iterations"][i] == st["n_iterations"]
        assert res["termination"][i] == st["termination"]
        assert np.isclose(res["cost"][i], cost, rtol=1e-8)
        n = int(trace[i, 0])
        assert n == len(st["trace_cost"])
        t = trace[i, 1 : 1 + 3 * n].reshape(n, 3)
        assert np.allclose(t[:, 0], st["trace_cost"], rtol=1e-8)
        assert np.allclose(t[:, 1], st["trace_radius"], rtol=1e-8)
        assert np.array_equal(t[:, 2].astype(int), st["trace_flag"])


def test_full_pipeline_pose_tolerance(rig, osub):
    """BASELINE config 2/4 unit: raw scan -> NDT -> associate -> solve in one call."""
    torch = rig.torch
    mp = R.default_matcher_params()
    g4 = synth.pose3_to_pose4(rig.prob["guess"])
    pose = torch.from_numpy(g4.copy()).to(rig.dev)
    res = torch.zeros((rig.B, 64), dtype=torch.uint8, device=rig.dev)
    ws = R.Maps(rig.ctx, rig.B, rig.mapp, rig.scan_cap, with_grid=False)
    R.scan_register_batch(rig.ctx, rig.points, rig.clu, rig.submaps, rig.fixed_idx, ws, mp, pose, res)
    rig.ctx.synchronize()
    pose = pose.cpu().numpy()
    op = to_oracle_params(mp)
    for i in range(rig.B):
        om = oracle_scan_map(rig.prob["scans"][i])
        rc, p4, cost, st = po.register_pair(osub[rig.prob["submap_of"][i]], om, op, g4[i])
        assert np.allclose(pose[i], p4, rtol=0, atol=1e-7)
        est = synth.pose4_to_pose3(pose[i])
        # and the registration actually works: within a few cm of the ground truth
        assert np.all(np.abs(est[:2] - rig.prob["truth"][i][:2]) < 0.1)
        assert abs(est[2] - rig.prob["truth"][i][2]) < 0.03


def test_host_pair_entry_point(rig, osub):
    from randt_slam_amd import host

    mp = R.default_matcher_params()
    g4 = synth.pose3_to_pose4(rig.prob["guess"][0])
    p, r = host.register_pair(rig.ctx, rig.submaps, int(rig.prob["submap_of"][0]), rig.scan_maps, 0, mp, g4)
    om = oracle_scan_map(rig.prob["scans"][0])
    rc, p4, cost, st = po.register_pair(osub[rig.prob["submap_of"][0]], om, to_oracle_params(mp), g4)
    assert np.allclose(p, p4, atol=1e-7)
    assert r["iterations"] == st["n_iterations"]


def test_full_batch_size_independent_properties(built):
    """BASELINE config 4 at FULL size (512 registrations): properties that need no oracle run --
    run-to-run bitwise determinism, invariance under a permutation of the batch, idempotence
    (re-registering from the solution stops immediately at the same pose), and agreement with
    the simulated ground truth."""
    import torch

    prob = synth.make_batch_problem(8, 64, 34)
    rig = GpuRig(prob)
    rig.build_submaps()
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD)
    g4 = synth.pose3_to_pose4(prob["guess"])
    ws = R.Maps(rig.ctx, rig.B, rig.mapp, rig.scan_cap, with_grid=False)

    def run(points, fidx, guess):
        pose = torch.from_numpy(guess.copy()).to(rig.dev)
        res = torch.zeros((rig.B, 64), dtype=torch.uint8, device=rig.dev)
        R.scan_register_batch(rig.ctx, points, rig.clu, rig.submaps, fidx, ws, mp, pose, res)
        rig.ctx.synchronize()
        return pose.cpu().numpy(), res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)

    p1, r1 = run(rig.points, rig.fixed_idx, g4)
    p2, r2 = run(rig.points, rig.fixed_idx, g4)
    assert np.array_equal(p1, p2) and np.array_equal(r1, r2)                       # deterministic reduction tree
    perm = np.random.default_rng(0).permutation(rig.B)
    pp, rp = run(rig.points[torch.from_numpy(perm).to(rig.dev)], rig.fixed_idx[torch.from_numpy(perm).to(rig.dev)], g4[perm])
    assert np.array_equal(pp, p1[perm])                                            # registrations are independent
    # near-idempotence: re-registering from the solution re-associates at the new pose (correspondences are
    # frozen per call, quirk A.7-8), so it may still move by the registration noise, but no further, and it
    # needs fewer iterations
    p3, r3 = run(rig.points, rig.fixed_idx, p1)
    d = np.abs(p3 - p1)
    assert np.median(d.max(axis=1)) < 0.03 and r3["iterations"].mean() < r1["iterations"].mean()
    # accuracy against the simulated truth
    est = synth.pose4_to_pose3(p1)
    err_t = np.hypot(est[:, 0] - prob["truth"][:, 0], est[:, 1] - prob["truth"][:, 1])
    err_r = np.abs(synth.wrap_angle(est[:, 2] - prob["truth"][:, 2]))
    assert np.median(err_t) < 0.02 and np.percentile(err_t, 95) < 0.1 and np.median(err_r) < 0.005
    assert (r1["status"] == 0).all() and (r1["termination"] >= 1).all() and (r1["termination"] <= 3).all()


def test_atomic_ranking_is_checked_in_every_launch_and_falls_back(built, monkeypatch):
    """k_ndt_build ranks points with one returning LDS atomic each, which is only right if colliding lanes are served in
    lane order.  The kernel re-checks that on a sample in every launch; a violation makes the workgroup rank again with
    ballots (bit-exact results) and the context stops using the atomic ranking.  RANDT_DEBUG_FORCE_MISRANK makes the check
    fail on purpose: the fallback must engage, be reported, and leave the cell statistics bit-identical to the oracle."""
    import ctypes as C

    import torch

    lib = R._capi.load()
    lib.randt_debug_lds_atomics_lane_ordered.argtypes = [C.c_void_p]
    lib.randt_debug_build_rank_fallbacks.argtypes = [C.c_void_p]
    prob = problem()
    dev = torch.device("cuda:0")
    pts = torch.from_numpy(prob["scans"]).to(dev)
    B = pts.shape[0]

    def build_and_check(ctx):
        maps = R.Maps(ctx, B, R.indoor_map_params(), 512, with_grid=True)
        R.ndt_build_batch(ctx, pts, R.indoor_cluster_params(), maps)
        ctx.synchronize()
        for i in range(B):
            om = oracle_scan_map(prob["scans"][i])
            cells, grid = maps.download(i)
            assert cells_equal(cells, om.cells()) and np.array_equal(grid, om.grid()), i

    healthy = R.Context(0, torch.cuda.current_stream().cuda_stream)
    assert lib.randt_debug_lds_atomics_lane_ordered(healthy._h) == 1          # the device passes the creation-time probe
    build_and_check(healthy)
    assert lib.randt_debug_build_rank_fallbacks(healthy._h) == 0              # ... and every in-launch check
    monkeypatch.setenv("RANDT_DEBUG_FORCE_MISRANK", "1")
    sick = R.Context(0, torch.cuda.current_stream().cuda_stream)
    monkeypatch.delenv("RANDT_DEBUG_FORCE_MISRANK")
    build_and_check(sick)                                                      # re-ranked in-kernel: still bit-exact
    n = lib.randt_debug_build_rank_fallbacks(sick._h)
    assert n >= B                                                              # every workgroup took the fallback
    build_and_check(sick)                                                      # the launcher has seen the report ...
    assert lib.randt_debug_lds_atomics_lane_ordered(sick._h) == 0              # ... and switched the context to ballots
    assert lib.randt_debug_build_rank_fallbacks(sick._h) == n                  # no further fallbacks: ballots from the start
    assert lib.randt_last_error(sick._h).decode() == ""                       # a downgrade is not an error: no stale text for a later failure to show
