"""-m gpu parity over the reference's OTHER shipped parameter sets and the kernel instantiations the indoor
defaults never reach (VERDICT r01 "untested HIP code on shipped parameter sets"):

* the general Barron branches of `ceres_loss_functions.cpp:19-39` (|alpha| <= 0.05 -> log, general power, alpha >= 2 ->
  identity) with the outdoor / mixed loss scale and GNC divisor (`config/parameters_outdoor.yaml:24-39`,
  `parameters_mixed.yaml:24-39`) -- the `AM2 = false` instantiations of k_solve / k_solve_window;
* the two-wavefront solve (`RANDT_SOLVE_BLOCK=128`) and the LDS-staged index grid (`RANDT_ASSOC_STAGE_GRID=1`);
* Oxford geometry: 3.5 m cells, 400 x 400-slot map, 10 m search window, 2 neighbours, min 10 points per cell,
  loop-closure refinement with 10 GNC steps at scale 0.5 (`config/parameters_oxford.yaml:44-47,52-57,59-64`).

Same bars as test_gpu_parity.py: identical iteration counts / termination types, per-iteration (cost, radius, flag) traces
to 1e-8 relative, pose within the north_star tolerance (asserted far tighter).  Traces run the one-registration-per-
workgroup kernel (tracing disables the RPB grouping); the default four-per-workgroup kernel is compared pose-for-pose.
"""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import synth
from test_gpu_parity import POSE_TOL_R, POSE_TOL_T, _solve_both
from test_gpu_window import _run_drive, drive  # noqa: F401  (module-scoped fixture re-used)
from util import GpuRig, oracle_scan_map, oracle_submap, problem, to_oracle_params

pytestmark = pytest.mark.gpu

# (name, alpha, loss scale, GNC divisor, gnc steps): outdoor / mixed odometry and loop-closure values + the two
# remaining branches of BarronLoss::Evaluate
LOSS_PRESETS = [
    ("outdoor", -1.0, 2.0, 1.1, 3),        # parameters_outdoor.yaml:25-29
    ("outdoor_loop", -1.0, 2.0, 1.1, 1),   # parameters_outdoor.yaml:7,9
    ("mixed", -1.5, 2.0, 1.1, 3),          # parameters_mixed.yaml:25-29
    ("cauchy_branch", 0.0, 1.5, 1.3, 2),   # |alpha| <= 0.05
    ("cauchy_edge", 0.05, 1.5, 1.3, 2),
    ("identity_branch", 2.0, 1.5, 1.3, 2),  # alpha >= 2
    ("positive_power", 1.0, 1.5, 1.3, 2),
]


@pytest.fixture(scope="module")
def rig(built):
    r = GpuRig(problem())
    r.build_submaps()
    r.build_scans()
    return r


@pytest.fixture(scope="module")
def osub(built):
    return [oracle_submap(sm) for sm in problem()["submaps"]]


def _check(pose, res, trace, ref, B, radius_rtol=1e-8):
    for i in range(B):
        p4, cost, st = ref[i]
        assert abs(pose[i, 2] - p4[2]) <= POSE_TOL_T and abs(pose[i, 3] - p4[3]) <= POSE_TOL_T
        dth = np.arctan2(pose[i, 1], pose[i, 0]) - np.arctan2(p4[1], p4[0])
        assert abs((dth + np.pi) % (2 * np.pi) - np.pi) <= POSE_TOL_R
        assert np.allclose(pose[i], p4, rtol=0, atol=1e-7), (i, pose[i], p4)
        assert res["n_residuals"][i] == st["n_residuals"]
        assert res["gnc_solves"][i] == st["n_solves"]
        assert res["iterations"][i] == st["n_iterations"], (i, res["iterations"][i], st["n_iterations"])
        assert res["termination"][i] == st["termination"]
        assert np.isclose(res["cost"][i], cost, rtol=1e-8)
        if trace is not None:
            n = int(trace[i, 0])
            assert n == len(st["trace_cost"])
            t = trace[i, 1: 1 + 3 * n].reshape(n, 3)
            assert np.allclose(t[:, 0], st["trace_cost"], rtol=1e-8)
            assert np.allclose(t[:, 1], st["trace_radius"], rtol=radius_rtol)
            assert np.array_equal(t[:, 2].astype(int), st["trace_flag"])


@pytest.mark.parametrize("name,alpha,scale,div,steps", LOSS_PRESETS, ids=[p[0] for p in LOSS_PRESETS])
@pytest.mark.parametrize("param", [R.PARAM_AMBIENT4, R.PARAM_MANIFOLD])
def test_pair_solve_general_barron_shapes(rig, osub, name, alpha, scale, div, steps, param):
    mp = R.default_matcher_params(parameterization=param, loss_alpha=alpha, loss_scale=scale, mu_scale=scale, gnc_divisor=div,
                                  gnc_steps=steps)
    pose, res, trace, ref = _solve_both(rig, osub, mp)
    _check(pose, res, trace, ref, rig.B)
    if alpha < 2.0:
        assert (res["gnc_solves"] == max(1, steps)).any() or steps == 1


@pytest.mark.parametrize("alpha,intensity", [(-1.0, 0), (-1.5, 1)])
def test_pair_solve_general_shape_grouped_kernel(rig, osub, alpha, intensity):
    """The default launch (RPB = 4 wavefronts = 4 registrations per workgroup, no tracing) of the AM2 = false kernel: the
    grouped variant exists for alpha = -2 only, so this is the plain one-per-workgroup launch -- pose / record level."""
    torch = rig.torch
    mp = R.default_matcher_params(loss_alpha=alpha, loss_scale=2.0, mu_scale=2.0, gnc_divisor=1.1, gnc_steps=3, use_intensity=intensity)
    g4 = synth.pose3_to_pose4(rig.prob["guess"])
    pose = torch.from_numpy(g4.copy()).to(rig.dev)
    res = torch.zeros((rig.B, 64), dtype=torch.uint8, device=rig.dev)
    R.register_batch(rig.ctx, rig.submaps, rig.fixed_idx, rig.scan_maps, 0, rig.B, mp, pose, res)
    rig.ctx.synchronize()
    res = res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)
    op = to_oracle_params(mp)
    ref = []
    for i in range(rig.B):
        rc, p4, cost, st = po.register_pair(osub[rig.prob["submap_of"][i]], oracle_scan_map(rig.prob["scans"][i]), op, g4[i])
        ref.append((p4, cost, st))
    _check(pose.cpu().numpy(), res, None, ref, rig.B)


@pytest.mark.parametrize("alpha", [-2.0, -1.0])
def test_two_wavefront_solve_block_128(built, osub, monkeypatch, alpha):
    """RANDT_SOLVE_BLOCK=128: two wavefronts per registration with the LDS cross-wave combine (solve.hip eval_pass WAVES = 2)."""
    monkeypatch.setenv("RANDT_SOLVE_BLOCK", "128")
    r = GpuRig(problem())                      # the knob is read when the context is created
    r.build_submaps()
    r.build_scans()
    mp = R.default_matcher_params(loss_alpha=alpha, gnc_steps=2)
    pose, res, trace, ref = _solve_both(r, osub, mp)
    _check(pose, res, trace, ref, r.B)


def test_association_with_lds_staged_grid(built, osub, monkeypatch):
    """RANDT_ASSOC_STAGE_GRID=1: the submap's index grid staged in LDS instead of gathered from L2 -- identical tables."""
    import torch

    monkeypatch.setenv("RANDT_ASSOC_STAGE_GRID", "1")
    r = GpuRig(problem())
    r.build_submaps()
    r.build_scans()
    for mahal, intensity in [(1, 1), (0, 1)]:
        mp = R.default_matcher_params(lookup_mahalanobis=mahal, use_intensity=intensity)
        k = mp.n_neighbours
        guess = torch.from_numpy(synth.pose3_to_pose4(r.prob["guess"])).to(r.dev)
        corr = torch.full((r.B, r.scan_cap, k), -7, dtype=torch.int32, device=r.dev)
        R.associate_batch(r.ctx, r.submaps, r.fixed_idx, r.scan_maps, 0, r.B, guess, mp, corr)
        r.ctx.synchronize()
        corr = corr.cpu().numpy()
        for i in range(r.B):
            om = oracle_scan_map(r.prob["scans"][i])
            oc, _ = po.associate(osub[r.prob["submap_of"][i]], om, synth.pose3_to_pose4(r.prob["guess"][i]), k, mahal, intensity)
            assert np.array_equal(corr[i, : om.n_cells], oc), f"pair {i}"


# ---------------------------------------------------------------- window solve, general loss -----------------------------
@pytest.mark.parametrize("alpha,scale,div", [(-1.0, 2.0, 1.1), (-1.5, 2.0, 1.1), (0.0, 1.5, 1.3)])
def test_window_drive_general_barron_shapes(drive, alpha, scale, div):  # noqa: F811
    _run_drive(drive, mp_over=dict(loss_alpha=alpha, loss_scale=scale, mu_scale=scale, gnc_divisor=div))


# ---------------------------------------------------------------- Oxford geometry ----------------------------------------
OXFORD = dict(resolution=3.5, size=400, max_neighbour=10.0, min_points=10, max_range=100.0, k=2, gnc_steps=10, scale=0.5, div=1.1)
SCALE = 7.0   # the indoor synthetic world blown up by 7: 0.5 m cells -> 3.5 m cells, 12 m range -> 84 m


def _oxford_maps():
    ox = OXFORD
    n_clusters = int((2.0 * ox["max_range"] / ox["resolution"]) ** 2)     # ndt_slam.cpp:691
    mapp = R.MapParams(ox["size"], ox["size"], ox["resolution"], 0.0, 0.0, ox["max_neighbour"], ox["min_points"], 0)
    clu = R.ClusterParams(n_clusters, ox["max_range"])
    return mapp, clu, n_clusters


def _oxford_omap(cap=None):
    ox = OXFORD
    return po.Map(ox["size"], ox["size"], ox["resolution"], (0.0, 0.0), ox["max_neighbour"], ox["min_points"], cap)


def _scale_scan(s):
    s = s.copy()
    s[:, :2] *= SCALE
    return s


def test_oxford_geometry_loop_closure_refinement(built):
    """parameters_oxford.yaml on the device: 160 000-slot submap, window radius int(10 / 3.5) = 2, two neighbours, ten GNC
    steps -- build, merge, association and the solve traces against the oracle."""
    import torch

    ox = OXFORD
    prob = problem(n_submaps=1, scans_per_submap=6, n_keyframes=20)
    mapp, clu, n_clusters = _oxford_maps()
    dev = torch.device("cuda:0")
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    sm = prob["submaps"][0]
    kf = np.stack([_scale_scan(s) for s in sm["kf_scans"]])
    kf_rel = sm["kf_rel"].copy()
    kf_rel[:, :2] *= SCALE
    scans = np.stack([_scale_scan(s) for s in prob["scans"]])
    guess = prob["guess"].copy()
    guess[:, :2] *= SCALE
    B = len(scans)
    # device
    sub = R.Maps(ctx, 1, mapp, 8192, with_grid=True)
    tmp = R.Maps(ctx, len(kf), mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(kf).to(dev), clu, tmp)
    sub.merge(0, tmp, 0, synth.pose3_to_pose4(kf_rel))
    smaps = R.Maps(ctx, B, mapp, 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(scans).to(dev), clu, smaps)
    # oracle
    osub = _oxford_omap()
    for t in range(len(kf)):
        m = _oxford_omap(512)
        m.build(kf[t], n_clusters, ox["max_range"])
        m.transform(synth.pose3_to_pose4(kf_rel[t]))
        osub.merge(m)
    cells, grid = sub.download(0)
    from util import cells_equal
    assert cells_equal(cells, osub.cells()) and np.array_equal(grid, osub.grid())
    assert osub.n_cells > 50
    mp = R.default_matcher_params(n_neighbours=ox["k"], gnc_steps=ox["gnc_steps"], loss_scale=ox["scale"], mu_scale=1.0,
                                  gnc_divisor=ox["div"])       # estimateLoopConstraint: scale argument in the loss, the
    op = to_oracle_params(mp)                                   # matcher's loss_function_scale (1) in the mu formula
    g4 = synth.pose3_to_pose4(guess)
    pose = torch.from_numpy(g4.copy()).to(dev)
    corr = torch.full((B, 512, ox["k"]), -1, dtype=torch.int32, device=dev)
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    tl = 3 * 1024 + 1
    trace = torch.zeros((B, tl), dtype=torch.float64, device=dev)
    fidx = torch.zeros(B, dtype=torch.int32, device=dev)
    R.associate_batch(ctx, sub, fidx, smaps, 0, B, pose, mp, corr)
    ctx.set_trace(trace, tl)
    R.solve_batch(ctx, sub, fidx, smaps, 0, B, corr, mp, pose, res)
    ctx.synchronize()
    ctx.set_trace(None, 0)
    corr = corr.cpu().numpy()
    res = res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)
    ref = []
    n_solves, stable, spread, cost_spread = [], [], [], []

    def oracle_self_spread(om, p4, cost, guess):
        """Yardstick: how far the oracle lands from ITSELF when one input moves by one ulp (each component of the guess up and
        down, the loss scale up and down, the GNC divisor) or the linear solver changes (QR <-> normal equations)."""
        dev, cdev = 0.0, 0.0
        for trial in range(12):
            o2, g = to_oracle_params(mp), guess.copy()
            if trial < 4:
                g[trial] = np.nextafter(g[trial], 10.0)
            elif trial < 8:
                g[trial - 4] = np.nextafter(g[trial - 4], -10.0)
            elif trial == 8:
                o2.loss_scale = np.nextafter(o2.loss_scale, 1.0)
            elif trial == 9:
                o2.loss_scale = np.nextafter(o2.loss_scale, 0.0)
            elif trial == 10:
                o2.gnc_divisor = np.nextafter(o2.gnc_divisor, 2.0)
            else:
                o2.linear_solver = 1 - o2.linear_solver
            _, q, c2, _ = po.register_pair(osub, om, o2, g)
            dev, cdev = max(dev, float(np.abs(q - p4).max())), max(cdev, abs(c2 - cost) / cost)
        return dev, cdev

    for i in range(B):
        om = _oxford_omap(512)
        om.build(scans[i], n_clusters, ox["max_range"])
        c, g = smaps.download(i)
        assert cells_equal(c, om.cells())
        oc, _ = po.associate(osub, om, g4[i], ox["k"], 1, 1)
        assert np.array_equal(corr[i, : om.n_cells], oc)
        rc, p4, cost, st = po.register_pair(osub, om, op, g4[i])
        sp, csp = oracle_self_spread(om, p4, cost, g4[i])
        ref.append((p4, cost, st))
        n_solves.append(st["n_solves"])
        spread.append(sp)
        cost_spread.append(csp)
        stable.append(sp < 1e-9)
    pose, trace = pose.cpu().numpy(), trace.cpu().numpy()
    assert sum(stable) >= B - 2
    # Registrations whose minimiser the oracle itself reproduces under one-ulp changes of its inputs: the full bar (1e-7 pose,
    # identical iteration counts / termination, traces to 1e-8).  The others (seen here: one pair with ~100 residuals that
    # needs 160+ LM iterations along a flat valley of the un-manifolded 4-parameter problem, where the twelve perturbed oracles
    # land up to 1.1e-2 m -- median 5.5e-3 -- away from the unperturbed one, at costs equal to 1e-5) are ill-conditioned in the
    # reference itself: there the device result has to be as close to the oracle as the oracle is to itself, and as good (cost).
    idx = [i for i in range(B) if stable[i]]
    # radii to 1e-6: a trust-region radius is a function of cost DIFFERENCES (rel = cost change / model change), which late in
    # a ten-step schedule are 1e-6 of the cost -- costs that agree to 1e-14 give radii that agree to ~1e-8 at best
    _check(pose[idx], res[idx], trace[idx], [ref[i] for i in idx], len(idx), radius_rtol=1e-6)
    for i in range(B):
        if stable[i]:
            continue
        p4, cost, st = ref[i]
        tol_t, tol_r = max(POSE_TOL_T, 2.0 * spread[i]), max(POSE_TOL_R, 2.0 * spread[i])
        assert abs(pose[i, 2] - p4[2]) <= tol_t and abs(pose[i, 3] - p4[3]) <= tol_t, (i, pose[i], p4, spread[i])
        dth = np.arctan2(pose[i, 1], pose[i, 0]) - np.arctan2(p4[1], p4[0])
        assert abs((dth + np.pi) % (2 * np.pi) - np.pi) <= tol_r
        assert np.isclose(res["cost"][i], cost, rtol=max(1e-6, 10.0 * cost_spread[i])) and res["n_residuals"][i] == st["n_residuals"]
    assert max(n_solves) >= 5          # the long GNC schedule really ran


def test_oxford_preset_odometry_drive_end_to_end(built):
    """parameters_oxford.yaml through the whole `processScan` call pattern on the device (not only the pair solve above): 3.5 m
    cells on the 400 x 400 map, 10 m search window, min 10 points per cell, two neighbours, constant-velocity fixed-lag window
    with covariance_scaling_factor 0.01 x motion_sqrtI diag(1, 1, 10, 1, 3, 0.1, 20, 60), ndt_weight 5000, GNC 2 steps / divisor
    1.1 at scale 1, rejection gate 5 m / 2 rad, every second scan a keyframe, 20-state submaps with 10 states of overlap
    (`config/parameters_oxford.yaml:41-47,52-57,59-87`) -- the synthetic world blown up by 7 and driven at 7 x the speed.  Pose
    by pose against the same loop on the oracle, and against the truth."""
    import torch
    from oracle_backend import OracleBackend
    from randt_slam_amd import odometry

    ox = OXFORD
    mapp, clu, n_clusters = _oxford_maps()
    world = synth.make_world()
    n_scans, dt = 56, 0.25
    traj = synth.make_trajectory(3300, n_scans, step=0.25)
    scans = [_scale_scan(synth.make_scan(world, traj[i], 14000 + i)) for i in range(n_scans)]
    truth = traj.copy()
    truth[:, :2] *= SCALE
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=2, gnc_divisor=1.1, loss_scale=1.0, mu_scale=1.0,
                                  n_neighbours=ox["k"])
    wp = R.window_params(motion_sqrtI_diag=(1, 1, 10, 1, 3, 0.1, 20, 60), covariance_scaling_factor=0.01, ndt_weight=5000.0,
                         reject_t=5.0, reject_r=2.0, smoothing_steps=3, const_vel=1)
    drive = dict(insertion_step=2, submap_size_poses=20, submap_overlap=10)
    oip = dict(size_x=ox["size"], size_y=ox["size"], resolution=ox["resolution"], max_neighbour_dist=ox["max_neighbour"],
               min_points_per_cell=ox["min_points"], n_clusters=n_clusters, max_range=ox["max_range"])
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    gpu = odometry.Odometry(odometry.HipBackend(ctx, mapp, clu), mp, wp, drive)
    cpu = odometry.Odometry(OracleBackend(params=oip), mp, wp, drive)
    origin_inv = synth.se2_inv3(truth[0])
    worst_t = worst_r = 0.0
    for i in range(n_scans):
        pg = gpu.process_scan(scans[i], i * dt)
        pc = cpu.process_scan(scans[i], i * dt)
        worst_t = max(worst_t, np.abs(pg[2:] - pc[2:]).max())
        worst_r = max(worst_r, abs(synth.wrap_angle(np.arctan2(pg[1], pg[0]) - np.arctan2(pc[1], pc[0]))))
        assert worst_t <= POSE_TOL_T and worst_r <= POSE_TOL_R, (i, pg, pc)
        rel = synth.se2_mul3(origin_inv, truth[i])
        est = synth.pose4_to_pose3(pg)
        assert np.all(np.abs(est[:2] - rel[:2]) < 0.25 * SCALE) and abs(synth.wrap_angle(est[2] - rel[2])) < 0.08, (i, est, rel)
    assert gpu.n_finished_submaps == cpu.n_finished_submaps >= 2           # roll-overs with overlap happened
    assert gpu.n_registrations == cpu.n_registrations and gpu.n_rejected == cpu.n_rejected
    print("oxford preset drive: max deviation GPU vs oracle %.3e m, %.3e rad; %d submaps" % (worst_t, worst_r, gpu.n_finished_submaps))
