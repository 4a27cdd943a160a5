"""-m gpu parity of the fixed-lag window path: randt_predict_state and
randt_register_window against the CPU oracle on a simulated drive, incl. the overlap case with two
fixed maps, the IMU factor, the constant-acceleration layout and the rejection gate."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import synth
from util import IP, cells_equal, oracle_scan_map, to_oracle_params

pytestmark = pytest.mark.gpu


def to_oracle_wp(wp):
    o = po.WindowParams()
    for name, _ in R.WindowParams._fields_:
        v = getattr(wp, name)
        if name == "motion_sqrtI":
            for i in range(64):
                o.motion_sqrtI[i] = v[i]
        else:
            setattr(o, name, v)
    return o


def _make_drive(env=None, n_scans=8):
    import os

    import torch

    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)   # the environment knobs are read at creation
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v

    world = synth.make_world()
    dt = 0.25
    traj = synth.make_trajectory(3100, n_scans + 34, step=0.25)
    origin_inv = synth.se2_inv3(traj[0])
    rel = np.array([synth.se2_mul3(origin_inv, p) for p in traj])
    rel[:, 2] = synth.wrap_angle(rel[:, 2])
    kf = [synth.make_scan(world, traj[t], 7000 + t) for t in range(0, 32, 4)]
    kf_rel = rel[0:32:4]
    scans = [synth.make_scan(world, traj[32 + i], 8000 + i) for i in range(n_scans)]

    # oracle side
    def omap(cap=None):
        return po.Map(IP["size_x"], IP["size_y"], IP["resolution"], (0, 0), IP["max_neighbour_dist"], IP["min_points_per_cell"], cap)

    osub, osub2 = omap(), omap()
    for i, s in enumerate(kf):
        m = oracle_scan_map(s)
        m.transform(synth.pose3_to_pose4(kf_rel[i]))
        osub.merge(m)
        if i % 2 == 0:
            osub2.merge(m)          # a sparser second "previous submap" for the overlap case
    oscans = [oracle_scan_map(s) for s in scans]
    # device side
    dev = torch.device("cuda:0")
    mapp, clu = R.indoor_map_params(), R.indoor_cluster_params()
    sub = R.Maps(ctx, 2, mapp, 10000, with_grid=True)
    tmp = R.Maps(ctx, len(kf), mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(np.stack(kf)).to(dev), clu, tmp)
    sub.merge(0, tmp, 0, synth.pose3_to_pose4(kf_rel))
    for i in range(0, len(kf), 2):
        sub.merge(1, tmp, i, synth.pose3_to_pose4(kf_rel[i:i + 1]))
    smaps = R.Maps(ctx, n_scans, mapp, 512, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(np.stack(scans)).to(dev), clu, smaps)
    ctx.synchronize()
    assert cells_equal(sub.download(1)[0], osub2.cells())
    return dict(ctx=ctx, sub=sub, smaps=smaps, osub=osub, osub2=osub2, oscans=oscans, truth=rel[32:32 + n_scans], dt=dt, torch=torch)


@pytest.fixture(scope="module")
def drive(built):
    return _make_drive()


@pytest.fixture(scope="module")
def drive_long(built):
    """Fourteen scans: windows of up to twelve optimised states (window_gen_big.hip)."""
    return _make_drive(n_scans=14)


@pytest.fixture(scope="module")
def drive_general(built):
    """The same drive on a context whose windows ALL take the general kernel (window_gen.hip)."""
    return _make_drive({"RANDT_WINDOW_GENERAL": "1"})


def test_predict_state_matches_oracle(built):
    rng = np.random.default_rng(0)
    for _ in range(10):
        th = rng.uniform(-3, 3)
        st = R.make_state([np.cos(th), np.sin(th), rng.normal(), rng.normal()], lin_vel=rng.normal(0, 1, 2), rot_vel=rng.normal(0, .5),
                          lin_acc=rng.normal(0, 1, 2), imu_bias=0.1, stamp=5.0)
        stamp = 5.0 + rng.choice([0.0, 0.1, 0.25, 1.0])
        a = R.predict_state(st, stamp)
        b = po.predict_state(st.astype(po.STATE_DTYPE), stamp)
        for f in a.dtype.names:
            assert np.allclose(a[f], b[f], rtol=0, atol=1e-15), f


def _run_drive(drive, n_fixed=1, use_imu=0, const_vel=1, trace=True, mp_over=None, param=R.PARAM_MANIFOLD, lag=3):
    torch, ctx = drive["torch"], drive["ctx"]
    vec = param in (R.PARAM_VECTOR, R.PARAM_ANALYTIC)
    mp = R.default_matcher_params(parameterization=param, gnc_steps=3, **(mp_over or {}))
    wp = R.window_params(use_imu=use_imu, const_vel=const_vel)
    op, owp = to_oracle_params(mp), to_oracle_wp(wp)
    truth, dt = drive["truth"], drive["dt"]
    s0 = R.make_state(synth.pose3_to_pose4(truth[0]), lin_vel=(0.8, 0.0), rot_vel=0.0, stamp=0.0)
    gs, os_ = [s0], [s0.astype(po.STATE_DTYPE)]
    gtrans = otrans = synth.pose3_to_pose4(truth[0])
    fixed_o = [drive["osub"], drive["osub2"]][:n_fixed]
    imu_all = []
    dev_trace = torch.zeros(3 * 512 + 1, dtype=torch.float64, device="cuda:0")
    for i in range(1, len(truth)):
        gs.append(R.predict_state(gs[-1], i * dt, R.PARAM_VECTOR if vec else R.PARAM_MANIFOLD))
        os_.append(po.predict_state(os_[-1], i * dt, vector=vec))
        for f in gs[-1].dtype.names:   # predictions from states that agree to the solve's tolerance
            assert np.allclose(gs[-1][f], os_[-1][f], rtol=0, atol=1e-6), f
        imu_all.append(synth.wrap_angle(truth[i][2] - truth[i - 1][2]) + 0.002)
        S = min(len(gs) - 1, lag)                                 # ndt_matcher.cpp:343 smoothing_steps_iter
        win = list(range(i - S + 1, i + 1))                       # scan indices of the optimised states
        imu = np.array(imu_all[-S:]) if use_imu else None
        if trace:
            ctx.set_trace(dev_trace, dev_trace.shape[0])
        g_states, gtrans, g_rej, g_res = R.register_window(ctx, drive["sub"], list(range(n_fixed)), drive["smaps"], win,
                                                           np.array(gs[-S - 1:], dtype=R.STATE_DTYPE), mp, wp, gtrans, imu)
        ctx.set_trace(None, 0)
        rc, o_states, otrans, o_st = po.register_window(fixed_o, [drive["oscans"][w] for w in win],
                                                        np.array(os_[-S - 1:], dtype=po.STATE_DTYPE), op, owp, otrans, imu)
        assert rc == int(g_rej) == 0
        # pose of every state of the window within the north_star tolerance (observed ~1e-9)
        for j in range(S + 1):
            assert np.abs(g_states[j]["pose"][2:] - o_states[j]["pose"][2:]).max() <= 1e-4
            assert abs(g_states[j]["rot"] - o_states[j]["rot"]) <= 1e-4
            assert np.allclose(g_states[j]["pose"], o_states[j]["pose"], atol=1e-7)
            assert np.allclose(g_states[j]["lin_vel"], o_states[j]["lin_vel"], atol=1e-6)
            assert np.isclose(g_states[j]["rot_vel"], o_states[j]["rot_vel"], atol=1e-6)
            assert np.allclose(g_states[j]["lin_acc"], o_states[j]["lin_acc"], atol=1e-5)
            assert np.isclose(g_states[j]["imu_bias"], o_states[j]["imu_bias"], atol=1e-7)
            g = g_states[j]   # both representations are in sync on return (local_fuser.cpp:141-150)
            assert np.allclose(g["pose"], [np.cos(g["rot"]), np.sin(g["rot"]), g["pos"][0], g["pos"][1]], atol=1e-12)
        assert np.allclose(gtrans, otrans, atol=1e-7)
        assert g_res["n_residuals"] == o_st["n_residuals"] and g_res["gnc_solves"] == o_st["n_solves"]
        assert g_res["iterations"] == o_st["n_iterations"] and g_res["termination"] == o_st["termination"]
        if trace:
            tr = dev_trace.cpu().numpy()
            n = int(tr[0])
            t = tr[1:1 + 3 * n].reshape(n, 3)
            assert n == len(o_st["trace_cost"])
            assert np.allclose(t[:, 0], o_st["trace_cost"], rtol=1e-7)
            assert np.array_equal(t[:, 2].astype(int), o_st["trace_flag"])
        for j in range(S + 1):
            gs[len(gs) - S - 1 + j] = g_states[j]
            os_[len(os_) - S - 1 + j] = o_states[j]
        est = synth.pose4_to_pose3(gtrans)
        assert np.all(np.abs(est[:2] - truth[i][:2]) < 0.08) and abs(synth.wrap_angle(est[2] - truth[i][2])) < 0.03
    return gs


def test_window_drive_matches_oracle(drive):
    gs = _run_drive(drive)
    assert abs(np.hypot(*gs[-1]["lin_vel"]) - 1.0) < 0.15       # the smoother recovered the 1 m/s body speed


@pytest.mark.parametrize("k", [9, 12, 16])
def test_window_with_more_than_eight_neighbours(drive, k):
    """the window entry refused n_results_kd_lookup > 8 although the association (WIDE instantiation) and the window
    kernels take any k <= 16: lifted; states, iteration counts and traces equal the oracle's like for the shipped k = 4."""
    _run_drive(drive, n_fixed=2, mp_over=dict(n_neighbours=k))


def test_general_window_kernel_with_twelve_neighbours(drive_general):
    _run_drive(drive_general, n_fixed=2, use_imu=1, const_vel=0, mp_over=dict(n_neighbours=12))


def test_window_overlap_two_fixed_maps(drive):
    _run_drive(drive, n_fixed=2)


def test_window_with_imu_factor(drive):
    _run_drive(drive, use_imu=1)


def test_window_constant_acceleration_layout(drive):
    _run_drive(drive, const_vel=0)


# ---- smoothing_steps > 3 (no shipped configuration; ndt_matcher.cpp:343 takes any lag): window_gen.hip
@pytest.mark.parametrize("lag,kw", [
    (4, dict()),
    (5, dict(n_fixed=2)),
    (7, dict(n_fixed=2, use_imu=1, const_vel=0)),               # the largest problem: 68 tangent dimensions, 14 NDT terms
    (6, dict(param=R.PARAM_VECTOR, use_imu=1)),
    (4, dict(param=R.PARAM_ANALYTIC, const_vel=0, mp_over=dict(use_intensity=0, loss_alpha=-1.0))),
])
def test_window_longer_lags_match_oracle(drive, lag, kw):
    _run_drive(drive, lag=lag, **kw)


# ---- smoothing_steps 8..12: the same kernel source compiled for the longer band (window_gen_big.hip; one Cholesky workspace)
@pytest.mark.parametrize("lag,kw", [
    (8, dict()),
    (10, dict(param=R.PARAM_VECTOR, use_imu=1)),
    (12, dict(n_fixed=2, use_imu=1, const_vel=0)),              # the largest problem: 113 tangent dimensions, 24 NDT terms
])
def test_window_lags_of_eight_to_twelve_match_oracle(drive_long, lag, kw):
    _run_drive(drive_long, lag=lag, **kw)


@pytest.mark.parametrize("kw", [dict(), dict(n_fixed=2, use_imu=1, const_vel=0), dict(param=R.PARAM_VECTOR, n_fixed=2)])
def test_general_window_kernel_on_three_state_windows(drive_general, kw):
    """The general kernel on the windows the tuned kernel normally takes: same oracle, same assertions (decision trace included)."""
    _run_drive(drive_general, **kw)


def test_window_lag_beyond_the_device_solver_is_refused(drive):
    ctx = drive["ctx"]
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    st = R.make_state(synth.pose3_to_pose4(drive["truth"][0]), stamp=0.0)
    states = np.array([st] * 14, dtype=R.STATE_DTYPE)           # 13 optimised states
    with pytest.raises(R.RandtError) as e:
        R.register_window(ctx, drive["sub"], [0], drive["smaps"], [i % 8 for i in range(13)], states, mp, R.window_params(), st["pose"])
    assert e.value.status == 3 and "1..12 optimised states" in str(e.value)                # RANDT_ERR_UNSUPPORTED


def test_window_rejection_gate(drive):
    ctx = drive["ctx"]
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    wp = R.window_params(reject_t=0.001)
    truth, dt = drive["truth"], drive["dt"]
    prev = R.make_state(synth.pose3_to_pose4(truth[0]), lin_vel=(1.0, 0.0), stamp=0.0)
    states = np.array([prev, R.predict_state(prev, dt)], dtype=R.STATE_DTYPE)
    out, trans, rej, res = R.register_window(ctx, drive["sub"], [0], drive["smaps"], [1], states, mp, wp, synth.pose3_to_pose4(truth[0]))
    assert rej and np.array_equal(out[1]["pose"], out[0]["pose"]) and np.all(out[1]["lin_vel"] == 0) and np.array_equal(trans, out[0]["pose"])


# ---- optimize_on_manifold: false -- parameter blocks pos[2], rot[1] (ndt_matcher.cpp:290-313, 330-335), MotionModelFactor /
# RotationalResidual / NDTFrameToMap{,Intensity}FactorResidual on them, vector-form prediction
def test_vector_predict_state_matches_oracle(built):
    rng = np.random.default_rng(1)
    for _ in range(10):
        th = rng.uniform(-3, 3)
        st = R.make_state([np.cos(th), np.sin(th), rng.normal(), rng.normal()], lin_vel=rng.normal(0, 1, 2), rot_vel=rng.normal(0, .5),
                          lin_acc=rng.normal(0, 1, 2), imu_bias=0.1, stamp=5.0)
        stamp = 5.0 + rng.choice([0.0, 0.1, 0.25, 1.0])
        a = R.predict_state(st, stamp, R.PARAM_VECTOR)
        b = po.predict_state(st.astype(po.STATE_DTYPE), stamp, vector=True)
        for f in a.dtype.names:
            assert np.allclose(a[f], b[f], rtol=0, atol=1e-15), f


def test_vector_window_drive_matches_oracle(drive):
    gs = _run_drive(drive, param=R.PARAM_VECTOR)
    assert abs(np.hypot(*gs[-1]["lin_vel"]) - 1.0) < 0.15


def test_vector_window_overlap_imu_and_constant_acceleration(drive):
    _run_drive(drive, n_fixed=2, param=R.PARAM_VECTOR)
    _run_drive(drive, use_imu=1, param=R.PARAM_VECTOR)
    _run_drive(drive, const_vel=0, param=R.PARAM_VECTOR)


def test_vector_window_two_dimensional_and_general_loss(drive):
    _run_drive(drive, param=R.PARAM_VECTOR, mp_over=dict(use_intensity=0))
    _run_drive(drive, param=R.PARAM_VECTOR, mp_over=dict(loss_alpha=-1.0))


def test_vector_odometry_drive_matches_oracle_loop(built):
    """The processScan call pattern with optimize_on_manifold: false end to end (vector prediction, vector window, keyframe
    merges, a submap roll-over) against the same loop on the oracle."""
    import torch

    from randt_slam_amd import odometry
    from oracle_backend import OracleBackend

    world = synth.make_world()
    n_scans, dt = 40, 0.25
    traj = synth.make_trajectory(3200, n_scans, step=0.25)
    scans = [synth.make_scan(world, traj[i], 9000 + i) for i in range(n_scans)]
    small = dict(submap_size_poses=24, submap_overlap=8)
    mp = R.default_matcher_params(parameterization=R.PARAM_VECTOR, gnc_steps=3)
    wp = R.window_params()
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    gpu = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp, small)
    cpu = odometry.Odometry(OracleBackend(), mp, wp, small)
    assert gpu.vector and cpu.vector
    origin_inv = synth.se2_inv3(traj[0])
    for i in range(n_scans):
        pg = gpu.process_scan(scans[i], i * dt)
        pc = cpu.process_scan(scans[i], i * dt)
        assert np.abs(pg[2:] - pc[2:]).max() <= 1e-4 and abs(synth.wrap_angle(np.arctan2(pg[1], pg[0]) - np.arctan2(pc[1], pc[0]))) <= 1e-4, (i, pg, pc)
        rel = synth.se2_mul3(origin_inv, traj[i])
        est = synth.pose4_to_pose3(pg)
        assert np.all(np.abs(est[:2] - rel[:2]) < 0.25) and abs(synth.wrap_angle(est[2] - rel[2])) < 0.08, (i, est, rel)
    assert gpu.n_finished_submaps == cpu.n_finished_submaps == 1 and gpu.n_rejected == cpu.n_rejected == 0


def test_analytic_functor_window_drive_matches_oracle(drive):
    """use_analytic_expressions_for_optimization: true -- (pos, rot) blocks, vector motion / IMU factors (the analytic ones have
    the same, correct, Jacobians) and the reference's hand-written NDT functor with its inexact rotation Jacobian, reproduced as
    written: same iterates as the oracle's restatement of it, 3-D and 2-D."""
    _run_drive(drive, param=R.PARAM_ANALYTIC)
    _run_drive(drive, n_fixed=2, use_imu=1, param=R.PARAM_ANALYTIC, mp_over=dict(use_intensity=0))


@pytest.mark.parametrize("kw", [dict(), dict(n_fixed=2, use_imu=1, const_vel=0), dict(param=R.PARAM_VECTOR, n_fixed=2), dict(lag=5), dict(lag=9)])
def test_window_batch_entry_is_bit_identical_to_the_single_entry(drive_long, kw):
    """Round-4 verdict, item 5: randt_register_window_batch = N independent windows in ONE association launch + ONE solve launch (a
    workgroup per window).  Every window of the batch must be exactly what randt_register_window gives for it -- states, pose,
    rejection flag, result record, bit for bit: windows over different scans, of different depth into the drive, started from
    different priors, on the tuned kernel (lag 3), the general one (5) and its long-band compilation (9)."""
    d = drive_long
    ctx = d["ctx"]
    lag, n_fixed, param = kw.get("lag", 3), kw.get("n_fixed", 1), kw.get("param", R.PARAM_MANIFOLD)
    use_imu, const_vel = kw.get("use_imu", 0), kw.get("const_vel", 1)
    vec = param in (R.PARAM_VECTOR, R.PARAM_ANALYTIC)
    mp = R.default_matcher_params(parameterization=param, gnc_steps=3)
    wp = R.window_params(use_imu=use_imu, const_vel=const_vel)
    truth, dt = d["truth"], d["dt"]
    rng = np.random.default_rng(3)
    n_scans = len(truth)
    W = n_scans - lag
    S = lag
    states = np.zeros((W, S + 1), dtype=R.STATE_DTYPE)
    midx = np.zeros((W, S), dtype=np.int32)
    trans = np.zeros((W, 4))
    imu = np.zeros((W, S)) if use_imu else None
    for w in range(W):                                       # window w optimises scans w + 1 .. w + S from a perturbed chain
        p = truth[w] + rng.normal(0, [0.03, 0.03, 0.004])
        st = R.make_state(synth.pose3_to_pose4(p), lin_vel=(0.8 + 0.05 * rng.normal(), 0.02 * rng.normal()), rot_vel=0.01 * rng.normal(), stamp=w * dt)
        states[w, 0] = st
        for j in range(1, S + 1):
            st = R.predict_state(st, (w + j) * dt, R.PARAM_VECTOR if vec else R.PARAM_MANIFOLD)
            states[w, j] = st
            midx[w, j - 1] = w + j
            if use_imu:
                imu[w, j - 1] = synth.wrap_angle(truth[w + j][2] - truth[w + j - 1][2]) + 0.002
        trans[w] = states[w, S]["pose"]
    fidx = np.tile(np.arange(n_fixed, dtype=np.int32), (W, 1))
    single = [R.register_window(ctx, d["sub"], fidx[w], d["smaps"], midx[w], states[w], mp, wp, trans[w], None if imu is None else imu[w]) for w in range(W)]
    bst, btr, brej, bres = R.register_window_batch(ctx, d["sub"], fidx, d["smaps"], midx, states, mp, wp, trans, imu)
    assert W >= 4
    for w in range(W):
        st, tr, rej, res = single[w]
        assert st.tobytes() == bst[w].tobytes(), w
        assert tr.tobytes() == btr[w].tobytes() and bool(rej) == bool(brej[w]), w
        assert res.tobytes() == bres[w].tobytes(), w
        assert res["n_residuals"] > 100 and res["iterations"] >= 2
    assert len({bres[w]["iterations"] for w in range(W)}) > 1     # the windows really are different problems


def test_window_batch_entry_edge_cases(drive):
    """Zero windows is a no-op; a map index outside its batch, a window shape the kernels do not take and a bad parameter set are
    refused before anything is enqueued, with the states left as they were."""
    d = drive
    ctx = d["ctx"]
    mp, wp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3), R.window_params()
    st = np.zeros((2, 3), dtype=R.STATE_DTYPE)
    for w in range(2):
        s = R.make_state(synth.pose3_to_pose4(d["truth"][w]), lin_vel=(0.8, 0.0), stamp=w * d["dt"])
        for j in range(3):
            st[w, j] = s
            s = R.predict_state(s, (w + j + 1) * d["dt"])
    fidx, midx, tr = np.zeros((2, 1), np.int32), np.array([[1, 2], [2, 3]], np.int32), np.stack([st[0, 2]["pose"], st[1, 2]["pose"]])
    out = R.register_window_batch(ctx, d["sub"], fidx[:0], d["smaps"], midx[:0], st[:0], mp, wp, tr[:0])
    assert out[0].shape == (0, 3) and len(out[3]) == 0
    before = st.copy()
    bad_m = midx.copy()
    bad_m[1, 1] = 10 ** 6
    with pytest.raises(R.RandtError) as e:
        R.register_window_batch(ctx, d["sub"], fidx, d["smaps"], bad_m, st, mp, wp, tr)
    assert e.value.status == R._capi.ERR_INVALID
    with pytest.raises(R.RandtError) as e:
        R.register_window_batch(ctx, d["sub"], np.zeros((2, 3), np.int32), d["smaps"], midx, st, mp, wp, tr)     # three fixed maps
    assert e.value.status == R._capi.ERR_UNSUPPORTED
    with pytest.raises(R.RandtError) as e:
        R.register_window_batch(ctx, d["sub"], fidx, d["smaps"], midx, st, R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, n_neighbours=17), wp, tr)
    assert e.value.status == R._capi.ERR_UNSUPPORTED
    assert st.tobytes() == before.tobytes()
    ok = R.register_window_batch(ctx, d["sub"], fidx, d["smaps"], midx, st, mp, wp, tr)                           # and the well-formed call goes through
    assert ok[3]["n_residuals"].min() > 100 and not ok[2].any()


def test_replicas_in_lock_step_equal_independent_odometry_loops(built):
    """ReplicaOdometry: R copies of the processScan call pattern advancing in lock-step (one build / window / merge launch per
    step for all of them) give, replica by replica, the very poses R independent Odometry objects give on the same scans -- over
    keyframe merges and a submap roll-over with overlap."""
    import torch

    from randt_slam_amd import odometry

    world = synth.make_world()
    n_scans, dt, n_rep = 44, 0.25, 3
    small = dict(submap_size_poses=24, submap_overlap=8)
    mp = R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3)
    wp = R.window_params()
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    drives = []
    for r in range(n_rep):                                    # three different vehicles
        traj = synth.make_trajectory(3200 + 17 * r, n_scans, step=0.25)
        drives.append(np.stack([synth.make_scan(world, traj[i], 9000 + 100 * r + i) for i in range(n_scans)]))
    lone = []
    for r in range(n_rep):
        odo = odometry.Odometry(odometry.HipBackend(ctx, R.indoor_map_params(), R.indoor_cluster_params()), mp, wp, small)
        lone.append(np.array([odo.process_scan(drives[r][i], i * dt).copy() for i in range(n_scans)]))
        assert odo.n_finished_submaps == 1
        n_reg = odo.n_registrations
    rep = odometry.ReplicaOdometry(ctx, n_rep, R.indoor_map_params(), R.indoor_cluster_params(), mp, wp, small)
    dev = torch.device("cuda:0")
    both = torch.from_numpy(np.stack(drives, 1)).to(dev)      # (n_scans, R, N, 4)
    for i in range(n_scans):
        poses = rep.process_scans(both[i], i * dt)
        for r in range(n_rep):
            assert poses[r].tobytes() == lone[r][i].tobytes(), (i, r, poses[r], lone[r][i])
    assert rep.n_finished_submaps == 1 and rep.n_registrations == n_rep * n_reg
