"""Row f-4 (back end): 2-D pose-graph optimisation, GlobalFuser::optimizePoseGraph (global_fuser.cpp:13-105).
CPU part: the oracle restatement against finite differences, an independent solver (scipy) and known answers;
-m gpu part: the HIP path (block-tridiagonal chain factor + dense Schur complement on the loop-closure poses) against
the oracle through the C ABI."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import host

ODOM_SQI = np.diag([10.0, 10.0, 50.0])   # local_fuser.cpp:203-205


def rel(a, b):
    """(trans.translation(), trans.log()(2)) of a^-1 * b for poses (x, y, yaw)."""
    c, s = np.cos(a[2]), np.sin(a[2])
    d = b[:2] - a[:2]
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], (b[2] - a[2] + np.pi) % (2 * np.pi) - np.pi])


def compose(a, m):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([a[0] + c * m[0] - s * m[1], a[1] + s * m[0] + c * m[1], a[2] + m[2]])


def make_graph(n, loops, seed=0, noise=(0.02, 0.02, 0.005), loop_weight=40.0, laps=1.0, radius=10.0):
    """Circular drive with noisy odometry edges (dead-reckoned initial guess) and the given loop-closure pairs."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0, 2 * np.pi * laps, n, endpoint=False)
    truth = np.stack([radius * np.cos(th), radius * np.sin(th), th + np.pi / 2], 1)
    ia, ib, meas, sq = [], [], [], []
    for i in range(n - 1):
        ia.append(i)
        ib.append(i + 1)
        meas.append(rel(truth[i], truth[i + 1]) + rng.normal(size=3) * noise)
        sq.append(ODOM_SQI)
    n_odom = len(ia)
    for a, b in loops:
        ia.append(a)
        ib.append(b)
        meas.append(rel(truth[a], truth[b]) + rng.normal(size=3) * np.array(noise) * 0.5)
        sq.append(np.eye(3) * loop_weight)
    x0 = [truth[0].copy()]
    for i in range(n_odom):
        x0.append(compose(x0[-1], meas[i]))
    return truth, np.array(x0), np.array(ia, np.int32), np.array(ib, np.int32), np.array(meas), np.array(sq)


def graph_cost(x, ia, ib, meas, sq):
    return 0.5 * sum(float(np.sum(po.pg_edge(x[a], x[b], m, s)[0] ** 2)) for a, b, m, s in zip(ia, ib, meas, sq))


# ------------------------------------------------------------------------------------------------- oracle
def test_edge_jacobians_match_finite_differences(built):
    rng = np.random.default_rng(1)
    for _ in range(10):
        pa, pb = rng.normal(size=3) * [5, 5, 2], rng.normal(size=3) * [5, 5, 2]
        m = rng.normal(size=3) * 0.3
        sq = ODOM_SQI + rng.normal(size=(3, 3))
        r, Ja, Jb = po.pg_edge(pa, pb, m, sq)
        # residual itself: sqrt_information * [R_a^T (p_b - p_a) - p_ab ; normalize(yaw_b - yaw_a - yaw_ab)]
        want = rel(pa, pb) - m
        want[2] = (want[2] + np.pi) % (2 * np.pi) - np.pi
        assert np.allclose(r, sq @ want, atol=1e-12)
        h = 1e-6
        for k in range(3):
            d = np.zeros(3)
            d[k] = h
            fa = (po.pg_edge(pa + d, pb, m, sq)[0] - po.pg_edge(pa - d, pb, m, sq)[0]) / (2 * h)
            fb = (po.pg_edge(pa, pb + d, m, sq)[0] - po.pg_edge(pa, pb - d, m, sq)[0]) / (2 * h)
            assert np.abs(fa - Ja[:, k]).max() < 1e-6 and np.abs(fb - Jb[:, k]).max() < 1e-6


def test_angle_residual_wraps_like_the_reference(built):
    # NormalizeAngle maps into [-pi, pi) (state_manifold.h:17-23): a relative yaw of +pi comes out as -pi
    r, _, _ = po.pg_edge([0, 0, 0], [0, 0, np.pi], [0, 0, 0], np.eye(3))
    assert r[2] == -np.pi
    r, _, _ = po.pg_edge([0, 0, 3.0], [0, 0, -3.0], [0, 0, 0], np.eye(3))
    assert abs(r[2] - (2 * np.pi - 6.0)) < 1e-15


def test_oracle_matches_an_independent_solver(built):
    from scipy.optimize import least_squares

    truth, x0, ia, ib, meas, sq = make_graph(80, [(0, 79), (5, 70), (0, 40), (20, 60)], seed=3)
    tight = po.pg_params(function_tolerance=1e-14, parameter_tolerance=1e-14, gradient_tolerance=1e-14, max_iterations=100)
    x, res = po.pose_graph_optimize(x0, ia, ib, meas, sq, len(x0), tight)
    assert res["n_residual_blocks"] == len(ia) and res["n_loop_closures"] == 4

    def fun(v):
        X = np.vstack([x0[:1], v.reshape(-1, 3)])
        return np.concatenate([po.pg_edge(X[a], X[b], m, s)[0] for a, b, m, s in zip(ia, ib, meas, sq)])

    sol = least_squares(fun, x0[1:].ravel(), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert np.abs(sol.x.reshape(-1, 3) - x[1:]).max() < 1e-7
    assert abs(sol.cost - res["final_cost"]) < 1e-9 * max(1.0, sol.cost)
    assert np.array_equal(x[0], x0[0])                                   # poses.begin() is constant (:48-49)
    # with the reference's default tolerances the solve stops earlier, on the function tolerance
    x_def, res_def = po.pose_graph_optimize(x0, ia, ib, meas, sq, len(x0))
    assert res_def["termination"] == 1 and res_def["iterations"] <= res["iterations"]
    assert np.abs(x_def - x).max() < 1e-3


def test_consistent_measurements_are_a_known_answer(built):
    # zero-noise edges: the minimiser is the ground truth whatever the starting point, cost 0
    truth, _, ia, ib, meas, sq = make_graph(60, [(0, 59), (10, 45)], seed=4, noise=(0, 0, 0))
    rng = np.random.default_rng(5)
    x0 = truth + rng.normal(size=truth.shape) * [0.3, 0.3, 0.05]
    x0[0] = truth[0]
    p = po.pg_params(function_tolerance=1e-16, parameter_tolerance=1e-14, gradient_tolerance=1e-14, max_iterations=100)
    x, res = po.pose_graph_optimize(x0, ia, ib, meas, sq, 60, p)
    assert np.abs(x - truth).max() < 1e-9 and res["final_cost"] < 1e-16


def test_max_update_index_drops_late_loop_edges(built):
    # an edge is used iff id_begin + 1 == id_end || id_end <= max_update_index (global_fuser.cpp:32)
    truth, x0, ia, ib, meas, sq = make_graph(50, [(0, 30), (2, 48)], seed=6)
    x_all, r_all = po.pose_graph_optimize(x0, ia, ib, meas, sq, 49)
    x_cut, r_cut = po.pose_graph_optimize(x0, ia, ib, meas, sq, 40)
    assert r_all["n_residual_blocks"] == 51 and r_cut["n_residual_blocks"] == 50
    x_ref, _ = po.pose_graph_optimize(x0, ia[:-1], ib[:-1], meas[:-1], sq[:-1], 49)
    assert np.array_equal(x_cut, x_ref) and not np.array_equal(x_cut, x_all)
    # an open chain without loop closures is already optimal: one iteration, gradient ~ 0, poses unchanged
    x_open, r_open = po.pose_graph_optimize(x0, ia[:49], ib[:49], meas[:49], sq[:49], 49)
    assert np.abs(x_open - x0).max() < 1e-9 and r_open["final_cost"] < 1e-18


def test_huber_loss_limits_an_outlier(built):
    truth, x0, ia, ib, meas, sq = make_graph(60, [(0, 59), (10, 45)], seed=7)
    meas = meas.copy()
    meas[-1] += [3.0, -2.0, 0.4]   # a false loop closure
    tight = dict(function_tolerance=1e-12, max_iterations=200)
    x_l2, _ = po.pose_graph_optimize(x0, ia, ib, meas, sq, 60, po.pg_params(**tight))
    x_hu, r_hu = po.pose_graph_optimize(x0, ia, ib, meas, sq, 60, po.pg_params(use_robust_loss=1, loss_scale=2.0, **tight))
    assert np.abs(x_hu - truth).max() < np.abs(x_l2 - truth).max()
    # cost reported = sum 1/2 rho(s) with Huber: s <= a^2 -> s, else 2 a sqrt(s) - a^2
    s = np.array([np.sum(po.pg_edge(x_hu[a], x_hu[b], m, q)[0] ** 2) for a, b, m, q in zip(ia, ib, meas, sq)])
    rho = np.where(s <= 4.0, s, 2 * 2.0 * np.sqrt(s) - 4.0)
    assert abs(0.5 * rho.sum() - r_hu["final_cost"]) < 1e-9 * r_hu["final_cost"]


def test_invalid_graphs_are_refused(built):
    with pytest.raises(ValueError):
        po.pose_graph_optimize(np.zeros((3, 3)), [0, 1], [1, 7], np.zeros((2, 3)), np.stack([np.eye(3)] * 2), 9)
    with pytest.raises(ValueError):
        po.pose_graph_optimize(np.zeros((3, 3)), [0, 1], [1, 1], np.zeros((2, 3)), np.stack([np.eye(3)] * 2), 9)


# ------------------------------------------------------------------------------------------------- GPU
def _gpu_vs_oracle(x0, ia, ib, meas, sq, mui, op=None, gp=None, tol=1e-7):
    import torch

    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    xo, ro = po.pose_graph_optimize(x0, ia, ib, meas, sq, mui, op)
    xg, rg = host.pose_graph_optimize(ctx, x0, ia, ib, meas, sq, mui, gp)
    assert rg["n_residual_blocks"] == ro["n_residual_blocks"] and rg["n_loop_closures"] == ro["n_loop_closures"]
    assert rg["termination"] == ro["termination"] and rg["iterations"] == ro["iterations"]
    assert abs(rg["initial_cost"] - ro["initial_cost"]) <= 1e-10 * ro["initial_cost"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-7 * max(ro["final_cost"], 1e-12)
    assert np.abs(xg - xo).max() < tol     # the 1e-4 m / 1e-4 rad bar of the path, with three digits to spare
    return xg, rg, xo, ro


@pytest.mark.gpu
def test_hip_pose_graph_matches_oracle(built):
    # chain + four loop closures, default (reference) options
    truth, x0, ia, ib, meas, sq = make_graph(120, [(0, 119), (5, 110), (0, 60), (20, 100)], seed=0)
    xg, rg, _, _ = _gpu_vs_oracle(x0, ia, ib, meas, sq, 120)
    assert rg["n_separator_poses"] == 4 and rg["final_cost"] < rg["initial_cost"] * 1e-3
    # tight tolerances: more iterations, rejected steps included
    kw = dict(function_tolerance=1e-13, parameter_tolerance=1e-12, max_iterations=60)
    _gpu_vs_oracle(x0, ia, ib, meas, sq, 120, po.pg_params(**kw), host.pg_params(**kw))
    # a run longer than the 128-pose segment cap: extra separators cut it, the step stays the exact Cholesky step
    truth, x0, ia, ib, meas, sq = make_graph(300, [(4, 290), (150, 20)], seed=12, laps=1.0, radius=25.0)
    xg, rg, _, _ = _gpu_vs_oracle(x0, ia, ib, meas, sq, 300)
    assert rg["n_separator_poses"] > 4
    # loop closures with the indoor weight (4e4 * I, parameters_indoor.yaml:10): ill-scaled normal equations
    truth, x0, ia, ib, meas, sq = make_graph(90, [(0, 89), (3, 80), (10, 50)], seed=11, loop_weight=4.0e4)
    _gpu_vs_oracle(x0, ia, ib, meas, sq, 90, tol=1e-6)


@pytest.mark.gpu
def test_hip_pose_graph_structure_cases(built):
    # no loop closures: pure block-tridiagonal path (no Schur complement)
    truth, x0, ia, ib, meas, sq = make_graph(64, [], seed=2)
    x0 = x0 + np.random.default_rng(3).normal(size=x0.shape) * 0.05
    xg, rg, _, _ = _gpu_vs_oracle(x0, ia, ib, meas, sq, 64)
    assert rg["n_separator_poses"] == 0
    # every variable pose is a separator (n_int = 0), an edge to the constant pose 0, adjacent separators,
    # duplicate odometry edges, a reversed chain edge
    rng = np.random.default_rng(8)
    truth, x0, ia, ib, meas, sq = make_graph(6, [(0, 2), (1, 3), (2, 4), (3, 5), (1, 4), (2, 5), (1, 5)], seed=9)
    _gpu_vs_oracle(x0, ia, ib, meas, sq, 6)
    truth, x0, ia, ib, meas, sq = make_graph(40, [(0, 20), (7, 8), (12, 11), (30, 5), (29, 31), (30, 32)], seed=10)
    xg, rg, _, _ = _gpu_vs_oracle(x0, ia, ib, meas, sq, 40)
    # max_update_index rule and full (non-diagonal) sqrt-information blocks
    sq2 = sq + rng.normal(size=sq.shape) * 0.5
    _gpu_vs_oracle(x0, ia, ib, meas, sq2, 25)
    # Huber loss with a false loop closure
    meas2 = meas.copy()
    meas2[-1] += [2.0, -1.0, 0.3]
    kw = dict(use_robust_loss=1, loss_scale=2.0)
    _gpu_vs_oracle(x0, ia, ib, meas2, sq, 40, po.pg_params(**kw), host.pg_params(**kw))
    # poses that no used edge references stay untouched
    x_pad = np.vstack([x0, [[5.0, 5.0, 0.1]]])
    xg, _, _, _ = _gpu_vs_oracle(x_pad, ia, ib, meas, sq, 40)
    assert np.array_equal(xg[-1], x_pad[-1]) and np.array_equal(xg[0], x_pad[0])


@pytest.mark.gpu
def test_hip_pose_graph_full_size_properties(built):
    """Oxford-sized graph (2200 keyframe nodes ~ 8800 scans / insertion_step 4, 60 loop closures): too large for the dense
    oracle, so check what any exact LM step must satisfy -- known answer, fixed first pose, monotone cost."""
    import torch

    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    n = 2200
    rng = np.random.default_rng(21)
    loops = [(int(a), int(a) + 1100 + int(o)) for a, o in zip(rng.integers(0, 1000, 60), rng.integers(-40, 40, 60))]
    truth, _, ia, ib, meas, sq = make_graph(n, loops, seed=22, noise=(0, 0, 0), laps=2.0, radius=60.0)
    x0 = [truth[0].copy()]
    for i in range(n - 1):
        x0.append(compose(x0[-1], meas[i] + rng.normal(size=3) * [0.01, 0.01, 0.001]))   # drifting dead reckoning
    x0 = np.array(x0)
    kw = dict(function_tolerance=1e-16, parameter_tolerance=1e-13, gradient_tolerance=1e-13, max_iterations=100)
    xg, rg = host.pose_graph_optimize(ctx, x0, ia, ib, meas, sq, n, host.pg_params(**kw))
    assert rg["n_residual_blocks"] == n - 1 + 60 and rg["n_separator_poses"] > 60
    assert np.array_equal(xg[0], x0[0])
    assert np.abs(xg - truth).max() < 1e-6 and rg["final_cost"] < 1e-12 * rg["initial_cost"]
    # noisy version: final cost below the initial one, and a second call from the optimum stays put
    truth, x0, ia, ib, meas, sq = make_graph(n, loops, seed=23, laps=2.0, radius=60.0)
    x1, r1 = host.pose_graph_optimize(ctx, x0, ia, ib, meas, sq, n)
    assert r1["final_cost"] < 0.05 * r1["initial_cost"] and r1["termination"] in (1, 2, 3)
    assert abs(graph_cost(x1, ia, ib, meas, sq) - r1["final_cost"]) < 1e-9 * r1["final_cost"]
    x2, r2 = host.pose_graph_optimize(ctx, x1, ia, ib, meas, sq, n)
    assert np.abs(x2 - x1).max() < 1e-2 and r2["final_cost"] <= r1["final_cost"] * (1 + 1e-12)


@pytest.mark.gpu
def test_hip_pose_graph_error_convention(built):
    import torch

    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    with pytest.raises(R.RandtError):
        host.pose_graph_optimize(ctx, np.zeros((3, 3)), [0, 1], [1, 7], np.zeros((2, 3)), np.stack([np.eye(3)] * 2), 9)
    # nothing to optimise: poses returned unchanged, zero residual blocks
    x, r = host.pose_graph_optimize(ctx, np.ones((3, 3)), [], [], np.zeros((0, 3)), np.zeros((0, 9)), 9)
    assert np.array_equal(x, np.ones((3, 3))) and r["n_residual_blocks"] == 0


# ------------------------------------------------------------------------------------------------- golden vector
def _golden():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posegraph_01.npz"))


def test_oracle_reaches_the_independently_computed_optimum(built):
    """tests/golden/posegraph_01.npz: minimiser from scipy on residuals written out in the generator script, not by the oracle."""
    g = _golden()
    tight = po.pg_params(function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-14, max_iterations=200)
    x, res = po.pose_graph_optimize(g["x0"], g["id_begin"], g["id_end"], g["meas"], g["sqrt_info"], len(g["x0"]), tight)
    assert abs(res["initial_cost"] - float(g["cost_init"])) < 1e-9 * float(g["cost_init"])
    assert abs(res["final_cost"] - float(g["cost_opt"])) < 1e-10 * max(1.0, float(g["cost_opt"]))
    assert np.abs(x - g["x_opt"]).max() < 1e-7


@pytest.mark.gpu
def test_hip_reaches_the_independently_computed_optimum(built):
    import torch

    g = _golden()
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    kw = dict(function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-14, max_iterations=200)
    x, res = host.pose_graph_optimize(ctx, g["x0"], g["id_begin"], g["id_end"], g["meas"], g["sqrt_info"], len(g["x0"]), host.pg_params(**kw))
    assert abs(res["final_cost"] - float(g["cost_opt"])) < 1e-10 * max(1.0, float(g["cost_opt"]))
    assert np.abs(x - g["x_opt"]).max() < 1e-7
    # reference options: stops on the function tolerance, within the path's 1e-4 bar of the true optimum... of the COST;
    # poses are within a millimetre
    xd, rd = host.pose_graph_optimize(ctx, g["x0"], g["id_begin"], g["id_end"], g["meas"], g["sqrt_info"], len(g["x0"]))
    assert rd["termination"] == 1 and np.abs(xd - g["x_opt"]).max() < 2e-3


@pytest.mark.gpu
def test_hip_pose_graph_beyond_2048_loop_closure_poses(built):
    """2600 poses carry loop closures (the separator limit was 2048 until round 4; the dense oracle would need hours): noise-free
    measurements make the truth a known answer, the first pose stays fixed, and the Schur complement is 7800 x 7800."""
    import torch

    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    n = 5200
    loops = [(2 * i, 2 * i + 2600) for i in range(1300)]                       # 2600 distinct poses
    truth, _, ia, ib, meas, sq = make_graph(n, loops, seed=31, noise=(0, 0, 0), laps=2.0, radius=80.0)
    rng = np.random.default_rng(32)
    x0 = [truth[0].copy()]
    for i in range(n - 1):
        x0.append(compose(x0[-1], meas[i] + rng.normal(size=3) * [0.004, 0.004, 0.0004]))   # drifting dead reckoning
    x0 = np.array(x0)
    kw = dict(function_tolerance=1e-16, parameter_tolerance=1e-13, gradient_tolerance=1e-13, max_iterations=60)
    xg, rg = host.pose_graph_optimize(ctx, x0, ia, ib, meas, sq, n, host.pg_params(**kw))
    assert rg["n_separator_poses"] > 2500 and rg["n_residual_blocks"] == n - 1 + 1300        # (pose 0 is constant, not a separator)
    assert np.array_equal(xg[0], x0[0])
    assert np.abs(xg - truth).max() < 1e-6 and rg["final_cost"] < 1e-12 * rg["initial_cost"]
