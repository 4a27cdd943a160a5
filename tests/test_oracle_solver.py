"""K1 / K3 / K7: known-answer registration, independent-solver cross-check (scipy) and GNC
schedules for the oracle's Ceres-faithful LM restatement."""
import numpy as np
import pytest
from scipy.optimize import least_squares

import pyoracle as po
from randt_slam_amd import synth
from util import IP, oracle_scan_map, oracle_submap, problem

F = np.float32


def synthetic_pair(rng, n=60, pose3=(0.4, -0.3, 0.25), noise=0.0):
    """fixed cells random; moving cells = T^-1 applied to the fixed cells (cell-level, float64 then
    cast), so the exact minimiser of the D2D objective with identity correspondences is T."""
    fixed = po.Map(100, 100, 0.5, (0, 0), 4.0, 5, n)
    moving = po.Map(100, 100, 0.5, (0, 0), 4.0, 5, n)
    fc = np.zeros(n, dtype=po.CELL_DTYPE)
    mc = np.zeros(n, dtype=po.CELL_DTYPE)
    th = pose3[2]
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    t = np.array([pose3[0], pose3[1], 0.0])
    for i in range(n):
        mean = np.array([rng.uniform(-10, 10), rng.uniform(-10, 10), rng.uniform(20, 80)])
        A = rng.normal(0, 1, (3, 3)) * [0.1, 0.1, 2.0]
        S = A @ A.T + np.diag([1e-3, 1e-3, 1e-1])
        fc[i]["mean"] = mean; fc[i]["cov"] = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]; fc[i]["n"] = 10
        mm = R.T @ (mean - t) + rng.normal(0, noise, 3) * [1, 1, 0]
        Sm = R.T @ S @ R
        mc[i]["mean"] = mm; mc[i]["cov"] = [Sm[0, 0], Sm[0, 1], Sm[0, 2], Sm[1, 1], Sm[1, 2], Sm[2, 2]]; mc[i]["n"] = 10
    g = np.full(10000, -1, dtype=np.int32)
    fixed.set(fc, g); moving.set(mc, g)
    corr = np.arange(n, dtype=np.int32).reshape(n, 1)
    return fixed, moving, corr


@pytest.mark.parametrize("param", [po.PARAM_MANIFOLD, po.PARAM_AMBIENT4, po.PARAM_VECTOR])
def test_known_answer_zero_noise(built, param):
    rng = np.random.default_rng(1)
    truth = (0.4, -0.3, 0.25)
    fixed, moving, corr = synthetic_pair(rng, pose3=truth)
    # the moving covariances make the exact D2D residual 2*... not zero-able?  residual d = R m + t - f = 0 at truth.
    guess = synth.pose3_to_pose4(np.array([0.1, 0.0, 0.15]))
    prm = po.default_params(parameterization=param, n_neighbours=1)
    rc, p4, st = po.solve_pair(fixed, moving, corr, prm, guess)
    est = synth.pose4_to_pose3(p4)
    assert rc == 0
    assert np.allclose(est, truth, atol=5e-6), est    # fp32 cell storage limits this, not the solver
    assert st["final_cost"] < 1e-6 * st["initial_cost"]


def objective_factory(fixed, moving, corr, a, alpha, mu):
    fcs, mcs = fixed.cells(), moving.cells()
    def full(c):
        return np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]], dtype=np.float64)
    data = [(mcs[i]["mean"].astype(float), full(mcs[i]["cov"]), fcs[j]["mean"].astype(float), full(fcs[j]["cov"]))
            for i in range(len(mcs)) for j in corr[i] if j >= 0]
    def resid(x):
        c, s = np.cos(x[2]), np.sin(x[2])
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]); t = np.array([x[0], x[1], 0])
        out = []
        for mm, mc, fm, fc in data:
            d = R @ mm + t - fm
            out.append(np.sqrt(d @ np.linalg.solve(R @ mc @ R.T + fc, d)))
        return np.array(out)
    def loss(z):
        rho = np.array([po.barron_scaled(s, a, alpha, mu, 1.0) for s in z])
        return rho.T
    return resid, loss


@pytest.mark.parametrize("alpha", [-2.0, -1.0])
def test_minimiser_matches_scipy(built, alpha):
    """Same fixed-correspondence robust objective, independent solver: with the stopping tolerances
    tightened, the oracle's LM and scipy's TRF land on the same (theta, tx, ty)."""
    rng = np.random.default_rng(2)
    fixed, moving, corr = synthetic_pair(rng, n=40, noise=0.05)
    guess3 = np.array([0.3, -0.2, 0.2])
    prm = po.default_params(parameterization=po.PARAM_MANIFOLD, n_neighbours=1, gnc_steps=1, loss_alpha=alpha,
                            function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-14)
    rc, p4, st = po.solve_pair(fixed, moving, corr, prm, synth.pose3_to_pose4(guess3))
    resid, loss = objective_factory(fixed, moving, corr, prm.loss_scale, alpha, 1.0)
    sol = least_squares(resid, guess3, loss=loss, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=500)
    est = synth.pose4_to_pose3(p4)
    assert np.allclose(est, sol.x, atol=2e-7), (est, sol.x)
    # and the costs agree (Ceres: 1/2 sum rho)
    z = resid(sol.x) ** 2
    assert np.isclose(st["trace_cost"][st["trace_flag"] < 2].min(), 0.5 * loss(z)[0].sum(), rtol=1e-9)


def test_qr_and_normal_equations_agree(built):
    prob = problem()
    sub = oracle_submap(prob["submaps"][0])
    scan = oracle_scan_map(prob["scans"][0])
    g = synth.pose3_to_pose4(prob["guess"][0])
    outs = []
    for ls in (po.LINSOLVE_QR, po.LINSOLVE_NORMAL):
        rc, p4, cost, st = po.register_pair(sub, scan, po.default_params(linear_solver=ls), g)
        outs.append((p4, st))
    assert np.allclose(outs[0][0], outs[1][0], atol=1e-9)
    assert outs[0][1]["n_iterations"] == outs[1][1]["n_iterations"]
    assert np.allclose(outs[0][1]["trace_cost"], outs[1][1]["trace_cost"], rtol=1e-9)


def test_gnc_schedules(built):
    """A.4: number of solves for the shipped parameter sets when the initial max residual is large."""
    prob = problem()
    sub = oracle_submap(prob["submaps"][0])
    scan = oracle_scan_map(prob["scans"][0])
    g = synth.pose3_to_pose4(prob["guess"][0])
    cases = [
        (dict(gnc_steps=3, gnc_divisor=1.3), 3, 1.3 ** 2),    # indoor odometry {1.69, 1.3, 1.0}
        (dict(gnc_steps=2, gnc_divisor=1.3), 2, 1.3),         # indoor loop {1.3, 1.0}
        (dict(gnc_steps=2, gnc_divisor=1.1), 2, 1.1),         # oxford odometry {1.1, 1.0}
        (dict(gnc_steps=10, gnc_divisor=1.1), 10, 1.1 ** 9),  # oxford loop: 10 solves
        (dict(gnc_steps=1, gnc_divisor=1.3), 1, 1.0),
    ]
    for over, n_solves, mu0 in cases:
        rc, p4, cost, st = po.register_pair(sub, scan, po.default_params(**over), g)
        assert st["n_solves"] == n_solves
        assert np.isclose(st["mu0"], mu0)
        assert (st["trace_flag"] == 0).sum() == n_solves     # each Solve restarts at iteration 0
    # small residuals -> a single solve at mu = 1
    rc, p4, cost, st = po.register_pair(sub, scan, po.default_params(mu_scale=1e3, gnc_steps=3), g)
    assert st["n_solves"] == 1 and st["mu0"] < 1.0


def test_lm_trace_invariants(built):
    prob = problem()
    sub = oracle_submap(prob["submaps"][1])
    scan = oracle_scan_map(prob["scans"][9])
    g = synth.pose3_to_pose4(prob["guess"][9])
    rc, p4, cost, st = po.register_pair(sub, scan, po.default_params(parameterization=po.PARAM_MANIFOLD), g)
    tc, tr, tf = st["trace_cost"], st["trace_radius"], st["trace_flag"]
    assert tf[0] == 0 and tr[0] == 1e4                        # initial_trust_region_radius
    for i in range(1, len(tf)):
        if tf[i] == 0:
            assert tr[i] == 1e4                               # fresh minimiser per GNC solve
        elif tf[i] == 1:
            last = [j for j in range(i) if tf[j] in (0, 1)][-1]
            assert tc[i] < tc[last]                           # monotonic accepted steps
            assert tr[i] <= 3 * tr[i - 1] * (1 + 1e-12)       # radius / max(1/3, ...)
        elif tf[i] == 2:
            assert tr[i] < tr[i - 1]
    assert st["termination"] in (1, 2, 3)
    assert np.isclose(np.hypot(p4[0], p4[1]), 1.0)            # manifold keeps the complex unit
    # the registration is right: a few cm from the ground truth
    est = synth.pose4_to_pose3(p4)
    assert np.all(np.abs(est[:2] - prob["truth"][9][:2]) < 0.1) and abs(est[2] - prob["truth"][9][2]) < 0.03


def test_no_residuals_leaves_pose_unchanged(built):
    sub = po.Map(100, 100, 0.5, (0, 0), 4.0, 5, 16)
    scan = po.Map(100, 100, 0.5, (0, 0), 4.0, 5, 16)
    g = np.array([0.9, 0.1, 1.0, 2.0])
    rc, p4, cost, st = po.register_pair(sub, scan, po.default_params(), g)
    assert st["n_residuals"] == 0 and np.array_equal(p4, g) and cost == 0.0
