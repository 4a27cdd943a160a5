"""K2 / K6: finite-difference check of the tangent-space Jacobian for every parameterisation,
closed forms of the Barron loss, and the Sophus SE(2) restatement."""
import numpy as np
import pytest

import pyoracle as po


def rand_spd(rng, d, scale):
    A = rng.normal(0, 1, (d, d)) * scale
    return A @ A.T + np.diag(np.full(d, 1e-3))


def num_jac(d, param, pose4, mm, mc, fm, fc, eps=1e-6):
    def r_at(p4):
        return po.ndt_residual(d, param, p4, mm, mc, fm, fc, want_jac=False)[0]
    J = []
    if param == po.PARAM_MANIFOLD:
        for i in range(3):
            e = np.zeros(3); e[i] = eps
            J.append((r_at(po.se2_mul(pose4, po.se2_exp(e))) - r_at(po.se2_mul(pose4, po.se2_exp(-e)))) / (2 * eps))
    elif param == po.PARAM_AMBIENT4:
        for i in range(4):
            e = np.zeros(4); e[i] = eps
            J.append((r_at(pose4 + e) - r_at(pose4 - e)) / (2 * eps))
    else:   # (pos, rot) blocks: PARAM_VECTOR (and PARAM_ANALYTIC, whose RESIDUAL is the same)
        th = np.arctan2(pose4[1], pose4[0])
        def r3(x):
            return r_at(np.array([np.cos(x[2]), np.sin(x[2]), x[0], x[1]]))
        x = np.array([pose4[2], pose4[3], th])
        for i in range(3):
            e = np.zeros(3); e[i] = eps
            J.append((r3(x + e) - r3(x - e)) / (2 * eps))
    return np.array(J)


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("param", [po.PARAM_MANIFOLD, po.PARAM_AMBIENT4, po.PARAM_VECTOR])
def test_jacobian_matches_finite_differences(built, d, param):
    rng = np.random.default_rng(10 * d + param)
    for theta in [0.0, 0.7, -2.5, 3.0]:        # far from 0 on purpose (the reference's analytic functors fail there)
        mm, fm = rng.normal(0, 2, d), rng.normal(0, 2, d)
        mc, fc = rand_spd(rng, d, 0.3), rand_spd(rng, d, 0.3)
        scale = 1.0 if param != po.PARAM_AMBIENT4 else 1.3   # non-unit complex in the ambient case
        pose4 = np.array([scale * np.cos(theta), scale * np.sin(theta), rng.normal(), rng.normal()])
        r, J = po.ndt_residual(d, param, pose4, mm, mc, fm, fc)
        Jn = num_jac(d, param, pose4, mm, mc, fm, fc)
        assert np.allclose(J, Jn, rtol=1e-5, atol=1e-7), (theta, J, Jn)
        # residual itself vs plain numpy
        th = np.arctan2(pose4[1], pose4[0])
        R = np.eye(d); R[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
        t = np.zeros(d); t[:2] = pose4[2:]
        dv = R @ mm + t - fm
        assert np.isclose(r, np.sqrt(dv @ np.linalg.solve(R @ mc @ R.T + fc, dv)), rtol=1e-12)


def test_zero_residual_guard(built):
    mm = np.array([1.0, 2.0, 30.0]); S = np.diag([.1, .2, 3.0])
    r, J = po.ndt_residual(3, po.PARAM_MANIFOLD, np.array([1.0, 0, 0, 0]), mm, S, mm, S)
    assert r == 0.0 and np.all(J == 0.0)


@pytest.mark.parametrize("alpha", [-2.0, -1.5, -1.0, 0.0, 1.0, 2.0])
def test_barron_loss_closed_forms(built, alpha):
    a, mu, w = 1.5, 1.3, 0.7
    b = mu * a * a
    for s in [0.0, 0.1, 1.0, 7.5, 120.0]:
        rho = po.barron_scaled(s, a, alpha, mu, w)
        if alpha >= 2:
            ref = [s, 1.0, 0.0]
        elif abs(alpha) <= 0.05:
            ref = [b * np.log1p(s / b), 1 / (1 + s / b), -(1 / b) / (1 + s / b) ** 2]
        else:
            f = abs(alpha - 2); u = s * 2 / (b * f) + 1; e = alpha / 2; pre = b * f / alpha; ts = 2 / (b * f)
            ref = [pre * (u ** e - 1), pre * e * u ** (e - 1) * ts, pre * e * (e - 1) * u ** (e - 2) * ts * ts]
        assert np.allclose(rho, np.array(ref) * w, rtol=1e-12, atol=1e-15)
        # rho' really is the derivative of rho
        h = 1e-6 * max(1.0, s)
        d = (po.barron_scaled(s + h, a, alpha, mu, w)[0] - po.barron_scaled(max(s - h, 0), a, alpha, mu, w)[0]) / (h + min(h, s))
        assert np.isclose(d, rho[1], rtol=1e-4, atol=1e-9)
    if alpha == -2.0:  # SURVEY A.3 closed form
        s = 3.0
        rho = po.barron_scaled(s, a, alpha, mu, 1.0)
        assert np.isclose(rho[0], s / (1 + s / (2 * b))) and np.isclose(rho[1], (1 + s / (2 * b)) ** -2)
        assert rho[2] < 0  # => Ceres applies plain sqrt(rho') re-weighting


def test_se2_restatement(built):
    rng = np.random.default_rng(0)
    def mat(p):
        return np.array([[p[0], -p[1], p[2]], [p[1], p[0], p[3]], [0, 0, 1]])
    for _ in range(20):
        xi = rng.normal(0, 1, 3)
        p = po.se2_exp(xi)
        assert np.isclose(p[0] ** 2 + p[1] ** 2, 1.0)
        assert np.allclose(po.se2_log(p), xi, atol=1e-12)
        q = po.se2_exp(rng.normal(0, 1, 3))
        assert np.allclose(mat(po.se2_mul(p, q)), mat(p) @ mat(q), atol=1e-12)
        assert np.allclose(mat(po.se2_mul(p, po.se2_inv(p))), np.eye(3), atol=1e-12)
    # small-angle branches
    assert np.allclose(po.se2_exp([0.3, -0.2, 1e-12]), [1, 1e-12, 0.3, -0.2], atol=1e-11)
    assert np.allclose(po.se2_log(po.se2_exp([0.3, -0.2, 1e-12])), [0.3, -0.2, 1e-12], atol=1e-11)
    # scipy cross-check of exp
    from scipy.linalg import expm
    xi = np.array([0.4, -1.1, 0.9])
    G = np.array([[0, -xi[2], xi[0]], [xi[2], 0, xi[1]], [0, 0, 0]])
    assert np.allclose(mat(po.se2_exp(xi)), expm(G), atol=1e-12)


@pytest.mark.parametrize("d", [2, 3])
def test_analytic_functor_jacobian_is_the_reference_formula_not_the_derivative(built, d):
    """`use_analytic_expressions_for_optimization: true` selects NDTFrameToMap{,Intensity}FactorResidualAnalytic
    (ceres_residuals.h:207-305).  PARAM_ANALYTIC reproduces its Jacobian AS WRITTEN -- re-derived here with numpy matrices,
    line by line from the functor's Evaluate() -- which equals the true derivative at theta = 0 and not elsewhere
    (the second rotation term uses R Sm (R J) where -(R J) Sm R^T is required)."""
    rng = np.random.default_rng(77 + d)
    for theta in [0.0, 0.7, -2.5, 3.0]:
        mm, fm = rng.normal(0, 2, d), rng.normal(0, 2, d)
        mc, fc = rand_spd(rng, d, 0.3), rand_spd(rng, d, 0.3)
        t2 = rng.normal(0, 1, 2)
        pose4 = np.array([np.cos(theta), np.sin(theta), t2[0], t2[1]])
        r, J = po.ndt_residual(d, po.PARAM_ANALYTIC, pose4, mm, mc, fm, fc)
        # the functor, transcribed: rotation, rotation_derivative = rotation * [[0,-1],[1,0]] (embedded), mean_diff, cov_inv
        rot = np.eye(d); rot[:2, :2] = [[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]]
        G = np.zeros((d, d)); G[0, 1], G[1, 0] = -1.0, 1.0
        rot_d = rot @ G
        trans = np.zeros(d); trans[:2] = t2
        md = rot @ mm + trans - fm
        ci = np.linalg.inv(rot @ mc @ rot.T + fc)
        res = np.sqrt(md @ ci @ md)
        ex, ey = np.zeros(d), np.zeros(d); ex[0], ey[1] = 1.0, 1.0
        Jref = np.array([md @ ci @ ex, md @ ci @ ey, md @ ci @ rot_d @ mm + md @ ci @ rot @ mc @ rot_d @ ci @ md]) / res
        assert np.isclose(r, res, rtol=1e-12) and np.allclose(J, Jref, rtol=1e-10, atol=1e-12), (theta, J, Jref)
        Jfd = num_jac(d, po.PARAM_VECTOR, pose4, mm, mc, fm, fc)
        assert np.allclose(J[:2], Jfd[:2], rtol=1e-5, atol=1e-7)                 # the translation columns are right
        if theta == 0.0:
            assert np.isclose(J[2], Jfd[2], rtol=1e-5, atol=1e-7)                # the rotation column only at theta = 0
        else:
            assert abs(J[2] - Jfd[2]) > 1e-3 * max(1.0, abs(Jfd[2]))
