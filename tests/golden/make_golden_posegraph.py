#!/usr/bin/env python3
"""Generates tests/golden/posegraph_01.npz: a 90-node pose graph (odometry chain + 5 loop closures, one of them with a
full sqrt-information block) together with its minimiser computed by an INDEPENDENT solver -- scipy's trust-region
least squares on residuals written here from the reference's formula (pose_graph_2d_error_term.h:44-60), not by the
oracle.  Pins the oracle's and the HIP path's optimum from outside.  Run from the repo root:
python tests/golden/make_golden_posegraph.py"""
import os

import numpy as np
from scipy.optimize import least_squares

N = 90
LOOPS = [(0, 89), (4, 80), (0, 45), (20, 70), (33, 61)]


def rel(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    d = b[:2] - a[:2]
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], (b[2] - a[2] + np.pi) % (2 * np.pi) - np.pi])


def residual(xa, xb, m, sq):
    e = rel(xa, xb) - m
    e[2] = e[2] - 2 * np.pi * np.floor((e[2] + np.pi) / (2 * np.pi))      # NormalizeAngle (state_manifold.h:17-23)
    return sq @ e


def main():
    rng = np.random.default_rng(2024)
    th = np.linspace(0, 2 * np.pi, N, endpoint=False)
    truth = np.stack([12 * np.cos(th), 8 * np.sin(th), np.arctan2(8 * np.cos(th), -12 * np.sin(th))], 1)
    ia, ib, meas, sq = [], [], [], []
    for i in range(N - 1):
        ia.append(i); ib.append(i + 1)
        meas.append(rel(truth[i], truth[i + 1]) + rng.normal(size=3) * [0.03, 0.03, 0.006])
        sq.append(np.diag([10.0, 10.0, 50.0]))
    for k, (a, b) in enumerate(LOOPS):
        ia.append(a); ib.append(b)
        meas.append(rel(truth[a], truth[b]) + rng.normal(size=3) * [0.01, 0.01, 0.002])
        sq.append(np.eye(3) * 40.0 if k else np.array([[40.0, 3.0, 0.0], [1.0, 35.0, 2.0], [0.0, 4.0, 60.0]]))
    x0 = [truth[0].copy()]
    for i in range(N - 1):
        a, m = x0[-1], meas[i]
        c, s = np.cos(a[2]), np.sin(a[2])
        x0.append(np.array([a[0] + c * m[0] - s * m[1], a[1] + s * m[0] + c * m[1], a[2] + m[2]]))
    x0 = np.array(x0)
    meas, sq = np.array(meas), np.array(sq)

    def fun(v):
        X = np.vstack([x0[:1], v.reshape(-1, 3)])
        return np.concatenate([residual(X[a], X[b], m, s) for a, b, m, s in zip(ia, ib, meas, sq)])

    sol = least_squares(fun, x0[1:].ravel(), xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
    x_opt = np.vstack([x0[:1], sol.x.reshape(-1, 3)])
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "posegraph_01.npz")
    np.savez_compressed(out, x0=x0, id_begin=np.array(ia, np.int32), id_end=np.array(ib, np.int32), meas=meas, sqrt_info=sq,
                        x_opt=x_opt, cost_opt=np.array(sol.cost), cost_init=np.array(0.5 * np.sum(fun(x0[1:].ravel()) ** 2)))
    print(out, os.path.getsize(out), "bytes; cost", 0.5 * np.sum(fun(x0[1:].ravel()) ** 2), "->", sol.cost, "grad", np.abs(sol.grad).max())


if __name__ == "__main__":
    main()
