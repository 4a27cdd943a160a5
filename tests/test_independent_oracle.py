"""The CPU oracle against the oracle-free fixture (tests/golden/independent_01.npz: numpy / scipy known answers, see
tests/golden/make_independent.py) -- the same three checks tests/test_gpu_independent.py applies to the HIP path, so
that oracle and kernels are each pinned to software neither of them shares code with."""
import os

import numpy as np

import pyoracle as po
from randt_slam_amd import synth
from util import IP

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "independent_01.npz")


def _maps(fixed, moving):
    n = len(fixed)
    f = po.Map(100, 100, 0.5, (0, 0), 4.0, 5, n)
    m = po.Map(100, 100, 0.5, (0, 0), 4.0, 5, n)
    f.set(fixed.astype(po.CELL_DTYPE), np.full(100 * 100, -1, dtype=np.int32))
    m.set(moving.astype(po.CELL_DTYPE), np.full(100 * 100, -1, dtype=np.int32))
    return f, m


def test_oracle_k1_and_k3(built):
    d = np.load(FIX)
    for i in range(len(d["k1_truth"])):
        f, m = _maps(d["k1_fixed"][i], d["k1_moving"][i])
        prm = po.default_params(parameterization=po.PARAM_MANIFOLD, n_neighbours=1, gnc_steps=2, function_tolerance=1e-14, parameter_tolerance=1e-13)
        rc, p4, st = po.solve_pair(f, m, d["k1_corr"][i], prm, synth.pose3_to_pose4(d["k1_guess"][i]))
        err = synth.pose4_to_pose3(p4) - d["k1_truth"][i]
        err[2] = (err[2] + np.pi) % (2 * np.pi) - np.pi
        assert np.abs(err[:2]).max() < 2e-5 and abs(err[2]) < 2e-6
    for i, alpha in enumerate(d["k3_alpha"]):
        f, m = _maps(d["k3_fixed"][i], d["k3_moving"][i])
        prm = po.default_params(parameterization=po.PARAM_MANIFOLD, n_neighbours=1, gnc_steps=1, loss_alpha=float(alpha), loss_scale=1.5, mu_scale=1.5,
                                use_intensity=int(d["k3_dim"][i] == 3), function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-14)
        rc, p4, st = po.solve_pair(f, m, d["k3_corr"][i], prm, synth.pose3_to_pose4(d["k3_guess"]))
        assert np.allclose(synth.pose4_to_pose3(p4), d["k3_solution"][i], atol=5e-7)
        assert np.isclose(st["final_cost"], d["k3_cost"][i], rtol=1e-8)


def test_oracle_k4(built):
    d = np.load(FIX)
    for s in range(len(d["k4_scans"])):
        m = po.Map(IP["size_x"], IP["size_y"], IP["resolution"], (0.0, 0.0), IP["max_neighbour_dist"], IP["min_points_per_cell"], 512)
        m.build(d["k4_scans"][s], IP["n_clusters"], IP["max_range"])
        n = int(d["k4_n_cells"][s])
        cells, ref = m.cells(), d["k4_cells"][s][:n]
        assert m.n_cells == n and np.array_equal(m.grid(), d["k4_grid"][s])
        assert np.array_equal(cells["mean"].view(np.uint32), ref["mean"].view(np.uint32))
        assert np.array_equal(cells["n"], ref["n"])
        for e in (2, 4, 5):
            assert np.array_equal(cells["cov"][:, e].view(np.uint32), ref["cov"][:, e].view(np.uint32))
