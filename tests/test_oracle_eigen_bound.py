"""What the one thing this image cannot pin would cost if it were wrong (round-3 verdict, item 7).

`Cell::updateCell` regularises the xy covariance through `Eigen::SelfAdjointEigenSolver<Matrix2f>` (iterative;
ndt_cell.cpp:102-112); the oracle -- and the HIP kernels, bit for bit -- use a closed-form 2 x 2 symmetric
eigen-decomposition instead (oracle/randt_oracle.c, SPEC DECISION 1).  No Eigen exists here to compare with, so the
difference is BOUNDED instead: every regularised xy covariance entry (xx, xy, yy) of every cell -- the scan cells of all
512 registrations of BASELINE config 4 AND the cells of the eight submaps -- is moved by +-1 and +-2 float32 ulp (the most
an iterative 2 x 2 solve can plausibly differ by: its result is backward-stable to a few ulp of the largest entry), in
six patterns (all +1, all -1, all +2, all -2, two seeded random mixes of {-2..2}), and the whole registration is redone
on the oracle.  Asserted: how many entries of the frozen correspondence tables flip (a near-tie of two Mahalanobis
distances in the top-k, the sharpest known consequence) and how far any pose moves.

The statement this test makes: IF Eigen's eigen-solver differs from the closed form by <= 2 ulp per covariance entry, THEN
the registration poses differ by less than 1e-5 m / 1e-5 rad (north_star tolerance: 1e-4; observed: 2.0e-6, one registration
whose LM path runs along a flat valley) -- and a correspondence flips in at most 16 of the 154 844 table entries (observed: 2,
in one registration, moving its pose by 1.9e-9).
"""
import numpy as np

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import synth
from util import oracle_scan_map, oracle_submap, to_oracle_params

N_REG = 512            # the whole config-4 batch
XY_ENTRIES = (0, 1, 3)  # cov = (xx, xy, xi, yy, yi, ii)


def _ulp_shift(a, steps):
    """float32 array moved by `steps` (integer array, may be negative) representable values."""
    out = a.astype(np.float32).copy()
    up = np.float32(np.inf)
    for s in (1, 2):
        m = np.abs(steps) >= s
        out[m] = np.nextafter(out[m], np.where(steps[m] > 0, up, -up).astype(np.float32))
    return out


def _perturbed(m, pattern, rng):
    cells = m.cells()
    cov = cells["cov"].copy()
    for e in XY_ENTRIES:
        steps = np.full(len(cells), pattern, dtype=np.int64) if pattern is not None else rng.integers(-2, 3, size=len(cells))
        cov[:, e] = _ulp_shift(cov[:, e], steps)
    cells["cov"] = cov
    out = m.copy()
    out.set(cells, m.grid())
    return out


def test_two_ulp_of_the_regularised_covariance_bound_the_pose():
    prob = synth.make_batch_problem(8, 64, 34)                 # bench.py's batch (BASELINE config 4)
    mp = R.default_matcher_params()
    op = to_oracle_params(mp)
    g4 = synth.pose3_to_pose4(prob["guess"])
    subs = [oracle_submap(sm) for sm in prob["submaps"]]
    scans = [oracle_scan_map(prob["scans"][i]) for i in range(N_REG)]
    k = mp.n_neighbours

    def run(fixed, moving):
        corr, pose = [], np.zeros((N_REG, 4))
        for i in range(N_REG):
            f = fixed[prob["submap_of"][i]]
            c, _ = po.associate(f, moving[i], g4[i], k, bool(mp.lookup_mahalanobis), bool(mp.use_intensity))
            rc, p4, st = po.solve_pair(f, moving[i], c, op, g4[i])
            assert rc == 0
            corr.append(np.asarray(c).copy())
            pose[i] = p4
        return corr, pose

    ref_corr, ref_pose = run(subs, scans)
    n_entries = sum(c.size for c in ref_corr)
    rng = np.random.default_rng(20260930)
    worst = {"flips": 0, "flipped_regs": 0, "dpose_same_corr": 0.0, "dpose_flipped": 0.0}
    for pattern in (1, -1, 2, -2, None, None):
        fixed = [_perturbed(m, pattern, rng) for m in subs]
        moving = [_perturbed(m, pattern, rng) for m in scans]
        corr, pose = run(fixed, moving)
        flips, flipped_regs = 0, 0
        for i in range(N_REG):
            d = int((corr[i] != ref_corr[i]).sum())
            flips += d
            flipped_regs += d > 0
            # ambient-4 blocks: compare the pose as (angle, translation)
            dth = np.arctan2(pose[i, 1], pose[i, 0]) - np.arctan2(ref_pose[i, 1], ref_pose[i, 0])
            dp = max(abs((dth + np.pi) % (2 * np.pi) - np.pi), np.abs(pose[i, 2:] - ref_pose[i, 2:]).max())
            key = "dpose_flipped" if d else "dpose_same_corr"
            worst[key] = max(worst[key], float(dp))
        worst["flips"] = max(worst["flips"], flips)
        worst["flipped_regs"] = max(worst["flipped_regs"], flipped_regs)
    print("eigen-solver bound over %d registrations, %d table entries: %s" % (N_REG, n_entries, worst))
    # with the frozen association unchanged a 2-ulp covariance change stays an order of magnitude below the tolerance
    assert worst["dpose_same_corr"] < 1e-5
    # a flipped index needs a near-tie of two float32 Mahalanobis distances: rare, and the two cells are then near-equivalent
    assert worst["flips"] <= 16 and worst["flipped_regs"] <= 4
    assert worst["dpose_flipped"] < 1e-5
