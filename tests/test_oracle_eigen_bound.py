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
from util import oracle_map, oracle_scan_map, oracle_submap, to_oracle_params

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


# --------------------------------------------------------------------------------------------------------------------------
# Round-4 verdict, item 7: the other two stand-ins, bounded the same way.
#
# (2) `Cell::transformCell` rotates the covariance with `trans.rotation()` (ndt_cell.cpp:117-123): Eigen extracts that rotation
#     from the affine matrix by a JacobiSVD, the oracle / the kernels use the exact blockdiag(R2, 1) (SPEC DECISION 2).  An SVD
#     of a matrix that IS a rotation up to float rounding returns it to a few ulp per entry, i.e. the transformed covariance to
#     a few ulp per entry.  Transformed covariances enter twice: every keyframe scan cell on its way into a submap
#     (transformMap -> mergeMapCell), and every moving cell at the association (transformCell by the guess, then
#     getClosestCells with Sigma_q + Sigma_c).  Both are perturbed here, all six entries, +-1 / +-2 ulp.
# (4) a residual of exactly zero gets a zero Jacobian row where the reference's autodiff of sqrt(0) produces NaN: counted over
#     every evaluation the oracle makes on the config-4 batch and on a fixed-lag odometry drive.  Expected, and asserted: never.
ALL_ENTRIES = (0, 1, 2, 3, 4, 5)


def _perturb_cov(m, pattern, rng, entries=ALL_ENTRIES):
    cells = m.cells()
    cov = cells["cov"].copy()
    for e in entries:
        steps = np.full(len(cells), pattern, dtype=np.int64) if pattern is not None else rng.integers(-2, 3, size=len(cells))
        cov[:, e] = _ulp_shift(cov[:, e], steps)
    cells["cov"] = cov
    m.set(cells, m.grid())


def test_two_ulp_of_every_transformed_covariance_bound_the_pose_and_no_residual_is_ever_zero():
    prob = synth.make_batch_problem(8, 64, 34)
    mp = R.default_matcher_params()
    op = to_oracle_params(mp)
    g4 = synth.pose3_to_pose4(prob["guess"])
    scans = [oracle_scan_map(prob["scans"][i]) for i in range(N_REG)]
    k = mp.n_neighbours
    ident = np.array([1.0, 0.0, 0.0, 0.0])

    def submaps(pattern, rng):
        out = []
        for sm in prob["submaps"]:
            sub = oracle_map()
            for t in range(len(sm["kf_scans"])):
                scan = oracle_scan_map(sm["kf_scans"][t])
                scan.transform(synth.pose3_to_pose4(sm["kf_rel"][t]))
                if pattern != 0:
                    _perturb_cov(scan, pattern, rng)           # what a rotation off by ulps would have produced
                sub.merge(scan)
            out.append(sub)
        return out

    def run(fixed, pattern, rng, via_transform):
        corr, pose = [], np.zeros((N_REG, 4))
        for i in range(N_REG):
            f = fixed[prob["submap_of"][i]]
            if via_transform:
                moved = scans[i].copy()
                moved.transform(g4[i])                          # Cell::transformCell(initial guess), ndt_matcher.cpp:207-209
                if pattern != 0:
                    _perturb_cov(moved, pattern, rng)
                c, _ = po.associate(f, moved, ident, k, bool(mp.lookup_mahalanobis), bool(mp.use_intensity))
            else:
                c, _ = po.associate(f, scans[i], g4[i], k, bool(mp.lookup_mahalanobis), bool(mp.use_intensity))
            rc, p4, st = po.solve_pair(f, scans[i], c, op, g4[i])   # the residuals take the UNtransformed cells and the pose
            assert rc == 0
            corr.append(np.asarray(c).copy())
            pose[i] = p4
        return corr, pose

    po.sqrt_zero_count(reset=True)
    ref_subs = submaps(0, None)
    ref_corr, ref_pose = run(ref_subs, 0, None, False)
    # the injection point is exact: transforming first and associating at the identity is the association at the guess
    chk_corr, chk_pose = run(ref_subs, 0, None, True)
    assert all(np.array_equal(a, b) for a, b in zip(ref_corr, chk_corr)) and np.array_equal(ref_pose, chk_pose)
    n_entries = sum(c.size for c in ref_corr)
    rng = np.random.default_rng(20260931)
    worst = {"flips": 0, "flipped_regs": 0, "dpose_same_corr": 0.0, "dpose_flipped": 0.0}
    for pattern in (1, -1, 2, -2, None):
        corr, pose = run(submaps(pattern, rng), pattern, rng, True)
        flips, flipped_regs = 0, 0
        for i in range(N_REG):
            d = int((corr[i] != ref_corr[i]).sum())
            flips += d
            flipped_regs += d > 0
            dth = np.arctan2(pose[i, 1], pose[i, 0]) - np.arctan2(ref_pose[i, 1], ref_pose[i, 0])
            dp = max(abs((dth + np.pi) % (2 * np.pi) - np.pi), np.abs(pose[i, 2:] - ref_pose[i, 2:]).max())
            key = "dpose_flipped" if d else "dpose_same_corr"
            worst[key] = max(worst[key], float(dp))
        worst["flips"] = max(worst["flips"], flips)
        worst["flipped_regs"] = max(worst["flipped_regs"], flipped_regs)
    print("rotation stand-in bound over %d registrations, %d table entries: %s" % (N_REG, n_entries, worst))
    assert worst["dpose_same_corr"] < 1e-5 and worst["dpose_flipped"] < 1e-5
    assert worst["flips"] <= 32 and worst["flipped_regs"] <= 8

    # (4) the sqrt(0) guard: seven passes over the batch above (3584 registrations, every LM iterate of every one) ...
    zero_pairs = po.sqrt_zero_count(reset=True)
    # ... and a fixed-lag drive through the processScan call pattern (window solves: motion factors + NDT terms)
    from oracle_backend import OracleBackend
    from randt_slam_amd import odometry

    world = synth.make_world()
    traj = synth.make_trajectory(3200, 30, step=0.25)
    odo = odometry.Odometry(OracleBackend(), R.default_matcher_params(parameterization=R.PARAM_MANIFOLD, gnc_steps=3), R.window_params(),
                            dict(submap_size_poses=20, submap_overlap=6))
    for i in range(30):
        odo.process_scan(synth.make_scan(world, traj[i], 9000 + i), 0.25 * i)
    zero_windows = po.sqrt_zero_count(reset=True)
    print("sqrt(0) guard fired %d times over the pair solves, %d times over the drive's %d window solves" % (zero_pairs, zero_windows, odo.n_registrations))
    assert zero_pairs == 0 and zero_windows == 0 and odo.n_registrations >= 28
