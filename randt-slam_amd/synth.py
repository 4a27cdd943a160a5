"""Synthetic radar world / scans for the bench and the parity tests.

Pure numpy, deterministic from seeds.  Not part of the hot path: it only manufactures inputs of
the shape BASELINE.json names (2000-point azimuth-ordered radar scans, indoor parameter set).

World (seed 1234): 40 m x 24 m outer rectangle centred at the origin, 12 interior wall segments
(length U[2,8] m, random orientation), 20 point reflectors (discs, r = 0.15 m); every surface has a
reflectivity I0 ~ U[20,80].
Scan (seed 1000+i): 400 azimuths ray-cast to the nearest surface within [0.6, 12] m; per hit 5
range bins (0.04 m apart, cf. beam_distance_increment_threshold,
config/ndt_radar_slam_base_parameters.yaml:33) with intensities I0*{0.6,0.8,1,0.8,0.6}+N(0,1);
position noise sigma (0.02 m default) -> exactly 2000 points (x, y, 0, intensity) float32.
"""
import numpy as np

MIN_RANGE = 0.6   # config/parameters_indoor.yaml:42
MAX_RANGE = 12.0  # config/parameters_indoor.yaml:43
BIN_SPACING = 0.04
BIN_PROFILE = np.array([0.6, 0.8, 1.0, 0.8, 0.6])


# ------------------------------------------------------------------ SE(2) helpers (numpy) ----
def pose3_to_pose4(p):
    p = np.asarray(p, dtype=np.float64)
    return np.stack([np.cos(p[..., 2]), np.sin(p[..., 2]), p[..., 0], p[..., 1]], axis=-1)


def pose4_to_pose3(p):
    p = np.asarray(p, dtype=np.float64)
    return np.stack([p[..., 2], p[..., 3], np.arctan2(p[..., 1], p[..., 0])], axis=-1)


def se2_mul3(a, b):
    """(x,y,theta) composition a*b."""
    ca, sa = np.cos(a[2]), np.sin(a[2])
    return np.array([a[0] + ca * b[0] - sa * b[1], a[1] + sa * b[0] + ca * b[1], a[2] + b[2]])


def se2_inv3(a):
    ca, sa = np.cos(a[2]), np.sin(a[2])
    return np.array([-(ca * a[0] + sa * a[1]), -(-sa * a[0] + ca * a[1]), -a[2]])


def se2_exp3(xi):
    """exp of (vx, vy, omega) as (x, y, theta)."""
    w = xi[2]
    if abs(w) < 1e-10:
        a, b = 1.0 - w * w / 6.0, 0.5 * w
    else:
        a, b = np.sin(w) / w, (1.0 - np.cos(w)) / w
    return np.array([a * xi[0] - b * xi[1], b * xi[0] + a * xi[1], w])


def wrap_angle(a):
    return (np.asarray(a) + np.pi) % (2 * np.pi) - np.pi


# ------------------------------------------------------------------ world -------------------
def make_world(seed=1234):
    rng = np.random.default_rng(seed)
    W, H = 40.0, 24.0
    x0, x1, y0, y1 = -W / 2, W / 2, -H / 2, H / 2
    segs = [(x0, y0, x1, y0), (x1, y0, x1, y1), (x1, y1, x0, y1), (x0, y1, x0, y0)]
    for _ in range(12):
        L = rng.uniform(2.0, 8.0)
        ang = rng.uniform(0, np.pi)
        cx = rng.uniform(x0 + 3, x1 - 3)
        cy = rng.uniform(y0 + 3, y1 - 3)
        dx, dy = 0.5 * L * np.cos(ang), 0.5 * L * np.sin(ang)
        segs.append((cx - dx, cy - dy, cx + dx, cy + dy))
    segs = np.array(segs, dtype=np.float64)
    seg_I0 = rng.uniform(20.0, 80.0, size=len(segs))
    discs = np.stack(
        [rng.uniform(x0 + 1, x1 - 1, 20), rng.uniform(y0 + 1, y1 - 1, 20), np.full(20, 0.15)], axis=1
    )
    disc_I0 = rng.uniform(20.0, 80.0, size=20)
    return dict(segs=segs, seg_I0=seg_I0, discs=discs, disc_I0=disc_I0, bounds=(x0, x1, y0, y1))


def raycast(world, origin, angles):
    """Nearest hit range and reflectivity for rays from origin (2,) at world-frame angles (A,)."""
    ox, oy = origin
    dx, dy = np.cos(angles)[:, None], np.sin(angles)[:, None]
    s = world["segs"]
    ex, ey = (s[:, 2] - s[:, 0])[None, :], (s[:, 3] - s[:, 1])[None, :]
    ax, ay = (s[:, 0] - ox)[None, :], (s[:, 1] - oy)[None, :]
    den = dx * ey - dy * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (ax * ey - ay * ex) / den
        u = (ax * dy - ay * dx) / den
    ok = (np.abs(den) > 1e-12) & (t > 0) & (u >= 0) & (u <= 1)
    t_seg = np.where(ok, t, np.inf)
    d = world["discs"]
    cx, cy, r = (d[:, 0] - ox)[None, :], (d[:, 1] - oy)[None, :], d[:, 2][None, :]
    b = dx * cx + dy * cy
    disc = b * b - (cx * cx + cy * cy - r * r)
    with np.errstate(invalid="ignore"):
        t_d = b - np.sqrt(disc)
    t_disc = np.where((disc >= 0) & (t_d > 0), t_d, np.inf)
    t_all = np.concatenate([t_seg, t_disc], axis=1)
    I_all = np.concatenate([world["seg_I0"], world["disc_I0"]])
    k = np.argmin(t_all, axis=1)
    rng_hit = t_all[np.arange(len(angles)), k]
    return rng_hit, I_all[k]


def make_scan(world, pose, seed, n_az=400, sigma=0.02, stride=4):
    """2000-point scan in the SENSOR frame from sensor pose (x, y, theta) in the world frame."""
    rng = np.random.default_rng(seed)
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False)
    r, I0 = raycast(world, pose[:2], az + pose[2])
    lo, hi = MIN_RANGE + 2 * BIN_SPACING + 0.01, MAX_RANGE - 2 * BIN_SPACING - 0.01
    bad = ~((r > lo) & (r < hi))
    for _ in range(64):
        if not bad.any():
            break
        az[bad] = rng.uniform(-np.pi, np.pi, bad.sum())
        rb, Ib = raycast(world, pose[:2], az[bad] + pose[2])
        r[bad], I0[bad] = rb, Ib
        bad = ~((r > lo) & (r < hi))
    if bad.any():
        raise RuntimeError("scan pose sees nothing within range")
    order = np.argsort(az, kind="stable")
    az, r, I0 = az[order], r[order], I0[order]
    rr = r[:, None] + (np.arange(5) - 2)[None, :] * BIN_SPACING
    x = rr * np.cos(az)[:, None] + rng.normal(0, 1, rr.shape) * sigma
    y = rr * np.sin(az)[:, None] + rng.normal(0, 1, rr.shape) * sigma
    inten = I0[:, None] * BIN_PROFILE[None, :] + rng.normal(0, 1, rr.shape)
    pts = np.zeros((n_az * 5, stride), dtype=np.float32)
    pts[:, 0] = x.reshape(-1)
    pts[:, 1] = y.reshape(-1)
    pts[:, stride - 1 if stride == 4 else 4] = inten.reshape(-1)
    return pts


def make_polar_scan(world, pose, seed, n_az=400, n_bins=3000, bin_size=0.0438, speckle=5.0, stride=4):
    """Oxford-RobotCar-shaped raw radar scan (BASELINE config 5): n_az azimuths x n_bins range bins,
    every bin a point (x, y, 0, intensity) in the SENSOR frame, azimuth after azimuth, range ascending
    (the organisation RadarPreprocessor::filterScan assumes, radar_preprocessor.cpp:61).
    Intensity = speckle U[0, speckle) everywhere + the 5-bin return profile of the first surface hit."""
    rng = np.random.default_rng(seed)
    az = -np.pi + (np.arange(n_az) + 0.5) * (2 * np.pi / n_az)
    r_hit, I0 = raycast(world, pose[:2], az + pose[2])
    r = (np.arange(n_bins) + 0.5) * bin_size
    inten = rng.uniform(0.0, speckle, (n_az, n_bins))
    hit_bin = np.floor(r_hit / bin_size).astype(np.int64)
    for k, w in enumerate(BIN_PROFILE):
        b = hit_bin + (k - 2)
        ok = np.isfinite(r_hit) & (b >= 0) & (b < n_bins)
        inten[np.nonzero(ok)[0], b[ok]] += I0[ok] * w
    pts = np.zeros((n_az, n_bins, stride), dtype=np.float32)
    pts[..., 0] = r[None, :] * np.cos(az)[:, None]
    pts[..., 1] = r[None, :] * np.sin(az)[:, None]
    pts[..., stride - 1 if stride == 4 else 4] = inten
    return pts


def make_trajectory(seed, n_poses, step=1.0):
    """Smooth seeded path (arc of an ellipse well inside the room); poses (n,3) in the world frame,
    heading along the tangent, arc-length spacing ~ step metres."""
    rng = np.random.default_rng(seed)
    a, b = rng.uniform(5.0, 8.0), rng.uniform(3.5, 5.0)
    cx, cy = rng.uniform(-7.0, 7.0), rng.uniform(-2.5, 2.5)
    phi = rng.uniform(0, 2 * np.pi)
    sgn = 1.0 if rng.uniform() < 0.5 else -1.0
    poses = []
    for _ in range(n_poses):
        x, y = cx + a * np.cos(phi), cy + b * np.sin(phi)
        tx, ty = -a * np.sin(phi) * sgn, b * np.cos(phi) * sgn
        poses.append((x, y, np.arctan2(ty, tx)))
        phi += sgn * step / np.hypot(tx, ty)
    return np.array(poses)


def perturb_pose(truth3, seed, dt=0.3, dtheta_deg=3.0):
    """Initial guess = truth o exp(delta), delta_t ~ U[-dt,dt] per axis, delta_theta ~ U[-3deg,3deg]."""
    rng = np.random.default_rng(seed)
    delta = np.array([rng.uniform(-dt, dt), rng.uniform(-dt, dt), np.deg2rad(rng.uniform(-dtheta_deg, dtheta_deg))])
    return se2_mul3(truth3, se2_exp3(delta))


# ------------------------------------------------------------------ parameter sets ----------
def indoor_params():
    """config/parameters_indoor.yaml + ndt_radar_slam_base_parameters.yaml, with the derivations of
    src/ndt_slam/ndt_slam.cpp:653-654,691."""
    res = 0.5
    return dict(
        resolution=res, size_x=int(50 / res), size_y=int(50 / res), min_points_per_cell=5,
        max_neighbour_dist=4.0, max_range=MAX_RANGE, min_range=MIN_RANGE,
        n_clusters=int((2.0 * MAX_RANGE / res) ** 2),
        n_neighbours=4, loss_alpha=-2.0, loss_scale=1.5, gnc_divisor=1.3, gnc_steps=3,
        loop_closure_gnc_steps=2, loop_closure_scale=1.5, max_iterations=200,
        insertion_step=4, submap_size_poses=135, submap_overlap=20, smoothing_steps=3,
        ndt_weight=5.0e4, covariance_scaling_factor=25.0,
    )


def make_batch_problem(n_submaps=8, scans_per_submap=64, n_keyframes=34, world_seed=1234, sigma=0.02,
                       scan_seed0=1000, guess_seed0=2000, traj_seed0=3000):
    """BASELINE config 4 inputs: for each submap a keyframe trajectory (+ its scans, to be merged by
    the caller with the a18 rule) and `scans_per_submap` query scans with truth pose and initial
    guess expressed in the submap frame (= first keyframe pose)."""
    world = make_world(world_seed)
    out = dict(world=world, submaps=[], scans=[], truth=[], guess=[], submap_of=[])
    i = 0
    for j in range(n_submaps):
        traj = make_trajectory(traj_seed0 + j, n_keyframes)
        origin_inv = se2_inv3(traj[0])
        kf_scans = [make_scan(world, traj[t], 500000 + 1000 * j + t, sigma=sigma) for t in range(n_keyframes)]
        kf_rel = np.array([se2_mul3(origin_inv, traj[t]) for t in range(n_keyframes)])
        out["submaps"].append(dict(traj=traj, kf_scans=kf_scans, kf_rel=kf_rel))
        rng = np.random.default_rng(traj_seed0 + 100 + j)
        for _ in range(scans_per_submap):
            t = rng.integers(0, n_keyframes)
            off = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), rng.uniform(-0.2, 0.2)])
            pose = se2_mul3(traj[t], off)
            rel = se2_mul3(origin_inv, pose)
            rel[2] = wrap_angle(rel[2])
            out["scans"].append(make_scan(world, pose, scan_seed0 + i, sigma=sigma))
            out["truth"].append(rel)
            out["guess"].append(perturb_pose(rel, guess_seed0 + i))
            out["submap_of"].append(j)
            i += 1
    out["scans"] = np.stack(out["scans"])
    out["truth"] = np.array(out["truth"])
    out["guess"] = np.array(out["guess"])
    out["submap_of"] = np.array(out["submap_of"], dtype=np.int32)
    return out
