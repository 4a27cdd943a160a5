"""Trajectory text formats and drift evaluation (evaluation half).

The reference publishes its Oxford results as KITTI pose rows (12 floats, row-major 3x4), TUM rows
(stamp x y z qx qy qz qw) and a `result.txt` table (oxford_results/*/{odom,slam}/est/{01,tum_01}.txt,
result.txt, errors/01.txt); the writer itself is not part of the reference repository.  This module
reads / writes those formats for planar poses [cos, sin, tx, ty] (= Sophus::SE2d::data(), the pose
type of the C ABI) and restates the evaluation behind `result.txt`: the KITTI odometry devkit's
segment drift (100..800 m, every 10th frame) plus aligned ATE, consecutive-frame RPE and bias.
tests/test_trajectory.py pins it to the reference's own numbers (tests/golden/oxford_eval_01.npz).
CPU-side by nature (sparse, serial, a few thousand poses)."""
import numpy as np

KITTI_LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)
KITTI_STEP = 10


# ---------------------------------------------------------------- formats
def pose4_to_kitti(pose4):
    """(N, 4) [c, s, tx, ty] -> (N, 12) row-major 3x4 [R | t] with z = 0."""
    p = np.asarray(pose4, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((p.shape[0], 12))
    out[:, 0], out[:, 1], out[:, 3] = p[:, 0], -p[:, 1], p[:, 2]
    out[:, 4], out[:, 5], out[:, 7] = p[:, 1], p[:, 0], p[:, 3]
    out[:, 10] = 1.0
    return out


def kitti_to_pose4(rows):
    r = np.asarray(rows, dtype=np.float64).reshape(-1, 12)
    c, s = r[:, 0], r[:, 4]
    n = np.hypot(c, s)
    return np.stack([c / n, s / n, r[:, 3], r[:, 7]], axis=1)


def write_kitti(path, pose4):
    with open(path, "w") as f:
        for row in pose4_to_kitti(pose4):
            f.write(" ".join("%.6f" % v for v in row) + "\n")


def read_kitti(path):
    return kitti_to_pose4(np.loadtxt(path).reshape(-1, 12))


def format_tum_row(stamp, pose4):
    """One TUM row in the reference's number formats: %.9f stamp, %.4f position, %.4g quaternion."""
    c, s, tx, ty = (float(v) for v in pose4)
    half = 0.5 * np.arctan2(s, c)
    return "%.9f %.4f %.4f %.4f 0 0 %.4g %.4g" % (stamp, tx, ty, 0.0, np.sin(half), np.cos(half))


def write_tum(path, stamps, pose4):
    with open(path, "w") as f:
        for t, p in zip(stamps, np.asarray(pose4).reshape(-1, 4)):
            f.write(format_tum_row(t, p) + "\n")


def read_tum(path_or_lines):
    lines = open(path_or_lines).read().splitlines() if isinstance(path_or_lines, str) else list(path_or_lines)
    a = np.array([[float(v) for v in ln.split()] for ln in lines if ln.strip()])
    yaw = 2.0 * np.arctan2(a[:, 6], a[:, 7])
    return a[:, 0], np.stack([np.cos(yaw), np.sin(yaw), a[:, 1], a[:, 2]], axis=1)


# ---------------------------------------------------------------- evaluation
def _mats(pose4):
    p = np.asarray(pose4, dtype=np.float64).reshape(-1, 4)
    T = np.tile(np.eye(3), (p.shape[0], 1, 1))
    T[:, 0, 0], T[:, 0, 1], T[:, 0, 2] = p[:, 0], -p[:, 1], p[:, 2]
    T[:, 1, 0], T[:, 1, 1], T[:, 1, 2] = p[:, 1], p[:, 0], p[:, 3]
    return T


def _mats_from_kitti(rows):
    """Use the 2x2 rotation block as written (not re-normalised): the published numbers were computed that way."""
    r = np.asarray(rows, dtype=np.float64).reshape(-1, 12)
    T = np.tile(np.eye(3), (r.shape[0], 1, 1))
    T[:, 0, 0], T[:, 0, 1], T[:, 0, 2] = r[:, 0], r[:, 1], r[:, 3]
    T[:, 1, 0], T[:, 1, 1], T[:, 1, 2] = r[:, 4], r[:, 5], r[:, 7]
    return T


def _rot_err(M):
    # devkit: acos((trace(R3) - 1) / 2) with R3 = diag-block(R2, 1)
    d = 0.5 * (M[..., 0, 0] + M[..., 1, 1] + 1.0 - 1.0)
    return np.arccos(np.clip(d, -1.0, 1.0))


def kitti_segment_errors(gt, est, lengths=KITTI_LENGTHS, step=KITTI_STEP):
    """Rows (first_frame, rot_err/len [rad/m], trans_err/len, len, speed) of the KITTI devkit (errors/01.txt).
    gt / est: (N, 3, 3) homogeneous planar transforms."""
    n = gt.shape[0]
    dist = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(gt[:, :2, 2], axis=0), axis=1))])
    rows = []
    for first in range(0, n, step):
        for L in lengths:
            idx = np.nonzero(dist[first:] > dist[first] + L)[0]
            if idx.size == 0:
                continue
            last = first + int(idx[0])
            dg = np.linalg.inv(gt[first]) @ gt[last]
            de = np.linalg.inv(est[first]) @ est[last]
            pe = np.linalg.inv(de) @ dg
            rows.append((first, _rot_err(pe) / L, np.linalg.norm(pe[:2, 2]) / L, L, L / (0.1 * (last - first + 1))))
    return np.array(rows).reshape(-1, 5)


def _align_rigid(a, b):
    """Least-squares rotation + translation taking points a onto b (Umeyama without scale)."""
    ma, mb = a.mean(0), b.mean(0)
    U, _, Vt = np.linalg.svd((a - ma).T @ (b - mb))
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[-1] *= -1
        R = Vt.T @ U.T
    return (R @ a.T).T + (mb - R @ ma)


def evaluate(gt, est):
    """The rows of the reference's result.txt.  gt / est: (N, 4) pose4 or (N, 12) KITTI rows."""
    G = _mats_from_kitti(gt) if np.asarray(gt).shape[-1] == 12 else _mats(gt)
    E = _mats_from_kitti(est) if np.asarray(est).shape[-1] == 12 else _mats(est)
    seg = kitti_segment_errors(G, E)
    rel_g = np.linalg.inv(G[:-1]) @ G[1:]
    rel_e = np.linalg.inv(E[:-1]) @ E[1:]
    err = np.linalg.inv(rel_g) @ rel_e
    tn = np.linalg.norm(err[:, :2, 2], axis=1)
    ang = np.degrees(_rot_err(err))
    al = _align_rigid(E[:, :2, 2], G[:, :2, 2])
    return {
        "trans_err_percent": float(seg[:, 2].mean() * 100.0) if seg.size else float("nan"),
        "rot_err_deg_per_100m": float(np.degrees(seg[:, 1].mean()) * 100.0) if seg.size else float("nan"),
        "ate_m": float(np.sqrt(((al - G[:, :2, 2]) ** 2).sum(1).mean())),
        "rpe_m": float(tn.mean()), "rpe_dev_m": float(tn.std()),
        "rpe_deg": float(ang.mean()), "rpe_dev_deg": float(ang.std()),
        "bias_x_m": float(err[:, 0, 2].mean()), "bias_y_m": float(err[:, 1, 2].mean()),
        "rmse_m": float(np.sqrt((tn ** 2).mean())),
        "segments": seg,
    }
