"""Builds tests/golden/oxford_eval_01.npz from the result DATA files the reference repository ships
(oxford_results/randt_eval_16-13-09/slam: estimated and ground-truth trajectories of Oxford sequence 01 and the
evaluation table computed from them).  Run in the build container only (/root/reference is not on the GPU box).

Stored: the planar entries of the 3553 KITTI pose rows as integers in units of 1e-6 (the files carry six
decimals, so this is exact), the first TUM rows verbatim, the published result.txt numbers and every 40th row
of errors/01.txt."""
import os
import numpy as np

SRC = "/root/reference/oxford_results/randt_eval_16-13-09/slam"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oxford_eval_01.npz")


def planar_micro(path):
    r = np.loadtxt(path).reshape(-1, 12)
    assert np.all(r[:, [2, 6, 8, 9, 11]] == 0) and np.all(r[:, 10] == 1), "not planar"
    q = np.rint(r[:, [0, 1, 3, 4, 5, 7]] * 1e6).astype(np.int64)
    assert np.abs(q / 1e6 - r[:, [0, 1, 3, 4, 5, 7]]).max() < 1e-9
    return q


result = {}
for line in open(os.path.join(SRC, "est/result.txt")):
    k, v = line.split(",")[:2]
    result[k.strip()] = float(v)
errors = np.loadtxt(os.path.join(SRC, "est/errors/01.txt"))
np.savez_compressed(
    OUT, est_micro=planar_micro(os.path.join(SRC, "est/01.txt")), gt_micro=planar_micro(os.path.join(SRC, "gt/01.txt")),
    tum_head=np.array(open(os.path.join(SRC, "est/tum_01.txt")).read().splitlines()[:12]),
    kitti_head=np.array(open(os.path.join(SRC, "est/01.txt")).read().splitlines()[:12]),
    result_keys=np.array(list(result.keys())), result_values=np.array(list(result.values())),
    errors_rows=errors[::40], errors_index=np.arange(errors.shape[0])[::40], errors_count=np.array(errors.shape[0]))
print(OUT, os.path.getsize(OUT), "bytes")
