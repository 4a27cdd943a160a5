"""Round-4 verdict item 4 (bounded experiment on k_solve's workgroup imbalance), step 1: how well can the LENGTH of a registration's
solve (its number of residual passes) be predicted from what is known before the first LM iteration?  Runs the config-4 batch
(512 registrations), reads the result records, and prints for each candidate predictor the share of the variance of `passes` a
linear fit explains (R^2), plus what perfect knowledge would buy: E[max of 4] / mean for random groups of four (what RPB = 4
pays today) against groups of four formed after sorting by the true length and by each predictor."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import synth  # noqa: E402

sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from util import GpuRig  # noqa: E402

prob = synth.make_batch_problem(8, 64, 34)
rig = GpuRig(prob)
rig.build_submaps()
mp = R.default_matcher_params()
g4 = synth.pose3_to_pose4(prob["guess"])
pose = torch.from_numpy(np.ascontiguousarray(g4)).to(rig.dev)
res = torch.zeros((rig.B, 64), dtype=torch.uint8, device=rig.dev)
ws = R.Maps(rig.ctx, rig.B, rig.mapp, rig.scan_cap, with_grid=False)
R.scan_register_batch(rig.ctx, rig.points, rig.clu, rig.submaps, rig.fixed_idx, ws, mp, pose, res)
rig.ctx.synchronize()
r = res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)
passes = r["n_evals"].astype(float)
trips = np.ceil(r["n_residuals"] / 64.0)
work = passes * trips                      # residual trips of a one-wavefront solve: what its duration is made of
dpose = np.abs(pose.cpu().numpy() - g4)
cells = ws.counts().astype(float)
pred = {
    "n_residuals": r["n_residuals"].astype(float),
    "trips = ceil(n_res / 64)": trips,
    "mu0 (first GNC stage's loss scale)": r["mu0"],
    "gnc_solves": r["gnc_solves"].astype(float),
    "initial_cost": r["initial_cost"],
    "initial_cost / n_residuals": r["initial_cost"] / np.maximum(1, r["n_residuals"]),
    "log initial_cost": np.log(np.maximum(r["initial_cost"], 1e-300)),
    "scan cells": cells,
    "(not known beforehand) |pose change| translation": np.hypot(dpose[:, 2], dpose[:, 3]),
}


def r2(x, y):
    x = (x - x.mean()) / (x.std() + 1e-300)
    return float(np.corrcoef(x, y)[0, 1] ** 2)


def group_cost(order, v, g=4):
    v = v[order]
    n = len(v) // g * g
    return float(v[:n].reshape(-1, g).max(axis=1).sum() * g / v[:n].sum())


rng = np.random.default_rng(0)
print("512 registrations: passes mean %.1f  min %d  max %d  std %.1f;  trips mean %.2f;  work = passes x trips mean %.1f std %.1f" %
      (passes.mean(), passes.min(), passes.max(), passes.std(), trips.mean(), work.mean(), work.std()))
print("%-52s %8s %8s | %s" % ("predictor", "R2 pass", "R2 work", "sum over groups of 4 of max(work) / sum(work), groups formed after sorting by the predictor"))
rand = np.mean([group_cost(rng.permutation(len(work)), work) for _ in range(200)])
print("%-52s %8s %8s | %.3f" % ("(batch order, what RPB = 4 does today)", "", "", group_cost(np.arange(len(work)), work)))
print("%-52s %8s %8s | %.3f" % ("(random groups)", "", "", rand))
print("%-52s %8s %8s | %.3f" % ("(the true work: perfect knowledge)", "1.000", "1.000", group_cost(np.argsort(-work), work)))
for k, v in pred.items():
    print("%-52s %8.3f %8.3f | %.3f" % (k, r2(v, passes), r2(v, work), group_cost(np.argsort(-v), work)))
X = np.stack([pred[k] for k in list(pred)[:8]] + [np.ones_like(passes)], 1)
X = (X - X.mean(0)) / (X.std(0) + 1e-300)
X[:, -1] = 1.0
for name, y in (("passes", passes), ("work", work)):
    coef, *_ = np.linalg.lstsq(X, y, rcond=None)
    fit = X @ coef
    print("all eight pre-solve predictors together, linear fit of %s: R2 = %.3f; grouping by the fit: %.3f" %
          (name, 1 - ((y - fit) ** 2).sum() / ((y - y.mean()) ** 2).sum(), group_cost(np.argsort(-fit), work)))
