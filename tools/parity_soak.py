"""Randomised parity soak (not part of the test suite): many random scenes / parameter sets through the HIP
build + association + solve and through the CPU oracle; reports any mismatch.  Usage: parity_soak.py [n_cases] [seed]   (SOAK_THROUGHPUT=1: the throughput-placement kernels)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle as po  # noqa: E402
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import synth  # noqa: E402
from util import cells_equal, to_oracle_params  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
VERBOSE = os.environ.get("SOAK_VERBOSE") == "1"
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
if os.environ.get("SOAK_THROUGHPUT") == "1":   # the register-capped build / association instantiations and the one-wavefront solve
    ctx.set_solve_mode(R._capi.SOLVE_THROUGHPUT)
bad = {"build": 0, "assoc": 0, "pose": 0, "iters": 0}
worst = 0.0
n_solved = n_well = bad_well = 0
worst_well = 0.0
for case in range(n_cases):
    # random map / clustering geometry around the shipped presets
    res = float(rng.choice([0.4, 0.5, 0.75, 1.2] if os.environ.get("SOAK_OLD_DRAW") == "1" else [0.25, 0.4, 0.5, 0.75, 1.2]))                 # (0.25 m: window radii beyond 7, the association's WIDE instantiation, round 4)
    size = int(rng.choice([60, 100, 140])) if res > 0.3 else int(rng.choice([120, 200]))
    max_range = float(rng.choice([10.0, 12.0, 20.0]))
    mapp_args = (size, size, res, float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(rng.choice([2.0, 4.0])) * res / 0.5 if rng.random() < 0.5 else 4.0,
                 int(rng.integers(2, 7)), 0)
    if int(mapp_args[5] / res) - 1 > 15:
        mapp_args = mapp_args[:5] + (15.9 * res,) + mapp_args[6:]
    n_clusters = int((2 * max_range / res) ** 2)
    mapp, clu = R.MapParams(*mapp_args), R.ClusterParams(n_clusters, max_range)
    world = synth.make_world(seed=int(rng.integers(0, 10000)))
    pose = np.array([rng.uniform(-6, 6), rng.uniform(-4, 4), rng.uniform(-3.1, 3.1)])
    n_az = int(rng.choice([100, 240, 400, 600]))                   # 500 .. 3000 points
    dpose = pose + np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.05, 0.05)])
    try:
        s_fix = synth.make_scan(world, pose, int(rng.integers(0, 1 << 30)), n_az=n_az)
        s_mov = synth.make_scan(world, dpose, int(rng.integers(0, 1 << 30)))
    except RuntimeError:
        continue                                                   # the random pose sees nothing
    if rng.random() < 0.3:                                        # ragged / partly out-of-range input
        s_mov = s_mov[: int(rng.integers(50, len(s_mov)))].copy()
        s_mov[::53, :2] *= 3.0
    fm = R.Maps(ctx, 1, mapp, 4096, with_grid=True)
    mm = R.Maps(ctx, 1, mapp, 1024, with_grid=False)
    R.ndt_build_batch(ctx, torch.from_numpy(s_fix[None]).to(dev), clu, fm)
    R.ndt_build_batch(ctx, torch.from_numpy(s_mov[None]).to(dev), clu, mm)

    def omap(cap):
        return po.Map(size, size, res, (mapp_args[3], mapp_args[4]), mapp_args[5], mapp_args[6], cap)

    of, om = omap(4096), omap(1024)
    of.build(s_fix, n_clusters, max_range)
    om.build(s_mov, n_clusters, max_range)
    cf, gf = fm.download(0)
    cm, _ = mm.download(0)
    if not (cells_equal(cf, of.cells()) and np.array_equal(gf, of.grid()) and cells_equal(cm, om.cells())):
        bad["build"] += 1
        print("case", case, "BUILD mismatch", mapp_args, n_clusters)
        continue
    if om.n_cells == 0 or of.n_cells == 0:
        continue
    k = int(rng.choice([1, 3, 4, 6] if os.environ.get("SOAK_OLD_DRAW") == "1" else [1, 3, 4, 6, 4, 4, 9, 12, 16]))
    mp = R.default_matcher_params(n_neighbours=k, lookup_mahalanobis=int(rng.random() < 0.7), use_intensity=int(rng.random() < 0.7),
                                  parameterization=int(rng.choice([R.PARAM_AMBIENT4, R.PARAM_MANIFOLD, R.PARAM_VECTOR, R.PARAM_ANALYTIC])),
                                  gnc_steps=int(rng.choice([1, 2, 3])))
    rel = synth.se2_mul3(synth.se2_inv3(pose), dpose)
    g4 = synth.pose3_to_pose4(rel + np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.04, 0.04)]))
    pose_t = torch.from_numpy(g4[None].copy()).to(dev)
    corr = torch.full((1, 1024, k), -1, dtype=torch.int32, device=dev)
    resu = torch.zeros((1, 64), dtype=torch.uint8, device=dev)
    fidx = torch.zeros(1, dtype=torch.int32, device=dev)
    R.associate_batch(ctx, fm, fidx, mm, 0, 1, pose_t, mp, corr)
    R.solve_batch(ctx, fm, fidx, mm, 0, 1, corr, mp, pose_t, resu)
    ctx.synchronize()
    oc, _ = po.associate(of, om, g4, k, mp.lookup_mahalanobis, mp.use_intensity)
    if not np.array_equal(corr.cpu().numpy()[0, : om.n_cells], oc):
        bad["assoc"] += 1
        print("case", case, "ASSOC mismatch", mapp_args, k)
        continue
    rc, p4, cost, st = po.register_pair(of, om, to_oracle_params(mp), g4)
    gp = pose_t.cpu().numpy()[0]
    r = resu.cpu().numpy().view(R.RESULT_DTYPE)[0]
    err = max(abs(gp[2] - p4[2]), abs(gp[3] - p4[3]), abs(synth.wrap_angle(np.arctan2(gp[1], gp[0]) - np.arctan2(p4[1], p4[0]))))
    worst = max(worst, err)
    n_solved += 1
    well = int(st["n_residuals"]) >= 100 and int(st["n_iterations"]) <= 100   # a well-posed registration
    n_well += well
    if well:
        worst_well = max(worst_well, err)
        bad_well += err > 1e-4
    it_g, it_o = int(r["iterations"].item() if hasattr(r["iterations"], "item") else r["iterations"]), int(st["n_iterations"])
    if VERBOSE and well and err > 1e-5:
        # yardstick: the oracle against ITSELF with the initial translation moved by one ulp -- is this registration a flat valley?
        g_ulp = g4.copy()
        g_ulp[2] = np.nextafter(g_ulp[2], np.inf)
        _, p_u, cost_u, st_u = po.register_pair(of, om, to_oracle_params(mp), g_ulp)
        err_u = max(abs(p_u[2] - p4[2]), abs(p_u[3] - p4[3]), abs(synth.wrap_angle(np.arctan2(p_u[1], p_u[0]) - np.arctan2(p4[1], p4[0]))))
        print("  oracle vs one-ulp-perturbed oracle: %.3e (iterations %d vs %d); costs gpu %.12g oracle %.12g perturbed %.12g" % (err_u, int(st_u["n_iterations"]), it_o if False else int(st["n_iterations"]), float(r["cost"]), cost, cost_u))
    if VERBOSE and well and err > 1e-6:
        print("well-posed case %d err %.3e k %d res %.2f param %d gnc %d n_res %d iters gpu %d oracle %d" % (case, err, k, res, mp.parameterization, mp.gnc_steps, int(st["n_residuals"]), int(r["iterations"]), int(st["n_iterations"])))
    if err > 1e-4 or it_g != it_o:
        bad["pose" if err > 1e-4 else "iters"] += 1
        if VERBOSE:
            print("case %d err %.3e | k %d mahal %d inten %d param %d gnc %d | n_res %d/%d | iters gpu %d oracle %d | term gpu %d oracle %d | cost gpu %.9g oracle %.9g | final gpu %.9g"
              % (case, err, k, mp.lookup_mahalanobis, mp.use_intensity, mp.parameterization, mp.gnc_steps, int(r["n_residuals"]), int(st["n_residuals"]),
                 it_g, it_o, int(r["termination"]), int(st["termination"]), float(r["cost"]), cost, float(r["final_cost"])))
print("cases", n_cases, "solved", n_solved, "mismatches", bad, "worst pose difference vs oracle %.3e" % worst)
print("well-posed (>= 100 residuals, <= 100 LM iterations): %d, beyond 1e-4: %d, worst %.3e" % (n_well, bad_well, worst_well))
