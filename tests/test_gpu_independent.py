"""-m gpu: the HIP path against known answers that were NOT produced by the CPU oracle (tests/golden/independent_01.npz,
made by tests/golden/make_independent.py from numpy / scipy alone; this file does not import the oracle either):
K1 zero-noise registrations with an exact minimiser, K3 scipy.optimize.least_squares minimisers of the robust
fixed-correspondence objective, K4 numpy-float32 cell statistics.  Removes the "kernel vs its author's oracle" loop for the
three core stages (VERDICT r01 item 4 ii)."""
import os
import sys

import numpy as np
import pytest

import randt_slam_amd as R
from randt_slam_amd import synth

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "independent_01.npz")


@pytest.fixture(scope="module")
def env(built):
    import torch

    assert "pyoracle" not in sys.modules or True   # (other test modules of the same session may have loaded it; this one never calls it)
    return torch, torch.device("cuda:0"), R.Context(0, torch.cuda.current_stream().cuda_stream), np.load(FIX)


def _solve(env, fixed, moving, corr, guess3, mp):
    torch, dev, ctx, _ = env
    B, n = fixed.shape
    mapp = R.indoor_map_params()
    fm = R.Maps(ctx, B, mapp, n, with_grid=True)
    mm = R.Maps(ctx, B, mapp, n, with_grid=False)
    for b in range(B):
        fm.upload(b, fixed[b].astype(R.CELL_DTYPE))
        mm.upload(b, moving[b].astype(R.CELL_DTYPE))
    pose = torch.from_numpy(synth.pose3_to_pose4(np.broadcast_to(guess3, (B, 3)).copy())).to(dev)
    c = torch.from_numpy(np.ascontiguousarray(corr, dtype=np.int32)).to(dev)
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    fidx = torch.arange(B, dtype=torch.int32, device=dev)
    R.solve_batch(ctx, fm, fidx, mm, 0, B, c, mp, pose, res)
    ctx.synchronize()
    return synth.pose4_to_pose3(pose.cpu().numpy()), res.cpu().numpy().view(R.RESULT_DTYPE).reshape(-1)


@pytest.mark.parametrize("param", [R.PARAM_MANIFOLD, R.PARAM_AMBIENT4, R.PARAM_VECTOR])
def test_k1_zero_noise_known_answer(env, param):
    d = env[3]
    mp = R.default_matcher_params(parameterization=param, n_neighbours=1, gnc_steps=2, function_tolerance=1e-14, parameter_tolerance=1e-13)
    for i in range(len(d["k1_truth"])):
        est, res = _solve(env, d["k1_fixed"][i:i + 1], d["k1_moving"][i:i + 1], d["k1_corr"][i:i + 1], d["k1_guess"][i], mp)
        err = est[0] - d["k1_truth"][i]
        err[2] = (err[2] + np.pi) % (2 * np.pi) - np.pi
        # the cells are stored in float32 (means ~10 m: 1e-6 m quantisation), the objective is exactly zero at the truth
        assert np.abs(err[:2]).max() < 2e-5 and abs(err[2]) < 2e-6, (i, err)
        assert res["n_residuals"][0] == 48 and res["final_cost"][0] < 1e-6 and res["status"][0] == 0


def test_k3_scipy_minimisers(env):
    d = env[3]
    for i, alpha in enumerate(d["k3_alpha"]):
        for param in (R.PARAM_MANIFOLD, R.PARAM_VECTOR):
            mp = R.default_matcher_params(parameterization=param, n_neighbours=1, gnc_steps=1, loss_alpha=float(alpha), loss_scale=1.5,
                                          mu_scale=1.5, use_intensity=int(d["k3_dim"][i] == 3), function_tolerance=1e-15,
                                          parameter_tolerance=1e-14, gradient_tolerance=1e-14)
            est, res = _solve(env, d["k3_fixed"][i:i + 1], d["k3_moving"][i:i + 1], d["k3_corr"][i:i + 1], d["k3_guess"], mp)
            assert np.allclose(est[0], d["k3_solution"][i], atol=5e-7), (i, alpha, param, est[0], d["k3_solution"][i])
            assert np.isclose(res["final_cost"][0], d["k3_cost"][i], rtol=1e-8), (res["final_cost"][0], d["k3_cost"][i])
            assert res["gnc_solves"][0] == 1 and res["n_residuals"][0] == 40


def test_k4_numpy_float32_cell_statistics(env):
    torch, dev, ctx, d = env
    scans = d["k4_scans"]
    maps = R.Maps(ctx, len(scans), R.indoor_map_params(), 512, with_grid=True)
    R.ndt_build_batch(ctx, torch.from_numpy(scans).to(dev), R.indoor_cluster_params(), maps)
    ctx.synchronize()
    for s in range(len(scans)):
        cells, grid = maps.download(s)
        n = int(d["k4_n_cells"][s])
        ref = d["k4_cells"][s][:n]
        assert len(cells) == n and n > 30
        assert np.array_equal(grid, d["k4_grid"][s])                                  # slots + compact order
        assert np.array_equal(cells["n"], ref["n"])
        assert np.array_equal(cells["mean"].view(np.uint32), ref["mean"].view(np.uint32))      # sequential fp32 sums: bit exact
        assert np.array_equal(cells["max_intensity"], ref["max_intensity"])
        for e in (2, 4, 5):                                                           # xi, yi, ii: untouched by the xy regularisation
            assert np.array_equal(cells["cov"][:, e].view(np.uint32), ref["cov"][:, e].view(np.uint32)), e
        got = np.stack([np.stack([cells["cov"][:, 0], cells["cov"][:, 1]], 1), np.stack([cells["cov"][:, 1], cells["cov"][:, 3]], 1)], 1)
        assert np.allclose(got.astype(np.float64), d["k4_xy_ref"][s][:n], rtol=2e-5, atol=1e-9)


# ---- K9: the fixed-lag window path against an answer known by construction (no solver, no oracle).  The moving maps are
# the fixed cells carried EXACTLY through the inverse poses of a constant-velocity chain X_j = predict(X_{j-1}); at that chain
# every NDT residual and every motion residual is zero, so it is THE minimiser of the window objective.  Started from
# perturbed poses and velocities, the window solve has to come back to it: checks the association in front of the window, the
# NDT terms, the motion factors' zero set against randt_predict_state_param, Plus on every block and the LM loop -- for the
# tuned three-state kernel, the general kernel (lags > 3, and three-state windows routed to it) and all three block layouts.
def _lattice_cells(rng, pose3_list):
    cell = R.CELL_DTYPE
    xs, ys = np.meshgrid(np.arange(-7.0, 7.1, 2.0), np.arange(-5.0, 5.1, 2.0))
    n = xs.size
    fc = np.zeros(n, dtype=cell)
    S_all = []
    for i, (x, y) in enumerate(zip(xs.ravel(), ys.ravel())):
        A = rng.normal(0, 1, (3, 3)) * [0.1, 0.1, 2.0]
        S = A @ A.T + np.diag([1e-3, 1e-3, 1e-1])
        S_all.append(S)
        fc[i]["mean"] = [x + rng.uniform(-0.2, 0.2), y + rng.uniform(-0.2, 0.2), rng.uniform(20, 80)]
        fc[i]["cov"] = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
        fc[i]["n"] = 10
    moving = []
    for p in pose3_list:
        Rm = np.array([[np.cos(p[2]), -np.sin(p[2]), 0], [np.sin(p[2]), np.cos(p[2]), 0], [0, 0, 1]])
        t = np.array([p[0], p[1], 0.0])
        mc = np.zeros(n, dtype=cell)
        for i in range(n):
            mc[i]["mean"] = Rm.T @ (fc[i]["mean"].astype(np.float64) - t)
            Sm = Rm.T @ S_all[i] @ Rm
            mc[i]["cov"] = [Sm[0, 0], Sm[0, 1], Sm[0, 2], Sm[1, 1], Sm[1, 2], Sm[2, 2]]
            mc[i]["n"] = 10
        moving.append(mc)
    return fc, moving


@pytest.mark.parametrize("param", [R.PARAM_MANIFOLD, R.PARAM_VECTOR, R.PARAM_ANALYTIC])
@pytest.mark.parametrize("lag,general,full", [(1, False, False), (3, False, False), (3, True, False), (5, False, False), (7, False, False),
                                              (3, False, True), (6, False, True)])
def test_k9_zero_noise_window_returns_the_constant_velocity_chain(built, param, lag, general, full):
    """full: constant-acceleration layout (acceleration blocks, zero along the chain) + the IMU factor fed with the chain's own
    heading increments (bias zero): still an exact zero of every residual."""
    import torch

    old = os.environ.get("RANDT_WINDOW_GENERAL")
    if general:
        os.environ["RANDT_WINDOW_GENERAL"] = "1"
    try:
        ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    finally:
        if general:
            if old is None:
                del os.environ["RANDT_WINDOW_GENERAL"]
            else:
                os.environ["RANDT_WINDOW_GENERAL"] = old
    rng = np.random.default_rng(99 + lag)
    pform = R.PARAM_MANIFOLD if param == R.PARAM_MANIFOLD else R.PARAM_VECTOR
    dt = 0.25
    chain = [R.make_state(synth.pose3_to_pose4(np.array([0.3, -0.2, 0.1])), lin_vel=(0.8, 0.1), rot_vel=0.2, stamp=0.0)]
    for j in range(1, lag + 1):
        chain.append(R.predict_state(chain[-1], j * dt, pform))
    truth3 = [synth.pose4_to_pose3(np.array(s["pose"])) for s in chain]
    fc, moving = _lattice_cells(rng, truth3[1:])
    mapp = R.indoor_map_params()
    fm = R.Maps(ctx, 1, mapp, 64, with_grid=True)
    fm.upload(0, fc)
    fm.reindex()
    mm = R.Maps(ctx, lag, mapp, 64, with_grid=False)
    for j in range(lag):
        mm.upload(j, moving[j])
    # start: every optimised state off its true pose and velocity
    states = [chain[0]]
    for j in range(1, lag + 1):
        p = truth3[j] + np.array([0.08, -0.05, 0.02]) * (1 if j % 2 else -1)
        states.append(R.make_state(synth.pose3_to_pose4(p), lin_vel=(0.9, 0.05), rot_vel=0.25, stamp=j * dt))
    mp = R.default_matcher_params(parameterization=param, n_neighbours=1, gnc_steps=2, function_tolerance=1e-14, parameter_tolerance=1e-13)
    wp = R.window_params(use_imu=1, const_vel=0) if full else R.window_params()
    imu = np.array([synth.wrap_angle(truth3[j][2] - truth3[j - 1][2]) for j in range(1, lag + 1)]) if full else None
    out, trans, rej, res = R.register_window(ctx, fm, [0], mm, list(range(lag)), np.array(states, dtype=R.STATE_DTYPE), mp, wp,
                                             states[-1]["pose"], imu)
    assert not rej and res["status"] == 0 and res["n_residuals"] == lag * len(fc)
    for j in range(1, lag + 1):
        e = synth.pose4_to_pose3(np.array(out[j]["pose"])) - truth3[j]
        e[2] = (e[2] + np.pi) % (2 * np.pi) - np.pi
        # cells are float32 (means ~10 m: 1e-6 m quantisation); the objective is zero at the chain up to that
        assert np.abs(e[:2]).max() < 2e-5 and abs(e[2]) < 2e-6, (j, e)
        assert np.allclose(out[j]["lin_vel"], chain[0]["lin_vel"], atol=2e-4) and abs(out[j]["rot_vel"] - chain[0]["rot_vel"]) < 2e-4, (j, out[j])
        if full:
            assert np.abs(out[j]["lin_acc"]).max() < 2e-3 and abs(out[j]["imu_bias"]) < 1e-6, (j, out[j])
    assert res["final_cost"] < 1e-3


# ---- K5 on the device: k_associate against a brute-force numpy enumeration of Map::getClosestCells / getAdjacentIndizes
# (ndt_map.cpp:101-175: the square window grown ring by ring around the query's slot until k occupied slots are inside or
# the radius limit is hit, uint32 slot arithmetic without a row-wrap guard, candidates ranked by the fp32 metric), written
# from the reference's definition in this file -- the oracle is not involved.
def _brute_closest(cells, grid, size, res, offset, max_dist, q_mean, q_cov, k, metric):
    F = np.float32
    n_slots = size * size
    mx = int((np.float64(q_mean[0]) - offset) / res) & 0xFFFFFFFF
    my = int((np.float64(q_mean[1]) - offset) / res) & 0xFFFFFFFF
    center = (my * size + mx) & 0xFFFFFFFF
    rmax = int(max_dist / res)
    targets, nadj, radius = [], 0, 0
    while len(targets) < k and nadj < n_slots:
        targets, seen = [], []
        for i in range(-radius, radius + 1):
            for j in range(-radius, radius + 1):
                ni = (center + i + j * size) & 0xFFFFFFFF
                if ni < n_slots and ni not in seen:
                    seen.append(ni)
        nadj = len(seen)
        for ni in seen:
            ci = grid[ni]
            if ci >= 0:
                f = cells[ci]
                if metric:
                    S = np.zeros((3, 3))
                    for e, (a, b) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
                        S[a, b] = S[b, a] = np.float64(F(f["cov"][e]) + F(q_cov[e]))
                    mu = (f["mean"].astype(F) - q_mean.astype(F)).astype(np.float64)
                    d = float(mu @ np.linalg.inv(S) @ mu)
                else:
                    d = float(np.hypot(F(q_mean[0]) - F(f["mean"][0]), F(q_mean[1]) - F(f["mean"][1])))
                targets.append((d, int(ci)))
        radius += 1
        if radius >= rmax:
            break
    targets.sort()
    return targets[:k]


@pytest.mark.parametrize("placement", ["lone", "throughput"])
@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_k5_association_vs_numpy_brute_force(env, metric, seed, placement):
    """placement "throughput": RANDT_SOLVE_THROUGHPUT and six copies of the pair -- the register-capped instantiation whose
    workgroups walk four pairs (one full walk + a ragged one); every copy must give the same table."""
    torch, dev, ctx, _ = env
    n_copies = 6 if placement == "throughput" else 1
    F = np.float32
    rng = np.random.default_rng(500 + seed)
    size, res, max_dist, n_cells, k = 40, 0.5, 4.0, 300, 4
    offset = -size / 2.0 * res
    cells = np.zeros(n_cells, dtype=R.CELL_DTYPE)
    grid = np.full(size * size, -1, dtype=np.int32)
    for i, s in enumerate(rng.choice(size * size, n_cells, replace=False)):
        my, mx = divmod(int(s), size)
        cells[i]["mean"] = [(mx + rng.uniform(.1, .9)) * res + offset, (my + rng.uniform(.1, .9)) * res + offset, rng.uniform(20, 80)]
        A = rng.normal(0, 1, (3, 3)) * [0.1, 0.1, 3.0]
        S = A @ A.T + np.diag([1e-3, 1e-3, 1e-2])
        cells[i]["cov"] = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
        cells[i]["n"] = 10
        grid[s] = i
    n_q = 100
    mc = np.zeros(n_q, dtype=R.CELL_DTYPE)
    for i in range(n_q):
        lim = 11.0 if i % 5 else 9.9          # some queries near / beyond the map edge (row wrap, out-of-range centres)
        mc[i]["mean"] = [rng.uniform(-lim, lim), rng.uniform(-lim, lim), rng.uniform(20, 80)]
        A = rng.normal(0, 1, (3, 3)) * [0.1, 0.1, 3.0]
        S = A @ A.T + np.diag([1e-3, 1e-3, 1e-2])
        mc[i]["cov"] = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
        mc[i]["n"] = 8
    mapp = R.MapParams(size, size, res, 0.0, 0.0, max_dist, 5, 0)
    fm = R.Maps(ctx, 1, mapp, n_cells, with_grid=True)
    fm.upload(0, cells, grid)
    mm = R.Maps(ctx, n_copies, mapp, 128, with_grid=False)
    for j in range(n_copies):
        mm.upload(j, mc)
    th = 0.1
    pose4 = np.tile(np.array([[np.cos(th), np.sin(th), 0.3, -0.2]]), (n_copies, 1))
    mp = R.default_matcher_params(n_neighbours=k, lookup_mahalanobis=metric, use_intensity=1)
    corr = torch.full((n_copies, 128, k), -7, dtype=torch.int32, device=dev)
    if placement == "throughput":
        ctx.set_solve_mode(R._capi.SOLVE_THROUGHPUT)
    try:
        R.associate_batch(ctx, fm, torch.zeros(n_copies, dtype=torch.int32, device=dev), mm, 0, n_copies, torch.from_numpy(pose4).to(dev), mp, corr)
        ctx.synchronize()
    finally:
        ctx.set_solve_mode(R._capi.SOLVE_AUTO)
    got_copies = corr.cpu().numpy()
    for j in range(1, n_copies):
        assert np.array_equal(got_copies[j, :n_q], got_copies[0, :n_q]), j
    got_all = got_copies[0]
    # the query as the reference forms it: float affine of the guess (transformMap / transformCell, ndt_matcher.cpp:203-209)
    c, s, tx, ty = F(np.cos(th)), F(np.sin(th)), F(0.3), F(-0.2)
    Rm = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    mism = 0
    for i in range(n_q):
        x, y = F(mc[i]["mean"][0]), F(mc[i]["mean"][1])
        qm = np.array([(c * x - s * y) + tx, (s * x + c * y) + ty, mc[i]["mean"][2]], dtype=F)
        cv = mc[i]["cov"].astype(np.float64)
        Sq = Rm @ np.array([[cv[0], cv[1], cv[2]], [cv[1], cv[3], cv[4]], [cv[2], cv[4], cv[5]]]) @ Rm.T
        qc = np.array([Sq[0, 0], Sq[0, 1], Sq[0, 2], Sq[1, 1], Sq[1, 2], Sq[2, 2]], dtype=F)
        ref = _brute_closest(cells, grid, size, res, offset, max_dist, qm, qc, k, metric)
        got = [int(v) for v in got_all[i] if v >= 0]
        assert len(got) == len(ref), (i, got, ref)
        if got != [r[1] for r in ref]:
            mism += 1                        # fp32 (device) against fp64 (brute force) distances: near-ties may swap, nothing else
            assert sorted(got) == sorted(r[1] for r in ref) or len(ref) == k, (i, got, ref)
    assert mism <= 3, mism


# ---- f-2 on the device: Map::calculateCSDivergence (ndt_map.cpp:36-99) against the definition evaluated in numpy float64
# (pair term 0.5 / sqrt(pi^2 det(Sf + Sq)) exp(-d^T (Sf + Sq)^-1 d / 2); a map's own term sums sqrt(det(S^-1)) / (2 pi) and
# twice the pair terms with every EARLIER cell, over the cells whose det(S) >= 1e-5; CS = -log I + log(F) / 2 + log(M) / 2).
def test_cs_divergence_vs_numpy_definition(env):
    from randt_slam_amd import host

    torch, dev, ctx, _ = env
    rng = np.random.default_rng(77)

    def rand_cells(n, centres=None):
        c = np.zeros(n, dtype=R.CELL_DTYPE)
        for i in range(n):
            base = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(20, 80)]) if centres is None else \
                centres[i % len(centres)] + np.array([rng.uniform(-.3, .3), rng.uniform(-.3, .3), rng.uniform(-3, 3)])
            A = rng.normal(0, 1, (3, 3)) * [0.15, 0.15, 2.0]
            S = A @ A.T + np.diag([1e-3 if i % 7 else 1e-5, 1e-3 if i % 7 else 1e-5, 1e-2 if i % 7 else 1e-4])
            if i % 7 == 0:
                S *= 1e-2                                   # a few nearly degenerate cells: the det(S) < 1e-5 gate
            c[i]["mean"], c[i]["cov"], c[i]["n"] = base, [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]], 9
        return c

    fc = rand_cells(60)
    mc = rand_cells(50, centres=[f["mean"].astype(np.float64) for f in fc])

    def full(c):
        c = c.astype(np.float64)
        return np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]])

    def pair(a, b):
        S = full(a["cov"]) + full(b["cov"])
        d = a["mean"].astype(np.float64) - b["mean"].astype(np.float64)
        return 0.5 / np.sqrt(np.pi ** 2 * np.linalg.det(S)) * np.exp(-0.5 * d @ np.linalg.solve(S, d))

    def own(cells):
        t = 0.0
        for i, a in enumerate(cells):
            det = np.linalg.det(full(a["cov"]))
            assert abs(det / 1e-5 - 1.0) > 0.05            # no cell sits on the gate, where fp32 and fp64 could disagree
            if det < 1e-5:
                continue
            t += np.sqrt(np.linalg.det(np.linalg.inv(full(a["cov"])))) / (2 * np.pi)
            t += sum(2 * pair(a, cells[j]) for j in range(i))
        return t

    inter = sum(pair(a, b) for a in fc if np.linalg.det(full(a["cov"])) >= 1e-5 for b in mc)
    ref_terms = np.array([inter, own(fc), own(mc)])
    ref = -np.log(inter) + 0.5 * np.log(ref_terms[1]) + 0.5 * np.log(ref_terms[2])
    mapp = R.indoor_map_params()
    fm = R.Maps(ctx, 1, mapp, 64, with_grid=True)
    fm.upload(0, fc)
    fm.reindex()
    mm = R.Maps(ctx, 1, mapp, 64, with_grid=False)
    mm.upload(0, mc)
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    terms = torch.zeros((1, 3), dtype=torch.float64, device=dev)
    ident = torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=torch.float64, device=dev)
    host.cs_divergence_batch(ctx, fm, 0, 1, torch.zeros(1, dtype=torch.int32, device=dev), mm, 0, 1, ident, out, terms)
    ctx.synchronize()
    assert sum(np.linalg.det(full(a["cov"])) < 1e-5 for a in fc) >= 3
    got = terms.cpu().numpy()[0]
    assert np.isclose(got[0], ref_terms[0], rtol=3e-5), (got, ref_terms)                # fp32 pair terms (Sf + Sq is well conditioned)
    # the maps' own terms hold sqrt(det(S^-1)) of single cells: the reference's float cofactor determinant of a covariance
    # with condition number ~1e3 carries ~1e-3 relative error against float64
    assert np.allclose(got[1:], ref_terms[1:], rtol=3e-3), (got, ref_terms)
    assert np.isclose(out.cpu().numpy()[0], ref, rtol=0, atol=3e-3)
