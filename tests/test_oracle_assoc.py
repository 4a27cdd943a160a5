"""K5: the oracle's window association vs an independent brute-force numpy re-implementation of
Map::getClosestCells / getAdjacentIndizes (ndt_map.cpp:101-175), including the no-row-wrap quirk."""
import numpy as np
import pytest

import pyoracle as po

F = np.float32


def random_map(rng, size=40, res=0.5, n_cells=300, max_dist=4.0):
    m = po.Map(size, size, res, (0, 0), max_dist, 5, size * size)
    cells = np.zeros(n_cells, dtype=po.CELL_DTYPE)
    grid = np.full(size * size, -1, dtype=np.int32)
    slots = rng.choice(size * size, n_cells, replace=False)
    for i, s in enumerate(slots):
        my, mx = divmod(int(s), size)
        cells[i]["mean"] = [(mx + rng.uniform(.1, .9)) * res - size / 2 * res, (my + rng.uniform(.1, .9)) * res - size / 2 * res,
                            rng.uniform(20, 80)]
        A = rng.normal(0, 1, (3, 3)) * [0.1, 0.1, 3.0]
        S = A @ A.T + np.diag([1e-3, 1e-3, 1e-2])
        cells[i]["cov"] = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
        cells[i]["n"] = 10
        grid[s] = i
    m.set(cells, grid)
    return m


def brute(m, q, k, metric):
    """q: transformed query cell (CELL_DTYPE scalar)."""
    c = m.c
    size_x, n_slots = c.size_x, c.size_x * c.size_y
    cells, grid = m.cells(), m.grid()
    mx = int((np.float64(q["mean"][0]) - c.offset_x) / c.res) & 0xFFFFFFFF
    my = int((np.float64(q["mean"][1]) - c.offset_y) / c.res) & 0xFFFFFFFF
    center = (my * size_x + mx) & 0xFFFFFFFF
    rmax = int(c.max_neighbour_dist / c.res)
    targets, nadj, radius = [], 0, 0
    while len(targets) < k and nadj < n_slots:
        targets, seen = [], []
        for i in range(-radius, radius + 1):
            for j in range(-radius, radius + 1):
                ni = (center + i + j * size_x) & 0xFFFFFFFF
                if ni < n_slots and ni not in seen:
                    seen.append(ni)
        nadj = len(seen)
        for ni in seen:
            ci = grid[ni]
            if ci >= 0:
                f = cells[ci]
                if metric:
                    S = np.zeros((3, 3), dtype=F)
                    idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
                    for e, (a, b) in enumerate(idx):
                        S[a, b] = S[b, a] = F(f["cov"][e]) + F(q["cov"][e])
                    mu = (f["mean"] - q["mean"]).astype(F)
                    d = float(mu.astype(np.float64) @ np.linalg.inv(S.astype(np.float64)) @ mu.astype(np.float64))
                else:
                    d = float(np.hypot(F(q["mean"][0]) - F(f["mean"][0]), F(q["mean"][1]) - F(f["mean"][1])))
                targets.append((d, int(ci)))
        radius += 1
        if radius >= rmax:
            break
    targets.sort()
    return targets[:k]


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("seed", [0, 1])
def test_association_vs_bruteforce(built, metric, seed):
    rng = np.random.default_rng(seed)
    fixed = random_map(rng)
    moving = po.Map(40, 40, 0.5, (0, 0), 4.0, 5, 128)
    mc = np.zeros(100, dtype=po.CELL_DTYPE)
    for i in range(100):
        # include queries near / beyond the map edge (row-wrap and out-of-range centres)
        lim = 11.0 if i % 5 else 9.9
        mc[i]["mean"] = [rng.uniform(-lim, lim), rng.uniform(-lim, lim), rng.uniform(20, 80)]
        A = rng.normal(0, 1, (3, 3)) * [0.1, 0.1, 3.0]
        S = A @ A.T + np.diag([1e-3, 1e-3, 1e-2])
        mc[i]["cov"] = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
        mc[i]["n"] = 8
    moving.set(mc, np.full(1600, -1, dtype=np.int32))
    pose = np.array([np.cos(0.1), np.sin(0.1), 0.3, -0.2])
    k = 4
    corr, n = po.associate(fixed, moving, pose, k, lookup_mahalanobis=metric, use_intensity=1)
    assert n == (corr >= 0).sum()
    mism = 0
    for i in range(100):
        if metric:
            q = po.cell_transform(mc[i], pose)
        else:
            q = mc[i].copy()
            aff = po.pose_to_affine_f(pose)
            x, y = F(mc[i]["mean"][0]), F(mc[i]["mean"][1])
            q["mean"][0] = (aff[0] * x - aff[1] * y) + aff[2]
            q["mean"][1] = (aff[1] * x + aff[0] * y) + aff[3]
        ref = brute(fixed, q, k, metric)
        got = [c for c in corr[i] if c >= 0]
        assert len(got) == len(ref)
        if got != [r[1] for r in ref]:
            # fp32 (oracle) vs fp64 (brute force) distances may swap near-ties only
            mism += 1
            assert sorted(got) == sorted(r[1] for r in ref) or len(ref) == k
    assert mism <= 3


def test_search_radius_is_bounded(built):
    # a lone far-away cell is never associated: the last evaluated radius is rmax-1 (= 7 for 4.0/0.5)
    fixed = po.Map(100, 100, 0.5, (0, 0), 4.0, 5, 16)
    cells = np.zeros(1, dtype=po.CELL_DTYPE)
    cells[0]["mean"] = [4.25, 0.25, 50]; cells[0]["cov"] = [.01, 0, 0, .01, 0, 1]; cells[0]["n"] = 9
    grid = np.full(10000, -1, dtype=np.int32)
    grid[fixed.coord_to_index(4.25, 0.25)] = 0
    fixed.set(cells, grid)
    def query(x):
        mv = po.Map(100, 100, 0.5, (0, 0), 4.0, 5, 4)
        q = cells.copy(); q[0]["mean"] = [x, 0.25, 50]
        mv.set(q, np.full(10000, -1, dtype=np.int32))
        return po.associate(fixed, mv, [1, 0, 0, 0], 4)[0][0]
    assert query(0.75)[0] == 0       # 7 slots away: inside the r = 7 window
    assert query(0.25)[0] == -1      # 8 slots away: never reached
