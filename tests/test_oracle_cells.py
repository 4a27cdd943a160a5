"""CPU tests pinning the oracle's cell statistics / clustering / merge against
independent numpy re-derivations.  The reference ships no tests, so these are the pins."""
import numpy as np
import pytest

import pyoracle as po
from util import IP, oracle_map

F = np.float32


def np_cell_stats(pts, ioff=3):
    """Sequential two-pass fp32 mean / population covariance exactly as ndt_cell.cpp:43-65 reads."""
    m = np.zeros(3, dtype=F)
    for p in pts:
        m = m + np.array([p[0], p[1], p[ioff]], dtype=F)
    n = F(len(pts))
    m = (m / n).astype(F)
    c = np.zeros(6, dtype=F)
    for p in pts:
        d = np.array([p[0], p[1], p[ioff]], dtype=F) - m
        c[0] += d[0] * d[0]; c[3] += d[1] * d[1]; c[5] += d[2] * d[2]
        c[1] += d[0] * d[1]; c[2] += d[0] * d[2]; c[4] += d[1] * d[2]
    return m, (c / n).astype(F)


def test_grid_labels_truncation_and_negative_labels(built):
    rng = np.random.default_rng(0)
    pts = np.zeros((4000, 4), dtype=F)
    pts[:, :2] = rng.uniform(-12, 12, (4000, 2))
    labels = po.grid_labels(pts, IP["n_clusters"], IP["max_range"])
    row = int(np.sqrt(IP["n_clusters"]))
    res = F(IP["max_range"]) * F(2) / F(row)
    exp = np.trunc(pts[:, 0] / res).astype(np.int32) + row * np.trunc(pts[:, 1] / res).astype(np.int32)
    assert row == 48 and res == F(0.5)
    assert np.array_equal(labels, exp)
    assert labels.min() < 0  # C truncation toward zero: negative labels, double-width cells on the axes
    a = po.grid_labels(np.array([[0.4, 0.4, 0, 0], [-0.4, -0.4, 0, 0]], dtype=F), IP["n_clusters"], IP["max_range"])
    assert a[0] == a[1] == 0


@pytest.mark.parametrize("k", [6, 9, 33, 200])
def test_cell_statistics_match_numpy_fp32(built, k):
    rng = np.random.default_rng(k)
    pts = np.zeros((k, 4), dtype=F)
    pts[:, 0] = rng.normal(3.0, 0.1, k)
    pts[:, 1] = rng.normal(-2.0, 0.05, k)
    pts[:, 3] = rng.uniform(20, 80, k)
    ok, cell = po.cell_from_points(pts, min_points=5)
    assert ok and cell["n"] == k
    m, c = np_cell_stats(pts)
    assert np.array_equal(cell["mean"], m)                       # bit exact
    assert cell["cov"][2] == c[2] and cell["cov"][4] == c[4]      # xi, yi untouched by the regularisation
    assert cell["cov"][5] == F(np.float64(c[5]) + 0.000001)       # += 0.000001 in double
    assert cell["max_intensity"] == pts[:, 3].max()
    # xy block: V diag(max(l0, 0.001 l1), l1) V^-1 against float64 eigh
    S = np.array([[c[0], c[1]], [c[1], c[3]]], dtype=np.float64)
    w, V = np.linalg.eigh(S)
    w[0] = max(w[0], 0.001 * w[1])
    ref = V @ np.diag(w) @ V.T
    got = np.array([[cell["cov"][0], cell["cov"][1]], [cell["cov"][1], cell["cov"][3]]], dtype=np.float64)
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-9)


def test_regularisation_of_degenerate_cluster(built):
    # collinear points: smallest eigenvalue is clamped to 0.001 * largest (ndt_cell.cpp:107)
    k = 12
    pts = np.zeros((k, 4), dtype=F)
    t = np.linspace(-0.2, 0.2, k)
    pts[:, 0] = 1.0 + t * np.cos(0.3)
    pts[:, 1] = 2.0 + t * np.sin(0.3)
    pts[:, 3] = 50
    ok, cell = po.cell_from_points(pts)
    S = np.array([[cell["cov"][0], cell["cov"][1]], [cell["cov"][1], cell["cov"][3]]], dtype=np.float64)
    w = np.linalg.eigvalsh(S)
    assert ok and w[0] > 0 and np.isclose(w[0] / w[1], 0.001, rtol=1e-3)
    assert cell["cov"][5] == F(1e-6)  # zero intensity variance + 1e-6


def test_min_points_gate(built):
    pts = np.random.default_rng(1).normal(0, 1, (6, 4)).astype(F)
    assert not po.cell_from_points(pts[:5], min_points=5)[0]   # n > min_points (ndt_cell.cpp:26)
    assert po.cell_from_points(pts[:6], min_points=5)[0]


def test_merge_uses_integer_division(built):
    rng = np.random.default_rng(2)
    a = po.cell_from_points(np.c_[rng.normal(0, .1, (7, 2)), np.zeros(7), rng.uniform(20, 80, 7)].astype(F))[1]
    b = po.cell_from_points(np.c_[rng.normal(.2, .1, (6, 2)), np.zeros(6), rng.uniform(20, 80, 6)].astype(F))[1]
    got = po.cell_merge(a, b)
    n, m = 7, 6
    wk = (n * m) // (n + m)           # 42 // 13 = 3, not 3.23 (ndt_cell.h:137)
    assert wk == 3
    d = (a["mean"] - b["mean"]).astype(np.float64)
    idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    ref = np.array([((n - 1) * float(a["cov"][e]) + (m - 1) * float(b["cov"][e]) + wk * d[i] * d[j]) / (n + m - 1)
                    for e, (i, j) in enumerate(idx)])
    assert got["n"] == 13
    assert np.allclose(got["cov"], ref, rtol=1e-5, atol=1e-9)
    assert np.allclose(got["mean"], (a["mean"].astype(float) * n + b["mean"].astype(float) * m) / (n + m), rtol=1e-6)
    # the real-valued weight would give a measurably different covariance
    ref_real = np.array([((n - 1) * float(a["cov"][e]) + (m - 1) * float(b["cov"][e]) + (n * m / (n + m)) * d[i] * d[j]) / (n + m - 1)
                         for e, (i, j) in enumerate(idx)])
    assert not np.allclose(got["cov"], ref_real, rtol=1e-5, atol=1e-9)


def test_build_orders_cells_by_label_and_keeps_point_order(built):
    rng = np.random.default_rng(3)
    centers = np.array([[3.25, 1.25], [-4.75, 2.25], [1.25, -6.75], [-2.75, -3.25]])
    pts = []
    for _ in range(12):
        for c in centers:
            pts.append([c[0] + rng.uniform(-.2, .2), c[1] + rng.uniform(-.2, .2), 0, rng.uniform(20, 80)])
    pts = np.array(pts, dtype=F)
    m = oracle_map(64)
    assert m.build(pts, IP["n_clusters"], IP["max_range"]) == 4
    labels = po.grid_labels(pts, IP["n_clusters"], IP["max_range"])
    cells = m.cells()
    for ci, lab in enumerate(sorted(set(labels.tolist()))):     # ascending label = std::map order
        sel = pts[labels == lab]                                  # input order inside the cluster
        mean, _ = np_cell_stats(sel)
        assert np.array_equal(cells[ci]["mean"], mean)
        slot = m.coord_to_index(mean[0], mean[1])
        assert m.grid()[slot] == ci


def test_slot_collision_last_writer_wins_both_cells_kept(built):
    # outdoor-like set-up: clustering grid (1.2308 m) and map grid (1.2 m) are misaligned, so two
    # clusters can land in one map slot (quirk A.7-5): the later one owns the slot, both stay in grid_.
    m = po.Map(41, 41, 1.2, (0, 0), 4.0, 3, 256)
    rng = np.random.default_rng(4)
    n_clusters, max_range = int((2 * 16 / 1.2) ** 2), 16.0
    # cluster cells [1.2308, 2.4615) and [2.4615, ...) in x; map slot [1.2+k*1.2): put means at ~2.41 and ~2.47
    a = np.c_[rng.uniform(2.38, 2.44, 8), rng.uniform(0.3, 0.5, 8), np.zeros(8), rng.uniform(20, 80, 8)]
    b = np.c_[rng.uniform(2.465, 2.50, 8), rng.uniform(0.3, 0.5, 8), np.zeros(8), rng.uniform(20, 80, 8)]
    pts = np.vstack([a, b]).astype(F)
    labels = po.grid_labels(pts, n_clusters, max_range)
    assert len(set(labels.tolist())) == 2
    assert m.build(pts, n_clusters, max_range) == 2
    cells = m.cells()
    s0 = m.coord_to_index(*cells[0]["mean"][:2]); s1 = m.coord_to_index(*cells[1]["mean"][:2])
    assert s0 == s1 and m.grid()[s0] == 1 and (m.grid() >= 0).sum() == 1


def test_transform_then_merge_inserts_and_merges(built):
    rng = np.random.default_rng(5)
    def blob(cx, cy, n=10):
        return np.c_[rng.normal(cx, .05, n), rng.normal(cy, .05, n), np.zeros(n), rng.uniform(20, 80, n)].astype(F)
    sub = oracle_map()
    s1 = oracle_map(16); s1.build(np.vstack([blob(1.25, 1.25), blob(5.25, -3.25)]), IP["n_clusters"], IP["max_range"])
    s2 = oracle_map(16); s2.build(np.vstack([blob(1.25, 1.25), blob(-2.75, 2.25)]), IP["n_clusters"], IP["max_range"])
    sub.merge(s1)
    assert sub.n_cells == 2
    sub.merge(s2)
    assert sub.n_cells == 3                       # one merged (same slot), one inserted
    assert sorted(sub.cells()["n"].tolist()) == [10, 10, 20]
    # transformMap leaves the index grid stale (quirk A.7-7)
    g0 = sub.grid().copy()
    sub.transform([np.cos(0.5), np.sin(0.5), 3.0, 1.0])
    assert np.array_equal(g0, sub.grid())
