"""-m gpu: the single-cell ABI behind the facade's Cell mutators (randt_cell_add_points, randt_cells_merge,
randt_cells_transform, randt_cells_mahalanobis) against the oracle's restatement of Cell::addPointCloud / updateCell /
operator+= / transformCell / mahalanobisSquared{,Intensity} (ndt_cell.cpp:25-176, ndt_cell.h:133-142): bit-exact fp32."""
import numpy as np
import pytest

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import host

pytestmark = pytest.mark.gpu
CELL = R.CELL_DTYPE


def _bits(a):
    return [np.asarray(a[f]).view(np.uint32).tolist() for f in ("mean", "cov", "n", "max_intensity")]


@pytest.fixture(scope="module")
def ctx(built):
    import torch

    return R.Context(0, torch.cuda.current_stream().cuda_stream)


def _cloud(rng, n, cx, cy, i0, s=0.05):
    p = np.zeros((n, 4), dtype=np.float32)
    p[:, 0] = cx + rng.normal(0, s, n)
    p[:, 1] = cy + rng.normal(0, 2 * s, n)
    p[:, 3] = i0 + rng.normal(0, 3, n)
    return p


def test_add_points_first_fill_gate_and_recursive_update(ctx):
    rng = np.random.default_rng(3)
    empty = np.zeros(1, dtype=CELL)[0]
    for n in (3, 5, 6, 40, 700):
        pts = _cloud(rng, n, 1.3, -2.1, 44.0)
        acc, cell = host.cell_add_points(ctx, empty, pts, 5)
        ok, ocell = po.cell_update(empty, pts, 5)
        assert acc == ok == (n > 5)
        assert _bits(cell) == _bits(ocell)
        if acc:                                           # == the one-voxel build path
            ok2, c2 = po.cell_from_points(pts, 5)
            assert ok2 and _bits(c2) == _bits(ocell)
        # a second cloud on the filled (or still empty) cell: recursive update + regularisation again
        more = _cloud(rng, 9, 1.35, -2.0, 60.0)
        acc2, cell2 = host.cell_add_points(ctx, cell, more, 5)
        okb, ocell2 = po.cell_update(ocell, more, 5)
        assert acc2 == okb and _bits(cell2) == _bits(ocell2)
        if n > 5:
            assert cell2["n"] == n + 9
    # PCL stride
    pts = _cloud(rng, 20, 0.2, 0.3, 30.0)
    pcl = np.zeros((20, 8), dtype=np.float32)
    pcl[:, :2], pcl[:, 4] = pts[:, :2], pts[:, 3]
    a, ca = host.cell_add_points(ctx, empty, pcl, 5)
    b, cb = host.cell_add_points(ctx, empty, pts, 5)
    assert a and b and _bits(ca) == _bits(cb)


def test_merge_transform_mahalanobis_bit_exact(ctx):
    rng = np.random.default_rng(4)
    empty = np.zeros(1, dtype=CELL)[0]
    cells = []
    for i in range(40):
        ok, c = po.cell_update(empty, _cloud(rng, int(rng.integers(6, 60)), rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(20, 80),
                                             s=rng.uniform(0.02, 0.3)), 5)
        assert ok
        cells.append(c)
    a, b = np.array(cells[:20], dtype=CELL), np.array(cells[20:], dtype=CELL)
    merged = host.cells_merge(ctx, a, b)
    for i in range(20):
        assert _bits(merged[i]) == _bits(po.cell_merge(a[i], b[i]))
    for th in (0.0, 0.3, -2.9):
        pose = [np.cos(th), np.sin(th), 1.5, -0.75]
        moved = host.cells_transform(ctx, a, pose)
        for i in range(20):
            assert _bits(moved[i]) == _bits(po.cell_transform(a[i], pose))
    for use_i in (True, False):
        d = host.cells_mahalanobis(ctx, a, b, use_i)
        ref = np.array([po.cell_mahalanobis(a[i], b[i], use_i) for i in range(20)])
        assert np.array_equal(d, ref) and (d > 0).all()
    # zero-length calls are fine
    assert len(host.cells_merge(ctx, a[:0], b[:0])) == 0
