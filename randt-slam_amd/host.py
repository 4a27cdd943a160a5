"""Host-side wrappers over the C ABI (include/randt.h).

Device memory is whatever the caller hands over: any object with ``data_ptr()`` (torch CUDA
tensors: PyTorch-ROCm is used as the device allocator / stream provider only) or a raw integer
device address.  Batch entry points are asynchronous on the context's HIP stream, exactly like the
``*_dev`` functions they bind.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (CELL_DTYPE, RESULT_DTYPE, STATE_DTYPE, BnbParams, ClusterParams, FilterParams, MapParams, MatcherParams, PgParams, PgResult,
                    ScParams, WindowParams)


class RandtError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = f"{where}: {_capi.load().randt_status_string(status).decode()} (status {status})"
        if detail:
            msg += f" -- {detail}"
        super().__init__(msg)


def _dptr(x):
    """Device (or host) address of x: torch tensor -> data_ptr(), numpy -> ctypes address, int, None."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(x))


def default_matcher_params(**over):
    """randt_matcher_params_default(): indoor loop-closure refinement values + Ceres 2.1.0 defaults."""
    p = MatcherParams()
    _capi.load().randt_matcher_params_default(C.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def indoor_map_params(center=(0.0, 0.0)):
    """config/parameters_indoor.yaml:16-18 + base yaml :59-60 through ndt_slam.cpp:653-654."""
    from .synth import indoor_params

    ip = indoor_params()
    return MapParams(ip["size_x"], ip["size_y"], ip["resolution"], center[0], center[1], ip["max_neighbour_dist"],
                     ip["min_points_per_cell"], 0)


def indoor_cluster_params():
    """ndt_slam.cpp:691: n_clusters = (2 * max_range / resolution)^2."""
    from .synth import indoor_params

    ip = indoor_params()
    return ClusterParams(ip["n_clusters"], ip["max_range"])


class Context:
    """randt_ctx: one per device / caller thread.  ``stream``: a hipStream_t handle (int), e.g.
    ``torch.cuda.current_stream().cuda_stream``; None = the null stream.  ``solve_mode``: _capi.SOLVE_THROUGHPUT for a
    caller that keeps several batches in flight on several contexts / streams of one GPU -- the default (SOLVE_AUTO) gives a
    lone small batch several wavefronts per registration, which assumes an otherwise idle device and costs ~2x throughput
    when it is not (set_solve_mode changes it later)."""

    def __init__(self, device=0, stream=None, solve_mode=None):
        self._lib = _capi.load()
        h = C.c_void_p()
        rc = self._lib.randt_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc:
            raise RandtError(rc, "randt_ctx_create")
        self._h = h
        self.device = device
        if solve_mode is not None:
            self.set_solve_mode(solve_mode)

    def _check(self, rc, where):
        if rc:
            raise RandtError(rc, where, self._lib.randt_last_error(self._h).decode())

    def set_stream(self, stream):
        self._check(self._lib.randt_ctx_set_stream(self._h, C.c_void_p(stream) if stream else None), "randt_ctx_set_stream")

    def synchronize(self):
        self._check(self._lib.randt_ctx_synchronize(self._h), "randt_ctx_synchronize")

    def set_solve_mode(self, mode):
        """randt_ctx_set_solve_mode: _capi.SOLVE_AUTO (default: small batches get several wavefronts per registration) or
        _capi.SOLVE_THROUGHPUT (always one wavefront each: several batches in flight on several contexts)."""
        self._check(self._lib.randt_ctx_set_solve_mode(self._h, int(mode)), "randt_ctx_set_solve_mode")

    def pool_stats(self):
        """randt_ctx_pool_stats: allocator / synchronisation counters of the context and what its storage pool holds."""
        st = _capi.PoolStats()
        self._check(self._lib.randt_ctx_pool_stats(self._h, C.byref(st)), "randt_ctx_pool_stats")
        return {k: int(getattr(st, k)) for k, _ in _capi.PoolStats._fields_ if k != "reserved"}

    def pool_trim(self):
        self._check(self._lib.randt_ctx_pool_trim(self._h), "randt_ctx_pool_trim")

    def set_trace(self, d_trace, max_len):
        self._check(self._lib.randt_ctx_set_trace(self._h, _dptr(d_trace), int(max_len)), "randt_ctx_set_trace")

    @classmethod
    def _borrowed(cls, handle, device):
        """A context owned by someone else (a Group member): same calls, never destroyed from here."""
        self = cls.__new__(cls)
        self._lib = _capi.load()
        self._h = C.c_void_p(handle)
        self.device = device
        self._borrow = True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrow", False):
                self._lib.randt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Maps:
    """randt_maps: a batch of device-resident NDT maps (compact 48-byte cells + int32 index grid).

    storage=(cells, counts, grid): externally owned device buffers (e.g. torch uint8 / int32 tensors
    so that torch.distributed can broadcast the submap tables); otherwise hipMalloc'ed by the library.
    """

    def __init__(self, ctx, n_maps, params, capacity, with_grid=True, storage=None, clear=True):
        self.ctx = ctx
        self._lib = ctx._lib
        self.n_maps, self.capacity, self.params = int(n_maps), int(capacity), params
        self.n_slots = params.size_x * params.size_y
        self.with_grid = bool(with_grid)
        self._storage = storage
        h = C.c_void_p()
        if storage is None:
            rc = self._lib.randt_maps_create(ctx._h, self.n_maps, C.byref(params), self.capacity, int(self.with_grid), C.byref(h))
            where = "randt_maps_create"
        else:
            cells, counts, grid = storage
            self.with_grid = grid is not None
            rc = self._lib.randt_maps_create_external(ctx._h, self.n_maps, C.byref(params), self.capacity, _dptr(cells),
                                                      _dptr(counts), _dptr(grid), C.byref(h))
            where = "randt_maps_create_external"
        ctx._check(rc, where)
        self._h = h
        if storage is not None and clear:
            self.clear()

    @staticmethod
    def storage_bytes(n_maps, params, capacity):
        lib = _capi.load()
        return lib.randt_maps_cells_bytes(n_maps, capacity), 4 * n_maps, lib.randt_maps_grid_bytes(n_maps, C.byref(params))

    def close(self):
        if getattr(self, "_h", None):
            # at interpreter shutdown the context may already be gone: leak rather than touch it
            if getattr(self.ctx, "_h", None):
                self._lib.randt_maps_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self, first=0, count=None):
        count = self.n_maps - first if count is None else count
        self.ctx._check(self._lib.randt_maps_clear(self._h, first, count), "randt_maps_clear")

    def upload(self, idx, cells, grid=None):
        cells = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
        g = None if grid is None else np.ascontiguousarray(grid, dtype=np.int32)
        if g is not None and g.size != self.n_slots:
            raise ValueError("grid size")
        self.ctx._check(self._lib.randt_maps_upload(self._h, idx, _dptr(cells), len(cells), _dptr(g)), "randt_maps_upload")

    def download(self, idx):
        cells = np.zeros(self.capacity, dtype=CELL_DTYPE)
        grid = np.empty(self.n_slots, dtype=np.int32) if self.with_grid else None
        n = C.c_int(0)
        self.ctx._check(self._lib.randt_maps_download(self._h, idx, _dptr(cells), self.capacity, C.byref(n), _dptr(grid)),
                        "randt_maps_download")
        return cells[: min(n.value, self.capacity)].copy(), grid

    def download_with_status(self, idx):
        """randt_maps_download without raising: (status, cells, grid).  The status of EARLIER asynchronous inserts is reported
        here once, with the outputs valid (include/randt.h, "deferred status")."""
        cells = np.zeros(self.capacity, dtype=CELL_DTYPE)
        grid = np.empty(self.n_slots, dtype=np.int32) if self.with_grid else None
        n = C.c_int(-1)
        rc = self._lib.randt_maps_download(self._h, idx, _dptr(cells), self.capacity, C.byref(n), _dptr(grid))
        return int(rc), cells[: max(0, min(n.value, self.capacity))].copy(), grid

    def counts(self, first=0, count=None):
        count = self.n_maps - first if count is None else count
        out = np.zeros(count, dtype=np.int32)
        self.ctx._check(self._lib.randt_maps_counts(self._h, first, count, _dptr(out)), "randt_maps_counts")
        return out

    def clone(self, first=0, count=None):
        """randt_maps_clone: a new library-owned batch holding a copy of maps [first, first + count) (Map's copy constructor)."""
        count = self.n_maps - first if count is None else count
        h = C.c_void_p()
        self.ctx._check(self._lib.randt_maps_clone(self._h, first, count, C.byref(h)), "randt_maps_clone")
        out = Maps.__new__(Maps)
        out.ctx, out._lib = self.ctx, self._lib
        out.n_maps, out.capacity, out.params, out.n_slots = int(count), self.capacity, self.params, self.n_slots
        out.with_grid, out._storage, out._h = self.with_grid, None, h
        return out

    def copy_from(self, src, dst_first=0, src_first=0, count=None):
        count = src.n_maps - src_first if count is None else count
        self.ctx._check(self._lib.randt_maps_copy(self._h, dst_first, src._h, src_first, count), "randt_maps_copy")

    def transform(self, first, poses4):
        p = np.ascontiguousarray(poses4, dtype=np.float64).reshape(-1, 4)
        self.ctx._check(self._lib.randt_maps_transform(self._h, first, len(p), _dptr(p)), "randt_maps_transform")

    def reindex(self, first=0, count=None):
        """randt_maps_reindex: rebuild the index grid from the cells' current means (not in the reference)."""
        count = self.n_maps - first if count is None else count
        self.ctx._check(self._lib.randt_maps_reindex(self._h, first, count), "randt_maps_reindex")

    def merge(self, fixed_idx, moving, moving_first, poses4):
        """Rolling-submap update: merge moving maps [moving_first, +len(poses4)) into self[fixed_idx]."""
        p = np.ascontiguousarray(poses4, dtype=np.float64).reshape(-1, 4)
        self.ctx._check(self._lib.randt_maps_merge(self._h, fixed_idx, moving._h, moving_first, len(p), _dptr(p)),
                        "randt_maps_merge")

    def merge_batch(self, fixed_first, n_fixed, moving, moving_first, poses4):
        """randt_maps_merge_batch: self[fixed_first + p] receives moving[moving_first + p * m + t] at poses4[p * m + t], one launch."""
        p = np.ascontiguousarray(poses4, dtype=np.float64).reshape(-1, 4)
        self.ctx._check(self._lib.randt_maps_merge_batch(self._h, fixed_first, n_fixed, moving._h, moving_first, len(p) // max(1, n_fixed), _dptr(p)),
                        "randt_maps_merge_batch")

    def insert_clusters(self, idx, points, offsets, intensity_index=None, wait=True):
        """randt_maps_insert_clusters: HierarchicalMap::addClusters, every cluster of a list in one launch.  points (n, stride),
        offsets (n_clusters + 1).  Returns the number of accepted clusters (wait=False: asynchronous, returns None)."""
        pts = np.ascontiguousarray(points, dtype=np.float32)
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        stride = int(pts.shape[1])
        ioff = (3 if stride == 4 else 4) if intensity_index is None else intensity_index
        acc = C.c_int(0)
        self.ctx._check(self._lib.randt_maps_insert_clusters(self._h, idx, _dptr(pts), _dptr(off), len(off) - 1, stride, ioff, C.byref(acc) if wait else None),
                        "randt_maps_insert_clusters")
        return int(acc.value) if wait else None

    def insert_cluster(self, idx, points, intensity_index=None):
        """Map::insertCluster: one cell from all `points` (n, stride) float32; returns True if the cell was accepted."""
        pts = np.ascontiguousarray(points, dtype=np.float32)
        stride = int(pts.shape[1])
        ioff = (3 if stride == 4 else 4) if intensity_index is None else intensity_index
        acc = C.c_int(0)
        self.ctx._check(self._lib.randt_maps_insert_cluster(self._h, idx, _dptr(pts), len(pts), stride, ioff, C.byref(acc)),
                        "randt_maps_insert_cluster")
        return bool(acc.value)

    def insert_cells(self, idx, cells, set_grid=False):
        """Map::insertCell (set_grid False) or append-and-index (True)."""
        cells = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
        self.ctx._check(self._lib.randt_maps_insert_cells(self._h, idx, _dptr(cells), len(cells), 1 if set_grid else 0),
                        "randt_maps_insert_cells")

    def closest_cells(self, idx, queries, k=4, lookup_mahalanobis=True, use_intensity=True):
        """Map::getClosestCells for every query cell: (len(queries), k) int32 compact indices, -1 padded."""
        q = np.ascontiguousarray(queries, dtype=CELL_DTYPE)
        out = np.full((len(q), k), -1, dtype=np.int32)
        self.ctx._check(self._lib.randt_closest_cells(self.ctx._h, self._h, idx, _dptr(q), len(q), k, 1 if lookup_mahalanobis else 0,
                                                      1 if use_intensity else 0, _dptr(out)), "randt_closest_cells")
        return out

    def device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.ctx._check(self._lib.randt_maps_device_ptrs(self._h, C.byref(a), C.byref(b), C.byref(c)), "randt_maps_device_ptrs")
        return a.value, b.value, c.value


# ------------------------------------------------------------------ single cells (Cell mutators) ----
def cell_add_points(ctx, cell, points, min_points_per_cell=5, intensity_index=None):
    """Cell::addPointCloud + updateCell on one cell (CELL_DTYPE scalar; n = 0: empty).  Returns (accepted, cell)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    stride = int(pts.shape[1])
    ioff = (3 if stride == 4 else 4) if intensity_index is None else intensity_index
    c = np.array([cell], dtype=CELL_DTYPE)
    acc = C.c_int(0)
    ctx._check(ctx._lib.randt_cell_add_points(ctx._h, _dptr(c), _dptr(pts), len(pts), stride, ioff, int(min_points_per_cell), C.byref(acc)),
               "randt_cell_add_points")
    return bool(acc.value), c[0]


def cells_merge(ctx, acc, other):
    """Cell::operator+= elementwise; returns the merged cells."""
    a = np.array(acc, dtype=CELL_DTYPE).reshape(-1).copy()
    b = np.ascontiguousarray(np.array(other, dtype=CELL_DTYPE).reshape(-1))
    ctx._check(ctx._lib.randt_cells_merge(ctx._h, _dptr(a), _dptr(b), len(a)), "randt_cells_merge")
    return a


def cells_transform(ctx, cells, pose4):
    """Cell::transformCell of every cell by one pose."""
    a = np.array(cells, dtype=CELL_DTYPE).reshape(-1).copy()
    p = np.ascontiguousarray(pose4, dtype=np.float64)
    ctx._check(ctx._lib.randt_cells_transform(ctx._h, _dptr(a), len(a), _dptr(p)), "randt_cells_transform")
    return a


def points_transform(ctx, points, pose4):
    """The point half of Cell::transformCellWithPointCloud: (n, stride) float32 points moved by one pose (fp32, on the device)."""
    a = np.ascontiguousarray(points, dtype=np.float32).copy()
    p = np.ascontiguousarray(pose4, dtype=np.float64)
    ctx._check(ctx._lib.randt_points_transform(ctx._h, _dptr(a), int(a.shape[0]), int(a.shape[1]), _dptr(p)), "randt_points_transform")
    return a


def cells_mahalanobis(ctx, self_cells, subtrahend_cells, use_intensity=True):
    """self.mahalanobisSquaredIntensity(subtrahend) (or mahalanobisSquared) elementwise, as float64."""
    a = np.ascontiguousarray(np.array(self_cells, dtype=CELL_DTYPE).reshape(-1))
    b = np.ascontiguousarray(np.array(subtrahend_cells, dtype=CELL_DTYPE).reshape(-1))
    out = np.zeros(len(a))
    ctx._check(ctx._lib.randt_cells_mahalanobis(ctx._h, _dptr(a), _dptr(b), len(a), 1 if use_intensity else 0, _dptr(out)),
               "randt_cells_mahalanobis")
    return out


def _shape3(points):
    if hasattr(points, "shape") and len(points.shape) == 3:
        return int(points.shape[0]), int(points.shape[1]), int(points.shape[2])
    raise ValueError("points must be a (n_scans, pitch_points, stride_floats) float32 device tensor")


def ndt_build_batch(ctx, points, cluster, out_maps, first_map=0, n_points=None, intensity_index=None):
    """randt_ndt_build_batch_dev.  points: (B, N, S) float32 on the device."""
    B, N, S = _shape3(points)
    ioff = (3 if S == 4 else 4) if intensity_index is None else intensity_index
    ctx._check(ctx._lib.randt_ndt_build_batch_dev(ctx._h, _dptr(points), B, N, _dptr(n_points), S, ioff, C.byref(cluster),
                                                  out_maps._h, first_map), "randt_ndt_build_batch_dev")


def ndt_build_pndt_batch(ctx, points, polar, beam_cov, cluster, out_maps, first_map=0, n_points=None, intensity_index=None):
    """randt_ndt_build_pndt_batch_dev: pNDT cells (Cell::updateCell with use_pndt).  points: (B, N, S) float32 and polar:
    (B, N, 2) float32 (angle, range) on the device; beam_cov: 3x3 (host)."""
    import numpy as np

    B, N, S = _shape3(points)
    assert tuple(polar.shape) == (B, N, 2) and polar.is_contiguous()
    ioff = (3 if S == 4 else 4) if intensity_index is None else intensity_index
    beam = np.ascontiguousarray(np.asarray(beam_cov, dtype=np.float32).reshape(9))
    ctx._check(ctx._lib.randt_ndt_build_pndt_batch_dev(ctx._h, _dptr(points), B, N, _dptr(n_points), S, ioff, _dptr(polar),
                                                       beam.ctypes.data_as(C.c_void_p), C.byref(cluster), out_maps._h, first_map),
               "randt_ndt_build_pndt_batch_dev")


def associate_batch(ctx, fixed, fixed_idx, moving, moving_first, n_pairs, guess4, mp, corr):
    ctx._check(ctx._lib.randt_associate_batch_dev(ctx._h, fixed._h, _dptr(fixed_idx), moving._h, moving_first, n_pairs,
                                                  _dptr(guess4), C.byref(mp), _dptr(corr)), "randt_associate_batch_dev")


def solve_batch(ctx, fixed, fixed_idx, moving, moving_first, n_pairs, corr, mp, pose4, results):
    ctx._check(ctx._lib.randt_solve_batch_dev(ctx._h, fixed._h, _dptr(fixed_idx), moving._h, moving_first, n_pairs,
                                              _dptr(corr), C.byref(mp), _dptr(pose4), _dptr(results)), "randt_solve_batch_dev")


def register_batch(ctx, fixed, fixed_idx, moving, moving_first, n_pairs, mp, pose4, results):
    ctx._check(ctx._lib.randt_register_batch_dev(ctx._h, fixed._h, _dptr(fixed_idx), moving._h, moving_first, n_pairs,
                                                 C.byref(mp), _dptr(pose4), _dptr(results)), "randt_register_batch_dev")


def scan_register_batch(ctx, points, cluster, fixed, fixed_idx, scan_maps, mp, pose4, results, n_points=None,
                        intensity_index=None):
    """Whole hot path: NDT build -> associate -> solve for a batch of raw scans."""
    B, N, S = _shape3(points)
    ioff = (3 if S == 4 else 4) if intensity_index is None else intensity_index
    ctx._check(ctx._lib.randt_scan_register_batch_dev(ctx._h, _dptr(points), B, N, _dptr(n_points), S, ioff, C.byref(cluster),
                                                      fixed._h, _dptr(fixed_idx), scan_maps._h, C.byref(mp), _dptr(pose4),
                                                      _dptr(results)), "randt_scan_register_batch_dev")


def register_pair(ctx, fixed, fixed_idx, moving, moving_idx, mp, pose4):
    """Host convenience (synchronous): Matcher::estimateLoopConstraint for one pair."""
    p = np.array(pose4, dtype=np.float64)
    res = np.zeros(1, dtype=RESULT_DTYPE)
    ctx._check(ctx._lib.randt_register_pair(ctx._h, fixed._h, fixed_idx, moving._h, moving_idx, C.byref(mp), _dptr(p), _dptr(res)),
               "randt_register_pair")
    return p, res[0]


def ndt_build_host(ctx, points, cluster, out_maps, map_idx, intensity_index=None):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    S = pts.shape[1] if pts.ndim == 2 else 4
    ioff = (3 if S == 4 else 4) if intensity_index is None else intensity_index
    ctx._check(ctx._lib.randt_ndt_build(ctx._h, _dptr(pts) if pts.size else None, int(pts.shape[0]), S, ioff, C.byref(cluster),
                                        out_maps._h, map_idx), "randt_ndt_build")


# ------------------------------------------------------------------ fixed-lag window (a16 / a17) ----
def make_state(pose4, lin_vel=(0.0, 0.0), rot_vel=0.0, lin_acc=(0.0, 0.0), imu_bias=0.0, stamp=0.0):
    """rc::navigation::ndt::State as a STATE_DTYPE scalar (both pose representations filled)."""
    st = np.zeros(1, dtype=STATE_DTYPE)[0]
    st["pose"] = pose4
    st["pos"] = pose4[2:]
    st["rot"] = np.arctan2(pose4[1], pose4[0])
    st["lin_vel"] = lin_vel
    st["rot_vel"] = rot_vel
    st["lin_acc"] = lin_acc
    st["imu_bias"] = imu_bias
    st["stamp"] = stamp
    return st


def window_params(motion_sqrtI_diag=(1, 1, 1, 1, 3, 0.1, 20, 60), covariance_scaling_factor=25.0, ndt_weight=5.0e4,
                  weight_imu=64.0, weight_imu_bias=6.0e5, reject_t=2.0, reject_r=2.0, smoothing_steps=3, use_imu=0,
                  const_vel=1):
    """indoor values: config/parameters_indoor.yaml:32-39 + ndt_radar_slam_base_parameters.yaml:36-47."""
    wp = WindowParams()
    M = np.diag(np.asarray(motion_sqrtI_diag, dtype=np.float64)) * covariance_scaling_factor
    for i, v in enumerate(M.reshape(-1)):
        wp.motion_sqrtI[i] = v
    wp.ndt_weight, wp.weight_imu, wp.weight_imu_bias = ndt_weight, weight_imu, weight_imu_bias
    wp.pose_reject_translation, wp.pose_reject_rotation = reject_t, reject_r
    wp.smoothing_steps, wp.use_imu, wp.use_constant_velocity_model = smoothing_steps, use_imu, const_vel
    return wp


def predict_state(last, stamp, parameterization=_capi.PARAM_MANIFOLD):
    """Matcher::predictTransform (constant-velocity prediction of the next State); PARAM_VECTOR = the (pos, rot) form the
    reference takes when optimize_on_manifold is false."""
    a = np.array([last], dtype=STATE_DTYPE)
    out = np.zeros(1, dtype=STATE_DTYPE)
    rc = _capi.load().randt_predict_state_param(_dptr(a), float(stamp), int(parameterization), _dptr(out))
    if rc:
        raise RandtError(rc, "randt_predict_state_param")
    return out[0]


def predict_states(last, stamp, parameterization=_capi.PARAM_MANIFOLD):
    """randt_predict_state_batch: predict_state for an array of independent states (one call)."""
    a = np.ascontiguousarray(last, dtype=STATE_DTYPE)
    out = np.zeros(len(a), dtype=STATE_DTYPE)
    rc = _capi.load().randt_predict_state_batch(_dptr(a), len(a), float(stamp), int(parameterization), _dptr(out))
    if rc:
        raise RandtError(rc, "randt_predict_state_batch")
    return out


def register_window(ctx, fixed, fixed_idx, moving, moving_idx, states, mp, wp, trans4, imu=None):
    """Matcher::estimateTransformCeres.  states: STATE_DTYPE array (S+1, oldest first).
    Returns (states_out, trans_out, rejected, result)."""
    st = np.array(states, dtype=STATE_DTYPE).copy()
    fi = np.ascontiguousarray(fixed_idx, dtype=np.int32)
    mi = np.ascontiguousarray(moving_idx, dtype=np.int32)
    t = np.array(trans4, dtype=np.float64)
    im = None if imu is None else np.ascontiguousarray(imu, dtype=np.float64)
    rej = C.c_int(0)
    res = np.zeros(1, dtype=RESULT_DTYPE)
    ctx._check(ctx._lib.randt_register_window(ctx._h, fixed._h, _dptr(fi), len(fi), moving._h, _dptr(mi), _dptr(st), len(st),
                                              _dptr(im), C.byref(mp), C.byref(wp), _dptr(t), C.byref(rej), _dptr(res)),
               "randt_register_window")
    return st, t, bool(rej.value), res[0]


def register_window_batch(ctx, fixed, fixed_idx, moving, moving_idx, states, mp, wp, trans4, imu=None):
    """randt_register_window_batch: W independent windows of one shape in one launch.  fixed_idx (W, n_fixed), moving_idx (W, S),
    states STATE_DTYPE (W, S + 1), trans4 (W, 4), imu None or (W, S).  Returns (states_out, trans_out, rejected (W,), results (W,))."""
    st = np.array(states, dtype=STATE_DTYPE).copy()
    W, n_states = st.shape
    fi = np.ascontiguousarray(fixed_idx, dtype=np.int32)
    fi = fi.reshape(W, fi.shape[-1] if fi.ndim == 2 else max(1, fi.size // max(1, W)))
    mi = np.ascontiguousarray(moving_idx, dtype=np.int32).reshape(W, n_states - 1)
    t = np.array(trans4, dtype=np.float64).reshape(W, 4).copy()
    im = None if imu is None else np.ascontiguousarray(imu, dtype=np.float64).reshape(W, n_states - 1)
    rej = np.zeros(W, dtype=np.int32)
    res = np.zeros(W, dtype=RESULT_DTYPE)
    ctx._check(ctx._lib.randt_register_window_batch(ctx._h, W, fixed._h, _dptr(fi), fi.shape[1], moving._h, _dptr(mi), _dptr(st), n_states,
                                                    _dptr(im), C.byref(mp), C.byref(wp), _dptr(t), _dptr(rej), _dptr(res)),
               "randt_register_window_batch")
    return st, t, rej.astype(bool), res


# ------------------------------------------------------------------ scan filter (f-1) ---------------
def filter_params(min_range=0.6, max_range=12.0, min_intensity=6.0, beam_thr=0.04, sensor_to_base=None):
    """indoor values: config/parameters_indoor.yaml:42-44, base yaml :33; identity sensor->base."""
    fp = FilterParams()
    fp.min_range, fp.max_range, fp.min_intensity, fp.beam_distance_increment_threshold = min_range, max_range, min_intensity, beam_thr
    T = np.eye(4, dtype=np.float32)[:3] if sensor_to_base is None else np.asarray(sensor_to_base, dtype=np.float32).reshape(3, 4)
    for i, v in enumerate(T.reshape(-1)):
        fp.sensor_to_base[i] = v
    return fp


def filter_scan_batch(ctx, raw, fp, out_points, out_counts, status, out_polar=None, peaks=None, peak_counts=None,
                      intensity_index=None):
    """randt_filter_scan_batch_dev.  raw: (n_scans, n_azimuths, n_bins, stride) float32 device tensor;
    out_points: (n_scans, pitch_out, 4)."""
    n_scans, n_az, n_bins, stride = (int(v) for v in raw.shape)
    ioff = (3 if stride == 4 else 4) if intensity_index is None else intensity_index
    ctx._check(ctx._lib.randt_filter_scan_batch_dev(ctx._h, _dptr(raw), n_scans, n_az, n_bins, stride, ioff, C.byref(fp),
                                                    _dptr(out_points), int(out_points.shape[1]), _dptr(out_counts), _dptr(out_polar),
                                                    _dptr(peaks), _dptr(peak_counts), _dptr(status)), "randt_filter_scan_batch_dev")


def filter_scan_host(ctx, raw, fp, capacity=8192, intensity_index=None, want_polar=True, want_peaks=True):
    """randt_filter_scan: RadarPreprocessor::filterScan for ONE raw scan in host memory, results on the host.
    raw: (n_azimuths, n_bins, stride) float32 numpy.  Returns (points [n][4], polar [n][2] or None, peaks [m][3] or None, n_kept, status)."""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    n_az, n_bins, stride = (int(v) for v in raw.shape)
    ioff = (3 if stride == 4 else 4) if intensity_index is None else intensity_index
    pts = np.zeros((capacity, 4), dtype=np.float32)
    pol = np.zeros((capacity, 2), dtype=np.float32) if want_polar else None
    pk = np.zeros((n_az, 3), dtype=np.float32) if want_peaks else None
    n, npk, st = C.c_int(0), C.c_int(0), C.c_int(0)
    ctx._check(ctx._lib.randt_filter_scan(ctx._h, _dptr(raw), n_az, n_bins, stride, ioff, C.byref(fp), _dptr(pts), capacity, C.byref(n), _dptr(pol),
                                          _dptr(pk), C.byref(npk), C.byref(st)), "randt_filter_scan")
    m = min(n.value, capacity)
    return pts[:m], (pol[:m] if want_polar else None), (pk[:npk.value] if want_peaks else None), n.value, st.value


def filter_build(ctx, raw, fp, clu, maps, map_idx=0, max_points=6144, intensity_index=None, wait=True):
    """randt_filter_build: raw polar scan in host memory -> filterScan -> clustering + NDT into maps[map_idx], on the device.
    Returns the filter's status (wait=False: asynchronous after the upload, returns None)."""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    n_az, n_bins, stride = (int(v) for v in raw.shape)
    ioff = (3 if stride == 4 else 4) if intensity_index is None else intensity_index
    st = C.c_int(0)
    ctx._check(ctx._lib.randt_filter_build(ctx._h, _dptr(raw), n_az, n_bins, stride, ioff, C.byref(fp), C.byref(clu), int(max_points), maps._h, int(map_idx),
                                           C.byref(st) if wait else None), "randt_filter_build")
    return st.value if wait else None


# ------------------------------------------------------------------ CS divergence (f-2) -------------
def cs_divergence_batch(ctx, fixed, fixed_first, fixed_count, fixed_idx, moving, moving_first, n_pairs, pose4, out, terms=None):
    """randt_cs_divergence_batch_dev (Map::calculateCSDivergence after transformMap)."""
    ctx._check(ctx._lib.randt_cs_divergence_batch_dev(ctx._h, fixed._h, fixed_first, fixed_count, _dptr(fixed_idx), moving._h,
                                                      moving_first, n_pairs, _dptr(pose4), _dptr(out), _dptr(terms)),
               "randt_cs_divergence_batch_dev")


def cs_divergence(ctx, fixed, fixed_idx, moving, moving_idx, pose4=None):
    """randt_cs_divergence: one pair, host result.  Returns (divergence, terms[3])."""
    out, terms = C.c_double(0), np.zeros(3)
    p = None if pose4 is None else np.ascontiguousarray(pose4, dtype=np.float64)
    ctx._check(ctx._lib.randt_cs_divergence(ctx._h, fixed._h, int(fixed_idx), moving._h, int(moving_idx),
                                            None if p is None else p.ctypes.data, C.byref(out), terms.ctypes.data), "randt_cs_divergence")
    return out.value, terms


# ------------------------------------------------------------------ pose graph (f-4) ----------------
def pg_params(**over):
    """Ceres Solver::Options defaults + global_fuser.cpp:52; use_robust_loss / loss_scale = GlobalFuserParameters."""
    p = PgParams()
    _capi.load().randt_pg_params_default(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def pose_graph_optimize(ctx, poses, id_begin, id_end, meas, sqrt_info, max_update_index, params=None):
    """randt_pose_graph_optimize (GlobalFuser::optimizePoseGraph).  poses [N][3] = (x, y, yaw); edges as parallel arrays
    (meas [E][3] = translation + log angle, sqrt_info [E][3][3]).  Returns (optimised poses, result dict)."""
    x = np.array(poses, dtype=np.float64, order="C").reshape(-1, 3)
    ia = np.ascontiguousarray(id_begin, dtype=np.int32)
    ib = np.ascontiguousarray(id_end, dtype=np.int32)
    m = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 3)
    sq = np.ascontiguousarray(sqrt_info, dtype=np.float64).reshape(-1, 9)
    if not (len(ia) == len(ib) == len(m) == len(sq)):
        raise ValueError("edge arrays differ in length")
    p = params if params is not None else pg_params()
    res = PgResult()
    ctx._check(ctx._lib.randt_pose_graph_optimize(ctx._h, len(x), x.ctypes.data, len(ia), ia.ctypes.data, ib.ctypes.data, m.ctypes.data,
                                                  sq.ctypes.data, int(max_update_index), C.byref(p), C.byref(res)),
               "randt_pose_graph_optimize")
    return x, {k: getattr(res, k) for k, _ in PgResult._fields_}


# ------------------------------------------------------------------ Scan Context (f-4) --------------
def sc_params(num_ring=20, num_sector=45, max_radius=15.0, num_exclude_recent=15, num_candidates=10, search_ratio=0.3,
              dist_thresh=0.6, assumed_drift=0.05, odom_eps=1.2, odom_weight=0.2, intensity_factor=0.04):
    """config/parameters_indoor.yaml "scan_context" (:45-58) through ndt_slam.cpp:515-552."""
    return ScParams(num_ring, num_sector, max_radius, num_exclude_recent, num_candidates, search_ratio, dist_thresh, assumed_drift,
                    odom_eps, odom_weight, intensity_factor)


def sc_make_batch(ctx, points, sp, desc, ring_keys, sector_keys, n_points=None, intensity_index=None):
    """randt_sc_make_batch_dev (SCManager::makeScancontext + keys).  points: (B, N, stride) float32 device tensor;
    desc (B, num_sector, num_ring), ring_keys (B, num_ring), sector_keys (B, num_sector): float64 device tensors."""
    B, N, S = (int(v) for v in points.shape)
    ioff = (3 if S == 4 else 4) if intensity_index is None else intensity_index
    ctx._check(ctx._lib.randt_sc_make_batch_dev(ctx._h, _dptr(points), B, N, _dptr(n_points), S, ioff, C.byref(sp), _dptr(desc),
                                                _dptr(ring_keys), _dptr(sector_keys)), "randt_sc_make_batch_dev")


def sc_detect_batch(ctx, sp, desc, ring_keys, pos, dist, query_ids, loop_id, yaw, min_dist=None):
    """randt_sc_detect_batch_dev (SCManager::detectLoopClosureID for a batch of query nodes)."""
    n_db = int(desc.shape[0])
    nq = int(loop_id.shape[0])
    ctx._check(ctx._lib.randt_sc_detect_batch_dev(ctx._h, C.byref(sp), _dptr(desc), _dptr(ring_keys), _dptr(pos), _dptr(dist), n_db,
                                                  _dptr(query_ids), nq, _dptr(loop_id), _dptr(yaw), _dptr(min_dist)),
               "randt_sc_detect_batch_dev")


class ScDatabase:
    """SCManager's state on the device (randt_sc_db_*): append keyframe scans, query loop closures."""

    def __init__(self, ctx, sp, capacity=64):
        self._ctx, self.sp = ctx, sp
        self._h = C.c_void_p()
        ctx._check(ctx._lib.randt_sc_db_create(ctx._h, C.byref(sp), int(capacity), C.byref(self._h)), "randt_sc_db_create")

    def __len__(self):
        return int(self._ctx._lib.randt_sc_db_size(self._h))

    def append(self, points, odom_position, traversed_distance, intensity_index=None):
        """makeAndSaveScancontextAndKeys: points (N, stride) float32 host array.  Returns the node id."""
        pts = np.ascontiguousarray(points, dtype=np.float32)
        n, stride = pts.shape
        ioff = (3 if stride == 4 else 4) if intensity_index is None else intensity_index
        pos = np.ascontiguousarray(odom_position, dtype=np.float64)
        node = C.c_int(-1)
        self._ctx._check(self._ctx._lib.randt_sc_db_append(self._h, pts.ctypes.data, n, stride, ioff, pos.ctypes.data, float(traversed_distance),
                                                          C.byref(node)), "randt_sc_db_append")
        return node.value

    def detect(self, node_id):
        """detectLoopClosureID: (loop_id or -1, yaw_diff_rad, min_dist)."""
        lid, yaw, md = C.c_int(-1), C.c_float(0), C.c_double(0)
        self._ctx._check(self._ctx._lib.randt_sc_db_detect(self._h, int(node_id), C.byref(lid), C.byref(yaw), C.byref(md)), "randt_sc_db_detect")
        return lid.value, yaw.value, md.value

    def download(self, node_id):
        d = np.zeros((self.sp.num_sector, self.sp.num_ring))
        rk, sk = np.zeros(self.sp.num_ring), np.zeros(self.sp.num_sector)
        self._ctx._check(self._ctx._lib.randt_sc_db_download(self._h, int(node_id), d.ctypes.data, rk.ctypes.data, sk.ctypes.data),
                         "randt_sc_db_download")
        return d, rk, sk

    def close(self):
        if self._h and getattr(self._ctx, "_h", None):
            self._ctx._lib.randt_sc_db_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ correlative search (f-3) --------
def bnb_params(window_linear=4.5, window_angular=0.45, linear_step=0.4, cost_threshold=0.82, max_px_accurate_range=4.0, n_iter=2):
    """config/ndt_radar_slam_base_parameters.yaml:50-56."""
    return BnbParams(window_linear, window_angular, linear_step, cost_threshold, max_px_accurate_range, n_iter, 0)


def eval_cost_batch(ctx, fixed, fixed_idx, moving, moving_idx, corr, mp, scale, poses4, cost, n_res=None):
    n = int(poses4.shape[0])
    ctx._check(ctx._lib.randt_eval_cost_batch_dev(ctx._h, fixed._h, fixed_idx, moving._h, moving_idx, _dptr(corr), C.byref(mp), float(scale),
                                                  _dptr(poses4), n, _dptr(cost), _dptr(n_res)), "randt_eval_cost_batch_dev")


def search_global(ctx, fixed, fixed_idx, moving, moving_idx, mp, bp, trans4, scale=1.5, window_linear=4.5, window_angular=0.45):
    """Matcher::estimateTransformGlobalBNB.  Returns (min_cost, pose4, n_evals)."""
    t = np.array(trans4, dtype=np.float64)
    mc, ne = C.c_double(0), C.c_int(0)
    ctx._check(ctx._lib.randt_search_global(ctx._h, fixed._h, fixed_idx, moving._h, moving_idx, C.byref(mp), C.byref(bp), float(scale),
                                            float(window_linear), float(window_angular), _dptr(t), C.byref(mc), C.byref(ne)),
               "randt_search_global")
    return mc.value, t, ne.value


# ------------------------------------------------------------------ multi-GPU group ----
def shard_range(n_items, world, rank):
    """randt_shard_range: contiguous split, remainders to the low ranks."""
    lo, hi = C.c_int(0), C.c_int(0)
    _capi.load().randt_shard_range(int(n_items), int(world), int(rank), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def group_unique_id():
    """randt_group_unique_id: 128 bytes rank 0 hands to every rank (as a numpy uint8 array)."""
    buf = np.zeros(_capi.UNIQUE_ID_BYTES, dtype=np.uint8)
    rc = _capi.load().randt_group_unique_id(_dptr(buf))
    if rc:
        raise RandtError(rc, "randt_group_unique_id")
    return buf


class Group:
    """randt_group: one context + stream per member GPU, map broadcast, sharded registration, row gather.

    Group(devices=[0, 1, ...])                       one process drives the listed devices (repeats = virtual ranks)
    Group(device=d, rank=r, world=G, unique_id=u)    one process per GPU (u from group_unique_id() on rank 0)
    Per-member arguments are lists with one entry per LOCAL member."""

    def __init__(self, devices=None, streams=None, transport=_capi.TRANSPORT_AUTO, device=None, rank=None, world=None, unique_id=None,
                 stream=None):
        self._lib = _capi.load()
        h = C.c_void_p()
        if devices is not None:
            n = len(devices)
            devs = (C.c_int * n)(*[int(d) for d in devices])
            sts = None if streams is None else (C.c_void_p * n)(*[C.c_void_p(s) if s else None for s in streams])
            rc = self._lib.randt_group_create(devs, n, sts, int(transport), C.byref(h))
            where = "randt_group_create"
        else:
            uid = None if unique_id is None else np.ascontiguousarray(unique_id, dtype=np.uint8)
            rc = self._lib.randt_group_create_rank(int(device), C.c_void_p(stream) if stream else None, int(rank), int(world), _dptr(uid),
                                                   C.byref(h))
            where = "randt_group_create_rank"
        if rc:  # the group object is gone; the library keeps the text of this thread's last failed creation
            raise RandtError(rc, where, (self._lib.randt_group_last_error(None) or b"").decode())
        self._h = h
        w, nl, fr, tr = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self._lib.randt_group_info(self._h, C.byref(w), C.byref(nl), C.byref(fr), C.byref(tr))
        self.world, self.n_local, self.first_rank, self.transport = w.value, nl.value, fr.value, tr.value
        dl = list(devices) if devices is not None else [device]
        self.ctxs = [Context._borrowed(self._lib.randt_group_ctx(self._h, i), dl[i]) for i in range(self.n_local)]

    def _check(self, rc, where):
        if rc:
            raise RandtError(rc, where, self._lib.randt_group_last_error(self._h).decode())

    def _arr(self, xs, handle=False):
        """Per-member pointer array (None entries allowed)."""
        if xs is None:
            return None
        assert len(xs) == self.n_local, "one entry per local member"
        vals = []
        for x in xs:
            if x is None:
                vals.append(None)
            elif handle:
                vals.append(x._h)
            else:
                vals.append(_dptr(x))
        return (C.c_void_p * self.n_local)(*vals)

    def close(self):
        if getattr(self, "_h", None):
            for c in self.ctxs:
                c._h = None
            self._lib.randt_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._lib.randt_group_synchronize(self._h), "randt_group_synchronize")

    def broadcast_maps(self, maps, first=0, count=None, root=0):
        count = maps[0].n_maps - first if count is None else count
        self._check(self._lib.randt_group_broadcast_maps(self._h, self._arr(maps, True), first, count, root), "randt_group_broadcast_maps")

    def allgather_rows(self, bufs, n_rows, row_bytes):
        self._check(self._lib.randt_group_allgather_rows(self._h, self._arr(bufs), int(n_rows), int(row_bytes)), "randt_group_allgather_rows")

    def register_batch(self, fixed, fixed_idx, moving, n_pairs, mp, pose4, results, gather=True):
        self._check(self._lib.randt_group_register_batch_dev(self._h, self._arr(fixed, True), self._arr(fixed_idx), self._arr(moving, True),
                                                             int(n_pairs), C.byref(mp), self._arr(pose4), self._arr(results), int(bool(gather))),
                    "randt_group_register_batch_dev")

    def scan_register_batch(self, points, cluster, fixed, fixed_idx, scan_maps, mp, pose4, results, gather=True, n_points=None,
                            intensity_index=None):
        B, N, S = _shape3(points[0])
        ioff = (3 if S == 4 else 4) if intensity_index is None else intensity_index
        self._check(self._lib.randt_group_scan_register_batch_dev(
            self._h, self._arr(points), B, N, self._arr(n_points), S, ioff, C.byref(cluster), self._arr(fixed, True), self._arr(fixed_idx),
            self._arr(scan_maps, True), C.byref(mp), self._arr(pose4), self._arr(results), int(bool(gather))),
            "randt_group_scan_register_batch_dev")

    def register_pairs(self, fixed, fixed_idx, moving, mp, pose4):
        """Host convenience: returns (poses (n,4) float64, results (n,) RESULT_DTYPE)."""
        p = np.array(pose4, dtype=np.float64).reshape(-1, 4).copy()
        fi = np.ascontiguousarray(fixed_idx, dtype=np.int32)
        res = np.zeros(len(p), dtype=RESULT_DTYPE)
        self._check(self._lib.randt_group_register_pairs(self._h, self._arr(fixed, True), _dptr(fi), self._arr(moving, True), len(p), C.byref(mp),
                                                         _dptr(p), _dptr(res)), "randt_group_register_pairs")
        return p, res
