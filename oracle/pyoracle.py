"""ctypes binding of the CPU oracle (oracle/randt_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  PARITY UNPINNED (see randt_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librandt_oracle.so")

CELL_DTYPE = np.dtype(
    [("mean", "<f4", (3,)), ("cov", "<f4", (6,)), ("n", "<u4"), ("max_intensity", "<f4"), ("reserved", "<u4")]
)
assert CELL_DTYPE.itemsize == 48

TRACE_MAX = 4096

PARAM_MANIFOLD, PARAM_AMBIENT4, PARAM_VECTOR, PARAM_ANALYTIC = 0, 1, 2, 3
LINSOLVE_QR, LINSOLVE_NORMAL = 0, 1


class OrcMap(C.Structure):
    _fields_ = [
        ("size_x", C.c_int32), ("size_y", C.c_int32),
        ("res", C.c_double), ("offset_x", C.c_double), ("offset_y", C.c_double),
        ("max_neighbour_dist", C.c_double),
        ("min_points", C.c_int32), ("cap", C.c_int32), ("n_cells", C.c_int32), ("n_dropped", C.c_int32),
        ("cells", C.c_void_p), ("grid", C.POINTER(C.c_int32)),
    ]


class MatcherParams(C.Structure):
    _fields_ = [
        ("loss_scale", C.c_double), ("mu_scale", C.c_double), ("loss_alpha", C.c_double),
        ("loss_weight", C.c_double), ("gnc_divisor", C.c_double),
        ("gnc_steps", C.c_int32), ("max_iterations", C.c_int32), ("n_neighbours", C.c_int32),
        ("lookup_mahalanobis", C.c_int32), ("use_intensity", C.c_int32), ("parameterization", C.c_int32),
        ("linear_solver", C.c_int32), ("max_consecutive_invalid_steps", C.c_int32),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
    ]


class SolveStats(C.Structure):
    _fields_ = [
        ("n_residuals", C.c_int32), ("n_solves", C.c_int32), ("n_iterations", C.c_int32),
        ("n_jac_evals", C.c_int32), ("n_cost_evals", C.c_int32), ("termination", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("max_raw_residual", C.c_double),
        ("mu0", C.c_double),
        ("trace_len", C.c_int32),
        ("trace_cost", C.c_double * TRACE_MAX), ("trace_radius", C.c_double * TRACE_MAX),
        ("trace_flag", C.c_int32 * TRACE_MAX),
    ]


class FilterParams(C.Structure):
    _fields_ = [("min_range", C.c_float), ("max_range", C.c_float), ("min_intensity", C.c_float),
                ("beam_distance_increment_threshold", C.c_float), ("sensor_to_base", C.c_float * 12)]


class BnbParams(C.Structure):
    _fields_ = [("csm_window_linear", C.c_double), ("csm_window_angular", C.c_double), ("csm_linear_step", C.c_double),
                ("csm_cost_threshold", C.c_double), ("csm_max_px_accurate_range", C.c_double), ("csm_n_iter", C.c_int32),
                ("reserved", C.c_int32)]


class State(C.Structure):
    _fields_ = [("pose", C.c_double * 4), ("pos", C.c_double * 2), ("rot", C.c_double), ("lin_vel", C.c_double * 2),
                ("rot_vel", C.c_double), ("lin_acc", C.c_double * 2), ("imu_bias", C.c_double), ("stamp", C.c_double)]


class WindowParams(C.Structure):
    _fields_ = [("motion_sqrtI", C.c_double * 64), ("ndt_weight", C.c_double), ("weight_imu", C.c_double),
                ("weight_imu_bias", C.c_double), ("pose_reject_translation", C.c_double), ("pose_reject_rotation", C.c_double),
                ("smoothing_steps", C.c_int32), ("use_imu", C.c_int32), ("use_constant_velocity_model", C.c_int32),
                ("reserved", C.c_int32)]


STATE_DTYPE = np.dtype([("pose", "<f8", (4,)), ("pos", "<f8", (2,)), ("rot", "<f8"), ("lin_vel", "<f8", (2,)), ("rot_vel", "<f8"),
                        ("lin_acc", "<f8", (2,)), ("imu_bias", "<f8"), ("stamp", "<f8")])
assert STATE_DTYPE.itemsize == C.sizeof(State) == 112


class PgParams(C.Structure):
    _fields_ = [
        ("use_robust_loss", C.c_int32), ("max_iterations", C.c_int32), ("max_consecutive_invalid_steps", C.c_int32),
        ("reserved", C.c_int32), ("loss_scale", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
    ]


class PgResult(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("iterations", C.c_int32), ("termination", C.c_int32), ("n_residual_blocks", C.c_int32), ("n_loop_closures", C.c_int32),
    ]


def build(force=False):
    src = os.path.join(_HERE, "randt_oracle.c")
    hdr = os.path.join(_HERE, "randt_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librandt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    P = C.POINTER
    L.orc_map_create.restype = P(OrcMap)
    L.orc_map_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    L.orc_map_destroy.argtypes = [P(OrcMap)]
    L.orc_map_clear.argtypes = [P(OrcMap)]
    L.orc_map_copy.argtypes = [P(OrcMap), P(OrcMap)]
    L.orc_grid_labels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    L.orc_ndt_build.restype = C.c_int
    L.orc_ndt_build.argtypes = [P(OrcMap), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_cell_from_points.restype = C.c_int
    L.orc_cell_from_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_cell_merge.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_cell_update.restype = C.c_int
    L.orc_cell_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_cell_mahalanobis.restype = C.c_double
    L.orc_cell_mahalanobis.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_pose_to_affine_f.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_cell_transform.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_map_transform.argtypes = [P(OrcMap), C.c_void_p]
    L.orc_map_merge.argtypes = [P(OrcMap), P(OrcMap)]
    L.orc_map_coord_to_index.restype = C.c_uint32
    L.orc_map_coord_to_index.argtypes = [P(OrcMap), C.c_float, C.c_float]
    L.orc_associate.restype = C.c_int
    L.orc_associate.argtypes = [P(OrcMap), P(OrcMap), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_barron_scaled.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
    L.orc_matcher_params_default.argtypes = [P(MatcherParams)]
    L.orc_ndt_residual.restype = C.c_double
    L.orc_ndt_residual.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6
    L.orc_solve_pair.restype = C.c_int
    L.orc_solve_pair.argtypes = [P(OrcMap), P(OrcMap), C.c_void_p, C.c_int, P(MatcherParams), C.c_void_p, P(SolveStats)]
    L.orc_register_pair.restype = C.c_int
    L.orc_register_pair.argtypes = [P(OrcMap), P(OrcMap), P(MatcherParams), C.c_void_p, C.c_void_p, P(SolveStats)]
    L.orc_register_batch.restype = C.c_int
    L.orc_register_batch.argtypes = [
        C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
        P(MatcherParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
    ]
    for name in ("orc_se2_exp", "orc_se2_log", "orc_se2_inv"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
    L.orc_se2_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_num_threads.restype = C.c_int
    L.orc_eval_cost_batch.argtypes = [P(OrcMap), P(OrcMap), C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int,
                                      C.c_void_p, P(C.c_int)]
    L.orc_search_global_bnb.restype = C.c_double
    L.orc_search_global_bnb.argtypes = [P(OrcMap), P(OrcMap), P(MatcherParams), P(BnbParams), C.c_double, C.c_double, C.c_double,
                                        C.c_void_p, P(C.c_int)]
    L.orc_sc_make.restype = None
    L.orc_sc_make.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_sc_distance.restype = C.c_double
    L.orc_sc_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, P(C.c_int)]
    L.orc_sc_detect.restype = C.c_int
    L.orc_sc_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(C.c_float), P(C.c_double)]
    L.orc_cs_divergence.restype = C.c_double
    L.orc_cs_divergence.argtypes = [P(OrcMap), P(OrcMap), C.c_void_p]
    L.orc_filter_scan.restype = C.c_int
    L.orc_filter_scan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(FilterParams), C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int, P(C.c_int)]
    L.orc_predict_state.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    L.orc_motion_residual.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_imu_residual.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.orc_predict_state_vec.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    L.orc_motion_residual_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_imu_residual_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.orc_register_window.restype = C.c_int
    L.orc_register_window.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, P(MatcherParams),
                                      P(WindowParams), C.c_void_p, P(SolveStats)]
    L.orc_pg_params_default.argtypes = [P(PgParams)]
    L.orc_pg_edge.argtypes = [C.c_void_p] * 7
    L.orc_pose_graph_optimize.restype = C.c_int
    L.orc_pose_graph_optimize.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          P(PgParams), P(PgResult)]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**over):
    p = MatcherParams()
    lib().orc_matcher_params_default(C.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Map:
    """Owning wrapper of an orc_map."""

    def __init__(self, size_x=100, size_y=100, res=0.5, center=(0.0, 0.0), max_neighbour_dist=4.0, min_points=5, cap=None):
        if cap is None:
            cap = size_x * size_y
        self._p = lib().orc_map_create(size_x, size_y, res, center[0], center[1], max_neighbour_dist, min_points, cap)
        if not self._p:
            raise MemoryError

    def __del__(self):
        if getattr(self, "_p", None) and lib is not None:   # at interpreter shutdown the module globals may be gone already
            lib().orc_map_destroy(self._p)
            self._p = None

    @property
    def c(self):
        return self._p.contents

    @property
    def n_cells(self):
        return self.c.n_cells

    @property
    def n_slots(self):
        return self.c.size_x * self.c.size_y

    def cells_view(self):
        buf = (C.c_char * (48 * self.c.cap)).from_address(self.c.cells)
        return np.frombuffer(buf, dtype=CELL_DTYPE)

    def cells(self):
        return self.cells_view()[: self.n_cells].copy()

    def grid(self):
        return np.ctypeslib.as_array(self.c.grid, shape=(self.n_slots,)).copy()

    def set(self, cells, grid):
        cells = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
        assert len(cells) <= self.c.cap
        self.cells_view()[: len(cells)] = cells
        self.c.n_cells = len(cells)
        g = np.ctypeslib.as_array(self.c.grid, shape=(self.n_slots,))
        g[:] = np.asarray(grid, dtype=np.int32)

    def clear(self):
        lib().orc_map_clear(self._p)

    def copy(self):
        m = Map(self.c.size_x, self.c.size_y, self.c.res, (0, 0), self.c.max_neighbour_dist, self.c.min_points, self.c.cap)
        lib().orc_map_copy(m._p, self._p)
        return m

    def build(self, pts, n_clusters, max_range, ioff=3, polar=None, beam_cov=None):
        """polar ([n, 2] angle / range) + beam_cov (3x3): the pNDT cells of ndt_cell.cpp:67-82 (use_pndt)."""
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        if polar is None:
            return lib().orc_ndt_build(self._p, _ptr(pts), pts.shape[0], pts.shape[1], ioff, int(n_clusters), float(max_range))
        polar = np.ascontiguousarray(polar, dtype=np.float32)
        beam = np.ascontiguousarray(beam_cov, dtype=np.float32).reshape(9)
        assert polar.shape == (pts.shape[0], 2)
        L = lib()
        L.orc_ndt_build_pndt.restype = C.c_int
        L.orc_ndt_build_pndt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        return L.orc_ndt_build_pndt(self._p, _ptr(pts), pts.shape[0], pts.shape[1], ioff, int(n_clusters), float(max_range),
                                    _ptr(polar), _ptr(beam))

    def transform(self, pose4):
        aff = pose_to_affine_f(pose4)
        lib().orc_map_transform(self._p, _ptr(aff))

    def merge(self, moving):
        lib().orc_map_merge(self._p, moving._p)

    def coord_to_index(self, x, y):
        return lib().orc_map_coord_to_index(self._p, float(x), float(y))


def grid_labels(pts, n_clusters, max_range, ioff=3):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty(pts.shape[0], dtype=np.int32)
    lib().orc_grid_labels(_ptr(pts), pts.shape[0], pts.shape[1], ioff, int(n_clusters), float(max_range), _ptr(out))
    return out


def cell_from_points(pts, min_points=5, ioff=3):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    cell = np.zeros(1, dtype=CELL_DTYPE)
    ok = lib().orc_cell_from_points(_ptr(cell), _ptr(pts), None, pts.shape[0], pts.shape[1], ioff, min_points)
    return bool(ok), cell[0]


def cell_update(cell, pts, min_points=5, ioff=3):
    """Cell::addPointCloud + updateCell on a cell that may already be filled; returns (accepted, cell)."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    c = np.array([cell], dtype=CELL_DTYPE)
    ok = lib().orc_cell_update(_ptr(c), _ptr(pts), pts.shape[0], pts.shape[1], ioff, min_points)
    return bool(ok), c[0]


def cell_mahalanobis(a, b, use_intensity=True):
    """a.mahalanobisSquared(b) / a.mahalanobisSquaredIntensity(b)"""
    x, y = np.array([a], dtype=CELL_DTYPE), np.array([b], dtype=CELL_DTYPE)
    return float(lib().orc_cell_mahalanobis(_ptr(x), _ptr(y), int(use_intensity)))


def cell_merge(dst, src):
    d = np.array([dst], dtype=CELL_DTYPE)
    s = np.array([src], dtype=CELL_DTYPE)
    lib().orc_cell_merge(_ptr(d), _ptr(s))
    return d[0]


def pose_to_affine_f(pose4):
    p = np.ascontiguousarray(pose4, dtype=np.float64)
    aff = np.empty(4, dtype=np.float32)
    lib().orc_pose_to_affine_f(_ptr(p), _ptr(aff))
    return aff


def cell_transform(cell, pose4):
    c = np.array([cell], dtype=CELL_DTYPE)
    aff = pose_to_affine_f(pose4)
    lib().orc_cell_transform(_ptr(c), _ptr(aff))
    return c[0]


def associate(fixed, moving, pose4, k=4, lookup_mahalanobis=True, use_intensity=True):
    p = np.ascontiguousarray(pose4, dtype=np.float64)
    corr = np.full((max(moving.n_cells, 1), k), -1, dtype=np.int32)
    n = lib().orc_associate(fixed._p, moving._p, _ptr(p), k, int(lookup_mahalanobis), int(use_intensity), _ptr(corr))
    return corr[: moving.n_cells], n


def barron_scaled(s, a, alpha, mu, weight=1.0):
    rho = np.empty(3)
    lib().orc_barron_scaled(float(s), float(a), float(alpha), float(mu), float(weight), _ptr(rho))
    return rho


def ndt_residual(d, parameterization, pose4, mm, mc, fm, fc, want_jac=True):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (pose4, mm, mc, fm, fc)]
    jac = np.zeros(4)
    r = lib().orc_ndt_residual(d, parameterization, *[_ptr(a) for a in arrs], _ptr(jac) if want_jac else None)
    nj = 4 if parameterization == PARAM_AMBIENT4 else 3
    return r, jac[:nj]


def stats_to_dict(st):
    n = st.trace_len
    return dict(
        n_residuals=st.n_residuals, n_solves=st.n_solves, n_iterations=st.n_iterations,
        n_jac_evals=st.n_jac_evals, n_cost_evals=st.n_cost_evals, termination=st.termination,
        initial_cost=st.initial_cost, final_cost=st.final_cost, max_raw_residual=st.max_raw_residual, mu0=st.mu0,
        trace_cost=np.array(st.trace_cost[:n]), trace_radius=np.array(st.trace_radius[:n]),
        trace_flag=np.array(st.trace_flag[:n]),
    )


def solve_pair(fixed, moving, corr, params, pose4):
    p = np.array(pose4, dtype=np.float64)
    corr = np.ascontiguousarray(corr, dtype=np.int32)
    st = SolveStats()
    rc = lib().orc_solve_pair(fixed._p, moving._p, _ptr(corr), corr.shape[1], C.byref(params), _ptr(p), C.byref(st))
    return rc, p, stats_to_dict(st)


def register_pair(fixed, moving, params, pose4):
    p = np.array(pose4, dtype=np.float64)
    cost = C.c_double(0)
    st = SolveStats()
    rc = lib().orc_register_pair(fixed._p, moving._p, C.byref(params), _ptr(p), C.byref(cost), C.byref(st))
    return rc, p, cost.value, stats_to_dict(st)


def register_batch(pts, fixed_maps, fixed_idx, params, guess4, n_clusters, max_range, ioff=3, n_threads=0, want_stats=False):
    """pts: (B, n, stride) float32.  Returns poses (B,4), cost (B,), iters (B,)."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    B, n, stride = pts.shape
    arr = (C.POINTER(OrcMap) * len(fixed_maps))(*[m._p for m in fixed_maps])
    fixed_idx = np.ascontiguousarray(fixed_idx, dtype=np.int32)
    guess4 = np.ascontiguousarray(guess4, dtype=np.float64)
    poses = np.zeros((B, 4))
    cost = np.zeros(B)
    iters = np.zeros(B, dtype=np.int32)
    stats = np.zeros((B, 4), dtype=np.int32) if want_stats else None
    fail = lib().orc_register_batch(
        B, _ptr(pts), n, stride, ioff, int(n_clusters), float(max_range), C.cast(arr, C.c_void_p), _ptr(fixed_idx),
        C.byref(params), _ptr(guess4), _ptr(poses), _ptr(cost), _ptr(iters), int(n_threads), _ptr(stats) if want_stats else None,
    )
    if want_stats:   # columns: n_residuals, n_solves, termination, passes
        return fail, poses, cost, iters, stats
    return fail, poses, cost, iters


def se2_exp(xi):
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    out = np.empty(4)
    lib().orc_se2_exp(_ptr(xi), _ptr(out))
    return out


def se2_log(p4):
    p4 = np.ascontiguousarray(p4, dtype=np.float64)
    out = np.empty(3)
    lib().orc_se2_log(_ptr(p4), _ptr(out))
    return out


def se2_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty(4)
    lib().orc_se2_mul(_ptr(a), _ptr(b), _ptr(out))
    return out


def se2_inv(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty(4)
    lib().orc_se2_inv(_ptr(a), _ptr(out))
    return out


def sqrt_zero_count(reset=False):
    """Evaluations at which orc_ndt_residual's sqrt(0) guard (zero Jacobian row where the reference's autodiff gives NaN) fired."""
    f = lib().orc_sqrt_zero_count
    f.restype = C.c_longlong
    f.argtypes = [C.c_int]
    return int(f(1 if reset else 0))


def set_eval_threads(n):
    """TIMING ONLY: residual blocks of one problem over n OpenMP threads (changes the summation order); 1 restores the oracle proper."""
    lib().orc_set_eval_threads(int(n))


def num_threads():
    return lib().orc_num_threads()


# ------------------------------------------------------------------ fixed-lag window ----------
def make_state(pose4, lin_vel=(0.0, 0.0), rot_vel=0.0, lin_acc=(0.0, 0.0), imu_bias=0.0, stamp=0.0):
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
    """indoor values: config/parameters_indoor.yaml:32-39 + base yaml :36-47."""
    wp = WindowParams()
    M = np.diag(np.asarray(motion_sqrtI_diag, dtype=np.float64)) * covariance_scaling_factor
    for i, v in enumerate(M.reshape(-1)):
        wp.motion_sqrtI[i] = v
    wp.ndt_weight, wp.weight_imu, wp.weight_imu_bias = ndt_weight, weight_imu, weight_imu_bias
    wp.pose_reject_translation, wp.pose_reject_rotation = reject_t, reject_r
    wp.smoothing_steps, wp.use_imu, wp.use_constant_velocity_model = smoothing_steps, use_imu, const_vel
    return wp


def predict_state(last, stamp, vector=False):
    a = np.array([last], dtype=STATE_DTYPE)
    out = np.zeros(1, dtype=STATE_DTYPE)
    (lib().orc_predict_state_vec if vector else lib().orc_predict_state)(_ptr(a), float(stamp), _ptr(out))
    return out[0]


def motion_residual(x0, x1, sqrtI, want_jac=True, vector=False):
    a = np.array([x0], dtype=STATE_DTYPE)
    b = np.array([x1], dtype=STATE_DTYPE)
    M = np.ascontiguousarray(sqrtI, dtype=np.float64).reshape(64)
    r = np.zeros(8)
    J = np.zeros((8, 16))
    (lib().orc_motion_residual_vec if vector else lib().orc_motion_residual)(_ptr(a), _ptr(b), _ptr(M), _ptr(r), _ptr(J) if want_jac else None)
    return r, J


def imu_residual(x0, x1, imu_rot, weight, bias_weight, want_jac=True, vector=False):
    a = np.array([x0], dtype=STATE_DTYPE)
    b = np.array([x1], dtype=STATE_DTYPE)
    r = np.zeros(2)
    J = np.zeros((2, 8))
    (lib().orc_imu_residual_vec if vector else lib().orc_imu_residual)(_ptr(a), _ptr(b), float(imu_rot), float(weight), float(bias_weight),
                                                                       _ptr(r), _ptr(J) if want_jac else None)
    return r, J


def register_window(fixed_maps, moving_maps, states, params, wparams, trans4, imu=None):
    """states: STATE_DTYPE array (S+1), oldest first; returns (rc, states_out, trans_out, stats)."""
    st = np.array(states, dtype=STATE_DTYPE).copy()
    fx = (C.POINTER(OrcMap) * len(fixed_maps))(*[m._p for m in fixed_maps])
    mv = (C.POINTER(OrcMap) * len(moving_maps))(*[m._p for m in moving_maps])
    t = np.array(trans4, dtype=np.float64)
    im = None if imu is None else np.ascontiguousarray(imu, dtype=np.float64)
    ss = SolveStats()
    rc = lib().orc_register_window(C.cast(fx, C.c_void_p), len(fixed_maps), C.cast(mv, C.c_void_p), _ptr(st), len(st),
                                   _ptr(im) if im is not None else None, C.byref(params), C.byref(wparams), _ptr(t), C.byref(ss))
    return rc, st, t, stats_to_dict(ss)


# ------------------------------------------------------------------ f-1 filterScan -------------
def filter_params(min_range=0.6, max_range=12.0, min_intensity=6.0, beam_thr=0.04, sensor_to_base=None):
    fp = FilterParams()
    fp.min_range, fp.max_range, fp.min_intensity, fp.beam_distance_increment_threshold = min_range, max_range, min_intensity, beam_thr
    T = np.eye(4, dtype=np.float32)[:3] if sensor_to_base is None else np.asarray(sensor_to_base, dtype=np.float32).reshape(3, 4)
    for i, v in enumerate(T.reshape(-1)):
        fp.sensor_to_base[i] = v
    return fp


def filter_scan(raw, fp, ioff=3, capacity=None):
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    n, stride = raw.shape
    capacity = n if capacity is None else capacity
    pts = np.zeros((capacity, 4), dtype=np.float32)
    polar = np.zeros((capacity, 2), dtype=np.float32)
    peaks = np.zeros((n, 3), dtype=np.float32)
    npk = C.c_int(0)
    cnt = lib().orc_filter_scan(_ptr(raw), n, stride, ioff, C.byref(fp), _ptr(pts), _ptr(polar), capacity, _ptr(peaks), n, C.byref(npk))
    return cnt, pts[:max(cnt, 0)], polar[:max(cnt, 0)], peaks[:npk.value]


# ------------------------------------------------------------------ f-2 CS divergence ----------
def cs_divergence(fixed, moving):
    terms = np.zeros(3)
    v = lib().orc_cs_divergence(fixed._p, moving._p, _ptr(terms))
    return v, terms


# ------------------------------------------------------------------ f-4 Scan Context -----------
class ScParams(C.Structure):
    _fields_ = [("num_ring", C.c_int), ("num_sector", C.c_int), ("max_radius", C.c_double), ("num_exclude_recent", C.c_int),
                ("num_candidates", C.c_int), ("search_ratio", C.c_double), ("dist_thresh", C.c_double), ("assumed_drift", C.c_double),
                ("odom_eps", C.c_double), ("odom_weight", C.c_double), ("intensity_factor", C.c_double)]


def sc_make(pts, sp, ioff=None):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    n, stride = pts.shape
    ioff = (3 if stride == 4 else 4) if ioff is None else ioff
    desc = np.zeros((sp.num_sector, sp.num_ring))
    rk, sk = np.zeros(sp.num_ring), np.zeros(sp.num_sector)
    lib().orc_sc_make(_ptr(pts), n, stride, ioff, C.byref(sp), _ptr(desc), _ptr(rk), _ptr(sk))
    return desc, rk, sk


def sc_distance(sp, sc1, sc2, pos1, pos2, dist1, dist2):
    sc1, sc2 = np.ascontiguousarray(sc1, dtype=np.float64), np.ascontiguousarray(sc2, dtype=np.float64)
    p1, p2 = np.ascontiguousarray(pos1, dtype=np.float64), np.ascontiguousarray(pos2, dtype=np.float64)
    sh = C.c_int(0)
    d = lib().orc_sc_distance(C.byref(sp), _ptr(sc1), _ptr(sc2), _ptr(p1), _ptr(p2), C.c_double(dist1), C.c_double(dist2), C.byref(sh))
    return d, sh.value


def sc_detect(sp, desc, ring_keys, pos, dist, node_id):
    desc = np.ascontiguousarray(desc, dtype=np.float64)
    rk = np.ascontiguousarray(ring_keys, dtype=np.float64)
    pos, dist = np.ascontiguousarray(pos, dtype=np.float64), np.ascontiguousarray(dist, dtype=np.float64)
    yaw, md = C.c_float(0), C.c_double(0)
    lid = lib().orc_sc_detect(C.byref(sp), _ptr(desc), _ptr(rk), _ptr(pos), _ptr(dist), desc.shape[0], int(node_id), C.byref(yaw), C.byref(md))
    return lid, yaw.value, md.value


# ------------------------------------------------------------------ f-3 correlative search -----
def bnb_params(window_linear=4.5, window_angular=0.45, linear_step=0.4, cost_threshold=0.82, max_px_accurate_range=4.0, n_iter=2):
    """config/ndt_radar_slam_base_parameters.yaml:50-56."""
    return BnbParams(window_linear, window_angular, linear_step, cost_threshold, max_px_accurate_range, n_iter, 0)


def eval_cost_batch(fixed, moving, corr, poses4, scale=1.5, alpha=-2.0, use_intensity=1):
    corr = np.ascontiguousarray(corr, dtype=np.int32)
    poses4 = np.ascontiguousarray(poses4, dtype=np.float64).reshape(-1, 4)
    cost = np.zeros(len(poses4))
    n = C.c_int(0)
    lib().orc_eval_cost_batch(fixed._p, moving._p, _ptr(corr), corr.shape[1], use_intensity, scale, alpha, _ptr(poses4), len(poses4),
                              _ptr(cost), C.byref(n))
    return cost, n.value


def search_global_bnb(fixed, moving, params, bp, trans4, scale=1.5, window_linear=4.5, window_angular=0.45):
    t = np.array(trans4, dtype=np.float64)
    n = C.c_int(0)
    mc = lib().orc_search_global_bnb(fixed._p, moving._p, C.byref(params), C.byref(bp), scale, window_linear, window_angular, _ptr(t), C.byref(n))
    return mc, t, n.value


def pg_params(**over):
    p = PgParams()
    lib().orc_pg_params_default(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def pg_edge(pose_a, pose_b, meas, sqrt_info):
    a = np.ascontiguousarray(pose_a, np.float64)
    b = np.ascontiguousarray(pose_b, np.float64)
    m = np.ascontiguousarray(meas, np.float64)
    sq = np.ascontiguousarray(sqrt_info, np.float64).reshape(9)
    r, Ja, Jb = np.zeros(3), np.zeros(9), np.zeros(9)
    lib().orc_pg_edge(_ptr(a), _ptr(b), _ptr(m), _ptr(sq), _ptr(r), _ptr(Ja), _ptr(Jb))
    return r, Ja.reshape(3, 3), Jb.reshape(3, 3)


def pose_graph_optimize(poses, id_begin, id_end, meas, sqrt_info, max_update_index, params=None):
    """poses [N][3] (x, y, yaw); returns (optimised copy, result dict)."""
    x = np.array(poses, np.float64, order="C").reshape(-1, 3)
    ia = np.ascontiguousarray(id_begin, np.int32)
    ib = np.ascontiguousarray(id_end, np.int32)
    m = np.ascontiguousarray(meas, np.float64).reshape(-1, 3)
    sq = np.ascontiguousarray(sqrt_info, np.float64).reshape(-1, 9)
    assert len(ia) == len(ib) == len(m) == len(sq)
    p = params if params is not None else pg_params()
    res = PgResult()
    rc = lib().orc_pose_graph_optimize(len(x), _ptr(x), len(ia), _ptr(ia), _ptr(ib), _ptr(m), _ptr(sq), int(max_update_index),
                                       C.byref(p), C.byref(res))
    if rc != 0:
        raise ValueError("orc_pose_graph_optimize: invalid graph")
    return x, {k: getattr(res, k) for k, _ in PgResult._fields_}
