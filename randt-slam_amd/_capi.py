"""ctypes declarations of the C ABI in include/randt.h (librandt_hip.so).

The library is the product; this file only binds it.  Loading fails loudly when the shared object
is missing -- there is no Python / CPU fallback for any compute entry point.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RANDT_LIB: developer A/B hook (an alternative build of the same library); the default is the in-tree build
LIB_PATH = os.environ.get("RANDT_LIB") or os.path.join(_HERE, "librandt_hip.so")

CELL_DTYPE = np.dtype(
    [("mean", "<f4", (3,)), ("cov", "<f4", (6,)), ("n", "<u4"), ("max_intensity", "<f4"), ("reserved", "<u4")]
)
RESULT_DTYPE = np.dtype(
    [("cost", "<f8"), ("final_cost", "<f8"), ("initial_cost", "<f8"), ("mu0", "<f8"),
     ("n_residuals", "<i4"), ("iterations", "<i4"), ("gnc_solves", "<i4"), ("termination", "<i4"),
     ("n_evals", "<i4"), ("status", "<i4"), ("reserved", "<i4", (2,))]
)
assert CELL_DTYPE.itemsize == 48 and RESULT_DTYPE.itemsize == 64

PARAM_MANIFOLD, PARAM_AMBIENT4, PARAM_VECTOR, PARAM_ANALYTIC = 0, 1, 2, 3
OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM, ERR_NODEVICE = range(6)


class MapParams(C.Structure):
    _fields_ = [
        ("size_x", C.c_int32), ("size_y", C.c_int32), ("resolution", C.c_double),
        ("center_x", C.c_double), ("center_y", C.c_double), ("max_neighbour_dist", C.c_double),
        ("min_points_per_cell", C.c_int32), ("reserved", C.c_int32),
    ]


class PoolStats(C.Structure):
    _fields_ = [("device_allocs", C.c_int64), ("device_frees", C.c_int64), ("stream_syncs", C.c_int64), ("pool_hits", C.c_int64),
                ("pool_bytes", C.c_int64), ("pool_blocks", C.c_int64), ("foreign_waits", C.c_int64), ("reserved", C.c_int64 * 1)]


class ClusterParams(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("max_range", C.c_float)]


class MatcherParams(C.Structure):
    _fields_ = [
        ("loss_scale", C.c_double), ("mu_scale", C.c_double), ("loss_alpha", C.c_double),
        ("loss_weight", C.c_double), ("gnc_divisor", C.c_double),
        ("gnc_steps", C.c_int32), ("max_iterations", C.c_int32), ("n_neighbours", C.c_int32),
        ("lookup_mahalanobis", C.c_int32), ("use_intensity", C.c_int32), ("parameterization", C.c_int32),
        ("max_consecutive_invalid_steps", C.c_int32), ("reserved", C.c_int32),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
    ]


class FilterParams(C.Structure):
    _fields_ = [("min_range", C.c_float), ("max_range", C.c_float), ("min_intensity", C.c_float),
                ("beam_distance_increment_threshold", C.c_float), ("sensor_to_base", C.c_float * 12)]


class ScParams(C.Structure):
    _fields_ = [("num_ring", C.c_int32), ("num_sector", C.c_int32), ("max_radius", C.c_double),
                ("num_exclude_recent", C.c_int32), ("num_candidates", C.c_int32), ("search_ratio", C.c_double),
                ("dist_thresh", C.c_double), ("assumed_drift", C.c_double), ("odom_eps", C.c_double),
                ("odom_weight", C.c_double), ("intensity_factor", C.c_double)]


class BnbParams(C.Structure):
    _fields_ = [("csm_window_linear", C.c_double), ("csm_window_angular", C.c_double), ("csm_linear_step", C.c_double),
                ("csm_cost_threshold", C.c_double), ("csm_max_px_accurate_range", C.c_double), ("csm_n_iter", C.c_int32),
                ("reserved", C.c_int32)]


class PgParams(C.Structure):
    _fields_ = [("use_robust_loss", C.c_int32), ("max_iterations", C.c_int32), ("max_consecutive_invalid_steps", C.c_int32),
                ("reserved", C.c_int32), ("loss_scale", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double)]


class PgResult(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int32), ("termination", C.c_int32),
                ("n_residual_blocks", C.c_int32), ("n_loop_closures", C.c_int32), ("n_separator_poses", C.c_int32),
                ("reserved", C.c_int32)]


class WindowParams(C.Structure):
    _fields_ = [("motion_sqrtI", C.c_double * 64), ("ndt_weight", C.c_double), ("weight_imu", C.c_double),
                ("weight_imu_bias", C.c_double), ("pose_reject_translation", C.c_double), ("pose_reject_rotation", C.c_double),
                ("smoothing_steps", C.c_int32), ("use_imu", C.c_int32), ("use_constant_velocity_model", C.c_int32),
                ("reserved", C.c_int32)]


STATE_DTYPE = np.dtype([("pose", "<f8", (4,)), ("pos", "<f8", (2,)), ("rot", "<f8"), ("lin_vel", "<f8", (2,)), ("rot_vel", "<f8"),
                        ("lin_acc", "<f8", (2,)), ("imu_bias", "<f8"), ("stamp", "<f8")])
assert STATE_DTYPE.itemsize == 112

# every symbol include/randt.h declares: name -> (restype, argtypes)
_V, _I, _P = C.c_void_p, C.c_int, C.POINTER
SYMBOLS = {
    "randt_version": (_I, []),
    "randt_status_string": (C.c_char_p, [_I]),
    "randt_last_error": (C.c_char_p, [_V]),
    "randt_ctx_create": (_I, [_I, _V, _P(_V)]),
    "randt_ctx_destroy": (_I, [_V]),
    "randt_ctx_set_stream": (_I, [_V, _V]),
    "randt_ctx_synchronize": (_I, [_V]),
    "randt_ctx_set_trace": (_I, [_V, _V, _I]),
    "randt_ctx_set_solve_mode": (_I, [_V, _I]),
    "randt_matcher_params_default": (None, [_P(MatcherParams)]),
    "randt_ctx_pool_stats": (_I, [_V, _P(PoolStats)]),
    "randt_ctx_pool_trim": (_I, [_V]),
    "randt_maps_create": (_I, [_V, _I, _P(MapParams), _I, _I, _P(_V)]),
    "randt_maps_create_external": (_I, [_V, _I, _P(MapParams), _I, _V, _V, _V, _P(_V)]),
    "randt_maps_destroy": (_I, [_V]),
    "randt_maps_cells_bytes": (C.c_size_t, [_I, _I]),
    "randt_maps_grid_bytes": (C.c_size_t, [_I, _P(MapParams)]),
    "randt_maps_info": (_I, [_V, _P(_I), _P(_I), _P(_I), _P(_I)]),
    "randt_maps_device_ptrs": (_I, [_V, _P(_V), _P(_V), _P(_V)]),
    "randt_maps_clear": (_I, [_V, _I, _I]),
    "randt_maps_upload": (_I, [_V, _I, _V, _I, _V]),
    "randt_maps_download": (_I, [_V, _I, _V, _I, _P(_I), _V]),
    "randt_maps_counts": (_I, [_V, _I, _I, _V]),
    "randt_maps_copy": (_I, [_V, _I, _V, _I, _I]),
    "randt_maps_clone": (_I, [_V, _I, _I, _P(_V)]),
    "randt_ndt_build_batch_dev": (_I, [_V, _V, _I, _I, _V, _I, _I, _P(ClusterParams), _V, _I]),
    "randt_ndt_build": (_I, [_V, _V, _I, _I, _I, _P(ClusterParams), _V, _I]),
    "randt_ndt_build_pndt_batch_dev": (_I, [_V, _V, _I, _I, _V, _I, _I, _V, _V, _P(ClusterParams), _V, _I]),
    "randt_maps_transform": (_I, [_V, _I, _I, _V]),
    "randt_maps_merge": (_I, [_V, _I, _V, _I, _I, _V]),
    "randt_maps_merge_batch": (_I, [_V, _I, _I, _V, _I, _I, _V]),
    "randt_maps_reindex": (_I, [_V, _I, _I]),
    "randt_maps_insert_cluster": (_I, [_V, _I, _V, _I, _I, _I, _P(_I)]),
    "randt_maps_insert_cells": (_I, [_V, _I, _V, _I, _I]),
    "randt_maps_insert_clusters": (_I, [_V, _I, _V, _V, _I, _I, _I, _P(_I)]),
    "randt_closest_cells": (_I, [_V, _V, _I, _V, _I, _I, _I, _I, _V]),
    "randt_cell_add_points": (_I, [_V, _V, _V, _I, _I, _I, _I, _P(_I)]),
    "randt_cells_merge": (_I, [_V, _V, _V, _I]),
    "randt_cells_transform": (_I, [_V, _V, _I, _V]),
    "randt_cells_mahalanobis": (_I, [_V, _V, _V, _I, _I, _V]),
    "randt_points_transform": (_I, [_V, _V, _I, _I, _V]),
    "randt_associate_batch_dev": (_I, [_V, _V, _V, _V, _I, _I, _V, _P(MatcherParams), _V]),
    "randt_solve_batch_dev": (_I, [_V, _V, _V, _V, _I, _I, _V, _P(MatcherParams), _V, _V]),
    "randt_register_batch_dev": (_I, [_V, _V, _V, _V, _I, _I, _P(MatcherParams), _V, _V]),
    "randt_scan_register_batch_dev": (_I, [_V, _V, _I, _I, _V, _I, _I, _P(ClusterParams), _V, _V, _V, _P(MatcherParams), _V, _V]),
    "randt_register_pair": (_I, [_V, _V, _I, _V, _I, _P(MatcherParams), _V, _V]),
    "randt_eval_cost_batch_dev": (_I, [_V, _V, _I, _V, _I, _V, _P(MatcherParams), C.c_double, _V, _I, _V, _V]),
    "randt_search_global": (_I, [_V, _V, _I, _V, _I, _P(MatcherParams), _P(BnbParams), C.c_double, C.c_double, C.c_double, _V,
                                 _P(C.c_double), _P(_I)]),
    "randt_cs_divergence_batch_dev": (_I, [_V, _V, _I, _I, _V, _V, _I, _I, _V, _V, _V]),
    "randt_cs_divergence": (_I, [_V, _V, _I, _V, _I, _V, _V, _V]),
    "randt_filter_scan_batch_dev": (_I, [_V, _V, _I, _I, _I, _I, _I, _P(FilterParams), _V, _I, _V, _V, _V, _V, _V]),
    "randt_filter_scan": (_I, [_V, _V, _I, _I, _I, _I, _P(FilterParams), _V, _I, _P(_I), _V, _V, _P(_I), _P(_I)]),
    "randt_filter_build": (_I, [_V, _V, _I, _I, _I, _I, _P(FilterParams), _P(ClusterParams), _I, _V, _I, _P(_I)]),
    "randt_sc_make_batch_dev": (_I, [_V, _V, _I, _I, _V, _I, _I, _P(ScParams), _V, _V, _V]),
    "randt_sc_detect_batch_dev": (_I, [_V, _P(ScParams), _V, _V, _V, _V, _I, _V, _I, _V, _V, _V]),
    "randt_sc_db_create": (_I, [_V, _P(ScParams), _I, _P(_V)]),
    "randt_sc_db_destroy": (None, [_V]),
    "randt_sc_db_size": (_I, [_V]),
    "randt_sc_db_append": (_I, [_V, _V, _I, _I, _I, _V, C.c_double, _P(_I)]),
    "randt_sc_db_detect": (_I, [_V, _I, _P(_I), _P(C.c_float), _P(C.c_double)]),
    "randt_sc_db_download": (_I, [_V, _I, _V, _V, _V]),
    "randt_pg_params_default": (None, [_P(PgParams)]),
    "randt_pose_graph_optimize": (_I, [_V, _I, _V, _I, _V, _V, _V, _V, _I, _P(PgParams), _P(PgResult)]),
    "randt_predict_state": (_I, [_V, C.c_double, _V]),
    "randt_predict_state_param": (_I, [_V, C.c_double, _I, _V]),
    "randt_predict_state_batch": (_I, [_V, _I, C.c_double, _I, _V]),
    "randt_register_window": (_I, [_V, _V, _V, _I, _V, _V, _V, _I, _V, _P(MatcherParams), _P(WindowParams), _V, _P(_I), _V]),
    "randt_register_window_batch": (_I, [_V, _I, _V, _V, _I, _V, _V, _V, _I, _V, _P(MatcherParams), _P(WindowParams), _V, _V, _V]),
    # multi-GPU group
    "randt_shard_range": (None, [_I, _I, _I, _P(_I), _P(_I)]),
    "randt_group_create": (_I, [_P(_I), _I, _P(_V), _I, _P(_V)]),
    "randt_group_unique_id": (_I, [_V]),
    "randt_group_create_rank": (_I, [_I, _V, _I, _I, _V, _P(_V)]),
    "randt_group_destroy": (_I, [_V]),
    "randt_group_info": (_I, [_V, _P(_I), _P(_I), _P(_I), _P(_I)]),
    "randt_group_ctx": (_V, [_V, _I]),
    "randt_group_last_error": (C.c_char_p, [_V]),
    "randt_group_synchronize": (_I, [_V]),
    "randt_group_broadcast_maps": (_I, [_V, _P(_V), _I, _I, _I]),
    "randt_group_allgather_rows": (_I, [_V, _P(_V), _I, C.c_size_t]),
    "randt_group_register_batch_dev": (_I, [_V, _P(_V), _P(_V), _P(_V), _I, _P(MatcherParams), _P(_V), _P(_V), _I]),
    "randt_group_scan_register_batch_dev": (_I, [_V, _P(_V), _I, _I, _P(_V), _I, _I, _P(ClusterParams), _P(_V), _P(_V), _P(_V),
                                                 _P(MatcherParams), _P(_V), _P(_V), _I]),
    "randt_group_register_pairs": (_I, [_V, _P(_V), _V, _P(_V), _I, _P(MatcherParams), _V, _V]),
}
TRANSPORT_AUTO, TRANSPORT_PEER, TRANSPORT_RCCL = 0, 1, 2
SOLVE_AUTO, SOLVE_THROUGHPUT, SOLVE_LATENCY = 0, 1, 2
UNIQUE_ID_BYTES = 128

_lib = None


def load():
    """dlopen librandt_hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C randt-slam_amd/csrc).  There is no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so, and a process that
    # maps both that copy and /opt/rocm's ends up with two HSA runtimes of which only the first sees
    # the GPU.  Importing torch first makes librandt_hip.so's DT_NEEDED libamdhip64 resolve to the
    # copy torch already mapped (torch is the device allocator / stream provider of this harness).
    try:
        import torch  # noqa: F401
    except Exception:  # pure C/C++ callers link /opt/rocm's runtime directly
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
