"""randt-slam_amd -- MI355X-native NDT scan-matching core (host-side Python binding).

The product is ``librandt_hip.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/randt.h``); this package binds it for the benchmark / parity harness and ships the
synthetic-scene generator.  Nothing here computes on the CPU: every compute entry point goes
through the shared library and fails loudly when it (or a GPU) is missing.
"""
from . import _capi, synth  # noqa: F401
from ._capi import (CELL_DTYPE, RESULT_DTYPE, PARAM_AMBIENT4, PARAM_ANALYTIC, PARAM_MANIFOLD, PARAM_VECTOR,  # noqa: F401
                    STATE_DTYPE, ClusterParams, MapParams, MatcherParams, WindowParams)
from .host import (Context, Group, Maps, RandtError, group_unique_id, shard_range, associate_batch, default_matcher_params, indoor_cluster_params,  # noqa: F401
                   indoor_map_params, make_state, ndt_build_batch, ndt_build_pndt_batch, predict_state, register_batch, register_window, register_window_batch,
                   scan_register_batch, solve_batch, window_params)

__all__ = [
    "Context", "Maps", "RandtError", "MapParams", "ClusterParams", "MatcherParams", "CELL_DTYPE", "RESULT_DTYPE",
    "PARAM_MANIFOLD", "PARAM_AMBIENT4", "PARAM_VECTOR", "PARAM_ANALYTIC", "default_matcher_params", "indoor_map_params",
    "indoor_cluster_params", "ndt_build_batch", "ndt_build_pndt_batch", "associate_batch", "solve_batch", "register_batch",
    "scan_register_batch", "synth", "STATE_DTYPE", "WindowParams", "make_state", "window_params", "predict_state",
    "register_window", "register_window_batch", "Group", "group_unique_id", "shard_range",
]
