// Internal declarations shared by the HIP translation units of librandt_hip.so (gfx950 only).
#pragma once
// issue priority (s_setprio, 0..3) of the latency-bound kernels that share the chip with the fp64-bound solve
#ifndef RANDT_LATENCY_KERNEL_PRIO
#define RANDT_LATENCY_KERNEL_PRIO 3
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <map>
#include <string>

#include "randt.h"

#define RANDT_WAVE 64

// Device-side view of a randt_maps batch (passed to kernels by value).
struct MapView {
  randt_cell* cells;  // [n_maps][cap]
  int32_t* counts;    // [n_maps]
  int32_t* grid;      // [n_maps][n_slots] or nullptr
  int32_t n_maps, cap, n_slots, size_x, size_y;
  int32_t rmax;       // static_cast<int>(max_neighbour_manhattan_distance / resolution), ndt_map.cpp:117
  int32_t min_points;
  int32_t pad_;
  double res, offset_x, offset_y;
};

struct randt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  // scratch (grown on demand, never inside a timed region after warm-up)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  void* h_pin = nullptr;     // pinned host memory: staging image of the window entries (one copy per direction), grown on demand
  size_t h_pin_bytes = 0;
  void* small = nullptr;     // 4 KB of device scratch for the synchronous host-level conveniences (lazily allocated)
  void* build_ws = nullptr;  // label scratch of k_ndt_build's fallback sort (its own buffer: callers stage points in ws)
  size_t build_ws_bytes = 0;
  void* build_wide_ws = nullptr;  // tiled build, scans whose labels span more than a tile's bins: sort workspace (ndt_build_big.hip)
  size_t build_wide_ws_bytes = 0;
  double* d_trace = nullptr;
  int trace_len = 0;
  int lds_limit = 160 * 1024;
  // kernel geometry knobs (RANDT_SOLVE_BLOCK, RANDT_ASSOC_STAGE_GRID; for experiments)
  int solve_block = 64;
  int solve_rpb = 4;         // independent registrations (one wavefront each) per workgroup in the pair solve: 1, 2, 4, 8
                             // (RANDT_SOLVE_RPB).  4 = one per SIMD of a CU: +9 % end to end over single-wavefront workgroups,
                             // which the dispatcher places unevenly when they arrive from 16 queues
  int solve_group = -1;      // the one-wavefront solve's workgroups take their registrations through k_solve_order's size-sorted order:
                             // -1 = for launches of >= 8 registrations per compute unit, 0 / 1 = never / always (RANDT_SOLVE_GROUP;
                             // round-5 experiment, profiles/experiments/r05_solve_length_grouping.md)
  void* order_ws = nullptr;  // the order itself (int32 per registration, grown on demand)
  size_t order_ws_bytes = 0;
  int solve_split = -1;      // wavefronts per registration in the pair solve: -1 = chosen from the batch size (solve.hip, split_width),
                             // 0 = never split, 2..8 = forced (RANDT_SOLVE_SPLIT; experiments)
  int solve_mode = 0;        // RANDT_SOLVE_AUTO / RANDT_SOLVE_THROUGHPUT / RANDT_SOLVE_LATENCY (randt_ctx_set_solve_mode)
  std::atomic<long long> last_enqueue_ns{0};  // CLOCK_MONOTONIC of this context's last kernel enqueue, 0 = known to have drained
                                              // (randt_note_enqueue / randt_device_shared: the placement decision of RANDT_SOLVE_AUTO)
  int last_placement = 0;    // debug: what the last pair-solve launch chose: split width W (>= 2), 0 = one wavefront per registration
  int n_cus = 256;           // compute units of the context's device
  int lds_atomics_lane_ordered = 0;  // device self-test at context creation (api.hip): same-address LDS atomics of one instruction
                                     // are served in ascending lane order -> the build kernels rank points with one atomic each
  int32_t* misrank_word = nullptr;   // pinned host word a build workgroup that had to re-rank sets to 1 (plain store, device-visible)
  int32_t* d_misrank_count = nullptr;  // device word: how many workgroups re-ranked with ballots so far (randt_debug_build_rank_fallbacks)
  int window_general = 0;            // RANDT_WINDOW_GENERAL=1: every window takes window_gen.hip (tests: the two kernels agree)
  int debug_force_misrank = 0;       // RANDT_DEBUG_FORCE_MISRANK=1: test hook, makes the in-kernel order check fail
  int build_tiled = 0;       // RANDT_BUILD_TILED=1: every scan through the multi-workgroup build (normally only > 7168 points)
  int assoc_tp_ppw = 4;      // pairs per association workgroup when batches share the chip (RANDT_ASSOC_TP_PPW)
  int assoc_tp_ch = 64;      // cells per chunk of the association when batches share the chip (16 / 32 / 48 / 64; RANDT_ASSOC_TP_CH)
  int assoc_stage_grid = 0;  // 1: stage the fixed map's index grid in LDS; 0: gather it from L2 (same speed alone, but 36 KB instead of 76 KB of LDS leaves room for co-running build workgroups: +2 % end to end)
  // ---- device storage pool (api.hip, randt_dev_alloc / randt_dev_release): the blocks of destroyed map batches and of the
  // host-level entries' temporaries are kept and handed out again.  A block goes back the moment its owner is destroyed,
  // WITHOUT a synchronisation: everything a context does is enqueued on its one stream, so whatever is enqueued on the
  // block's next owner runs after the previous owner's last kernel.  A batch that OTHER contexts have been handed (as the
  // fixed / moving side of a registration, the source of a copy or merge, ...) remembers them (randt_maps::foreign), and its
  // destruction makes the owner's stream wait -- on the device, no host wait -- for a marker event recorded on each of their
  // streams before the block is parked: the block's next owner also runs behind every foreign reader.
  std::multimap<size_t, void*> pool_free;  // block size -> block
  size_t pool_bytes = 0;                   // bytes parked in pool_free
  size_t pool_cap = (size_t)1 << 30;       // park at most this much (RANDT_POOL_MAX_BYTES); beyond it blocks are really freed
  randt_pool_stats stats{};                // allocator / synchronisation counters (randt_ctx_pool_stats)
  // ---- pinned, device-visible argument ring (randt_pin_take): poses, indices and scan points of the host-level entries are
  // written here by the host and read by the device (kernels directly, or an async copy) -- no pageable staging copy, no
  // synchronisation "because the buffer is reused": a segment is only reused after the event recorded behind its last user
  char* pin_ring = nullptr;
  static constexpr int kPinSegs = 4;
  static constexpr size_t kPinSegBytes = 512 * 1024;
  int pin_cur = 0;
  size_t pin_off = 0;
  hipEvent_t pin_ev[kPinSegs] = {nullptr, nullptr, nullptr, nullptr};
  bool pin_pending[kPinSegs] = {false, false, false, false};
  randt_maps* tmp_cluster = nullptr;  // one-cell scratch map of randt_maps_insert_cluster (created once per context)
  size_t merge_lds_granted = 0;       // dynamic LDS k_maps_merge has been granted on this context's device (mapops.hip)
  hipEvent_t marker_ev = nullptr;     // "everything this context has enqueued so far", recorded when a batch ANOTHER context owns
                                      // and this one has used is destroyed (randt_maps_destroy waits for it on the owner's stream)
};

// RANDT_SOLVE_AUTO's question "does this batch have the device to itself?" (api.hip).  Every launcher of the hot path notes its
// enqueue; a context that is about to choose a placement asks whether ANOTHER context of this process has work in flight on the
// same device: one that enqueued within the last 100 us (cannot have drained unless the library itself synchronised it, which
// resets its stamp), or -- for older stamps -- one whose stream hipStreamQuery reports busy (asked once per stamp).
void randt_note_enqueue(randt_ctx* ctx);
bool randt_device_shared(randt_ctx* ctx);
// the placement of a batch that shares the chip: the caller said so, or the library sees it
inline bool randt_throughput_placement(randt_ctx* ctx) {
  if (ctx->solve_mode == RANDT_SOLVE_THROUGHPUT) return true;
  if (ctx->solve_mode == RANDT_SOLVE_LATENCY) return false;
  return randt_device_shared(ctx);
}

// counted wrappers (randt_ctx_pool_stats reports them; tests assert a steady-state scan of the drop-in drive makes none)
hipError_t randt_dev_alloc(randt_ctx* ctx, void** p, size_t bytes, size_t* granted = nullptr);  // pool first, hipMalloc otherwise
void randt_dev_release(randt_ctx* ctx, void* p, size_t bytes);                                   // back to the pool (bytes = what was granted)
hipError_t randt_hip_malloc(randt_ctx* ctx, void** p, size_t bytes);  // plain hipMalloc, counted (grow-only workspaces)
hipError_t randt_hip_free(randt_ctx* ctx, void* p);
hipError_t randt_sync(randt_ctx* ctx);                                // hipStreamSynchronize(ctx->stream), counted
// bytes of pinned host memory the device can read under the same address, valid until the stream has passed the work
// enqueued next; nullptr if the request is larger than a segment (callers fall back to a pageable copy + synchronisation)
void* randt_pin_take(randt_ctx* ctx, size_t bytes);

struct randt_maps {
  randt_ctx* ctx = nullptr;
  randt_map_params p{};
  MapView v{};
  bool owns = false;
  void* block = nullptr;    // owns: ONE pooled block [cells | counts | 2 deferred-status words | grid]
  size_t block_bytes = 0;
  bool deferred_pending = false;  // an asynchronous insert has run since the last synchronising read (api.hip, deferred_status)
  // contexts other than `ctx` that were handed this (library-owned) batch through an entry point: their streams may still read
  // or write it when the owner destroys it (randt_note_user / wait_for_foreign_users in api.hip).  Lock-free: several contexts
  // may use one read-only batch from several threads.
  static constexpr int kForeignSlots = 32;
  mutable std::atomic<randt_ctx*> foreign[kForeignSlots] = {};
  mutable std::atomic<bool> foreign_overflow{false};  // more users than slots: every live context counts as one
};
void randt_note_foreign_user(randt_ctx* user, const randt_maps* m);
// called by every entry point that enqueues work of `user`'s stream on batch `m`
inline void randt_note_user(randt_ctx* user, const randt_maps* m) {
  if (m && user && m->ctx != user && m->owns) randt_note_foreign_user(user, m);
}

// Parameters of the solve kernel (POD copy of randt_matcher_params + derived values).
struct SolveParams {
  double loss_a, mu_scale, alpha, weight, gnc_div;
  double mu_cap, mu_stop;  // gnc_div^(gnc_steps - 1) and 1 / sqrt(gnc_div) (ndt_matcher.cpp:388-397,475-483), formed once on the host
  double ftol, gtol, ptol, r0, rmax, rmin, min_rel, dmin, dmax;
  int32_t gnc_steps, max_it, k, max_invalid;
};

int randt_set_error(randt_ctx* ctx, int status, const char* what, hipError_t e);
#define RANDT_HIP_CHECK(ctx, call)                                          \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess) return randt_set_error((ctx), RANDT_ERR_HIP, #call, e__); \
  } while (0)

// Makes the context's device current for the duration of an entry point and restores the caller's afterwards: a process
// (or thread) may hold contexts on several GPUs, and the caller's current device may have been changed behind our back
// (e.g. torch.cuda.set_device).  Allocations and launches of a context always land on ITS device.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const randt_ctx* ctx) {
    if (!ctx) return;
    if (hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// launchers implemented in the kernel TUs
int launch_ndt_build(randt_ctx* ctx, const float* d_points, int n_scans, int pitch, const int32_t* d_n_points,
                     int stride, int ioff, const randt_cluster_params* cp, const MapView& out, int first_map, const float* d_polar = nullptr,
                     const float* beam_cov9 = nullptr);
size_t ndt_build_big_ws_bytes(int n_scans, int pitch, int with_polar = 0);
int launch_ndt_build_big(randt_ctx* ctx, const float* d_points, int n_scans, int pitch, const int32_t* d_n_points, int stride,
                         int ioff, const randt_cluster_params* cp, const MapView& out, int first_map, void* d_ws, const float* d_polar = nullptr,
                         const float* beam_cov9 = nullptr);
int launch_ndt_build_big_wide(randt_ctx* ctx, const float* d_points, int n, int stride, int ioff, const randt_cluster_params* cp,
                              const MapView& out, int map, const float* d_polar, const float* beam_cov9);
int launch_maps_transform(randt_ctx* ctx, const MapView& m, int first, int count, const double* d_pose4);
int launch_maps_reindex(randt_ctx* ctx, const MapView& m, int first, int count);
int launch_maps_append(randt_ctx* ctx, const MapView& dst, int dst_idx, const MapView& src, int src_idx, int set_grid,
                       int32_t* d_status, int accumulate /* status[] += instead of = */);
int launch_maps_merge(randt_ctx* ctx, const MapView& fixed, int fixed_idx, const MapView& moving, int moving_first,
                      int n_moving, const double* d_pose4, int n_pairs = 1);
int launch_associate(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving,
                     int moving_first, int n_pairs, const double* d_guess4, int k, int lookup_mahalanobis,
                     int use_intensity, int32_t* d_corr, const int32_t* d_moving_idx = nullptr);

#define RANDT_CS_SELF_OUTER 16  // csdiv.hip: outer cells per workgroup of the self-term kernel (= one partial sum; api.hip sizes the workspace)
// fixed-lag window problem description (kernel argument, by value)
#define RANDT_WIN_MAX_STATES 13  // <= 12 optimised states + the constant one (window.hip takes <= 3 + 1, window_gen.hip 4..7, window_gen_big.hip 8..12)
#define RANDT_WIN_MAX_TERMS 24   // (state, fixed map) NDT terms: 12 states x 2 fixed maps
struct WinDesc {
  int S, n_terms, n_tan, n_amb, use_imu, const_vel, k, d3;
  int vec, pad_;  // vec: (pos[2], rot) parameter blocks instead of the SE(2) manifold (optimize_on_manifold: false); pad_: 1 = RANDT_PARAM_ANALYTIC
  int term_state[RANDT_WIN_MAX_TERMS], term_moving[RANDT_WIN_MAX_TERMS], term_fixed[RANDT_WIN_MAX_TERMS];
  int off_tan[RANDT_WIN_MAX_STATES][5], off_amb[RANDT_WIN_MAX_STATES][5];  // pose, v, w, a, b; -1 = constant
  double sqrtI[64];
  double imu[RANDT_WIN_MAX_STATES];
  double raw_dt[RANDT_WIN_MAX_STATES];  // stamp[j] - stamp[j-1], index j
  double w_imu, w_bias, ndt_weight;
};
// n_windows windows of ONE shape (desc = the first one's: S, n_terms, layout) in one launch, one workgroup each; window w's
// descriptor is d_desc[w], its correspondence tables / states / result lie w * corr_stride ints / w * state_stride doubles / w
// records behind the first window's
int launch_solve_window(randt_ctx* ctx, const MapView& fixed, const MapView& moving, const WinDesc& desc, const WinDesc* d_desc,
                        const int32_t* d_corr, const randt_matcher_params* mp, double* d_states /* (S+1) x 12 */,
                        randt_result* d_result, int n_windows = 1, int corr_stride = 0, int state_stride = 0);
// window_gen.hip: the general kernel (4..7 optimised states); launch_solve_window routes to it, and it routes 8..12 states to
// the second compilation of the same source (window_gen_big.hip)
int launch_solve_window_gen(randt_ctx* ctx, const MapView& fixed, const MapView& moving, const WinDesc& desc, const WinDesc* d_desc,
                            const int32_t* d_corr, const SolveParams& P, double* d_states, randt_result* d_result, int n_windows,
                            int corr_stride, int state_stride);
int launch_solve_window_gen_big(randt_ctx* ctx, const MapView& fixed, const MapView& moving, const WinDesc& desc, const WinDesc* d_desc,
                                const int32_t* d_corr, const SolveParams& P, double* d_states, randt_result* d_result, int n_windows,
                                int corr_stride, int state_stride);
int launch_solve(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving,
                 int moving_first, int n_pairs, const int32_t* d_corr, const randt_matcher_params* mp,
                 double* d_pose4, randt_result* d_results);

int launch_filter_scan(randt_ctx* ctx, const float* d_raw, int n_scans, int n_az, int n_bins, int stride, int ioff,
                       const randt_filter_params* fp, float* d_out_pts, int pitch_out, int32_t* d_out_counts, float* d_polar,
                       float* d_peaks, int32_t* d_peak_counts, int32_t* d_status, void* d_scratch /* n_scans * n_az * 32 B */);

int launch_cs_divergence(randt_ctx* ctx, const MapView& fixed, int fixed_first, int fixed_count, const int32_t* d_fixed_idx,
                         const MapView& moving, int moving_first, int n_pairs, const double* d_pose4, double* d_partial,
                         double* d_out, double* d_terms);

int launch_sc_make(randt_ctx* ctx, const float* d_points, int n_scans, int pitch, const int32_t* d_n_points, int stride, int ioff,
                   const randt_sc_params* p, double* d_desc, double* d_ring_keys, double* d_sector_keys);
size_t sc_detect_ws_bytes(int n_queries, int n_db, int n_cand);  // workspace launch_sc_detect needs behind d_ws
int launch_sc_detect(randt_ctx* ctx, const randt_sc_params* p, const double* d_desc, const double* d_ring_keys, const double* d_pos,
                     const double* d_dist, int n_db, const int32_t* d_query_ids, int n_queries, float* d_ws, int32_t* d_loop_id,
                     float* d_yaw, double* d_min_dist);

int launch_points_transform(randt_ctx* ctx, float* d_pts, int n, int stride, const double* d_pose4);
int launch_cells_op(randt_ctx* ctx, int op, randt_cell* d_a, const randt_cell* d_b, int n, const double* d_pose4, double* d_out);
int launch_maps_insert_clusters(randt_ctx* ctx, const MapView& dst, int dst_idx, const float* d_pts, const int32_t* d_offsets, int n_clusters,
                                int stride, int ioff, int32_t* d_status, int accumulate, int32_t* d_n_accepted);
int launch_cell_update(randt_ctx* ctx, randt_cell* d_cell, const float* d_pts, int k, int stride, int ioff, int min_points,
                       int32_t* d_accepted);

int launch_eval_cost(randt_ctx* ctx, const MapView& fixed, int fmap, const MapView& moving, int mmap, const int32_t* d_corr, int k,
                     int use_intensity, double scale, double alpha, const double* d_poses4, int n_poses, double* d_cost,
                     int32_t* d_n_res);
