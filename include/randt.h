/*
 * randt.h -- C ABI of librandt_hip.so: MI355X-native (gfx950, hand-written HIP) NDT scan-matching
 * core that drops in behind RaNDT-SLAM's ndt_representation / ndt_registration API.
 *
 * The reference (IGMR-RWTH/RaNDT-SLAM) has no FFI/plugin layer: the seam is the C++ public API
 * that LocalFuser calls.  Every entry point below cites the reference interface
 * it replaces; paths are relative to /root/reference/ros/ndt_radar_slam/.  The reference-side
 * binding a maintainer would add is shown in INTEGRATION.md; include/randt_facade.hpp mirrors the
 * reference's Cell / Map / Matcher classes on top of this ABI.
 *
 * Conventions
 *   - every function returns a randt_status (0 = ok); nothing throws across the ABI;
 *   - poses are double[4] = [cos, sin, tx, ty], the memory layout of Sophus::SE2d::data()
 *     (include/ndt_slam/trajectory_representation.h:14, ndt_matcher.cpp:292);
 *   - cells are float (ndt_cell.h:167-168), solves are double (ndt_matcher.cpp:231);
 *   - pointers named d_* are DEVICE pointers (HBM resident), h_* are host pointers;
 *   - all work is enqueued on the context's HIP stream; *_dev entry points do not synchronise;
 *   - a context is single-caller (the reference Matcher is not re-entrant either).
 */
#ifndef RANDT_H
#define RANDT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RANDT_VERSION 100

typedef enum randt_status {
  RANDT_OK = 0,
  RANDT_ERR_INVALID = 1,      /* bad argument */
  RANDT_ERR_HIP = 2,          /* HIP runtime error (see randt_last_error) */
  RANDT_ERR_UNSUPPORTED = 3,  /* size beyond the kernels' limits */
  RANDT_ERR_NOMEM = 4,
  RANDT_ERR_NODEVICE = 5      /* no gfx950 device visible: the library never falls back to CPU */
} randt_status;

/* 48-byte NDT cell record.  Numeric payload of rc::navigation::ndt::Cell
 * (include/ndt_representation/ndt_cell.h:164-168): mean (x, y, intensity), upper triangle of the
 * 3x3 covariance in the order of ndt_msgs/msg/Covariance.msg (xx,xy,xi,yy,yi,ii), point count,
 * max intensity (ndt_cell.h:166). */
typedef struct randt_cell {
  float mean[3];
  float cov[6];
  uint32_t n;
  float max_intensity;
  uint32_t reserved;
} randt_cell;

/* rc::navigation::ndt::NDTMapParameters (include/ndt_slam/ndt_slam_parameters.h:17-28) + the
 * centre passed to Map::initialize (ndt_map.cpp:7-21).  size_* are in CELLS (ndt_slam.cpp:653-654). */
typedef struct randt_map_params {
  int32_t size_x, size_y;
  double resolution;
  double center_x, center_y;
  double max_neighbour_dist; /* max_neighbour_manhattan_distance */
  int32_t min_points_per_cell;
  int32_t reserved;
} randt_map_params;

/* RadarPreprocessorParameters::n_clusters / max_range as consumed by Grid::cluster
 * (src/radar_preprocessing/grid.cpp:7-14, cluster_generator.h:40-42). */
typedef struct randt_cluster_params {
  int32_t n_clusters;
  float max_range;
} randt_cluster_params;

/* Parameter blocks the NDT residuals hang on (ndt_matcher.cpp:217-246, 290-313):
 *   MANIFOLD  SE(2) pose with Sophus' manifold             (optimize_on_manifold: true; the window solve's default)
 *   AMBIENT4  the un-manifolded [c, s, tx, ty] block estimateLoopConstraint really optimises in that configuration
 *   VECTOR    pos[2] + rot[1], autodiff functors           (optimize_on_manifold: false)
 *   ANALYTIC  pos[2] + rot[1] with the reference's hand-written functors NDTFrameToMap{,Intensity}FactorResidualAnalytic
 *             (use_analytic_expressions_for_optimization: true; ceres_residuals.h:207-305).  Their rotation Jacobian is not the
 *             derivative for theta != 0 and is reproduced as written, because Ceres iterates with exactly that
 *             row; the analytic motion / IMU factors (:794-889, :372-419) have correct Jacobians = the VECTOR factors. */
enum { RANDT_PARAM_MANIFOLD = 0, RANDT_PARAM_AMBIENT4 = 1, RANDT_PARAM_VECTOR = 2, RANDT_PARAM_ANALYTIC = 3 };

enum {
  RANDT_TERM_NONE = 0,
  RANDT_TERM_CONVERGENCE_FUNCTION = 1,
  RANDT_TERM_CONVERGENCE_PARAMETER = 2,
  RANDT_TERM_CONVERGENCE_GRADIENT = 3,
  RANDT_TERM_CONVERGENCE_RADIUS = 4,
  RANDT_TERM_NO_CONVERGENCE = 5,
  RANDT_TERM_FAILURE = 6
};

/* Subset of rc::navigation::ndt::NDTMatcherParameters (ndt_slam_parameters.h:52-84) used by the
 * pair registration, plus the Ceres 2.1.0 Solver::Options the reference leaves at their defaults
 * (ndt_matcher.cpp:457-464).  Fill with randt_matcher_params_default() first. */
typedef struct randt_matcher_params {
  double loss_scale;   /* BarronLoss 'a': loss_function_scale, or `scale` of estimateLoopConstraint */
  double mu_scale;     /* loss_function_scale used in the gnc_mu formula (ndt_matcher.cpp:388,475) */
  double loss_alpha;   /* loss_function_convexity */
  double loss_weight;  /* ScaledLoss factor (ndt_matcher.cpp:392: ndt_weight/(n_cells*k); :479: 1) */
  double gnc_divisor;  /* gnc_control_parameter_divisor */
  int32_t gnc_steps;   /* gnc_steps / max_gnc_steps */
  int32_t max_iterations;
  int32_t n_neighbours;        /* n_results_kd_lookup */
  int32_t lookup_mahalanobis;
  int32_t use_intensity;       /* use_intensity_as_dimension */
  int32_t parameterization;    /* RANDT_PARAM_* ; AMBIENT4 = what estimateLoopConstraint really optimises */
  int32_t max_consecutive_invalid_steps;
  int32_t reserved;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius, max_radius, min_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
} randt_matcher_params;

/* Per-registration output (64 bytes). `cost` is what estimateLoopConstraint returns
 * (summary.final_cost / summary.num_residual_blocks, ndt_matcher.cpp:492). */
typedef struct randt_result {
  double cost;
  double final_cost;
  double initial_cost;
  double mu0;
  int32_t n_residuals;
  int32_t iterations;   /* minimizer iterations summed over the GNC solves (incl. each iteration 0) */
  int32_t gnc_solves;
  int32_t termination;  /* RANDT_TERM_* of the last solve */
  int32_t n_evals;      /* residual(+Jacobian) passes over the correspondence set */
  int32_t status;       /* 0 ok, 1 no residuals ("WARNING: NO RESIDUALS ADDED!", ndt_matcher.cpp:454), 2 numerical failure */
  int32_t reserved[2];
} randt_result;

/* rc::navigation::ndt::State (include/ndt_slam/trajectory_representation.h:12-22) as a POD:
 * both pose representations, velocities, acceleration, IMU bias, stamp. 14 doubles. */
typedef struct randt_state {
  double pose[4]; /* Sophus::SE2d data: cos, sin, tx, ty */
  double pos[2];
  double rot;
  double lin_vel[2];
  double rot_vel;
  double lin_acc[2];
  double imu_bias;
  double stamp;
} randt_state;

/* Fixed-lag smoother part of NDTMatcherParameters (ndt_slam_parameters.h:52-84). */
typedef struct randt_window_params {
  double motion_sqrtI[64];  /* covariance_scaling_factor * motion_sqrtI, row-major 8x8 (ndt_matcher.cpp:99) */
  double ndt_weight;
  double weight_imu, weight_imu_bias;
  double pose_reject_translation, pose_reject_rotation;
  int32_t smoothing_steps;
  int32_t use_imu;
  int32_t use_constant_velocity_model;
  int32_t reserved;
} randt_window_params;

typedef struct randt_ctx randt_ctx;
typedef struct randt_maps randt_maps; /* a batch of device-resident NDT maps with common parameters */

/* ------------------------------------------------------------------ context ----------------- */
int randt_version(void);
const char* randt_status_string(int status);
const char* randt_last_error(const randt_ctx* ctx);
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the null stream. */
int randt_ctx_create(int device, void* stream, randt_ctx** out);
int randt_ctx_destroy(randt_ctx* ctx);
int randt_ctx_set_stream(randt_ctx* ctx, void* stream);
int randt_ctx_synchronize(randt_ctx* ctx);
/* Debug/parity hook: when set, each solve writes its per-iteration (cost, radius, flag) triplets
 * (max_len per registration; first double of each block = number written).  NULL disables. */
int randt_ctx_set_trace(randt_ctx* ctx, double* d_trace, int max_len);
/* How the pair solve lays a batch out on the device (results are bit-identical either way):
 *   RANDT_SOLVE_AUTO (default)  chosen per launch from the batch size AND from what else the process has in flight on the
 *                               device: a batch too small to give every SIMD a registration of its own (<= 3 per compute
 *                               unit: one loop-closure burst, the per-GPU share of a multi-GPU split) gets several
 *                               wavefronts per registration, which shortens the batch's latency (512 registrations: 148 ->
 *                               117 us) -- unless another context of this process has work in flight on the same device
 *                               (it enqueued within the last 100 us, or its stream reports busy): then the batch takes the
 *                               throughput placement below, as if the caller had asked for it.  Work of OTHER processes or of
 *                               the caller's own kernels is not seen;
 *   RANDT_SOLVE_THROUGHPUT      always one wavefront per registration: for callers that keep several batches in flight on
 *                               several contexts / streams (bench.py's 16-stream region), where the chip is full anyway and
 *                               helper wavefronts only take slots away from other batches;
 *   RANDT_SOLVE_LATENCY         never look at other contexts: the latency placement whenever the batch size allows it. */
enum { RANDT_SOLVE_AUTO = 0, RANDT_SOLVE_THROUGHPUT = 1, RANDT_SOLVE_LATENCY = 2 };
int randt_ctx_set_solve_mode(randt_ctx* ctx, int mode);
void randt_matcher_params_default(randt_matcher_params* p);
/* Device storage pool of a context.  The reference copies Maps by value several times per scan
 * (local_fuser.cpp:128-130,135,173-178; ndt_map.h:129-131) and inserts clusters one by one
 * (ndt_hierarchical_map.cpp:28-33); behind this ABI every such copy is a randt_maps_create / randt_maps_clone +
 * randt_maps_destroy.  Their storage therefore comes from a per-context pool: a destroyed batch parks its block (no
 * hipFree, no synchronisation -- the next owner's work is enqueued on the same stream behind the previous owner's), a
 * created batch takes a parked block of a fitting size (no hipMalloc).  In steady state a scan of the drop-in drive makes
 * no allocator call at all; the counters below are how tests / bench.py check that.
 *   device_allocs / device_frees  hipMalloc / hipFree calls this context has made (pool misses, workspace growth)
 *   stream_syncs                  host waits on the context's stream (hipStreamSynchronize / a pinned segment's event)
 *   pool_hits                     allocations served from parked blocks
 *   pool_bytes / pool_blocks      what is parked right now
 *   foreign_waits                 see below
 * randt_ctx_pool_trim really frees everything parked (synchronises); the pool never parks more than 1 GiB
 * (RANDT_POOL_MAX_BYTES).
 * BATCHES SHARED BETWEEN CONTEXTS.  A batch may be handed to entry points of OTHER contexts (the fixed / moving side of their
 * registrations, the source of a copy or a merge, the target of their builds): the batch remembers them, and
 * randt_maps_destroy makes the owner's stream wait -- on the device, not on the host -- for everything those contexts have
 * enqueued so far before the block is parked (foreign_waits counts these waits), so the block's next owner cannot overtake
 * a reader on another stream.  What stays the caller's job: not to destroy a batch from one thread while another thread is
 * inside a call that uses it, and not to enqueue NEW work on a destroyed batch.  Caller-provided storage (*_external) is
 * never pooled; its lifetime is the caller's altogether. */
typedef struct randt_pool_stats {
  int64_t device_allocs, device_frees, stream_syncs, pool_hits;
  int64_t pool_bytes, pool_blocks;
  int64_t foreign_waits; /* device-side waits of randt_maps_destroy for OTHER contexts' streams that used the batch (below) */
  int64_t reserved[1];
} randt_pool_stats;
int randt_ctx_pool_stats(const randt_ctx* ctx, randt_pool_stats* out);
int randt_ctx_pool_trim(randt_ctx* ctx);

/* ------------------------------------------------------------------ maps -------------------- */
/* Replaces Map::initialize (ndt_map.cpp:7-21) for n_maps maps at once.  Storage per map:
 * cell_capacity x 48 B compact cells (grid_), one int32 count, size_x*size_y int32 index grid
 * (grid_indizes_, -1 = empty; omitted when with_grid = 0: scan maps that are only ever the MOVING
 * side need none).  Storage is hipMalloc'ed, or caller-provided (*_external) so that e.g. a torch
 * tensor / RCCL buffer can back it. */
int randt_maps_create(randt_ctx* ctx, int n_maps, const randt_map_params* p, int cell_capacity,
                      int with_grid, randt_maps** out);
int randt_maps_create_external(randt_ctx* ctx, int n_maps, const randt_map_params* p, int cell_capacity,
                               void* d_cells, void* d_counts, void* d_grid /* nullable */, randt_maps** out);
int randt_maps_destroy(randt_maps* m);
size_t randt_maps_cells_bytes(int n_maps, int cell_capacity);
size_t randt_maps_grid_bytes(int n_maps, const randt_map_params* p);
int randt_maps_info(const randt_maps* m, int* n_maps, int* cell_capacity, int* n_slots, int* with_grid);
int randt_maps_device_ptrs(const randt_maps* m, void** d_cells, void** d_counts, void** d_grid);
/* Map::clear (ndt_map.cpp:252-259) + re-initialise the index grid; async on the stream. */
int randt_maps_clear(randt_maps* m, int first, int count);
/* Host <-> device copies of one map (synchronous).  h_grid may be NULL. */
int randt_maps_upload(randt_maps* m, int idx, const randt_cell* h_cells, int n_cells, const int32_t* h_grid);
int randt_maps_download(randt_maps* m, int idx, randt_cell* h_cells, int max_cells, int* n_cells, int32_t* h_grid);
int randt_maps_counts(randt_maps* m, int first, int count, int32_t* h_counts);
/* Device copy of whole maps (Map copy-construction, local_fuser.cpp:43-44,128-129): one launch for cells, counts and
 * index grids of all `count` maps, async on dst's stream. */
int randt_maps_copy(randt_maps* dst, int dst_first, const randt_maps* src, int src_first, int count);
/* Map's copy constructor in one call: a new batch with the geometry, capacity and (if src has one) index grid of src,
 * holding a copy of maps [first, first + count) -- pooled storage, ONE launch, no clearing pass, async. */
int randt_maps_clone(const randt_maps* src, int first, int count, randt_maps** out);

/* ------------------------------------------------------------------ NDT build (a1-a4) ------- */
/* Replaces RadarPreprocessor::processScan's clustering + HierarchicalMap::addClusters:
 * Grid::cluster (grid.cpp:7-14), ClusterGenerator::labelClouds (radar_preprocessor.cpp:151-169),
 * Map::insertCluster (ndt_map.cpp:238-245), Cell::addPointCloud/updateCell (ndt_cell.cpp:25-114).
 * d_points: n_scans scans, each `pitch_points` points of `stride_floats` floats (x at 0, y at 1,
 * intensity at `intensity_index`; pcl::PointXYZI is stride 8 / index 4, packed xyzI is 4 / 3).
 * d_n_points (nullable): per-scan point count <= pitch_points (ragged batches).
 * Scan s is written to out map (first_map + s), which is cleared first. */
int randt_ndt_build_batch_dev(randt_ctx* ctx, const float* d_points, int n_scans, int pitch_points,
                              const int32_t* d_n_points, int stride_floats, int intensity_index,
                              const randt_cluster_params* cp, randt_maps* out, int first_map);
/* Host convenience for one scan: the points are staged through the context's pinned ring (the host buffer is free when the
 * call returns) and the build is enqueued -- no synchronisation (scans too large for a ring segment, > 512 KB, take a pageable
 * copy and wait for it). */
int randt_ndt_build(randt_ctx* ctx, const float* h_points, int n_points, int stride_floats,
                    int intensity_index, const randt_cluster_params* cp, randt_maps* out, int map_idx);
/* The same with pNDT cells: Cell::updateCell's `params_.use_pndt` branch (ndt_cell.cpp:67-82, 102; NDTCellParameters
 * {beam_cov, use_pndt}, ndt_slam_parameters.h:12-15; false in every shipped configuration).  d_polar:
 * [n_scans][pitch_points][2] floats = (angle, range) of every point -- what RadarPreprocessor::filterScan emits beside the
 * cloud (radar_preprocessor.cpp:116; randt_filter_scan_batch_dev's d_polar) and labelClouds deals out per cluster
 * (:164-167); beam_cov9: row-major 3x3 sensor covariance (angle, range, intensity), host memory.  Every point adds
 * J beam_cov J^T with J = d(x, y, i) / d(angle, range, i); the cell covariance is the sample covariance plus the mean of
 * those, and the eigenvalue regularisation is skipped.  Always the multi-workgroup (tiled) build; fp32, parity with the
 * oracle to rounding of sin / cos (spec decision 10). */
int randt_ndt_build_pndt_batch_dev(randt_ctx* ctx, const float* d_points, int n_scans, int pitch_points,
                                   const int32_t* d_n_points, int stride_floats, int intensity_index, const float* d_polar,
                                   const float* beam_cov9, const randt_cluster_params* cp, randt_maps* out, int first_map);

/* ------------------------------------------------------------------ transform / merge (a9,a18) */
/* Map::transformMap (ndt_map.cpp:177-182, Cell::transformCell ndt_cell.cpp:117-123); like the
 * reference it leaves the index grid stale.  h_pose4: one pose per map (copied before the call returns; asynchronous). */
int randt_maps_transform(randt_maps* m, int first, int count, const double* h_pose4);
/* Rolling-submap update: for t = 0..n_moving-1, transform moving map (moving_first + t) by
 * h_pose4[t] (Map::transformMapWithPointCloud, local_fuser.cpp:175-177) and merge it into
 * fixed map fixed_idx with Map::mergeMapCell (ndt_map.cpp:191-207, Cell::operator+= ndt_cell.h:133-142),
 * strictly in order.  The moving maps themselves are not modified.  Asynchronous (the poses are copied before the call returns). */
int randt_maps_merge(randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_first,
                     int n_moving, const double* h_pose4);
/* The same for n_fixed independent submaps in ONE launch (replicas of the odometry advancing in lock-step): fixed map
 * (fixed_first + p) receives moving maps [moving_first + p * n_moving_each, + n_moving_each) at h_pose4[p * n_moving_each + t],
 * each merge exactly randt_maps_merge's. */
int randt_maps_merge_batch(randt_maps* fixed, int fixed_first, int n_fixed, const randt_maps* moving, int moving_first,
                           int n_moving_each, const double* h_pose4);
/* NOT in the reference: rebuild the index grid of maps [first, first+count) from the cells' current means (async).
 * Map::transformMap leaves grid_indizes_ stale (ndt_map.cpp:177-182), so a transformed map answers getClosestCells
 * through the slots its cells USED to occupy -- which is what randt_maps_transform reproduces.  Callers that want a
 * transformed map to be searchable again (the fixed submap hand-over, see DESIGN "reference quirks") call this. */
int randt_maps_reindex(randt_maps* m, int first, int count);

/* Cell-by-cell edits and single-cell queries of the reference's Map, host-level conveniences (a few tiny launches and
 * a synchronisation each; the batched entries above are the hot path):
 *  - randt_maps_insert_cluster: Map::insertCluster (ndt_map.cpp:238-245) -- ONE cell from all the points
 *    (Cell::addPointCloud / updateCell), appended if accepted (n > min_points), its mean's slot pointed at it.
 *    *accepted (nullable) = 1 if a cell was added; a cluster whose mean lies outside the index grid is dropped like
 *    the batched build drops it (the reference's std::vector::at throws there).  With accepted = NULL on a library-owned
 *    batch the call is ASYNCHRONOUS (HierarchicalMap::addClusters inserts hundreds of clusters per scan,
 *    ndt_hierarchical_map.cpp:28-33): nothing is read back, and a cluster that could not be placed (outside the grid ->
 *    RANDT_ERR_INVALID, capacity exhausted -> RANDT_ERR_UNSUPPORTED) is reported ONCE by the next synchronising read of the
 *    batch -- randt_maps_counts / randt_maps_download return that status with their outputs valid.
 *  - randt_maps_insert_cells: Map::insertCell (ndt_map.h:137-140) for set_grid = 0 (cells appended, index grid
 *    untouched); set_grid = 1 also points each cell's slot at it (the tail of insertCluster).
 *  - randt_closest_cells: Map::getClosestCells (ndt_map.cpp:101-151) for n_queries query cells: h_out[q][k] compact
 *    cell indices in ascending (distance, index) order, -1 padded.  lookup_mahalanobis = 0 is the Vector2f overload
 *    (Euclidean distance of the query mean to the cell means), 1 the Cell overload (mahalanobisSquaredIntensity). */
int randt_maps_insert_cluster(randt_maps* m, int idx, const float* h_points, int n_points, int stride_floats,
                              int intensity_index, int* accepted);
int randt_maps_insert_cells(randt_maps* m, int idx, const randt_cell* h_cells, int n_cells, int set_grid);
/* HierarchicalMap::addClusters (ndt_hierarchical_map.cpp:28-33) in ONE call and ONE launch: Map::insertCluster for every cluster
 * of a list, in order -- cluster c = points [h_offsets[c], h_offsets[c + 1]) of h_points (n_clusters + 1 offsets).  The same
 * cells, order and index grid as n_clusters randt_maps_insert_cluster calls (a later cluster wins a shared slot).
 * n_accepted != NULL: synchronous, *n_accepted = clusters that became cells, unplaceable ones reported by the status;
 * n_accepted == NULL on a library-owned batch: asynchronous, reported by the next synchronising read (see above). */
int randt_maps_insert_clusters(randt_maps* m, int idx, const float* h_points, const int32_t* h_offsets, int n_clusters,
                               int stride_floats, int intensity_index, int* n_accepted);
int randt_closest_cells(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_cell* h_queries, int n_queries,
                        int k, int lookup_mahalanobis, int use_intensity, int32_t* h_out);

/* Single cells (the mutators of rc::navigation::ndt::Cell, for callers that edit cells one by one; host buffers, a tiny
 * launch and a synchronisation each -- same device arithmetic as the batched kernels):
 *  - randt_cell_add_points: Cell::addPointCloud + updateCell (ndt_cell.cpp:25-114) on a cell that may already hold a
 *    distribution (n > 0: the recursive update :84-89, then the regularisation :102-112 again).  *accepted = 1 if the
 *    points were taken (n + n_points > min_points_per_cell); Cell::addPoint + updateCell is the same call per batch.
 *  - randt_cells_merge: Cell::operator+= (ndt_cell.h:133-142), h_acc[i] += h_other[i].
 *  - randt_cells_transform: Cell::transformCell (ndt_cell.cpp:117-123) of every cell by one pose.
 *  - randt_cells_mahalanobis: h_out[i] = h_self[i].mahalanobisSquaredIntensity(h_subtrahend[i]) (use_intensity = 1,
 *    ndt_cell.cpp:165-169) or .mahalanobisSquared (0, :158-162); fp32 like the reference, returned as double. */
int randt_cell_add_points(randt_ctx* ctx, randt_cell* h_cell, const float* h_points, int n_points, int stride_floats,
                          int intensity_index, int min_points_per_cell, int* accepted);
int randt_cells_merge(randt_ctx* ctx, randt_cell* h_acc, const randt_cell* h_other, int n);
int randt_cells_transform(randt_ctx* ctx, randt_cell* h_cells, int n, const double h_pose4[4]);
int randt_cells_mahalanobis(randt_ctx* ctx, const randt_cell* h_self, const randt_cell* h_subtrahend, int n, int use_intensity,
                            double* h_out);
/* The point half of Cell::transformCellWithPointCloud (ndt_cell.cpp:126-136): pcl::transformPointCloud of a cell's generating
 * points (Cell::getPointCloud, ndt_cell.h:146-148) by the 2-D pose, fp32, in place; x at 0, y at 1, z at 2 of every point,
 * the other floats (intensity) untouched.  The cell statistics move with randt_cells_transform. */
int randt_points_transform(randt_ctx* ctx, float* h_points, int n_points, int stride_floats, const double h_pose4[4]);

/* ------------------------------------------------------------------ association (a7,a8,a10) -- */
/* Association half of Matcher::addNDTFactor (ndt_matcher.cpp:200-215,249-253) with
 * Map::getClosestCells / getAdjacentIndizes (ndt_map.cpp:101-175) and
 * Cell::mahalanobisSquaredIntensity (ndt_cell.cpp:172-176).
 * Pair p registers moving map (moving_first + p) against fixed map d_fixed_idx[p] from guess
 * d_guess4[p].  d_corr: n_pairs x cell_capacity(moving) x k int32, -1 padded.
 * Limits (RANDT_ERR_UNSUPPORTED beyond; the reference has none): window
 * int(max_neighbour_dist / resolution) <= 16, k = mp->n_neighbours <= 16; maps narrower than the window only up to a
 * window of 8. */
int randt_associate_batch_dev(randt_ctx* ctx, const randt_maps* fixed, const int32_t* d_fixed_idx,
                              const randt_maps* moving, int moving_first, int n_pairs,
                              const double* d_guess4, const randt_matcher_params* mp, int32_t* d_corr);

/* ------------------------------------------------------------------ solve (a11-a15) ---------- */
/* GNC loop + Ceres-LM of Matcher::estimateLoopConstraint (ndt_matcher.cpp:457-492) over frozen
 * correspondences: residual NDTFrameToMap{,Intensity}FactorResidual{,SE2} (ceres_residuals.h:421-552),
 * BarronLoss/ScaledLoss (ceres_loss_functions.cpp:19-39), whole iteration loop on device.
 * d_pose4 in: initial guess, out: refined pose. */
int randt_solve_batch_dev(randt_ctx* ctx, const randt_maps* fixed, const int32_t* d_fixed_idx,
                          const randt_maps* moving, int moving_first, int n_pairs, const int32_t* d_corr,
                          const randt_matcher_params* mp, double* d_pose4, randt_result* d_results);

/* ------------------------------------------------------------------ registration ------------- */
/* Matcher::estimateLoopConstraint (ndt_matcher.cpp:426-493) for a batch: associate + solve. */
int randt_register_batch_dev(randt_ctx* ctx, const randt_maps* fixed, const int32_t* d_fixed_idx,
                             const randt_maps* moving, int moving_first, int n_pairs,
                             const randt_matcher_params* mp, double* d_pose4, randt_result* d_results);
/* Whole hot path for a batch of raw scans: NDT build -> associate -> solve
 * (LocalFuser::processScan's preprocessing + detectLoopClosures' refinement, local_fuser.cpp:102-105,335).
 * scan_maps: workspace maps (>= n_scans, with or without grid) that receive the scan NDTs. */
int randt_scan_register_batch_dev(randt_ctx* ctx, const float* d_points, int n_scans, int pitch_points,
                                  const int32_t* d_n_points, int stride_floats, int intensity_index,
                                  const randt_cluster_params* cp, const randt_maps* fixed,
                                  const int32_t* d_fixed_idx, randt_maps* scan_maps,
                                  const randt_matcher_params* mp, double* d_pose4, randt_result* d_results);
/* Host convenience: one pair, synchronous (double Matcher::estimateLoopConstraint(trans, old, new, ...)). */
int randt_register_pair(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_maps* moving,
                        int moving_idx, const randt_matcher_params* mp, double h_pose4[4], randt_result* h_result);

/* ------------------------------------------------------------------ correlative search (f-3) - */
/* csm_* members of NDTMatcherParameters (ndt_slam_parameters.h:76-83). */
typedef struct randt_bnb_params {
  double csm_window_linear, csm_window_angular, csm_linear_step, csm_cost_threshold, csm_max_px_accurate_range;
  int32_t csm_n_iter, reserved;
} randt_bnb_params;
/* ceres::Problem::Evaluate with the loss applied (ndt_matcher.cpp:561-576) at n_poses poses for one
 * frozen correspondence set d_corr (cell_capacity(moving) x k): d_cost[p] = sum 1/2 rho(s),
 * rho = BarronLoss(scale, loss_alpha) (mu = 1, unscaled, :517).  d_n_res (nullable): residual count. */
int randt_eval_cost_batch_dev(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_idx,
                              const int32_t* d_corr, const randt_matcher_params* mp, double scale, const double* d_poses4,
                              int n_poses, double* d_cost, int32_t* d_n_res);
/* Matcher::estimateTransformGlobalBNB (ndt_matcher.cpp:495-608): association with 4 neighbours at the
 * guess, then the breadth-first coarse-to-fine pose grid; every level is evaluated in one launch.
 * h_trans4 in: guess, out: best pose (identity if no pose is below csm_cost_threshold, like the
 * reference).  *min_cost_out: the reference's return value; *n_evals (nullable): poses evaluated. */
int randt_search_global(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_idx,
                        const randt_matcher_params* mp, const randt_bnb_params* bp, double scale,
                        double search_window_size_linear, double search_window_size_angular, double h_trans4[4],
                        double* min_cost_out, int* n_evals);

/* ------------------------------------------------------------------ CS divergence (f-2) ------ */
/* Map::calculateCSDivergence (src/ndt_representation/ndt_map.cpp:42-99) for a batch of pairs: pair p =
 * fixed map d_fixed_idx[p] (must lie in [fixed_first, fixed_first + fixed_count)) vs moving map
 * (moving_first + p), the moving map first transformed by d_pose4[p] like local_fuser.cpp:338
 * (d_pose4 may be NULL = already transformed).  The fixed maps' self terms are computed once per map.
 * d_out[p] = -log(interaction) + 0.5 log(fixed term) + 0.5 log(moving term); d_terms (nullable):
 * the three sums per pair.  The moving maps' CAPACITY is bounded by the LDS that holds a transformed moving map
 * (40 B per cell: <= ~3700 cells; scan maps have a few hundred): beyond it RANDT_ERR_UNSUPPORTED. */
int randt_cs_divergence_batch_dev(randt_ctx* ctx, const randt_maps* fixed, int fixed_first, int fixed_count,
                                  const int32_t* d_fixed_idx, const randt_maps* moving, int moving_first, int n_pairs,
                                  const double* d_pose4, double* d_out, double* d_terms);

/* ------------------------------------------------------------------ Scan Context (f-4) -------- */
/* ScanContextParameters (src/ndt_slam/ndt_slam.cpp:515-552; config/parameters_*.yaml "scan_context"). */
typedef struct randt_sc_params {
  int32_t num_ring, num_sector;        /* PC_NUM_RING (<= 64), PC_NUM_SECTOR (<= 128) */
  double max_radius;                   /* PC_MAX_RADIUS */
  int32_t num_exclude_recent, num_candidates;  /* NUM_EXCLUDE_RECENT, NUM_CANDIDATES_FROM_TREE (<= 32) */
  double search_ratio, dist_thresh, assumed_drift, odom_eps, odom_weight, intensity_factor;
} randt_sc_params;
/* SCManager::makeScancontext + makeRingkeyFromScancontext + makeSectorkeyFromScancontext
 * (src/local_fuser/Scancontext/Scancontext.cpp:156-237) for a batch of keyframe scans (same point layout as
 * randt_ndt_build_batch_dev).  d_desc: [n_scans][num_sector][num_ring] doubles (one sector = one contiguous column, like
 * the reference's column-major MatrixXd), d_ring_keys [n_scans][num_ring], d_sector_keys [n_scans][num_sector]. */
int randt_sc_make_batch_dev(randt_ctx* ctx, const float* d_points, int n_scans, int points_pitch, const int32_t* d_n_points,
                            int stride_floats, int intensity_index, const randt_sc_params* p, double* d_desc, double* d_ring_keys,
                            double* d_sector_keys);
/* SCManager::detectLoopClosureID (Scancontext.cpp:261-341) for a batch of query nodes over a database of n_db nodes
 * (descriptors / ring keys as produced above, odometry positions [n_db][2], traversed distances [n_db]).  Query q is node
 * d_query_ids[q] (NULL: q) and searches nodes [0, node - num_exclude_recent].  d_loop_id[q] = matched node or -1,
 * d_yaw[q] = relative yaw [rad], d_min_dist (nullable) = best combined distance.  The reference's KD-tree (rebuilt only
 * every tree_making_period queries) is replaced by an exact search over the current database. */
int randt_sc_detect_batch_dev(randt_ctx* ctx, const randt_sc_params* p, const double* d_desc, const double* d_ring_keys,
                              const double* d_pos, const double* d_dist, int n_db, const int32_t* d_query_ids, int n_queries,
                              int32_t* d_loop_id, float* d_yaw, double* d_min_dist);

/* Device-resident Scan Context database = the state SCManager keeps (polarcontexts_, polarcontext_invkeys_mat_,
 * odom_positions_, distances_; include/local_fuser/Scancontext.h:88-99).  Nodes are appended in keyframe order. */
typedef struct randt_sc_db randt_sc_db;
int randt_sc_db_create(randt_ctx* ctx, const randt_sc_params* p, int initial_capacity, randt_sc_db** out);
void randt_sc_db_destroy(randt_sc_db* db);
int randt_sc_db_size(const randt_sc_db* db);
/* SCManager::makeAndSaveScancontextAndKeys (Scancontext.cpp:240-258): descriptor + keys of one HOST scan, stored as node
 * randt_sc_db_size() together with its odometry position and traversed distance.  Returns the status; *node_id (nullable)
 * receives the new node's index. */
int randt_sc_db_append(randt_sc_db* db, const float* h_points, int n_points, int stride_floats, int intensity_index,
                       const double odom_position[2], double traversed_distance, int* node_id);
/* SCManager::detectLoopClosureID (Scancontext.cpp:261-341) for one node of the database: *loop_id = match or -1,
 * *yaw_diff_rad as in the reference's return value, *min_dist (nullable). */
int randt_sc_db_detect(randt_sc_db* db, int node_id, int* loop_id, float* yaw_diff_rad, double* min_dist);
/* host copies of one node (debug / tests): h_desc [num_sector][num_ring], h_ring_key [num_ring], h_sector_key [num_sector]
 * (each nullable) */
int randt_sc_db_download(const randt_sc_db* db, int node_id, double* h_desc, double* h_ring_key, double* h_sector_key);

/* One pair, host result (Map::calculateCSDivergence as LocalFuser::detectLoopClosures calls it, local_fuser.cpp:338-339):
 * h_pose4 (nullable) transforms the moving map first; *out = the divergence; h_terms (nullable) = the three sums. */
int randt_cs_divergence(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_idx,
                        const double* h_pose4, double* out, double* h_terms);

/* ------------------------------------------------------------------ scan filter (f-1) -------- */
/* RadarPreprocessorParameters used by filterScan + initial_transform_radar_baselink_ as a row-major
 * 3x4 matrix (radar_preprocessor.cpp:7-28,124). */
typedef struct randt_filter_params {
  float min_range, max_range, min_intensity, beam_distance_increment_threshold;
  float sensor_to_base[12];
} randt_filter_params;
/* RadarPreprocessor::filterScan (src/radar_preprocessing/radar_preprocessor.cpp:45-125) for a batch of
 * polar-organised raw scans: d_raw = n_scans x n_azimuths x n_bins points of stride_floats floats
 * (x, y, z, .., intensity at intensity_index), azimuth after azimuth, range ascending -- the layout
 * the reference assumes (:61).  Output per scan: up to pitch_out filtered points as packed x y z I
 * in the base frame (feed them to randt_ndt_build_batch_dev with d_n_points = d_out_counts),
 * optional polar (angle, range) pairs, optional per-azimuth peak detections (angle, range,
 * intensity = max_detections).  d_status[s]: 0 ok, 1 input not azimuth-organised, 2 output overflow. */
int randt_filter_scan_batch_dev(randt_ctx* ctx, const float* d_raw, int n_scans, int n_azimuths, int n_bins,
                                int stride_floats, int intensity_index, const randt_filter_params* fp,
                                float* d_out_points, int pitch_out, int32_t* d_out_counts, float* d_out_polar,
                                float* d_peaks, int32_t* d_peak_counts, int32_t* d_status);
/* Host conveniences for ONE raw scan in host memory -- what RadarPreprocessor::processScan is handed (a sensor message,
 * radar_preprocessor.cpp:30-43).  The raw scan is uploaded (19.2 MB for 400 x 3000 bins: the PCIe copy is most of the call;
 * it is waited for, the host buffer is free on return), device buffers come from the context's storage pool.
 *  - randt_filter_scan: filterScan alone, results back on the host (synchronous): up to `capacity` filtered points as packed
 *    x y z I (*n_out = how many were written; *status = 2 if the filter kept more than `capacity`), optional polar pairs, optional
 *    per-azimuth peaks (n_azimuths x 3, *n_peaks of them).  *status as d_status above.
 *  - randt_filter_build: filterScan + clustering + NDT of the kept points into map `map_idx` of `out`, everything on the
 *    device, nothing read back unless `status` is given (then the call waits and reports the filter's status; NULL: asynchronous
 *    after the upload).  max_points: capacity of the intermediate point buffer (<= 7168 keeps the one-workgroup build). */
int randt_filter_scan(randt_ctx* ctx, const float* h_raw, int n_azimuths, int n_bins, int stride_floats, int intensity_index,
                      const randt_filter_params* fp, float* h_out_points, int capacity, int* n_out, float* h_out_polar,
                      float* h_peaks, int* n_peaks, int* status);
int randt_filter_build(randt_ctx* ctx, const float* h_raw, int n_azimuths, int n_bins, int stride_floats, int intensity_index,
                       const randt_filter_params* fp, const randt_cluster_params* cp, int max_points, randt_maps* out, int map_idx,
                       int* status);


/* ------------------------------------------------------------------ fixed-lag window (a16, a17) */
/* Matcher::predictTransform, optimize_on_manifold branch (ndt_matcher.cpp:22-59) with predictSE2
 * (ceres_residuals.h:62-83): constant-velocity prediction of the next state.  Host-side O(1) math. */
int randt_predict_state(const randt_state* last, double stamp, randt_state* next);
/* The same for either state representation: RANDT_PARAM_MANIFOLD = predictSE2 (what randt_predict_state does),
 * RANDT_PARAM_VECTOR = the (pos[2], rot) form `predict` (ceres_residuals.h:25-55, 91-123) that Matcher::predictTransform
 * takes when optimize_on_manifold is false (ndt_matcher.cpp:27-41): mid-point heading, NormalizeAngle'd rotation,
 * pose = Sophus::SE2d(rot, pos). */
int randt_predict_state_param(const randt_state* last, double stamp, int parameterization, randt_state* next);
/* The same for n independent states (the replicas of randt_register_window_batch): next[i] = prediction of last[i] to `stamp`. */
int randt_predict_state_batch(const randt_state* last, int n, double stamp, int parameterization, randt_state* next);
/* Matcher::estimateTransformCeres (ndt_matcher.cpp:322-424): fixed-lag smoother over n_states = S+1
 * states (oldest first; its pose is held constant), S <= 12, n_fixed <= 2 (the shipped lag smoothing_steps: 3 runs the kernel
 * tuned for it, window.hip; lags 4..7 -- ndt_matcher.cpp:343 takes any -- the general kernel, window_gen.hip; 8..12 the same
 * source compiled for the longer band, window_gen_big.hip; beyond: RANDT_ERR_UNSUPPORTED).  Per state j = 1..S: MotionModelFactorSE2 to
 * its predecessor (ceres_residuals.h:621-679), optional RotationalResidualSE2 (:338-370, h_imu[j-1]),
 * and NDT factors of moving map moving_idx[j-1] against every fixed map (association at the state's
 * own pose), robustified by Scaled(Barron) with weight ndt_weight / (n_cells * k); GNC loop as in the
 * pair registration; whole LM loop on the device.  h_states in/out (both pose representations are
 * synchronised on return, cf. local_fuser.cpp:141-150); h_trans4 in: prior pose for the rejection
 * gate (ndt_matcher.cpp:339-340,411-422), out: newest pose.  *rejected = 1 if the gate fired.
 * mp->parameterization (RANDT_PARAM_ANALYTIC = VECTOR with the analytic NDT functor): RANDT_PARAM_MANIFOLD (optimize_on_manifold: true, the shipped configuration: SE(2) pose blocks with
 * Sophus' manifold, MotionModelFactorSE2 / RotationalResidualSE2) or RANDT_PARAM_VECTOR (optimize_on_manifold: false: parameter
 * blocks pos[2] and rot[1] with plain addition, MotionModelFactor :554-619 / RotationalResidual :307-336 /
 * NDTFrameToMap{,Intensity}FactorResidual :421-451,486-518; pos / rot of the states are the variables, the association still
 * starts from the states' `pose` members like the reference's, ndt_matcher.cpp:364).
 * Limit: mp->n_neighbours <= 16 (RANDT_ERR_UNSUPPORTED beyond), like the pair registration. */
int randt_register_window(randt_ctx* ctx, const randt_maps* fixed, const int32_t* h_fixed_idx, int n_fixed,
                          const randt_maps* moving, const int32_t* h_moving_idx, randt_state* h_states, int n_states,
                          const double* h_imu, const randt_matcher_params* mp, const randt_window_params* wp,
                          double h_trans4[4], int* rejected, randt_result* h_result);
/* NOT in the reference (its node runs one odometry): n_windows INDEPENDENT fixed-lag windows of one shape -- the same number of
 * states and of fixed maps, e.g. R replicas of the sequential path ("replicas only") advancing in lock-step, or one
 * window re-solved from n_windows priors -- in ONE association launch and ONE solve launch (a workgroup per window: the single
 * entry uses 1 of the chip's 256 compute units), one pinned image per direction, one synchronisation.  Window w is exactly what
 * randt_register_window computes for its slice of the window-major arrays (bit-identical): h_fixed_idx[w][n_fixed],
 * h_moving_idx[w][n_states - 1], h_states[w][n_states] (in / out), h_imu[w][n_states - 1] (nullable), h_trans4[w][4] (in: prior,
 * out: newest pose), rejected[w], h_results[w] (both nullable).  All windows read their maps from the two batches `fixed` / `moving`. */
int randt_register_window_batch(randt_ctx* ctx, int n_windows, const randt_maps* fixed, const int32_t* h_fixed_idx, int n_fixed,
                                const randt_maps* moving, const int32_t* h_moving_idx, randt_state* h_states, int n_states,
                                const double* h_imu, const randt_matcher_params* mp, const randt_window_params* wp,
                                double* h_trans4, int* rejected, randt_result* h_results);

/* ------------------------------------------------------------------ pose graph (f-4) */
/* GlobalFuser::optimizePoseGraph (src/global_fuser/global_fuser.cpp:13-105): 2-D pose graph with
 * PoseGraph2dErrorTerm residuals (include/global_fuser/pose_graph_2d_error_term.h:33-80), optional
 * ceres::HuberLoss(loss_function_scale) (:17-23), first pose constant (:48-49), Ceres 2.1.0 trust-region LM.
 * The reference's SPARSE_NORMAL_CHOLESKY step (:54-57) is computed exactly on the device: block-tridiagonal
 * Cholesky along the odometry chain, one lane per right-hand side, dense Schur complement on the poses that
 * loop closures touch (see csrc/posegraph.hip).  The defaults are Ceres' Solver::Options defaults plus
 * max_num_iterations = 200000 (:52). */
#define RANDT_PG_MAX_SEPARATORS 16384  /* the dense Schur complement takes 2 x (3 n)^2 x 8 B of device memory: 39 GB at the limit */
typedef struct randt_pg_params {
  int32_t use_robust_loss;  /* GlobalFuserParameters::use_robust_loss */
  int32_t max_iterations;
  int32_t max_consecutive_invalid_steps;
  int32_t reserved;
  double loss_scale;        /* GlobalFuserParameters::loss_function_scale */
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius, max_radius, min_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
} randt_pg_params;
typedef struct randt_pg_result {
  double initial_cost, final_cost;
  int32_t iterations;         /* Summary::iterations.size() */
  int32_t termination;        /* RANDT_TERM_* */
  int32_t n_residual_blocks;  /* edges that pass the max_update_index rule (:32) */
  int32_t n_loop_closures;    /* edges.size() + 1 - poses.size(), the count the reference prints (:26) */
  int32_t n_separator_poses;  /* poses eliminated through the dense Schur complement */
  int32_t reserved;
} randt_pg_result;
void randt_pg_params_default(randt_pg_params* p);
/* h_poses [n_poses][3] = (pos.x, pos.y, rot) in/out (pose i = key i of the reference's std::map<int, Pose>);
 * edge e: h_id_begin/h_id_end, h_meas [E][3] = (trans.translation(), trans.log()(2)), h_sqrt_info [E][9]
 * row-major.  An edge is used iff id_begin + 1 == id_end || id_end <= max_update_index.  Host buffers: the graph
 * lives on the host in the reference and is a few hundred KB.  RANDT_ERR_UNSUPPORTED if more than
 * RANDT_PG_MAX_SEPARATORS poses carry loop closures (2048 until round 4); a device allocation that fails is reported
 * as the HIP error it is. */
int randt_pose_graph_optimize(randt_ctx* ctx, int n_poses, double* h_poses, int n_edges, const int32_t* h_id_begin,
                              const int32_t* h_id_end, const double* h_meas, const double* h_sqrt_info,
                              int max_update_index, const randt_pg_params* p, randt_pg_result* out);

/* ------------------------------------------------------------------ multi-GPU group (8e) ----- */
/* Independent scan-to-submap registrations -- the loop-closure candidates LocalFuser::detectLoopClosures hands to
 * Matcher::estimateLoopConstraint one by one (src/local_fuser/local_fuser.cpp:329-339, 370-397), a submap batch, a BNB
 * pose grid -- share nothing but the read-only fixed maps, so a batch splits contiguously over the GPUs of a node with
 * NO data-path collective.  A group owns one context (and stream) per member GPU and the only two exchanges the path
 * has: a broadcast of map tables (cells + counts + index grid, 520 KB per indoor submap) from the member that built
 * them, once per submap epoch, and a gather of the 32-byte poses / 64-byte result records.  Transport: RCCL over xGMI
 * (ncclBroadcast; librccl is opened at run time, it is not a link-time dependency of this library) or -- one process
 * only -- hipMemcpyPeerAsync fan-out, which also serves "virtual ranks" that share one device (tests, a 1-GPU box).
 *
 * Two ways to form a group:
 *   - ONE process drives n devices (the reference's single-process node): randt_group_create(devices, n, ...);
 *     members are ranks 0..n-1, all local.
 *   - one process per GPU (torch.distributed / MPI launchers): rank 0 calls randt_group_unique_id, the launcher ships the
 *     128 bytes to every rank, every rank calls randt_group_create_rank; each process then holds ONE local member.
 * Per-member arguments below are arrays with one entry per LOCAL member (n_local; 1 in the one-process-per-GPU mode),
 * e.g. `randt_maps* const* maps`: maps[i] lives on member i's device, created on randt_group_ctx(g, i).
 * Member r of a world of G owns items [lo, hi) = randt_shard_range(n, G, r): contiguous, remainders to the low ranks. */
typedef struct randt_group randt_group;
enum { RANDT_TRANSPORT_AUTO = 0, RANDT_TRANSPORT_PEER = 1, RANDT_TRANSPORT_RCCL = 2 };
#define RANDT_UNIQUE_ID_BYTES 128
void randt_shard_range(int n_items, int world, int rank, int* lo, int* hi);
/* devices[i] = HIP device of member i (repeats allowed with the PEER transport); streams (nullable, or entries NULL):
 * a hipStream_t per member to enqueue on, otherwise the group creates its own non-blocking streams.
 * AUTO = RCCL when n > 1, all devices are distinct and librccl can be opened, PEER otherwise. */
int randt_group_create(const int* devices, int n, void* const* streams, int transport, randt_group** out);
int randt_group_unique_id(void* out128);
int randt_group_create_rank(int device, void* stream, int rank, int world, const void* unique_id128, randt_group** out);
int randt_group_destroy(randt_group* g);
int randt_group_info(const randt_group* g, int* world, int* n_local, int* first_rank, int* transport);
randt_ctx* randt_group_ctx(randt_group* g, int local_member);
/* Text of the group's last failure.  g == NULL: why THIS THREAD's last randt_group_create / randt_group_create_rank failed
 * (the RCCL / HIP error text -- the group object no longer exists then); "" if it succeeded. */
const char* randt_group_last_error(const randt_group* g);
int randt_group_synchronize(randt_group* g);
/* Cells, counts and index grids of maps [first, first + count) of every member's batch := those of rank `root`'s batch
 * (all batches must have the same geometry).  Asynchronous on the members' streams. */
int randt_group_broadcast_maps(randt_group* g, randt_maps* const* maps, int first, int count, int root);
/* Rows [lo_r, hi_r) = randt_shard_range(n_rows, world, r) of d_rows[member r] -> the same rows of every member's buffer
 * (each buffer holds n_rows rows of row_bytes bytes): the gather of per-registration outputs.  Asynchronous. */
int randt_group_allgather_rows(randt_group* g, void* const* d_rows, int n_rows, size_t row_bytes);
/* Matcher::estimateLoopConstraint (ndt_matcher.cpp:426-493) for a batch SHARDED over the group: member r runs pairs
 * [lo_r, hi_r) of the batch exactly like randt_register_batch_dev / randt_scan_register_batch_dev would (same kernels,
 * bit-identical results), on its own device and stream.  Every member's arrays describe the WHOLE batch (n_pairs /
 * n_scans entries; a member reads only its own rows: moving maps [lo_r, hi_r) of moving[r], d_fixed_idx[r][lo_r..],
 * d_pose4[r][4 lo_r ..]); scan_maps[r] is a workspace of >= hi_r - lo_r maps.  gather != 0: poses and results are
 * all-gathered afterwards, so every member holds the full batch's outputs.  Asynchronous. */
int randt_group_register_batch_dev(randt_group* g, randt_maps* const* fixed, const int32_t* const* d_fixed_idx,
                                   randt_maps* const* moving, int n_pairs, const randt_matcher_params* mp,
                                   double* const* d_pose4, randt_result* const* d_results, int gather);
int randt_group_scan_register_batch_dev(randt_group* g, const float* const* d_points, int n_scans, int pitch_points,
                                        const int32_t* const* d_n_points /* nullable */, int stride_floats, int intensity_index,
                                        const randt_cluster_params* cp, randt_maps* const* fixed,
                                        const int32_t* const* d_fixed_idx, randt_maps* const* scan_maps,
                                        const randt_matcher_params* mp, double* const* d_pose4,
                                        randt_result* const* d_results, int gather);
/* Host convenience (synchronous) for a caller that holds poses on the host, like LocalFuser does: the same sharded
 * registration of pre-built moving maps; h_fixed_idx[p] selects pair p's fixed map, h_pose4 in: guesses, out: refined
 * poses, h_results (nullable) out -- identical on every rank on return. */
int randt_group_register_pairs(randt_group* g, randt_maps* const* fixed, const int32_t* h_fixed_idx, randt_maps* const* moving,
                               int n_pairs, const randt_matcher_params* mp, double* h_pose4, randt_result* h_results);

#ifdef __cplusplus
}
#endif
#endif /* RANDT_H */
