// randt_facade.hpp -- header-only C++17 mirror of the reference's public classes on top of the C ABI.
//
// The reference's seam is the C++ API LocalFuser calls:
//   rc::navigation::ndt::Cell     include/ndt_representation/ndt_cell.h:16-170
//   rc::navigation::ndt::Map      include/ndt_representation/ndt_map.h:14-199
//   rc::navigation::ndt::Matcher  include/ndt_registration/ndt_matcher.h:46-87
//   SCManager                     include/local_fuser/Scancontext.h:50-103 (loop-closure candidates)
//   GlobalFuser / Pose / Constraint  include/global_fuser/global_fuser.h:32-60, include/ndt_slam/
//                                 trajectory_representation.h:25-52 (pose-graph back end)
// This header keeps their names, argument meaning and error behaviour (void/double returns, a
// warning on std::cout, "keep the previous pose" on failure) but is free of Eigen / Sophus / PCL /
// Ceres: vectors are std::array, poses are the 4 doubles of Sophus::SE2d::data().  A ROS node built
// against the reference headers is re-pointed with the adapter shown in INTEGRATION.md.
//
// Everything numeric happens on the GPU inside librandt_hip.so; there is no CPU fallback.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <iostream>
#include <limits>
#include <map>
#include <mutex>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "randt.h"

namespace randt {

using Vector2f = std::array<float, 2>;
using Vector3f = std::array<float, 3>;
using Matrix2f = std::array<float, 4>;  // row-major
using Matrix3f = std::array<float, 9>;  // row-major

// Sophus::SE2d stand-in: data() = [cos, sin, tx, ty] (trajectory_representation.h:14).
struct SE2d {
  double d[4] = {1.0, 0.0, 0.0, 0.0};
  SE2d() = default;
  SE2d(double theta, double tx, double ty) : d{std::cos(theta), std::sin(theta), tx, ty} {}
  double* data() { return d; }
  const double* data() const { return d; }
  double angle() const { return std::atan2(d[1], d[0]); }  // so2().log()
  std::array<double, 2> translation() const { return {d[2], d[3]}; }
  SE2d operator*(const SE2d& o) const {
    SE2d r;
    r.d[0] = d[0] * o.d[0] - d[1] * o.d[1];
    r.d[1] = d[0] * o.d[1] + d[1] * o.d[0];
    const double n = std::sqrt(r.d[0] * r.d[0] + r.d[1] * r.d[1]);
    r.d[0] /= n;
    r.d[1] /= n;
    r.d[2] = d[2] + d[0] * o.d[2] - d[1] * o.d[3];
    r.d[3] = d[3] + d[1] * o.d[2] + d[0] * o.d[3];
    return r;
  }
  SE2d inverse() const {
    SE2d r;
    r.d[0] = d[0];
    r.d[1] = -d[1];
    r.d[2] = -(d[0] * d[2] + d[1] * d[3]);
    r.d[3] = -(-d[1] * d[2] + d[0] * d[3]);
    return r;
  }
};

// NDTCellParameters / NDTMapParameters / RadarPreprocessorParameters / NDTMatcherParameters
// (include/ndt_slam/ndt_slam_parameters.h:11-84), fields used on the path only.
struct NDTMapParameters {
  int size_x = 100, size_y = 100;  // cells (ndt_slam.cpp:653-654)
  double resolution = 0.5;
  double max_neighbour_manhattan_distance = 4.0;
  int min_points_per_cell = 5;
};
struct RadarPreprocessorParameters {
  int n_clusters = 2304;  // (2*max_range/resolution)^2, ndt_slam.cpp:691
  double max_range = 12.0;
};
// rc::navigation::ndt::NDTMatcherParameters (include/ndt_slam/ndt_slam_parameters.h:52-84), every member under its
// reference name; motion_sqrtI is the 8 x 8 matrix row-major.  Defaults = config/parameters_indoor.yaml:24-39 +
// config/ndt_radar_slam_base_parameters.yaml:21-48 (the reference struct itself has no defaults: readParameters fills it).
struct NDTMatcherParameters {
  std::array<double, 64> motion_sqrtI = [] {
    std::array<double, 64> m{};
    const double d[8] = {1, 1, 1, 1, 3, 0.1, 20, 60};  // base yaml :36-43
    for (int i = 0; i < 8; ++i) m[i * 8 + i] = d[i];
    return m;
  }();
  double covariance_scaling_factor = 25.0;
  double weight_kinematics = 1.0;  // unused by the reference as well
  double weight_imu = 64.0, weight_imu_bias = 6.0e5, initial_imu_bias = 0.0;
  int gnc_steps = 3;
  int smoothing_steps = 3;
  double loss_function_convexity = -2.0, loss_function_scale = 1.5, gnc_control_parameter_divisor = 1.3;
  int max_iteration = 200;
  double pose_reject_translation = 2.0, pose_reject_rotation = 2.0;
  int n_results_kd_lookup = 4;
  double ndt_weight = 5.0e4;
  bool use_analytic_expressions_for_optimization = false;
  bool use_intensity_as_dimension = true, use_constant_velocity_model = true, optimize_on_manifold = true, lookup_mahalanobis = true;
  bool use_imu = false;  // indoor preset: true; the IMU increments then come through predictTransform's initial_angle_guess
  double csm_window_linear = 4.5, csm_window_angular = 0.45, csm_linear_step = 0.4, csm_cost_threshold = 0.82;
  double csm_max_px_accurate_range = 4.0;
  bool csm_ignore_overlap = false;
  int csm_n_iter = 2;
};

// Error policy of the facade.  The reference never throws and has no status codes: void / double returns, a warning on
// std::cout, "keep the previous value" on failure.  kKeepPrevious (default) reproduces that: a failing ABI
// call prints a warning, leaves the object as it was and records the status in last_status(); kThrow raises
// std::runtime_error instead (what the tests of this repository use to see failures).
enum class ErrorPolicy { kKeepPrevious, kThrow };
inline ErrorPolicy& error_policy() {
  static ErrorPolicy p = ErrorPolicy::kKeepPrevious;
  return p;
}
inline int& last_status() {
  static thread_local int s = RANDT_OK;
  return s;
}
// The first failure since clear_errors() (last_status() is overwritten by the next successful call): a caller that cannot
// check after every call -- the reference's call sites check nothing -- reads this once per scan.
inline int& first_error() {
  static thread_local int s = RANDT_OK;
  return s;
}
inline void clear_errors() { last_status() = first_error() = RANDT_OK; }
// what a failed call returns where the reference returns a scalar (a cost, a divergence): never a plausible value
inline double failed_value() { return std::numeric_limits<double>::quiet_NaN(); }
// returns true if the call succeeded
inline bool facade_check(int rc, const char* what, randt_ctx* ctx) {
  last_status() = rc;
  if (rc == RANDT_OK) return true;
  if (first_error() == RANDT_OK) first_error() = rc;
  const std::string msg = std::string(what) + ": " + randt_status_string(rc) + " (" + (ctx ? randt_last_error(ctx) : "") + ")";
  if (error_policy() == ErrorPolicy::kThrow) throw std::runtime_error(msg);
  std::cout << "WARNING: " << msg << " -- previous value kept\n";
  return false;
}

class Context {
 public:
  // solve_mode: RANDT_SOLVE_AUTO (default) lets the library give a lone batch of up to two registrations per CU several
  // wavefronts per registration, which assumes the device is otherwise idle; a caller that keeps several batches in flight
  // (several contexts / streams on one GPU) says RANDT_SOLVE_THROUGHPUT here, or loses ~2x throughput
  explicit Context(int device = 0, void* stream = nullptr, int solve_mode = RANDT_SOLVE_AUTO) {
    // without a device every later call on this context fails with a status (there is no CPU fallback)
    if (facade_check(randt_ctx_create(device, stream, &ctx_), "randt_ctx_create", nullptr) && solve_mode != RANDT_SOLVE_AUTO)
      facade_check(randt_ctx_set_solve_mode(ctx_, solve_mode), "randt_ctx_set_solve_mode", ctx_);
  }
  void setSolveMode(int solve_mode) { facade_check(randt_ctx_set_solve_mode(ctx_, solve_mode), "randt_ctx_set_solve_mode", ctx_); }
  ~Context() { randt_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  randt_ctx* get() const { return ctx_; }

 private:
  randt_ctx* ctx_ = nullptr;
};

// Host copy of one cell = rc::navigation::ndt::Cell (ndt_cell.h:16-170): getters on the record, mutators through the
// ABI (the arithmetic runs on the device, randt_cell_add_points / randt_cells_{merge,transform,mahalanobis}).
// A cell without a context can be read but not modified.
// COST: every mutator / metric below is ONE device round trip + stream synchronisation (~15-20 us) for ONE cell -- right for
// the reference's occasional single-cell uses, wrong for loops over a map's cells (NDTSlam::createVisualizationMsg style).
// The batched forms of the same operations, one launch for any number of cells:
//   transformCell x n          randt_cells_transform(ctx, cells, n, pose4)      or the whole map: Map::transformMap / randt_maps_transform
//   operator+= x n             randt_cells_merge(ctx, dst, src, n)              or cell-by-grid-slot: Map::mergeMap / randt_maps_merge
//   mahalanobisSquared x n     randt_cells_mahalanobis(ctx, a, b, n, out)
//   addPointCloud per cell     a whole scan at once: Map::addClusters / randt_ndt_build_batch_dev (the hot path)
//   transformCellWithPointCloud  randt_points_transform(ctx, points, n, stride, pose4) on any point set
class Cell {
 public:
  Cell() = default;
  explicit Cell(const randt_cell& c) : c_(c) {}
  Cell(const randt_cell& c, std::shared_ptr<Context> ctx, int min_points_per_cell = 0)
      : c_(c), ctx_(std::move(ctx)), min_points_(min_points_per_cell) {}

  // void initialize(const int& min_points_per_cell, const NDTCellParameters& params) (ndt_cell.cpp:7-11); use_pndt is
  // false in every shipped configuration and not built
  void initialize(std::shared_ptr<Context> ctx, const int& min_points_per_cell) {
    ctx_ = std::move(ctx);
    min_points_ = min_points_per_cell;
    c_ = randt_cell{};
    pending_.clear();
    pending_polar_.clear();
    points_.clear();
    polar_points_.clear();
  }
  // void addPoint(const pcl::PointXYZI& point, const std::pair<double, double>& angle_dist) (ndt_cell.cpp:19-23): queued
  // until updateCell, like points_to_add_
  void addPoint(float x, float y, float intensity, const std::pair<double, double>& angle_dist = {0.0, 0.0}) {
    pending_.insert(pending_.end(), {x, y, 0.0f, intensity});
    pending_polar_.push_back(angle_dist);
  }
  // bool addPointCloud(const pcl::PointCloud<pcl::PointXYZI>&, ...) (ndt_cell.cpp:25-34): points = n x stride floats
  bool addPointCloud(const float* points, int n, int stride, int intensity_index, const std::pair<double, double>* angle_dists = nullptr) {
    if (static_cast<long long>(c_.n) + static_cast<long long>(pending_.size() / 4) + n <= static_cast<long long>(min_points_)) return false;
    for (int i = 0; i < n; ++i)
      addPoint(points[i * stride], points[i * stride + 1], points[i * stride + intensity_index], angle_dists ? angle_dists[i] : std::pair<double, double>{0.0, 0.0});
    updateCell();  // "use recursive update equation"
    return true;
  }
  // void updateCell(void) (ndt_cell.cpp:36-114)
  void updateCell() {
    if (pending_.empty() || !ctx_) return;
    randt_cell next = c_;
    int accepted = 0;
    if (facade_check(randt_cell_add_points(ctx_->get(), &next, pending_.data(), static_cast<int>(pending_.size() / 4), 4, 3,
                                           min_points_, &accepted), "randt_cell_add_points", ctx_->get()) && accepted) {
      c_ = next;
      // points_ += points_to_add_; polar_points_.insert(...) (ndt_cell.cpp:95-98): the cell keeps the points it was made of
      points_.insert(points_.end(), pending_.begin(), pending_.end());
      polar_points_.insert(polar_points_.end(), pending_polar_.begin(), pending_polar_.end());
      pending_.clear();  // points_to_add_.clear(); below the gate the points stay queued, like in the reference
      pending_polar_.clear();
    }
  }
  // void clearCell(void)
  void clearCell() {
    c_ = randt_cell{};
    pending_.clear();
    pending_polar_.clear();
    points_.clear();
    polar_points_.clear();
  }
  // void transformCell(const Eigen::Affine2f& trans) (ndt_cell.cpp:117-123)
  void transformCell(const SE2d& trans) {
    if (!ctx_) return;
    randt_cell next = c_;
    if (facade_check(randt_cells_transform(ctx_->get(), &next, 1, trans.data()), "randt_cells_transform", ctx_->get())) c_ = next;
  }
  // void transformCellWithPointCloud(const Eigen::Affine2f& trans) (ndt_cell.cpp:126-136): the statistics as in transformCell,
  // the generating points through pcl::transformPointCloud's arithmetic on the device (randt_points_transform)
  void transformCellWithPointCloud(const SE2d& trans) {
    if (!ctx_) return;
    randt_cell next = c_;
    std::vector<float> moved = points_;
    if (!facade_check(randt_cells_transform(ctx_->get(), &next, 1, trans.data()), "randt_cells_transform", ctx_->get())) return;
    if (!moved.empty() &&
        !facade_check(randt_points_transform(ctx_->get(), moved.data(), static_cast<int>(moved.size() / 4), 4, trans.data()), "randt_points_transform", ctx_->get()))
      return;
    c_ = next;
    points_.swap(moved);
  }
  // const pcl::PointCloud<pcl::PointXYZI>& getPointCloud() const (ndt_cell.h:146-148): the points that contributed, packed
  // x y z I (z = 0); cells handed out by Map::getCells() carry none (device maps keep statistics only; the reference's only
  // reader is the dead Map::mergeMapPoints, ndt_map.cpp:209-236)
  const std::vector<float>& getPointCloud() const { return points_; }
  // const std::vector<std::pair<double, double>> getAngleDists() const (ndt_cell.h:152-154)
  std::vector<std::pair<double, double>> getAngleDists() const { return polar_points_; }
  // Cell& operator+=(const Cell& m_cell) (ndt_cell.h:133-142)
  Cell& operator+=(const Cell& m_cell) {
    if (!ctx_) return *this;
    randt_cell next = c_;
    if (facade_check(randt_cells_merge(ctx_->get(), &next, &m_cell.c_, 1), "randt_cells_merge", ctx_->get())) c_ = next;
    return *this;
  }
  // double mahalanobisSquared(const Cell& subtrahend) const / mahalanobisSquaredIntensity (ndt_cell.cpp:158-169)
  double mahalanobisSquared(const Cell& subtrahend) const { return mahalanobis(subtrahend, 0); }
  double mahalanobisSquaredIntensity(const Cell& subtrahend) const { return mahalanobis(subtrahend, 1); }

  Vector2f getMean() const { return {c_.mean[0], c_.mean[1]}; }
  void getMean(Vector2f& mean) const { mean = getMean(); }
  Vector3f getIntensityMean() const { return {c_.mean[0], c_.mean[1], c_.mean[2]}; }
  void getIntensityMean(Vector3f& mean) const { mean = getIntensityMean(); }
  Matrix2f getCov() const { return {c_.cov[0], c_.cov[1], c_.cov[1], c_.cov[3]}; }
  void getCov(Matrix2f& cov) const { cov = getCov(); }
  Matrix3f getIntensityCov() const {
    return {c_.cov[0], c_.cov[1], c_.cov[2], c_.cov[1], c_.cov[3], c_.cov[4], c_.cov[2], c_.cov[4], c_.cov[5]};
  }
  void getIntensityCov(Matrix3f& cov) const { cov = getIntensityCov(); }
  void getMeanAndCov(Vector2f& mean, Matrix2f& cov) const {
    mean = getMean();
    cov = getCov();
  }
  void getIntensityMeanAndCov(Vector3f& mean, Matrix3f& cov) const {
    mean = getIntensityMean();
    cov = getIntensityCov();
  }
  double getMeanIntensity() const { return c_.mean[2]; }
  double getMaxIntensity() const { return c_.max_intensity; }
  size_t getNumCells() const { return c_.n; }  // sic: number of POINTS (ndt_cell.cpp:178-180)
  const randt_cell& raw() const { return c_; }

 private:
  double mahalanobis(const Cell& subtrahend, int use_intensity) const {
    double out = 0.0;
    if (!ctx_) return out;
    facade_check(randt_cells_mahalanobis(ctx_->get(), &c_, &subtrahend.c_, 1, use_intensity, &out), "randt_cells_mahalanobis", ctx_->get());
    return out;
  }
  randt_cell c_{};
  std::shared_ptr<Context> ctx_;
  int min_points_ = 0;
  std::vector<float> pending_;  // points_to_add_ as packed x y z I
  std::vector<std::pair<double, double>> pending_polar_;  // polar_points_to_add_
  std::vector<float> points_;   // points_ (ndt_cell.h:158)
  std::vector<std::pair<double, double>> polar_points_;
};

// rc::navigation::ndt::State (include/ndt_slam/trajectory_representation.h:12-22)
struct State {
  SE2d pose;
  std::array<double, 2> pos{0.0, 0.0};
  double rot = 0.0;
  std::array<double, 2> lin_vel{0.0, 0.0};
  double rot_vel = 0.0;
  std::array<double, 2> lin_acc{0.0, 0.0};
  double imu_bias = 0.0;
  double stamp = 0.0;
};

// rc::navigation::ndt::Map: one device-resident NDT map.
// COPIES SHARE THEIR STORAGE UNTIL ONE OF THEM IS WRITTEN (round 5).  The reference copies Maps by value ~7 times per scan
// (local_fuser.cpp:128-136,173-178) and writes to almost none of the copies: fmap / mmap / the deques' elements are only read,
// smoothed_map / global_map are never used.  A copy here is a shared_ptr copy; the first mutator called on a Map whose storage
// is shared clones it first (randt_maps_clone: pooled block, one launch) -- value semantics as far as any caller of this class
// can tell, no device work for copies that are only read.  (Writes THROUGH THE C ABI on handle() bypass this: use
// mutable_handle().)
class Map {
  struct Storage {
    randt_maps* m = nullptr;
    std::uint64_t id = 0, version = 0;   // identity of the storage object and a counter of writes to it (Matcher's staging cache)
    bool known_nonempty = false;
    std::vector<randt_cell> host_cells;  // the cells as last downloaded, valid while host_cells_version == version + 1
    std::uint64_t host_cells_version = 0;
    ~Storage() {
      if (m) randt_maps_destroy(m);  // the block goes back to the context's pool: no hipFree, no synchronisation
    }
  };
  static std::uint64_t next_id() {
    static std::uint64_t n = 0;
    return ++n;
  }

 public:
  Map() = default;
  Map(const Map&) = default;             // shares the storage (copy-on-write)
  Map& operator=(const Map&) = default;
  Map(Map&&) noexcept = default;
  Map& operator=(Map&&) noexcept = default;
  ~Map() = default;

  // Map::initialize (ndt_map.cpp:7-21)
  void initialize(std::shared_ptr<Context> ctx, const NDTMapParameters& p, double center_x, double center_y,
                  int cell_capacity = 0) {
    s_.reset();
    ctx_ = std::move(ctx);
    params_.size_x = p.size_x;
    params_.size_y = p.size_y;
    params_.resolution = p.resolution;
    params_.center_x = center_x;
    params_.center_y = center_y;
    params_.max_neighbour_dist = p.max_neighbour_manhattan_distance;
    params_.min_points_per_cell = p.min_points_per_cell;
    params_.reserved = 0;
    cap_ = cell_capacity > 0 ? cell_capacity : p.size_x * p.size_y;
    create();
  }

  // RadarPreprocessor::processScan's clustering + HierarchicalMap::addClusters
  // (radar_preprocessor.cpp:34-37, ndt_hierarchical_map.cpp:28-33): the whole filtered scan at once.
  // points: n x stride floats (pcl::PointXYZI: stride 8, intensity at 4).
  void addScan(const float* points, int n, int stride, int intensity_index, const RadarPreprocessorParameters& rp) {
    randt_cluster_params cp{rp.n_clusters, static_cast<float>(rp.max_range)};
    if (!writable()) return;
    s_->known_nonempty = false;  // the build replaces the map's content
    check(randt_ndt_build(ctx_->get(), points, n, stride, intensity_index, &cp, s_->m, 0), "randt_ndt_build");
  }

  // void insertCluster(const pcl::PointCloud<pcl::PointXYZI>& cluster, const std::vector<...>& angle_dists)
  //                                                                                    (ndt_map.cpp:238-245)
  // One cell from all the points of an already separated cluster; appended and indexed if it is accepted.
  void insertCluster(const float* points, int n, int stride, int intensity_index) {
    if (!writable()) return;
    check(randt_maps_insert_cluster(s_->m, 0, points, n, stride, intensity_index, nullptr), "randt_maps_insert_cluster");
  }
  // the reference's loop over insertCluster (ndt_hierarchical_map.cpp:28-33) in one call: cluster c = points
  // [offsets[c], offsets[c + 1]); asynchronous like insertCluster
  void insertClusters(const float* points, const std::vector<int>& offsets, int stride, int intensity_index) {
    if (offsets.size() < 2 || !writable()) return;
    static_assert(sizeof(int) == sizeof(int32_t), "offsets are passed as int32");
    check(randt_maps_insert_clusters(s_->m, 0, points, reinterpret_cast<const int32_t*>(offsets.data()), static_cast<int>(offsets.size()) - 1, stride,
                                     intensity_index, nullptr), "randt_maps_insert_clusters");
  }
  // int insertCell(const Cell& cell): grid_.push_back(cell), index grid untouched (ndt_map.h:137-140)
  int insertCell(const Cell& cell) {
    if (!writable()) return -1;
    check(randt_maps_insert_cells(s_->m, 0, &cell.raw(), 1, 0), "randt_maps_insert_cells");
    return static_cast<int>(get_n_cells()) - 1;
  }
  // void update(): Cell::updateCell on every cell (ndt_map.cpp:247-251).  Cells are always held in their updated
  // form on the device (the build folds addPointCloud + updateCell), so there is nothing left to do.
  void update() {}

  // unsigned int coordinateToIndex(const Eigen::Vector2f& point) const (ndt_map.h:87-90,181-184): index arithmetic
  // on the map's geometry, unsigned like the reference's.
  unsigned int coordinateToIndex(const Vector2f& point) const {
    const double offset_x = -static_cast<double>(params_.size_x) / 2.0 * params_.resolution + params_.center_x;
    const double offset_y = -static_cast<double>(params_.size_y) / 2.0 * params_.resolution + params_.center_y;
    const unsigned int mx = static_cast<unsigned int>((point[0] - offset_x) / params_.resolution);
    const unsigned int my = static_cast<unsigned int>((point[1] - offset_y) / params_.resolution);
    return my * static_cast<unsigned int>(params_.size_x) + mx;
  }

  // void getClosestCells(const Eigen::Vector2f& query_pt, const int& n_neighbours, std::vector<size_t>& indizes) const
  // void getClosestCells(const Cell& query_cell, const int& n_neighbours, std::vector<size_t>& indizes) const
  //                                                                                    (ndt_map.cpp:101-151)
  void getClosestCells(const Vector2f& query_pt, const int& n_neighbours, std::vector<size_t>& indizes) const {
    randt_cell q{};
    q.mean[0] = query_pt[0];
    q.mean[1] = query_pt[1];
    closest(q, n_neighbours, 0, indizes);
  }
  void getClosestCells(const Cell& query_cell, const int& n_neighbours, std::vector<size_t>& indizes) const {
    closest(query_cell.raw(), n_neighbours, 1, indizes);
  }

  unsigned int get_n_cells() const {
    int32_t n = 0;
    check(randt_maps_counts(handle(), 0, 1, &n), "randt_maps_counts");
    return static_cast<unsigned int>(n);
  }
  // A map only ever gains cells until it is cleared or rebuilt, so once a non-zero count has been read it is remembered:
  // LocalFuser::processScan's per-scan "_current_submap.isEmpty()" (local_fuser.cpp:123) then costs no device round trip.
  bool isEmpty() const {
    if (!s_) return true;
    if (s_->known_nonempty) return false;
    s_->known_nonempty = get_n_cells() != 0;
    return !s_->known_nonempty;
  }

  // The cell records on the host: ONE download per content of the map (remembered until the next write through this class), so
  // that loops over getCellMeanAndCovariance(i) / getPointsInCell(i) -- NDTSlam::createVisualizationMsg, ndt_slam.cpp:370-393 --
  // cost one device round trip, not one per cell.
  const std::vector<randt_cell>& hostCells() const {
    static const std::vector<randt_cell> none;
    if (!s_ || !s_->m) return none;
    if (s_->host_cells_version != s_->version + 1) {
      std::vector<randt_cell> raw(cap_);
      int n = 0;
      check(randt_maps_download(s_->m, 0, raw.data(), cap_, &n, nullptr), "randt_maps_download");  // (a deferred insert status: the outputs are valid)
      raw.resize(static_cast<size_t>(std::min(n, cap_)));
      s_->host_cells.swap(raw);
      s_->host_cells_version = s_->version + 1;
    }
    return s_->host_cells;
  }
  std::vector<Cell> getCells() const {
    const std::vector<randt_cell>& raw = hostCells();
    std::vector<Cell> out;
    out.reserve(raw.size());
    for (const randt_cell& c : raw) out.emplace_back(c, ctx_, params_.min_points_per_cell);
    return out;
  }

  std::vector<int> getGridIndizes() const {
    std::vector<int32_t> g(static_cast<size_t>(params_.size_x) * params_.size_y);
    int n = 0;
    check(randt_maps_download(handle(), 0, nullptr, 0, &n, g.data()), "randt_maps_download");
    return std::vector<int>(g.begin(), g.end());
  }

  // Map::getCellMeanAndCovariance (ndt_map.cpp:33-40)
  bool getCellMeanAndCovariance(unsigned int index, Vector3f& mean, Matrix3f& cov) const {
    const std::vector<randt_cell>& cells = hostCells();
    if (index < cells.size()) {
      const Cell c(cells[index]);
      mean = c.getIntensityMean();
      cov = c.getIntensityCov();
      return true;
    }
    std::cout << "WARNING: requested cell out of range!" << "\n";
    return false;
  }

  // the 2-D overload (ndt_map.cpp:23-31)
  bool getCellMeanAndCovariance(unsigned int index, Vector2f& mean, Matrix2f& cov) const {
    const std::vector<randt_cell>& cells = hostCells();
    if (index < cells.size()) {
      const Cell c(cells[index]);
      mean = c.getMean();
      cov = c.getCov();
      return true;
    }
    std::cout << "WARNING: requested cell out of range!" << "\n";
    return false;
  }
  // const size_t getPointsInCell(size_t i) const (ndt_map.h:66-68)
  size_t getPointsInCell(size_t i) const { return hostCells().at(i).n; }

  // Map::transformMap (ndt_map.cpp:177-182); index grid stays stale like in the reference
  void transformMap(const SE2d& trans) {
    if (writable()) check(randt_maps_transform(s_->m, 0, 1, trans.data()), "randt_maps_transform");
  }
  // Map::transformMapWithPointCloud (ndt_map.cpp:184-189): the per-cell point clouds only feed the OGM, which is not
  // part of this path -- the cell statistics move exactly as in transformMap
  void transformMapWithPointCloud(const SE2d& trans) { transformMap(trans); }
  // NOT in the reference: make a transformed map searchable again (DESIGN "reference quirks")
  void reindex() {
    if (writable()) check(randt_maps_reindex(s_->m, 0, 1), "randt_maps_reindex");
  }

  // Map::mergeMapCell (ndt_map.cpp:191-207): moving_map is expected already transformed, as in
  // local_fuser.cpp:177,190; mergeMapCellAt fuses the transform.
  void mergeMapCell(const Map& moving_map) { mergeMapCellAt(moving_map, SE2d()); }
  void mergeMapCellAt(const Map& moving_map, const SE2d& pose) {
    if (!writable()) return;
    s_->known_nonempty = s_->known_nonempty || (moving_map.s_ && moving_map.s_->known_nonempty);
    check(randt_maps_merge(s_->m, 0, moving_map.handle(), 0, 1, pose.data()), "randt_maps_merge");
  }

  void clear() {
    if (!s_) return;
    if (s_.use_count() > 1) {  // the copies keep the content; this map gets a fresh (cleared) batch instead of a clone to clear
      s_.reset();
      create();
      return;
    }
    ++s_->version;
    s_->known_nonempty = false;
    check(randt_maps_clear(s_->m, 0, 1), "randt_maps_clear");
  }

  // double Map::calculateCSDivergence(const Map& m_map)                       (ndt_map.cpp:42-99)
  // (the moving map already transformed by the caller, like local_fuser.cpp:338-339)
  double calculateCSDivergence(const Map& m_map) const {
    double v = 0.0;
    if (!check(randt_cs_divergence(ctx_->get(), handle(), 0, m_map.handle(), 0, nullptr, &v, nullptr), "randt_cs_divergence"))
      return failed_value();  // NaN: "identical maps" (0) would pass a loop-closure gate
    return v;
  }

  // read access for the C ABI (registration, copies); writes through it are invisible to the maps that share the storage
  randt_maps* handle() const { return s_ ? s_->m : nullptr; }
  // write access: detaches from the copies first
  randt_maps* mutable_handle() {
    if (!writable()) return nullptr;
    s_->known_nonempty = false;  // whatever is written through the handle may replace the content: isEmpty() asks again
    return s_->m;
  }
  // identity and write counter of the storage behind this map: equal pairs = equal device content (Matcher's staging cache)
  std::uint64_t storage_id() const { return s_ ? s_->id : 0; }
  std::uint64_t storage_version() const { return s_ ? s_->version : 0; }
  const std::shared_ptr<Context>& context() const { return ctx_; }
  const randt_map_params& params() const { return params_; }
  int capacity() const { return cap_; }

 private:
  void create() {
    s_ = std::make_shared<Storage>();
    s_->id = next_id();
    check(randt_maps_create(ctx_->get(), 1, &params_, cap_, 1, &s_->m), "randt_maps_create");
  }
  // before a write: a storage of its own (clone if copies share it), write counter bumped
  bool writable() {
    if (!s_ || !s_->m) return false;
    if (s_.use_count() > 1) {
      auto fresh = std::make_shared<Storage>();
      fresh->id = next_id();
      fresh->known_nonempty = s_->known_nonempty;
      if (!check(randt_maps_clone(s_->m, 0, 1, &fresh->m), "randt_maps_clone")) return false;
      s_ = std::move(fresh);
    }
    ++s_->version;
    return true;
  }
  bool check(int rc, const char* what) const { return facade_check(rc, what, ctx_ ? ctx_->get() : nullptr); }
  void closest(const randt_cell& q, int n_neighbours, int mahalanobis, std::vector<size_t>& indizes) const {
    if (n_neighbours <= 0) return;
    std::vector<int32_t> out(static_cast<size_t>(n_neighbours), -1);
    check(randt_closest_cells(ctx_->get(), handle(), 0, &q, 1, n_neighbours, mahalanobis, 1, out.data()), "randt_closest_cells");
    for (int32_t v : out)
      if (v >= 0) indizes.push_back(static_cast<size_t>(v));  // appended, like the reference's push_back
  }
  std::shared_ptr<Context> ctx_;
  randt_map_params params_{};
  int cap_ = 0;
  std::shared_ptr<Storage> s_;
};

// rc::navigation::ndt::RadarPreprocessor (include/radar_preprocessing/radar_preprocessor.h:20-60), the data path: filterScan on a
// raw polar scan in host memory (a sensor message's cloud: n_azimuths x n_bins points, azimuth after azimuth, range ascending --
// the layout the reference's filterScan assumes, radar_preprocessor.cpp:61) and processScan = filterScan + clustering + NDT.
struct RadarFilterParameters {  // the filterScan members of RadarPreprocessorParameters (ndt_slam_parameters.h:30-50)
  float min_range = 0.6f, max_range = 12.0f, min_intensity = 6.0f, beam_distance_increment_threshold = 0.04f;
  std::array<float, 12> sensor_to_base{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // initial_transform_radar_baselink_, row-major 3 x 4
};
class RadarPreprocessor {
 public:
  void initialize(std::shared_ptr<Context> ctx, const RadarPreprocessorParameters& clustering, const RadarFilterParameters& filter) {
    ctx_ = std::move(ctx);
    clustering_ = clustering;
    fp_.min_range = filter.min_range;
    fp_.max_range = filter.max_range;
    fp_.min_intensity = filter.min_intensity;
    fp_.beam_distance_increment_threshold = filter.beam_distance_increment_threshold;
    std::copy(filter.sensor_to_base.begin(), filter.sensor_to_base.end(), fp_.sensor_to_base);
  }
  // void filterScan(cloud_in, cloud_out, polar_points, max_detections)                    (radar_preprocessor.cpp:45-125)
  // cloud_out: packed x y z I in the base frame; polar_points: (angle, range) per kept point; max_detections: (angle, range,
  // intensity) of every flushed azimuth.  Returns false (outputs untouched) if the cloud is not azimuth-organised or the call failed.
  bool filterScan(const float* raw, int n_azimuths, int n_bins, int stride, int intensity_index, std::vector<float>& cloud_out,
                  std::vector<std::pair<double, double>>& polar_points, std::vector<std::array<double, 3>>& max_detections) const {
    int capacity = 8192;
    for (;;) {
      std::vector<float> pts(static_cast<size_t>(capacity) * 4), pol(static_cast<size_t>(capacity) * 2), pk(static_cast<size_t>(n_azimuths) * 3);
      int n = 0, npk = 0, status = 0;
      if (!facade_check(randt_filter_scan(ctx_->get(), raw, n_azimuths, n_bins, stride, intensity_index, &fp_, pts.data(), capacity, &n, pol.data(), pk.data(),
                                          &npk, &status), "randt_filter_scan", ctx_->get()))
        return false;
      if (status == 2 && capacity < n_azimuths * n_bins) {  // more kept points than the buffer: once more with a larger one
        capacity = std::min(capacity * 4, n_azimuths * n_bins);
        continue;
      }
      if (status != 0) {
        std::cout << "WARNING: filterScan: the cloud is not organised azimuth after azimuth -- previous value kept\n";
        return false;
      }
      cloud_out.assign(pts.begin(), pts.begin() + 4 * static_cast<size_t>(n));
      polar_points.clear();
      for (int i = 0; i < n; ++i) polar_points.emplace_back(pol[2 * i], pol[2 * i + 1]);
      max_detections.clear();
      for (int i = 0; i < npk; ++i) max_detections.push_back({pk[3 * i], pk[3 * i + 1], pk[3 * i + 2]});
      return true;
    }
  }
  // RadarPreprocessor::processScan + HierarchicalMap::addClusters (local_fuser.cpp:102-105) without the host round trip of the
  // filtered points: raw scan up, filter -> clustering -> NDT on the device into scan_ndt.  Returns the filter's status check.
  bool processScan(const float* raw, int n_azimuths, int n_bins, int stride, int intensity_index, Map& scan_ndt, int max_points = 6144) const {
    randt_cluster_params cp{clustering_.n_clusters, static_cast<float>(clustering_.max_range)};
    int status = 0;
    randt_maps* m = scan_ndt.mutable_handle();
    if (!m || !facade_check(randt_filter_build(ctx_->get(), raw, n_azimuths, n_bins, stride, intensity_index, &fp_, &cp, max_points, m, 0, &status),
                            "randt_filter_build", ctx_->get()))
      return false;
    if (status != 0) std::cout << "WARNING: processScan: filter status " << status << (status == 2 ? " (more kept points than max_points)" : " (cloud not azimuth-organised)") << "\n";
    return status == 0;
  }

 private:
  std::shared_ptr<Context> ctx_;
  RadarPreprocessorParameters clustering_;
  randt_filter_params fp_{};
};

// rc::navigation::ndt::HierarchicalMap, the NDT side only (include/ndt_representation/ndt_hierarchical_map.h): the OGM
// ray tracing is outside this path, so the class is the pass-through LocalFuser uses to fill a scan's NDT map
// (ndt_hierarchical_map.cpp:28-33 addClusters -> Map::insertCluster per cluster; :35-37 getMap; clear / transform).
class HierarchicalMap {
 public:
  void initialize(std::shared_ptr<Context> ctx, const NDTMapParameters& p, double center_x, double center_y, int cell_capacity = 0) {
    ndt_map_.initialize(std::move(ctx), p, center_x, center_y, cell_capacity);
    is_empty = true;  // ndt_hierarchical_map.cpp:15
  }
  // void addClusters(const std::vector<pcl::PointCloud<pcl::PointXYZI>>& clusters, ...): cluster c = points
  // [offsets[c], offsets[c+1]) of one n x stride array, inserted in order like the reference's loop
  void addClusters(const float* points, const std::vector<int>& offsets, int stride, int intensity_index) {
    ndt_map_.insertClusters(points, offsets, stride, intensity_index);  // one launch; the same map as one insertCluster per cluster
    is_empty = false;  // :32 -- whatever the clusters amounted to
  }
  // the whole filtered scan at once (clustering on the device): what RadarPreprocessor::processScan + addClusters amount to
  void addScan(const float* points, int n, int stride, int intensity_index, const RadarPreprocessorParameters& rp) {
    ndt_map_.addScan(points, n, stride, intensity_index, rp);
    is_empty = false;
  }
  const Map& getMap() const { return ndt_map_; }
  Map& getMap() { return ndt_map_; }
  void clear() { ndt_map_.clear(); }  // (:7-10: the flag is NOT touched; LocalFuser::initializeNewSubmap re-initialises right behind it)
  void transformMap(const SE2d& trans) { ndt_map_.transformMap(trans); }                            // ndt_hierarchical_map.cpp:74-76
  void transformMapWithPointCloud(const SE2d& trans) { ndt_map_.transformMapWithPointCloud(trans); }  // :78-80
  // void mergeMapCell(const HierarchicalMap& m_map) (:68-72): the NDT layer's Map::mergeMapCell
  void mergeMapCell(const HierarchicalMap& m_map) {
    ndt_map_.mergeMapCell(m_map.ndt_map_);
    is_empty = false;  // :71 -- also when the merged map held no cell
  }
  // inline bool isEmpty() const { return is_empty; } (ndt_hierarchical_map.h:85-87): a FLAG -- true from initialize() until the
  // first addClusters / mergeMapCell, whatever those put into the map -- not Map::isEmpty()'s cell count (ndt_map.h:144-146).
  // LocalFuser::processScan's "first scan of the submap" test reads this one (local_fuser.cpp:108).
  bool isEmpty() const { return is_empty; }
  // void transformMapToOrigin(const Sophus::SE2d& new_origin) (:82-85) / getOrigin (h:91-93): the submap's origin in the global
  // frame -- bookkeeping for the OGM layer (LocalFuser::updateSubmaps, local_fuser.cpp:72,78); the cells do not move
  void transformMapToOrigin(const SE2d& new_origin) { origin_in_global_frame_ = new_origin; }
  const SE2d& getOrigin() const { return origin_in_global_frame_; }

 private:
  Map ndt_map_;
  SE2d origin_in_global_frame_;
  bool is_empty = true;
};

// Several GPUs behind one caller: the C ABI's multi-GPU group (randt_group_*, csrc/group.hip) for the batched loop
// registration below.  devices = HIP device indices of the members (member 0 must be the device the facade Maps live
// on; a repeated index gives "virtual ranks" that share one GPU).
class DeviceGroup {
 public:
  explicit DeviceGroup(const std::vector<int>& devices, int transport = RANDT_TRANSPORT_AUTO) {
    if (!facade_check(randt_group_create(devices.data(), static_cast<int>(devices.size()), nullptr, transport, &g_), "randt_group_create", nullptr))
      std::cout << "WARNING: randt_group_create: " << randt_group_last_error(nullptr) << std::endl;  // the RCCL / HIP text survives the object
    if (g_) randt_group_info(g_, &world_, &n_local_, nullptr, &transport_);
  }
  ~DeviceGroup() { randt_group_destroy(g_); }
  DeviceGroup(const DeviceGroup&) = delete;
  DeviceGroup& operator=(const DeviceGroup&) = delete;
  randt_group* get() const { return g_; }
  int size() const { return world_; }
  int transport() const { return transport_; }

 private:
  randt_group* g_ = nullptr;
  int world_ = 0, n_local_ = 0, transport_ = 0;
};

// rc::navigation::ndt::Matcher (include/ndt_registration/ndt_matcher.h:46-87): the public methods under the reference's
// signatures (Sophus::SE2d -> SE2d, Eigen -> std::array); everything numeric goes through the C ABI.
class Matcher {
 public:
  Matcher() = default;
  Matcher(const Matcher&) = delete;
  Matcher& operator=(const Matcher&) = delete;
  ~Matcher() { release_stage(); }

  // void Matcher::initialize(NDTMatcherParameters parameters)                          (ndt_matcher.cpp:7-16)
  void initialize(const NDTMatcherParameters& parameters) { parameters_ = parameters; }

  // double Matcher::estimateLoopConstraint(Sophus::SE2d& trans, const Map& old_ndt, Map& new_ndt,
  //   int max_gnc_steps, bool use_intensity_as_dimension, double scale)   (ndt_matcher.cpp:426-493)
  // Same contract: trans is in/out, the return value is final_cost / num_residual_blocks; with no
  // residuals it prints the reference's warning and leaves trans untouched.
  double estimateLoopConstraint(SE2d& trans, const Map& old_ndt, Map& new_ndt, int max_gnc_steps,
                                bool use_intensity_as_dimension, double scale, randt_result* stats = nullptr) const {
    randt_matcher_params mp;
    if (!loop_params(max_gnc_steps, use_intensity_as_dimension, scale, &mp, old_ndt.context()->get())) return failed_value();
    randt_result r{};
    int rc = randt_register_pair(old_ndt.context()->get(), old_ndt.handle(), 0, new_ndt.handle(), 0, &mp, trans.data(), &r);
    if (stats) *stats = r;
    if (!facade_check(rc, "randt_register_pair", old_ndt.context()->get())) return failed_value();  // trans untouched: outputs are written on success only
    if (r.n_residuals == 0) std::cout << "WARNING: NO RESIDUALS ADDED!" << std::endl;
    return r.cost;
  }

  // NOT in the reference (its detectLoopClosures registers the candidates one by one, local_fuser.cpp:329-339,370-397):
  // the same call for a batch of independent candidates, sharded over the GPUs of `group`.  Pair p registers
  // *moving_ndts[p] against *fixed_ndts[fixed_of_pair[p]] from trans[p] (in/out).  The maps are staged into one batch
  // per member GPU and broadcast (RCCL / peer copies), every member registers its contiguous share, the poses come
  // back to the caller; results are bit-identical to estimateLoopConstraint pair by pair.  Returns the costs.
  std::vector<double> estimateLoopConstraintBatch(DeviceGroup& group, std::vector<SE2d>& trans, const std::vector<const Map*>& fixed_ndts,
                                                  const std::vector<int>& fixed_of_pair, const std::vector<const Map*>& moving_ndts,
                                                  int max_gnc_steps, bool use_intensity_as_dimension, double scale,
                                                  std::vector<randt_result>* stats = nullptr) const {
    const int n_pairs = static_cast<int>(moving_ndts.size()), n_fixed = static_cast<int>(fixed_ndts.size());
    std::vector<double> cost(static_cast<size_t>(n_pairs), failed_value());
    randt_group* g = group.get();
    randt_matcher_params mp;
    if (!g || n_pairs == 0 || n_fixed == 0 || static_cast<int>(trans.size()) != n_pairs || static_cast<int>(fixed_of_pair.size()) != n_pairs) {
      facade_check(RANDT_ERR_INVALID, "estimateLoopConstraintBatch: group / argument sizes", nullptr);
      return cost;
    }
    if (!loop_params(max_gnc_steps, use_intensity_as_dimension, scale, &mp, randt_group_ctx(g, 0))) return cost;
    int world = 0, n_local = 0;
    randt_group_info(g, &world, &n_local, nullptr, nullptr);
    int fcap = 1, mcap = 1;
    for (const Map* m : fixed_ndts) fcap = std::max(fcap, m->capacity());
    for (const Map* m : moving_ndts) mcap = std::max(mcap, m->capacity());
    // one parameter set per batch (the reference has one, with a zero centre): differently centred or scaled maps would be
    // associated in the wrong geometry without an error
    for (const Map* m : fixed_ndts)
      if (std::memcmp(&m->params(), &fixed_ndts[0]->params(), sizeof(randt_map_params)) != 0) {
        facade_check(RANDT_ERR_INVALID, "estimateLoopConstraintBatch: the fixed maps must share one geometry (centre, resolution, window)", nullptr);
        return cost;
      }
    for (const Map* m : moving_ndts)
      if (std::memcmp(&m->params(), &moving_ndts[0]->params(), sizeof(randt_map_params)) != 0) {
        facade_check(RANDT_ERR_INVALID, "estimateLoopConstraintBatch: the moving maps must share one geometry", nullptr);
        return cost;
      }
    std::vector<randt_maps*> fb(static_cast<size_t>(n_local), nullptr), mb(static_cast<size_t>(n_local), nullptr);
    auto cleanup = [&] {
      randt_group_synchronize(g);
      for (randt_maps* m : fb) randt_maps_destroy(m);
      for (randt_maps* m : mb) randt_maps_destroy(m);
    };
    int rc = RANDT_OK;
    for (int i = 0; i < n_local && !rc; ++i) {
      rc = randt_maps_create(randt_group_ctx(g, i), n_fixed, &fixed_ndts[0]->params(), fcap, 1, &fb[i]);
      if (!rc) rc = randt_maps_create(randt_group_ctx(g, i), n_pairs, &moving_ndts[0]->params(), mcap, 0, &mb[i]);
    }
    // stage on member 0 (the facade Maps' device), then one broadcast per batch
    for (int f = 0; f < n_fixed && !rc; ++f) {
      randt_ctx_synchronize(fixed_ndts[f]->context()->get());
      rc = randt_maps_copy(fb[0], f, fixed_ndts[f]->handle(), 0, 1);
    }
    for (int p = 0; p < n_pairs && !rc; ++p) {
      randt_ctx_synchronize(moving_ndts[p]->context()->get());
      rc = randt_maps_copy(mb[0], p, moving_ndts[p]->handle(), 0, 1);
    }
    if (!rc) rc = randt_group_broadcast_maps(g, fb.data(), 0, n_fixed, 0);
    if (!rc) rc = randt_group_broadcast_maps(g, mb.data(), 0, n_pairs, 0);
    std::vector<double> pose(4 * static_cast<size_t>(n_pairs));
    std::vector<randt_result> res(static_cast<size_t>(n_pairs));
    std::vector<int32_t> fidx(fixed_of_pair.begin(), fixed_of_pair.end());
    for (int p = 0; p < n_pairs; ++p) std::copy(trans[p].d, trans[p].d + 4, pose.begin() + 4 * p);
    if (!rc) rc = randt_group_register_pairs(g, fb.data(), fidx.data(), mb.data(), n_pairs, &mp, pose.data(), res.data());
    if (rc != RANDT_OK) {
      std::cout << "WARNING: batched loop registration failed: " << randt_group_last_error(g) << std::endl;
      facade_check(rc, "estimateLoopConstraintBatch", randt_group_ctx(g, 0));
      cleanup();
      return cost;
    }
    for (int p = 0; p < n_pairs; ++p) {
      if (res[p].n_residuals == 0) {
        std::cout << "WARNING: NO RESIDUALS ADDED!" << std::endl;  // pose untouched, like the single call
      }
      std::copy(pose.begin() + 4 * p, pose.begin() + 4 * p + 4, trans[p].d);
      cost[p] = res[p].cost;
    }
    if (stats) *stats = res;
    cleanup();
    last_status() = RANDT_OK;
    return cost;
  }

  // double Matcher::estimateTransformGlobalBNB(Sophus::SE2d& trans, const Map& fixed_ndt, Map& moving_ndt,
  //   bool use_intensity_as_dimension, double scale, double search_window_size_linear,
  //   double search_window_size_angular)                                       (ndt_matcher.cpp:495-608)
  // The csm_* members come from the parameters given to initialize(), like in the reference.
  double estimateTransformGlobalBNB(SE2d& trans, const Map& fixed_ndt, Map& moving_ndt, bool use_intensity_as_dimension, double scale,
                                    double search_window_size_linear, double search_window_size_angular) const {
    randt_bnb_params csm{};
    csm.csm_window_linear = parameters_.csm_window_linear;
    csm.csm_window_angular = parameters_.csm_window_angular;
    csm.csm_linear_step = parameters_.csm_linear_step;
    csm.csm_cost_threshold = parameters_.csm_cost_threshold;
    csm.csm_max_px_accurate_range = parameters_.csm_max_px_accurate_range;
    csm.csm_n_iter = parameters_.csm_n_iter;
    return estimateTransformGlobalBNB(trans, fixed_ndt, moving_ndt, use_intensity_as_dimension, scale, search_window_size_linear,
                                      search_window_size_angular, csm);
  }
  // the same with explicit csm_* values (not a reference signature)
  double estimateTransformGlobalBNB(SE2d& trans, const Map& fixed_ndt, Map& moving_ndt, bool use_intensity_as_dimension, double scale,
                                    double search_window_size_linear, double search_window_size_angular,
                                    const randt_bnb_params& csm) const {
    randt_matcher_params mp;
    randt_matcher_params_default(&mp);
    mp.loss_alpha = parameters_.loss_function_convexity;
    mp.lookup_mahalanobis = parameters_.lookup_mahalanobis ? 1 : 0;
    mp.use_intensity = use_intensity_as_dimension ? 1 : 0;
    double min_cost = 0.0;
    const int rc = randt_search_global(fixed_ndt.context()->get(), fixed_ndt.handle(), 0, moving_ndt.handle(), 0, &mp, &csm, scale,
                                       search_window_size_linear, search_window_size_angular, trans.data(), &min_cost, nullptr);
    if (!facade_check(rc, "randt_search_global", fixed_ndt.context()->get())) return failed_value();
    return min_cost;
  }

  void resetMatcher() { imu_constraints_.clear(); }  // ndt_matcher.cpp:18-20

  // void Matcher::predictTransform(const double& initial_angle_guess, const double& stamp,
  //                                std::vector<State>& trajectory)            (ndt_matcher.cpp:22-59)
  // optimize_on_manifold = false (or the analytic flag) takes the reference's vector-form `predict` (:27-41).
  void predictTransform(const double& initial_angle_guess, const double& stamp, std::vector<State>& trajectory) {
    if (trajectory.empty()) return;
    randt_state last = toAbi(trajectory.back()), next;
    const bool vec = parameters_.use_analytic_expressions_for_optimization || !parameters_.optimize_on_manifold;
    randt_predict_state_param(&last, stamp, vec ? RANDT_PARAM_VECTOR : RANDT_PARAM_MANIFOLD, &next);
    trajectory.push_back(fromAbi(next));
    imu_constraints_.push_back(initial_angle_guess);
  }

  // void Matcher::estimateTransformCeres(Sophus::SE2d& trans, std::vector<State>& trajectory,
  //      const double& initial_angle_guess, const double& stamp, const std::deque<Map>& fixed_ndts,
  //      const std::deque<Map>& moving_ndts)                                   (ndt_matcher.cpp:322-424)
  // THE REFERENCE SIGNATURE.  The window's maps (<= 2 fixed, the newest smoothing_steps moving ones) are copied on the
  // device into two internal batches (Map is one device-resident map; the window kernel wants its maps side by side);
  // every window parameter comes from initialize(), like in the reference.
  void estimateTransformCeres(SE2d& trans, std::vector<State>& trajectory, const double& initial_angle_guess, const double& stamp,
                              const std::deque<Map>& fixed_ndts, const std::deque<Map>& moving_ndts, randt_result* stats = nullptr) {
    if (trajectory.size() < 2 || fixed_ndts.empty() || moving_ndts.empty()) return;
    const size_t S = std::min(trajectory.size() - 1, static_cast<size_t>(parameters_.smoothing_steps));  // :343
    if (moving_ndts.size() < S) {
      facade_check(RANDT_ERR_INVALID, "estimateTransformCeres: fewer moving maps than window states", nullptr);
      return;
    }
    const std::shared_ptr<Context>& ctx = fixed_ndts.front().context();
    const int nf = static_cast<int>(fixed_ndts.size());
    int fcap = 1, mcap = 1;
    for (const Map& m : fixed_ndts) fcap = std::max(fcap, m.capacity());
    for (size_t i = 1; i <= S; ++i) mcap = std::max(mcap, moving_ndts.end()[-static_cast<long>(i)].capacity());
    for (const Map& m : fixed_ndts)
      if (std::memcmp(&m.params(), &fixed_ndts.front().params(), sizeof(randt_map_params)) != 0) {
        facade_check(RANDT_ERR_INVALID, "estimateTransformCeres: the fixed maps of a window must share one geometry (centre, resolution, window)", nullptr);
        return;
      }
    for (size_t i = 1; i <= S; ++i)
      if (std::memcmp(&moving_ndts.end()[-static_cast<long>(i)].params(), &moving_ndts.back().params(), sizeof(randt_map_params)) != 0) {
        facade_check(RANDT_ERR_INVALID, "estimateTransformCeres: the moving maps of a window must share one geometry", nullptr);
        return;
      }
    if (!ensure_stage(ctx, fixed_ndts.front().params(), fcap, nf, moving_ndts.back().params(), mcap, static_cast<int>(S))) return;
    // A staging slot that already holds a map's current content (same storage, same write counter) is not copied again: from
    // one scan to the next the window keeps S - 1 of its S scan maps, and the submap only changes on keyframe scans.
    std::vector<int32_t> fslots, mslots;
    std::vector<const Map*> fneed, mneed;
    for (const Map& m : fixed_ndts) fneed.push_back(&m);
    for (size_t i = S; i >= 1; --i) mneed.push_back(&moving_ndts.end()[-static_cast<long>(i)]);  // oldest first
    int rc = place_in_stage(stage_fixed_, stage_fixed_keys_, fneed, fslots);
    if (!rc) rc = place_in_stage(stage_moving_, stage_moving_keys_, mneed, mslots);
    if (!facade_check(rc, "randt_maps_copy (window staging)", ctx->get())) return;
    estimateTransformCeres(trans, trajectory, initial_angle_guess, stamp, stage_fixed_, fslots, stage_moving_, mslots, ctx->get(), window_params(),
                           stats);
  }

  // The same on maps that already live side by side in device batches (no staging copies): slots of `fixed_batch` /
  // `moving_batch` instead of deques (moving_slots: the scan window, oldest first).  Not a reference signature.
  void estimateTransformCeres(SE2d& trans, std::vector<State>& trajectory, const double& /*initial_angle_guess*/,
                              const double& /*stamp*/, randt_maps* fixed_batch, const std::vector<int32_t>& fixed_slots,
                              randt_maps* moving_batch, const std::vector<int32_t>& moving_slots, randt_ctx* ctx,
                              const randt_window_params& wp, randt_result* stats = nullptr) {
    if (trajectory.size() < 2) return;
    const size_t S = std::min(trajectory.size() - 1, static_cast<size_t>(wp.smoothing_steps > 0 ? wp.smoothing_steps : parameters_.smoothing_steps));  // :343
    randt_matcher_params mp;
    randt_matcher_params_default(&mp);
    mp.loss_scale = mp.mu_scale = parameters_.loss_function_scale;
    mp.loss_alpha = parameters_.loss_function_convexity;
    mp.gnc_divisor = parameters_.gnc_control_parameter_divisor;
    mp.gnc_steps = parameters_.gnc_steps;
    mp.max_iterations = parameters_.max_iteration;
    mp.n_neighbours = parameters_.n_results_kd_lookup;
    mp.lookup_mahalanobis = parameters_.lookup_mahalanobis ? 1 : 0;
    mp.use_intensity = parameters_.use_intensity_as_dimension ? 1 : 0;
    // optimize_on_manifold: false -> the (pos[2], rot) problem of ndt_matcher.cpp:290-313,330-335
    mp.parameterization = analytic() ? RANDT_PARAM_ANALYTIC : (parameters_.optimize_on_manifold ? RANDT_PARAM_MANIFOLD : RANDT_PARAM_VECTOR);
    std::vector<randt_state> st(S + 1);
    for (size_t j = 0; j <= S; ++j) st[j] = toAbi(trajectory.end()[-(long)(S + 1) + (long)j]);
    std::vector<int32_t> mv(moving_slots.end() - (long)S, moving_slots.end());  // moving_ndts.end()[-i], i = S..1
    std::vector<double> imu;
    if (wp.use_imu && imu_constraints_.size() > S)
      for (size_t i = S; i >= 1; --i) imu.push_back(imu_constraints_.end()[-(long)i - 1]);  // sic: one step older (:360)
    int rejected = 0;
    randt_result r{};
    int rc = randt_register_window(ctx, fixed_batch, fixed_slots.data(), (int)fixed_slots.size(), moving_batch, mv.data(), st.data(),
                                   (int)st.size(), imu.empty() ? nullptr : imu.data(), &mp, &wp, trans.data(), &rejected, &r);
    if (stats) *stats = r;
    if (!facade_check(rc, "randt_register_window", ctx)) return;  // trajectory and trans as they were
    for (size_t j = 0; j <= S; ++j) trajectory.end()[-(long)(S + 1) + (long)j] = fromAbi(st[j]);
  }

  // randt_window_params from the parameters given to initialize() (ndt_matcher.cpp:99: covariance_scaling_factor * motion_sqrtI)
  randt_window_params window_params() const {
    randt_window_params wp{};
    for (int i = 0; i < 64; ++i) wp.motion_sqrtI[i] = parameters_.covariance_scaling_factor * parameters_.motion_sqrtI[i];
    wp.ndt_weight = parameters_.ndt_weight;
    wp.weight_imu = parameters_.weight_imu;
    wp.weight_imu_bias = parameters_.weight_imu_bias;
    wp.pose_reject_translation = parameters_.pose_reject_translation;
    wp.pose_reject_rotation = parameters_.pose_reject_rotation;
    wp.smoothing_steps = parameters_.smoothing_steps;
    wp.use_imu = parameters_.use_imu ? 1 : 0;
    wp.use_constant_velocity_model = parameters_.use_constant_velocity_model ? 1 : 0;
    return wp;
  }

  static randt_state toAbi(const State& s) {
    randt_state a{};
    std::copy(s.pose.d, s.pose.d + 4, a.pose);
    a.pos[0] = s.pos[0]; a.pos[1] = s.pos[1]; a.rot = s.rot;
    a.lin_vel[0] = s.lin_vel[0]; a.lin_vel[1] = s.lin_vel[1]; a.rot_vel = s.rot_vel;
    a.lin_acc[0] = s.lin_acc[0]; a.lin_acc[1] = s.lin_acc[1]; a.imu_bias = s.imu_bias; a.stamp = s.stamp;
    return a;
  }
  static State fromAbi(const randt_state& a) {
    State s;
    std::copy(a.pose, a.pose + 4, s.pose.d);
    s.pos = {a.pos[0], a.pos[1]}; s.rot = a.rot;
    s.lin_vel = {a.lin_vel[0], a.lin_vel[1]}; s.rot_vel = a.rot_vel;
    s.lin_acc = {a.lin_acc[0], a.lin_acc[1]}; s.imu_bias = a.imu_bias; s.stamp = a.stamp;
    return s;
  }

 private:
  // `use_analytic_expressions_for_optimization: true` selects the reference's hand-written functors
  // (ceres_residuals.h:207-305, 372-419, 794-889) on (pos, rot) blocks whatever optimize_on_manifold says
  // (ndt_matcher.cpp:225-231, 330-335).  The NDT functors' rotation Jacobian is not the derivative for theta != 0 (SURVEY
  // a12); it is reproduced as written (RANDT_PARAM_ANALYTIC), because those are the iterates the reference takes.  A warning
  // says so once per Matcher.
  bool analytic() const {
    if (parameters_.use_analytic_expressions_for_optimization && !warned_analytic_) {
      std::cout << "WARNING: use_analytic_expressions_for_optimization: the reference's analytic NDT Jacobian is inexact for rotated "
                   "poses; reproduced as written" << std::endl;
      warned_analytic_ = true;
    }
    return parameters_.use_analytic_expressions_for_optimization;
  }
  bool loop_params(int max_gnc_steps, bool use_intensity_as_dimension, double scale, randt_matcher_params* mp, randt_ctx* ctx) const {
    (void)ctx;
    randt_matcher_params_default(mp);
    mp->loss_scale = scale;                              // BarronLoss(scale, ...)            (:479)
    mp->mu_scale = parameters_.loss_function_scale;      // gnc_mu uses the odometry scale    (:475)
    mp->loss_alpha = parameters_.loss_function_convexity;
    mp->loss_weight = 1.0;                               // ScaledLoss(..., 1, ...)          (:479)
    mp->gnc_divisor = parameters_.gnc_control_parameter_divisor;
    mp->gnc_steps = max_gnc_steps;
    mp->max_iterations = parameters_.max_iteration;
    mp->n_neighbours = parameters_.n_results_kd_lookup;
    mp->lookup_mahalanobis = parameters_.lookup_mahalanobis ? 1 : 0;
    mp->use_intensity = use_intensity_as_dimension ? 1 : 0;
    // optimize_on_manifold = true: the residuals hang on an un-manifolded 4-vector; false: (pos, rot); the analytic
    // flag: (pos, rot) with the hand-written functors (ndt_matcher.cpp:428-433: no manifold either way)
    mp->parameterization = analytic() ? RANDT_PARAM_ANALYTIC : (parameters_.optimize_on_manifold ? RANDT_PARAM_AMBIENT4 : RANDT_PARAM_VECTOR);
    return true;
  }
  using StageKey = std::pair<std::uint64_t, std::uint64_t>;  // (Map::storage_id, Map::storage_version) of what a staging slot holds
  static int place_in_stage(randt_maps* batch, std::vector<StageKey>& keys, const std::vector<const Map*>& need, std::vector<int32_t>& slots) {
    slots.assign(need.size(), -1);
    std::vector<char> used(keys.size(), 0);
    for (size_t i = 0; i < need.size(); ++i) {  // what is already there
      const StageKey k{need[i]->storage_id(), need[i]->storage_version()};
      if (k.first == 0) continue;
      for (size_t sl = 0; sl < keys.size(); ++sl)
        if (!used[sl] && keys[sl] == k) {
          slots[i] = static_cast<int32_t>(sl);
          used[sl] = 1;
          break;
        }
    }
    for (size_t i = 0; i < need.size(); ++i) {  // the rest: one device copy each into a slot nobody needs
      if (slots[i] >= 0) continue;
      size_t sl = 0;
      while (sl < keys.size() && used[sl]) ++sl;
      if (sl == keys.size()) return RANDT_ERR_INVALID;
      const int rc = randt_maps_copy(batch, static_cast<int>(sl), need[i]->handle(), 0, 1);
      if (rc) return rc;
      keys[sl] = StageKey{need[i]->storage_id(), need[i]->storage_version()};
      used[sl] = 1;
      slots[i] = static_cast<int32_t>(sl);
    }
    return RANDT_OK;
  }
  // internal batches the deque overload of estimateTransformCeres copies the window's maps into
  bool ensure_stage(const std::shared_ptr<Context>& ctx, const randt_map_params& fp, int fcap, int n_fixed, const randt_map_params& mpar,
                    int mcap, int n_moving) {
    // a staging batch is reused only for maps of the SAME geometry (centre, resolution, window, min points: the association
    // and the index arithmetic read them from the batch) -- not merely the same slot count
    auto fits = [](randt_maps* b, const randt_map_params& have, const randt_map_params& p, int cap_needed, int n) {
      if (!b) return false;
      int nm = 0, cap = 0, slots = 0;
      randt_maps_info(b, &nm, &cap, &slots, nullptr);
      return nm >= n && cap >= cap_needed && std::memcmp(&have, &p, sizeof(randt_map_params)) == 0;
    };
    if (stage_ctx_ != ctx) release_stage();
    int rc = RANDT_OK;
    if (!fits(stage_fixed_, stage_fixed_params_, fp, fcap, n_fixed)) {
      randt_maps_destroy(stage_fixed_);
      stage_fixed_ = nullptr;
      rc = randt_maps_create(ctx->get(), std::max(2, n_fixed), &fp, fcap, 1, &stage_fixed_);
      stage_fixed_params_ = fp;
      stage_fixed_keys_.assign(static_cast<size_t>(std::max(2, n_fixed)), StageKey{0, 0});
    }
    if (!rc && !fits(stage_moving_, stage_moving_params_, mpar, mcap, n_moving)) {
      randt_maps_destroy(stage_moving_);
      stage_moving_ = nullptr;
      rc = randt_maps_create(ctx->get(), std::max(4, n_moving), &mpar, mcap, 0, &stage_moving_);
      stage_moving_params_ = mpar;
      stage_moving_keys_.assign(static_cast<size_t>(std::max(4, n_moving)), StageKey{0, 0});
    }
    stage_ctx_ = ctx;
    return facade_check(rc, "randt_maps_create (window staging)", ctx->get());
  }
  void release_stage() {
    randt_maps_destroy(stage_fixed_);
    randt_maps_destroy(stage_moving_);
    stage_fixed_ = stage_moving_ = nullptr;
    stage_fixed_keys_.clear();
    stage_moving_keys_.clear();
    stage_ctx_.reset();
  }

  NDTMatcherParameters parameters_;
  mutable bool warned_analytic_ = false;
  std::vector<double> imu_constraints_;
  std::shared_ptr<Context> stage_ctx_;
  randt_maps *stage_fixed_ = nullptr, *stage_moving_ = nullptr;
  randt_map_params stage_fixed_params_{}, stage_moving_params_{};  // what the staging batches were created with
  std::vector<StageKey> stage_fixed_keys_, stage_moving_keys_;     // what each staging slot holds
};

// rc::navigation::ndt::ScanContextParameters (include/ndt_slam/ndt_slam_parameters.h; ndt_slam.cpp:515-552)
struct ScanContextParameters {
  int PC_NUM_RING = 20, PC_NUM_SECTOR = 45;
  double PC_MAX_RADIUS = 15.0;
  int NUM_EXCLUDE_RECENT = 15, NUM_CANDIDATES_FROM_TREE = 10;
  double SEARCH_RATIO = 0.3, SC_DIST_THRES = 0.6;
  int TREE_MAKING_PERIOD_ = 10;  // accepted for source compatibility; the device search always sees the current database
  double assumed_drift = 0.05, odom_eps = 1.2, odom_weight = 0.2, intensity_factor = 0.04;
};

// SCManager (include/local_fuser/Scancontext.h:50-103): the user-side API LocalFuser calls
// (local_fuser.cpp:30,207,284,323); the descriptor database lives on the device.
class SCManager {
 public:
  SCManager() = default;
  SCManager(const SCManager&) = delete;
  SCManager& operator=(const SCManager&) = delete;
  ~SCManager() { randt_sc_db_destroy(db_); }

  void initialize(const std::shared_ptr<Context>& ctx, const ScanContextParameters& params) {
    ctx_ = ctx;
    randt_sc_params p{};
    p.num_ring = params.PC_NUM_RING;
    p.num_sector = params.PC_NUM_SECTOR;
    p.max_radius = params.PC_MAX_RADIUS;
    p.num_exclude_recent = params.NUM_EXCLUDE_RECENT;
    p.num_candidates = params.NUM_CANDIDATES_FROM_TREE;
    p.search_ratio = params.SEARCH_RATIO;
    p.dist_thresh = params.SC_DIST_THRES;
    p.assumed_drift = params.assumed_drift;
    p.odom_eps = params.odom_eps;
    p.odom_weight = params.odom_weight;
    p.intensity_factor = params.intensity_factor;
    randt_sc_db_destroy(db_);
    db_ = nullptr;
    const int rc = randt_sc_db_create(ctx_->get(), &p, 256, &db_);
    facade_check(rc, "randt_sc_db_create", ctx_->get());
  }

  // void makeAndSaveScancontextAndKeys(pcl::PointCloud<SCPointType>::Ptr scan, Eigen::Vector2d& odom_position,
  //                                    double& traversed_distance)                       (Scancontext.cpp:240-258)
  void makeAndSaveScancontextAndKeys(const float* points, int n, int stride, int intensity_index,
                                     const std::array<double, 2>& odom_position, const double& traversed_distance) {
    const int rc = randt_sc_db_append(db_, points, n, stride, intensity_index, odom_position.data(), traversed_distance, nullptr);
    facade_check(rc, "randt_sc_db_append", ctx_ ? ctx_->get() : nullptr);
  }

  // std::pair<int, float> detectLoopClosureID(int node_id): nearest node index or -1, relative yaw  (:261-341)
  std::pair<int, float> detectLoopClosureID(int node_id) const {
    int loop_id = -1;
    float yaw = 0.f;
    const int rc = randt_sc_db_detect(db_, node_id, &loop_id, &yaw, nullptr);
    if (!facade_check(rc, "randt_sc_db_detect", ctx_ ? ctx_->get() : nullptr)) return {-1, 0.f};  // "no loop found"
    return {loop_id, yaw};
  }

  int size() const { return randt_sc_db_size(db_); }

 private:
  std::shared_ptr<Context> ctx_;
  randt_sc_db* db_ = nullptr;
};

// Pose / Constraint (include/ndt_slam/trajectory_representation.h:25-52): graph node and edge.  Matrices row-major.
struct Pose {
  SE2d pose;
  std::array<double, 2> pos{0.0, 0.0};
  double rot = 0.0;
  std::array<double, 9> cov{};
  std::array<double, 4> cov_pos_pos{};
  std::array<double, 2> cov_pos_rot{};
  double cov_rot_rot = 0.0;
  double traversed_dist = 0.0;
};
struct Constraint {
  int id_begin = 0, id_end = 0;
  SE2d trans;
  std::array<double, 9> sqrt_information{};
};
struct GlobalFuserParameters {  // include/ndt_slam/ndt_slam_parameters.h:134-139
  double loss_function_scale = 60.0;
  bool use_robust_loss = false;
};

// GlobalFuser (include/global_fuser/global_fuser.h:32-60, src/global_fuser/global_fuser.cpp:7-105)
class GlobalFuser {
 public:
  void initialize(const std::shared_ptr<Context>& ctx, GlobalFuserParameters parameters) {
    ctx_ = ctx;
    parameters_ = parameters;
    n_optimized_constraints_ = 0;
    n_optimized_poses_ = 0;
  }

  // void optimizePoseGraph(std::map<int, Pose>& poses_ref, const std::vector<Constraint>& edges, std::mutex& poses_mutex,
  //                        int max_update_index)                                        (global_fuser.cpp:13-105)
  // Same protocol: copy the nodes under the mutex, optimise the copy, write back keys 0..size-1 under the mutex.  Node
  // keys must be 0..size-1 (the reference's write-back loop assumes it, :100-102).  On a solver error the nodes are left
  // as they were and a warning goes to std::cout.
  void optimizePoseGraph(std::map<int, Pose>& poses_ref, const std::vector<Constraint>& edges, std::mutex& poses_mutex,
                         int max_update_index) {
    std::unique_lock<std::mutex> lock(poses_mutex);
    std::map<int, Pose> poses(poses_ref);
    lock.unlock();
    const int n = static_cast<int>(poses.size());
    const int loop_closures = static_cast<int>(edges.size()) + 1 - n;
    std::cout << "detected " << loop_closures << " loop closure constraints so far" << std::endl;
    std::vector<double> x(3 * static_cast<size_t>(n)), meas(3 * edges.size()), sqi(9 * edges.size());
    std::vector<int32_t> ia(edges.size()), ib(edges.size());
    for (int i = 0; i < n; ++i) {
      const Pose& p = poses.at(i);
      x[3 * i + 0] = p.pos[0];
      x[3 * i + 1] = p.pos[1];
      x[3 * i + 2] = p.rot;
    }
    for (size_t e = 0; e < edges.size(); ++e) {
      ia[e] = edges[e].id_begin;
      ib[e] = edges[e].id_end;
      meas[3 * e + 0] = edges[e].trans.d[2];
      meas[3 * e + 1] = edges[e].trans.d[3];
      meas[3 * e + 2] = edges[e].trans.angle();  // trans.log()(2)
      std::copy(edges[e].sqrt_information.begin(), edges[e].sqrt_information.end(), sqi.begin() + 9 * e);
    }
    randt_pg_params pp;
    randt_pg_params_default(&pp);
    pp.use_robust_loss = parameters_.use_robust_loss ? 1 : 0;
    pp.loss_scale = parameters_.loss_function_scale;
    randt_pg_result res{};
    const int rc = randt_pose_graph_optimize(ctx_->get(), n, x.data(), static_cast<int>(edges.size()), ia.data(), ib.data(),
                                             meas.data(), sqi.data(), max_update_index, &pp, &res);
    if (rc != RANDT_OK) {
      std::cout << "pose graph optimization failed: " << randt_last_error(ctx_->get()) << std::endl;
      return;
    }
    std::cout << "Ceres-style report: iterations " << res.iterations << ", initial cost " << res.initial_cost << ", final cost "
              << res.final_cost << ", termination " << res.termination << '\n';
    for (int i = 0; i < n; ++i) {
      Pose& p = poses.at(i);
      p.pos = {x[3 * i + 0], x[3 * i + 1]};
      p.rot = x[3 * i + 2];
      p.pose = SE2d(p.rot, p.pos[0], p.pos[1]);  // Sophus::SE2d(rot, pos) (:85)
      p.cov = {p.cov_pos_pos[0], p.cov_pos_pos[1], p.cov_pos_rot[0], p.cov_pos_pos[2], p.cov_pos_pos[3], p.cov_pos_rot[1],
               p.cov_pos_rot[0], p.cov_pos_rot[1], p.cov_rot_rot};  // :86-89
    }
    n_optimized_constraints_ = static_cast<int>(edges.size());
    n_optimized_poses_ = n;
    lock.lock();
    for (int i = 0; i < n; ++i) poses_ref.at(i) = poses.at(i);
  }

 private:
  std::shared_ptr<Context> ctx_;
  int n_optimized_poses_ = 0, n_optimized_constraints_ = 0;
  GlobalFuserParameters parameters_;
};

}  // namespace randt
