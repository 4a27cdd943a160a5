// randt_local_fuser.hpp -- the reference's front end as a reusable C++ class on top of randt_facade.hpp (round 5).
//
// rc::navigation::ndt::LocalFuser (include/local_fuser/local_fuser.h:46-141) orchestrates scan -> NDT -> predict -> fixed-lag
// registration -> keyframe queue -> rolling submap -> graph node, submap roll-over with overlap, loop-closure candidates and the
// current submap's origin after a pose-graph optimisation.  This header holds its DATA PATH -- processScan (local_fuser.cpp:99-300),
// initializeNewSubmap (:40-63), detectLoopClosures (:318-350, Scan Context branch), getTransform, submapComplete, and
// NDTSlam::optimizePoseGraph's call into GlobalFuser (ndt_slam.cpp:351-361) with the pose part of updateSubmaps (local_fuser.cpp:65-88)
// -- written against the facade classes exactly as the reference writes it against its own Matcher / Map / HierarchicalMap /
// SCManager / GlobalFuser: Maps by value in the same places, the reference-signature Matcher::estimateTransformCeres.  Not here: ROS
// messages / TF / timers, IMU message handling (pass the yaw increment yourself), OGM ray tracing (SURVEY: out of scope).
//
// tests/cpp/local_fuser_drive.cpp drives it (host buffers in, poses out); tests/test_gpu_local_fuser_cpp.py holds it to the Python
// harness (randt-slam_amd/odometry.py / slam.py), which tests/test_gpu_odometry.py / test_gpu_slam.py hold to the CPU oracle.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include "randt_facade.hpp"

namespace randt {

inline SE2d se2_mul(const SE2d& a, const SE2d& b) {  // Sophus SE2 product, complex re-normalised (std::hypot, like so2.hpp)
  SE2d r;
  const double re = a.d[0] * b.d[0] - a.d[1] * b.d[1], im = a.d[0] * b.d[1] + a.d[1] * b.d[0];
  const double n = std::hypot(re, im);
  r.d[0] = re / n;
  r.d[1] = im / n;
  r.d[2] = a.d[2] + a.d[0] * b.d[2] - a.d[1] * b.d[3];
  r.d[3] = a.d[3] + a.d[1] * b.d[2] + a.d[0] * b.d[3];
  return r;
}
inline SE2d se2_inv(const SE2d& a) {
  SE2d r;
  r.d[0] = a.d[0];
  r.d[1] = -a.d[1];
  r.d[2] = -(r.d[0] * a.d[2] - r.d[1] * a.d[3]);
  r.d[3] = -(r.d[1] * a.d[2] + r.d[0] * a.d[3]);
  return r;
}

// LocalFuserParameters (include/ndt_slam/ndt_slam_parameters.h:86-132), the members the data path reads; defaults = the indoor
// preset (config/parameters_indoor.yaml + config/ndt_radar_slam_base_parameters.yaml).
struct LocalFuserParameters {
  NDTMapParameters ndt_map_parameters;
  RadarPreprocessorParameters preprocessor_parameters;
  RadarFilterParameters filter_parameters;
  NDTMatcherParameters ndt_matcher_parameters;
  ScanContextParameters scan_context_parameters;
  GlobalFuserParameters global_fuser_parameters;
  int submap_size_poses = 135, submap_overlap = 20, insertion_step = 4;
  bool use_scan_context_as_loop_closure = false;      // the graph / loop-closure layer on top of the odometry
  int loop_closure_gnc_steps = 2;
  bool use_intensity_in_loop_closure = true;          // ndt_slam.cpp:614-616: defaults to ndt_matcher.use_intensity_as_dimension
  double loop_closure_scale = 1.5, loop_closure_max_cs_divergence = 3.6, loop_closure_weight = 4.0e4;
  std::array<double, 3> odometry_sqrt_information{10.0, 10.0, 50.0};   // local_fuser.cpp:203-205, :264-266 (diagonal)
  int scan_cell_capacity = 512;                        // cells a scan's NDT can hold (a 2000-point scan has <= 333)
};

// rc::navigation::ndt::LocalFuser (include/local_fuser/local_fuser.h:46-141, src/local_fuser/local_fuser.cpp), the data path:
// the members it touches under their reference names, Maps held BY VALUE exactly where the reference holds them.
class LocalFuser {
 public:
  enum Insertion { kAddScan = 0, kAddClusters = 1, kInsertClusterLoop = 2 };

  // void LocalFuser::initialize(...)                                                                  (local_fuser.cpp:10-38)
  void initialize(std::shared_ptr<Context> ctx, const LocalFuserParameters& parameters) {
    ctx_ = std::move(ctx);
    parameters_ = parameters;
    map_parameters_ = parameters.ndt_map_parameters;
    preprocessor_parameters_ = parameters.preprocessor_parameters;
    matcher_parameters_ = parameters.ndt_matcher_parameters;
    submap_size_poses_ = parameters.submap_size_poses;
    submap_overlap_ = parameters.submap_overlap;
    insertion_step_ = parameters.insertion_step;
    ndt_matcher_.initialize(matcher_parameters_);
    last_imu_bias_ = matcher_parameters_.initial_imu_bias;                                        // :36
    n_finished_submaps_ = 0;                                                                      // :37
    _preprocessor.initialize(ctx_, preprocessor_parameters_, parameters.filter_parameters);
    _current_submap.initialize(ctx_, map_parameters_, 0.0, 0.0);
    current_submap_is_empty_ = true;   // HierarchicalMap::initialize (ndt_hierarchical_map.cpp:15)
    if (parameters.use_scan_context_as_loop_closure) {
      slam_ = true;
      sc_manager_.initialize(ctx_, parameters.scan_context_parameters);
      global_fuser_.initialize(ctx_, parameters.global_fuser_parameters);
    }
  }

  SE2d getTransform() const { return se2_mul(current_global_transform_, current_transform_); }  // local_fuser.h:113-127

  // ---- the graph / loop-closure layer (use_scan_context_as_loop_closure) ---------------------------------------------------
  struct LoopLog { int query, candidate; double cs; bool accepted; };
  const std::map<int, Pose>& nodes() const { return nodes_; }
  const std::vector<LoopLog>& loopLog() const { return loop_log_; }
  const std::vector<Constraint>& edges() const { return edges_; }

  // LocalFuser::detectLoopClosures, Scan Context branch (local_fuser.cpp:318-350)
  int detectLoopClosures() {
    int added = 0;
    while (!_next_maps_to_search_loop.empty()) {
      const int q = _next_maps_to_search_loop.front();
      _next_maps_to_search_loop.pop_front();
      const std::pair<int, float> det = sc_manager_.detectLoopClosureID(q);   // :323
      const int lid = det.first;
      if (lid == -1 || submap_idzs_.at(q) == submap_idzs_.at(lid)) continue;
      const int sub_i = submap_idzs_.at(lid);
      if (!submaps_.count(sub_i)) continue;  // submaps_.at() would throw: the candidate's submap is still being built
      const SE2d root = nodes_.at(root_nodes_.at(sub_i)).pose;
      SE2d trans = se2_mul(se2_mul(se2_inv(root), nodes_.at(lid).pose), SE2d(-static_cast<double>(det.second), 0.0, 0.0));   // :333
      Map f_loop_map = submaps_.at(sub_i);   // :329  (copies: values)
      Map m_loop_map = scans_.at(q);         // :332
      ndt_matcher_.estimateLoopConstraint(trans, f_loop_map, m_loop_map, parameters_.loop_closure_gnc_steps, parameters_.use_intensity_in_loop_closure,
                                          parameters_.loop_closure_scale);   // :335
      m_loop_map.transformMap(trans);                                                     // :338
      const double cs = f_loop_map.calculateCSDivergence(m_loop_map);                     // :339
      const bool ok = cs < parameters_.loop_closure_max_cs_divergence;                    // :340 (parameters_indoor.yaml:8)
      loop_log_.push_back({q, lid, cs, ok});
      if (ok) {                                                                           // :341-347
        Constraint c;
        c.id_begin = root_nodes_.at(sub_i);
        c.id_end = q;
        c.trans = trans;
        const double w = parameters_.loop_closure_weight;
        c.sqrt_information = {w, 0, 0, 0, w, 0, 0, 0, w};                                 // loop_closure_weight * I
        edges_.push_back(c);
        ++added;
      }
    }
    return added;
  }

  // NOT in the reference (its loop registers the candidates one by one): the same search with the registrations of ALL pending
  // queries as ONE batch, sharded over the GPUs of `group` -- north_star's multi-GPU unit ("independent scan-to-submap
  // registrations ... loop-closure candidates shard across the GPUs"; Matcher::estimateLoopConstraintBatch: staging, broadcast of
  // the candidate submaps, contiguous shards, gather).  The candidates of different queries do not depend on each other (an
  // accepted edge never feeds a later registration), and the batched registrations are bit-identical to the single calls, so the
  // graph comes out exactly as from detectLoopClosures() called at the same moments.  Pays when several queries are pending --
  // a search timer slower than the keyframe rate (ndt_slam.cpp:363-365), offline replays.
  int detectLoopClosuresBatched(DeviceGroup& group, int* n_candidates = nullptr) {
    struct Candidate { int q, lid, sub_i; };
    std::vector<Candidate> cand;
    std::vector<SE2d> trans;
    while (!_next_maps_to_search_loop.empty()) {
      const int q = _next_maps_to_search_loop.front();
      _next_maps_to_search_loop.pop_front();
      const std::pair<int, float> det = sc_manager_.detectLoopClosureID(q);   // :323
      const int lid = det.first;
      if (lid == -1 || submap_idzs_.at(q) == submap_idzs_.at(lid)) continue;
      const int sub_i = submap_idzs_.at(lid);
      if (!submaps_.count(sub_i)) continue;
      const SE2d root = nodes_.at(root_nodes_.at(sub_i)).pose;
      cand.push_back({q, lid, sub_i});
      trans.push_back(se2_mul(se2_mul(se2_inv(root), nodes_.at(lid).pose), SE2d(-static_cast<double>(det.second), 0.0, 0.0)));   // :333
    }
    if (n_candidates) *n_candidates = static_cast<int>(cand.size());
    if (cand.empty()) return 0;
    std::vector<const Map*> fixed, moving;
    std::vector<int> fixed_of_pair;
    std::map<int, int> slot_of_submap;
    for (const Candidate& c : cand) {
      if (!slot_of_submap.count(c.sub_i)) {
        slot_of_submap[c.sub_i] = static_cast<int>(fixed.size());
        fixed.push_back(&submaps_.at(c.sub_i));
      }
      fixed_of_pair.push_back(slot_of_submap.at(c.sub_i));
      moving.push_back(&scans_.at(c.q));
    }
    ndt_matcher_.estimateLoopConstraintBatch(group, trans, fixed, fixed_of_pair, moving, parameters_.loop_closure_gnc_steps,
                                             parameters_.use_intensity_in_loop_closure, parameters_.loop_closure_scale);   // :335, all at once
    int added = 0;
    for (size_t p = 0; p < cand.size(); ++p) {   // the gate and the edges in query order, like the sequential loop
      const Candidate& c = cand[p];
      Map m_loop_map = scans_.at(c.q);
      m_loop_map.transformMap(trans[p]);                                                                  // :338
      const double cs = submaps_.at(c.sub_i).calculateCSDivergence(m_loop_map);                           // :339
      const bool ok = cs < parameters_.loop_closure_max_cs_divergence;
      loop_log_.push_back({c.q, c.lid, cs, ok});
      if (ok) {
        Constraint e;
        e.id_begin = root_nodes_.at(c.sub_i);
        e.id_end = c.q;
        e.trans = trans[p];
        const double w = parameters_.loop_closure_weight;
        e.sqrt_information = {w, 0, 0, 0, w, 0, 0, 0, w};
        edges_.push_back(e);
        ++added;
      }
    }
    return added;
  }

  // NDTSlam::optimizePoseGraph (ndt_slam.cpp:351-361) + the pose part of LocalFuser::updateSubmaps (local_fuser.cpp:65-88)
  void optimizePoseGraph() {
    if (nodes_.empty() || edges_.empty() || submap_idzs_.back() <= 0) return;
    const int n_nodes_per_submap = static_cast<int>(std::ceil((submap_size_poses_ - (matcher_parameters_.smoothing_steps - 1)) / static_cast<double>(insertion_step_)));
    const int max_update_index = static_cast<int>((nodes_.size() - 1) / n_nodes_per_submap) * n_nodes_per_submap;
    global_fuser_.optimizePoseGraph(nodes_, edges_, nodes_mutex_, max_update_index);
    current_global_transform_ = nodes_.at(root_nodes_.at(n_finished_submaps_)).pose;
  }
  bool submapComplete() const { return static_cast<int>(_trajectory.size()) >= submap_size_poses_; }
  int finishedSubmaps() const { return n_finished_submaps_; }

  // local_fuser.cpp:40-63
  void initializeNewSubmap(const SE2d& initial_transform) {
    _last_state = _trajectory.back();
    const SE2d old_submap_to_new_submap = se2_mul(se2_inv(current_global_transform_), initial_transform);  // :45, name and all
    if (slam_) submaps_[n_finished_submaps_] = _current_submap;                                    // :43 submaps_.insert(...)
    _last_submap_transformed = _current_submap;                                                    // :44 (a copy)
    _last_submap_transformed.transformMap(old_submap_to_new_submap);                               // :46 (index grid left stale, like there)
    _next_maps_to_insert.clear();
    _next_scans_to_insert.clear();
    ndt_matcher_.resetMatcher();                                                                   // :51 imu_constraints_ of the old submap must not feed the new one's IMU factors
    _map_window.clear();
    current_transform_ = SE2d();
    current_global_transform_ = initial_transform;
    _current_submap.clear();
    current_submap_is_empty_ = true;   // :55 _current_submap.initialize(...) sets HierarchicalMap::is_empty
    _trajectory.clear();
    ++n_finished_submaps_;
  }

  // local_fuser.cpp:99-300, data path only.  points: n_points records of `stride` floats, intensity at `intensity_index`
  // the same on a RAW polar scan (n_azimuths x n_bins points, azimuth after azimuth): RadarPreprocessor::processScan's filterScan
  // runs first (radar_preprocessor.cpp:45-125), on the device
  void processPolarScan(const float* raw, int n_azimuths, int n_bins, int stride, int intensity_index, double stamp, double imu_yaw_increment = 0.0) {
    polar_az_ = n_azimuths;
    polar_bins_ = n_bins;
    processScan(raw, n_azimuths * n_bins, stride, intensity_index, stamp, kAddScan, imu_yaw_increment);
    polar_az_ = polar_bins_ = 0;
  }
  // imu_yaw_increment: the heading change since the last scan from the IMU (what the reference extracts from the two orientation
  // quaternions, local_fuser.cpp:107-121; used when ndt_matcher_parameters.use_imu is set), 0 otherwise
  void processScan(const float* points, int n_points, int stride, int intensity_index, double stamp, int cluster_by_cluster = kAddScan,
                   double imu_yaw_increment = 0.0) {
    yaw_ = imu_yaw_increment;
    HierarchicalMap current_scan;  // :103-105
    current_scan.initialize(ctx_, map_parameters_, 0.0, 0.0, parameters_.scan_cell_capacity);
    if (polar_az_ > 0 && slam_) {
      // the loop search wants the FILTERED cloud on the host (SCManager keys of a keyframe, local_fuser.cpp:207): filterScan's
      // outputs come back once, the scan's NDT is built from them
      std::vector<std::pair<double, double>> polar_detections;
      std::vector<std::array<double, 3>> peak_detections;
      if (_preprocessor.filterScan(points, polar_az_, polar_bins_, stride, intensity_index, filtered_, polar_detections, peak_detections)) {
        points = filtered_.data();
        n_points = static_cast<int>(filtered_.size() / 4);
        stride = 4;
        intensity_index = 3;
        current_scan.addScan(points, n_points, stride, intensity_index, preprocessor_parameters_);
      }
    } else if (polar_az_ > 0) {
      _preprocessor.processScan(points, polar_az_, polar_bins_, stride, intensity_index, current_scan.getMap());  // :102 filterScan + clustering + NDT
    } else if (cluster_by_cluster) {
      // RadarPreprocessor::processScan's clustering on the host, like the reference: Grid::cluster (grid.cpp:7-14) ...
      const int row_size = static_cast<int>(std::sqrt(static_cast<double>(preprocessor_parameters_.n_clusters)));
      const float resolution = static_cast<float>(preprocessor_parameters_.max_range) * 2 / (row_size);
      std::vector<int> labels(static_cast<size_t>(n_points));
      for (int i = 0; i < n_points; ++i)
        labels[i] = static_cast<int>(points[static_cast<size_t>(i) * stride] / resolution) +
                    row_size * static_cast<int>(points[static_cast<size_t>(i) * stride + 1] / resolution);
      // ... and ClusterGenerator::labelClouds (radar_preprocessor.cpp:151-169): clusters in ascending label order, points in cloud order
      std::vector<int> sorted = labels;
      std::sort(sorted.begin(), sorted.end());
      sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
      std::map<int, int> dense;
      for (size_t c = 0; c < sorted.size(); ++c) dense[sorted[c]] = static_cast<int>(c);
      std::vector<int> size(sorted.size(), 0), offsets(sorted.size() + 1, 0);
      for (int i = 0; i < n_points; ++i) ++size[dense[labels[i]]];
      for (size_t c = 0; c < sorted.size(); ++c) offsets[c + 1] = offsets[c] + size[c];
      std::vector<int> at(offsets.begin(), offsets.end() - 1);
      clustered_.resize(static_cast<size_t>(n_points) * stride);
      for (int i = 0; i < n_points; ++i) {
        const int c = dense[labels[i]];
        std::memcpy(&clustered_[static_cast<size_t>(at[c]++) * stride], points + static_cast<size_t>(i) * stride, sizeof(float) * stride);
      }
      if (cluster_by_cluster > 1) {  // the loop spelled out: one Map::insertCluster call per cluster (ndt_hierarchical_map.cpp:29-32)
        for (size_t c = 0; c + 1 < offsets.size(); ++c)
          current_scan.getMap().insertCluster(clustered_.data() + static_cast<size_t>(offsets[c]) * stride, offsets[c + 1] - offsets[c], stride, intensity_index);
      } else {
        current_scan.addClusters(clustered_.data(), offsets, stride, intensity_index);  // HierarchicalMap::addClusters: the whole list in one call
      }
    } else {
      current_scan.addScan(points, n_points, stride, intensity_index, preprocessor_parameters_);  // clustering + NDT of the scan in one call
    }
    const Map& scan_ndt = current_scan.getMap();
    cur_points_ = points;
    cur_n_ = n_points;
    cur_stride_ = stride;
    cur_ioff_ = intensity_index;
    process(scan_ndt, stamp);
    if (submapComplete()) {  // ndt_slam.cpp:211-223
      initializeNewSubmap(getTransform());
      process(scan_ndt, stamp);
    }
  }

 private:
  void process(const Map& scan_ndt, double stamp) {
    // :108 `!_current_submap.isEmpty()` is HierarchicalMap's FLAG (ndt_hierarchical_map.h:85-87): false from the first mergeMapCell
    // on, whatever was merged -- a submap whose first scan produced no cell still counts as started
    if (!current_submap_is_empty_) {
      ndt_matcher_.predictTransform(yaw_, stamp, _trajectory);  // :125
      // every copy the reference makes is made here (Maps by value, local_fuser.cpp:128-136)
      Map fmap = _current_submap;                              // :128  Map fmap = _current_submap.getMap();
      Map mmap = scan_ndt;                                     // :129  Map mmap = current_scan.getMap();
      _map_window.push_back(mmap);                             // :130
      std::deque<Map> fixed_ndts;                              // :131-136
      fixed_ndts.push_back(fmap);
      if (static_cast<int>(_trajectory.size()) < submap_overlap_ && n_finished_submaps_ > 0) {
        Map old_fmap = _last_submap_transformed;               // :134
        fixed_ndts.push_back(old_fmap);
      }
      ndt_matcher_.estimateTransformCeres(current_transform_, _trajectory, yaw_, stamp, fixed_ndts, _map_window);  // :139
      const int n = static_cast<int>(_trajectory.size());
      if (static_cast<int>(_map_window.size()) >= matcher_parameters_.smoothing_steps) _map_window.pop_front();  // :152-154
      if (n % insertion_step_ == 0) {                                                                             // :155-161
        _next_maps_to_insert.push_back(scan_ndt);
        if (slam_) _next_scans_to_insert.emplace_back(cur_points_, cur_points_ + static_cast<size_t>(cur_n_) * cur_stride_);
      }
      const int insertion_delay = matcher_parameters_.smoothing_steps + 1;                                       // ndt_slam.cpp:580
      if (n >= insertion_delay + insertion_step_ && (n - insertion_delay) % insertion_step_ == 0) {              // :164
        const SE2d smoothed = _trajectory.end()[-insertion_delay - 1].pose;                                       // :165-166
        Map smoothed_map = _next_maps_to_insert.front();   // :173 (unused there as well)
        Map global_map = _current_submap;                  // :174 (unused there as well)
        _last_scan_kept = _next_maps_to_insert.front();    // :176 scans_[current_node_id_] = ... "before transforming"
        if (slam_) {                                        // :192-222 node + odometry edge, :207 Scan Context keys
          const int nid = addNode(se2_mul(current_global_transform_, smoothed), _next_maps_to_insert.front(), _next_scans_to_insert.front());
          _next_scans_to_insert.pop_front();
          _next_maps_to_search_loop.push_back(nid);
        }
        _next_maps_to_insert.front().transformMap(smoothed);   // :177
        _last_merged_map = _next_maps_to_insert.front();       // :178
        _current_submap.mergeMapCell(_next_maps_to_insert.front());  // :190
        current_submap_is_empty_ = false;
        _next_maps_to_insert.pop_front();                      // :223
      }
    } else {
      // first scan of the submap (:225-295)
      State st;
      st.pose = current_transform_;
      st.pos = {current_transform_.d[2], current_transform_.d[3]};
      st.rot = current_transform_.angle();
      if (n_finished_submaps_ == 0) {          // :231-236 (velocities and acceleration zero: State's defaults)
        st.imu_bias = last_imu_bias_;          // = ndt_matcher_parameters.initial_imu_bias (:36)
      } else {                                 // :237-242
        st.lin_vel = _last_state.lin_vel;
        st.rot_vel = _last_state.rot_vel;
        st.lin_acc = _last_state.lin_acc;
        st.imu_bias = _last_state.imu_bias;
      }
      st.stamp = stamp;
      _trajectory.push_back(st);
      if (slam_) {                             // :247-279 root node of the submap
        const int nid = addNode(current_global_transform_, scan_ndt, std::vector<float>(cur_points_, cur_points_ + static_cast<size_t>(cur_n_) * cur_stride_));
        root_nodes_[n_finished_submaps_] = nid;
      }
      Map first = scan_ndt;
      first.transformMap(current_transform_);  // :281
      _current_submap.mergeMapCell(first);     // :293
      current_submap_is_empty_ = false;        // ndt_hierarchical_map.cpp:71
    }
  }

  int addNode(const SE2d& pose, const Map& scan, const std::vector<float>& points) {
    const int nid = static_cast<int>(nodes_.size());
    Pose p;
    p.pose = pose;
    p.pos = {pose.d[2], pose.d[3]};
    p.rot = pose.angle();
    if (nid > 0) {                                       // :199-205, :258-267
      Constraint c;
      c.id_begin = nid - 1;
      c.id_end = nid;
      c.trans = se2_mul(se2_inv(nodes_.at(nid - 1).pose), pose);
      c.sqrt_information = {parameters_.odometry_sqrt_information[0], 0, 0, 0, parameters_.odometry_sqrt_information[1], 0, 0, 0,
                            parameters_.odometry_sqrt_information[2]};   // :203-205
      edges_.push_back(c);
      p.traversed_dist = nodes_.at(nid - 1).traversed_dist + std::hypot(c.trans.d[2], c.trans.d[3]);
    }
    nodes_[nid] = p;
    submap_idzs_.push_back(n_finished_submaps_);
    scans_[nid] = scan;                                  // kept alive for loop registration (a value: shares the storage)
    sc_manager_.makeAndSaveScancontextAndKeys(points.data(), static_cast<int>(points.size()) / cur_stride_, cur_stride_, cur_ioff_, {pose.d[2], pose.d[3]},
                                              p.traversed_dist);   // :207, :281
    return nid;
  }

  std::shared_ptr<Context> ctx_;
  LocalFuserParameters parameters_;
  bool slam_ = false;
  SCManager sc_manager_;
  GlobalFuser global_fuser_;
  std::mutex nodes_mutex_;
  std::map<int, Pose> nodes_;
  std::vector<Constraint> edges_;
  std::vector<int> submap_idzs_;
  std::map<int, int> root_nodes_;
  std::map<int, Map> scans_, submaps_;
  std::deque<int> _next_maps_to_search_loop;
  std::deque<std::vector<float>> _next_scans_to_insert;
  std::vector<LoopLog> loop_log_;
  double yaw_ = 0.0;
  const float* cur_points_ = nullptr;
  int cur_n_ = 0, cur_stride_ = 4, cur_ioff_ = 3;
  NDTMapParameters map_parameters_;                    // indoor preset
  RadarPreprocessorParameters preprocessor_parameters_;
  NDTMatcherParameters matcher_parameters_;
  Matcher ndt_matcher_;
  RadarPreprocessor _preprocessor;
  int polar_az_ = 0, polar_bins_ = 0;
  Map _current_submap, _last_submap_transformed, _last_scan_kept, _last_merged_map;
  std::deque<Map> _map_window, _next_maps_to_insert;
  std::vector<float> clustered_, filtered_;
  std::vector<State> _trajectory;
  State _last_state;
  SE2d current_transform_, current_global_transform_;
  bool current_submap_is_empty_ = true;                // HierarchicalMap::is_empty of _current_submap
  int submap_size_poses_ = 135, submap_overlap_ = 20, insertion_step_ = 4, n_finished_submaps_ = 0;
  double last_imu_bias_ = 0.0;
};

}  // namespace randt
