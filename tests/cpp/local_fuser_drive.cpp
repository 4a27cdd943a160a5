// A C++ caller of the facade with the call pattern of the reference's front end -- LocalFuser::processScan /
// initializeNewSubmap (src/local_fuser/local_fuser.cpp:40-63, 99-300) and NDTSlam::radarCb's submap roll-over
// (src/ndt_slam/ndt_slam.cpp:211-223), data path only -- written against include/randt_facade.hpp exactly as a maintainer
// would write it against the reference's own Matcher / Map classes: Maps held by value in deques, the reference-signature
// Matcher::estimateTransformCeres(trans, trajectory, angle, stamp, fixed_ndts, moving_ndts), Map::mergeMapCell after
// Map::transformMap, Matcher::predictTransform.  (round-3 verdict, missing 5: the host side of north_star is C++.)
//
//   local_fuser_drive <scans.bin> <poses.txt> [submap_size_poses submap_overlap] [--xyzi8] [--clusters] [--timing WARM]
//   scans.bin: int32 n_scans, int32 n_points, float32 [n_scans][n_points][4] (x y z intensity), stamps = 0.25 s apart
//   poses.txt: one line per scan, the global pose [cos sin tx ty] with 17 significant digits
//   --xyzi8     hand the scans over as pcl::PointXYZI records (32 bytes: x y z pad intensity pad pad pad), the reference's own
//               host layout, instead of packed x y z I
//   --clusters  the reference's own insertion: Grid::cluster + labelClouds on the host (grid.cpp:7-14,
//               radar_preprocessor.cpp:151-169), then HierarchicalMap::addClusters (ndt_hierarchical_map.cpp:28-33) on the cluster
//               list -- instead of the whole scan in one call (Map::addScan); --cluster-loop: the same with one
//               Map::insertCluster call per cluster
//   --polar A B every "scan" of scans.bin is a RAW POLAR scan of A azimuths x B bins (n_points = A * B, BASELINE config 5):
//               RadarPreprocessor::processScan -- filterScan (radar_preprocessor.cpp:45-125) + clustering + NDT -- runs on the device
//               from the host buffer (RadarPreprocessor facade, randt_filter_build)
//   --slam F    the whole SLAM loop (row f-4), like randt-slam_amd/slam.py: graph nodes / odometry edges (local_fuser.cpp:192-222,
//               247-279), SCManager keys per node, LocalFuser::detectLoopClosures after every scan (:318-350: Scan Context
//               candidate -> Matcher::estimateLoopConstraint against the candidate's finished submap -> Map::transformMap +
//               calculateCSDivergence gate -> loop edge), GlobalFuser::optimizePoseGraph every 40 scans (ndt_slam.cpp:351-361) and
//               the current submap's origin following its root node (local_fuser.cpp:78-79).  Writes the graph to file F:
//               "node x y rot" per node, "loop query candidate cs accepted" per checked candidate.
//   --timing W  after W untimed scans: wall time per scan and the context's allocator / synchronisation counters per scan
//               (randt_ctx_pool_stats) over the rest of the drive, as one JSON line on stdout (bench.py: cpp_local_fuser_drive)
// tests/test_gpu_local_fuser_cpp.py runs it beside the Python harness (randt-slam_amd/odometry.py) on the same drive.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "randt_facade.hpp"

using namespace randt;

namespace {

SE2d mul(const SE2d& a, const SE2d& b) {  // Sophus SE2 product, complex re-normalised
  SE2d r;
  const double re = a.d[0] * b.d[0] - a.d[1] * b.d[1], im = a.d[0] * b.d[1] + a.d[1] * b.d[0];
  const double n = std::hypot(re, im);
  r.d[0] = re / n;
  r.d[1] = im / n;
  r.d[2] = a.d[2] + a.d[0] * b.d[2] - a.d[1] * b.d[3];
  r.d[3] = a.d[3] + a.d[1] * b.d[2] + a.d[0] * b.d[3];
  return r;
}
SE2d inv(const SE2d& a) {
  SE2d r;
  r.d[0] = a.d[0];
  r.d[1] = -a.d[1];
  r.d[2] = -(r.d[0] * a.d[2] - r.d[1] * a.d[3]);
  r.d[3] = -(r.d[1] * a.d[2] + r.d[0] * a.d[3]);
  return r;
}

// the members of LocalFuser the data path touches, under their reference names
class LocalFuser {
 public:
  LocalFuser(std::shared_ptr<Context> ctx, int submap_size_poses, int submap_overlap)
      : ctx_(std::move(ctx)), submap_size_poses_(submap_size_poses), submap_overlap_(submap_overlap) {
    ndt_matcher_.initialize(matcher_parameters_);
    _current_submap.initialize(ctx_, map_parameters_, 0.0, 0.0);
  }

  SE2d getTransform() const { return mul(current_global_transform_, current_transform_); }  // local_fuser.h:113-127

  // ---- the SLAM layer (--slam) --------------------------------------------------------------------------------------------
  struct LoopLog { int query, candidate; double cs; bool accepted; };
  void enableSlam() {
    slam_ = true;
    ScanContextParameters sp;
    sp.PC_MAX_RADIUS = 20.0;   // the drive of bench.py's slam_loop / tests/test_gpu_slam.py
    sp.SC_DIST_THRES = 0.5;
    sc_manager_.initialize(ctx_, sp);
    global_fuser_.initialize(ctx_, GlobalFuserParameters());
  }
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
      SE2d trans = mul(mul(inv(root), nodes_.at(lid).pose), SE2d(-static_cast<double>(det.second), 0.0, 0.0));   // :333
      Map f_loop_map = submaps_.at(sub_i);   // :329  (copies: values)
      Map m_loop_map = scans_.at(q);         // :332
      ndt_matcher_.estimateLoopConstraint(trans, f_loop_map, m_loop_map, 2, true, 1.5);   // :335 (loop_closure_gnc_steps, loop_closure_scale)
      m_loop_map.transformMap(trans);                                                     // :338
      const double cs = f_loop_map.calculateCSDivergence(m_loop_map);                     // :339
      const bool ok = cs < 3.6;                                                           // loop_closure_max_cs_divergence (parameters_indoor.yaml:8)
      loop_log_.push_back({q, lid, cs, ok});
      if (ok) {                                                                           // :341-347
        Constraint c;
        c.id_begin = root_nodes_.at(sub_i);
        c.id_end = q;
        c.trans = trans;
        c.sqrt_information = {40.0, 0, 0, 0, 40.0, 0, 0, 0, 40.0};                        // loop_closure_weight * I (the drive's 40)
        edges_.push_back(c);
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
    const SE2d old_submap_to_new_submap = mul(inv(current_global_transform_), initial_transform);  // :45, name and all
    if (slam_) submaps_[n_finished_submaps_] = _current_submap;                                    // :43 submaps_.insert(...)
    _last_submap_transformed = _current_submap;                                                    // :44 (a copy)
    _last_submap_transformed.transformMap(old_submap_to_new_submap);                               // :46 (index grid left stale, like there)
    _next_maps_to_insert.clear();
    _next_scans_to_insert.clear();
    _map_window.clear();
    current_transform_ = SE2d();
    current_global_transform_ = initial_transform;
    _current_submap.clear();
    _trajectory.clear();
    ++n_finished_submaps_;
  }

  // local_fuser.cpp:99-300, data path only.  points: n_points records of `stride` floats, intensity at `intensity_index`
  void setPolar(int n_azimuths, int n_bins) {
    polar_az_ = n_azimuths;
    polar_bins_ = n_bins;
    _preprocessor.initialize(ctx_, preprocessor_parameters_, RadarFilterParameters());
  }
  void processScan(const float* points, int n_points, int stride, int intensity_index, int cluster_by_cluster, double stamp) {
    HierarchicalMap current_scan;  // :103-105
    current_scan.initialize(ctx_, map_parameters_, 0.0, 0.0, 512);
    if (polar_az_ > 0) {
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
    if (!_current_submap.isEmpty()) {  // :123
      ndt_matcher_.predictTransform(0.0, stamp, _trajectory);  // :125
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
      ndt_matcher_.estimateTransformCeres(current_transform_, _trajectory, 0.0, stamp, fixed_ndts, _map_window);  // :139
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
          const int nid = addNode(mul(current_global_transform_, smoothed), _next_maps_to_insert.front(), _next_scans_to_insert.front());
          _next_scans_to_insert.pop_front();
          _next_maps_to_search_loop.push_back(nid);
        }
        _next_maps_to_insert.front().transformMap(smoothed);   // :177
        _last_merged_map = _next_maps_to_insert.front();       // :178
        _current_submap.mergeMapCell(_next_maps_to_insert.front());  // :190
        _next_maps_to_insert.pop_front();                      // :223
      }
    } else {
      // first scan of the submap (:225-295)
      State st;
      st.pose = current_transform_;
      st.pos = {current_transform_.d[2], current_transform_.d[3]};
      st.rot = current_transform_.angle();
      if (n_finished_submaps_ > 0) {
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
      c.trans = mul(inv(nodes_.at(nid - 1).pose), pose);
      c.sqrt_information = {10.0, 0, 0, 0, 10.0, 0, 0, 0, 50.0};   // :203-205
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
  std::vector<float> clustered_;
  std::vector<State> _trajectory;
  State _last_state;
  SE2d current_transform_, current_global_transform_;
  int submap_size_poses_, submap_overlap_, insertion_step_ = 4, n_finished_submaps_ = 0;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s scans.bin poses.txt [submap_size_poses submap_overlap]\n", argv[0]);
    return 2;
  }
  std::ifstream in(argv[1], std::ios::binary);
  int32_t n_scans = 0, n_points = 0;
  in.read(reinterpret_cast<char*>(&n_scans), 4);
  in.read(reinterpret_cast<char*>(&n_points), 4);
  if (!in || n_scans <= 0 || n_points <= 0) {
    std::fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  std::vector<float> scans(static_cast<size_t>(n_scans) * n_points * 4);
  in.read(reinterpret_cast<char*>(scans.data()), static_cast<std::streamsize>(scans.size() * sizeof(float)));
  int size_poses = 135, overlap = 20, n_pos = 0, warm = -1;
  bool xyzi8 = false;
  std::string slam_file;
  int polar_az = 0, polar_bins = 0;
  int clusters = 0;  // 1: HierarchicalMap::addClusters (the list in one call), 2: one Map::insertCluster call per cluster
  for (int a = 3; a < argc; ++a) {
    const std::string arg = argv[a];
    if (arg == "--xyzi8") xyzi8 = true;
    else if (arg == "--clusters") clusters = 1;
    else if (arg == "--cluster-loop") clusters = 2;
    else if (arg == "--timing" && a + 1 < argc) warm = std::atoi(argv[++a]);
    else if (arg == "--slam" && a + 1 < argc) slam_file = argv[++a];
    else if (arg == "--polar" && a + 2 < argc) { polar_az = std::atoi(argv[++a]); polar_bins = std::atoi(argv[++a]); }
    else if (n_pos == 0) { size_poses = std::atoi(argv[a]); ++n_pos; }
    else if (n_pos == 1) { overlap = std::atoi(argv[a]); ++n_pos; }
  }
  int stride = 4, ioff = 3;
  if (xyzi8) {  // pcl::PointXYZI: x y z 1.0f | intensity pad pad pad
    std::vector<float> wide(static_cast<size_t>(n_scans) * n_points * 8, 0.f);
    for (size_t i = 0; i < static_cast<size_t>(n_scans) * n_points; ++i) {
      wide[8 * i + 0] = scans[4 * i + 0];
      wide[8 * i + 1] = scans[4 * i + 1];
      wide[8 * i + 2] = scans[4 * i + 2];
      wide[8 * i + 3] = 1.f;
      wide[8 * i + 4] = scans[4 * i + 3];
    }
    scans.swap(wide);
    stride = 8;
    ioff = 4;
  }

  auto ctx = std::make_shared<Context>(0);
  if (last_status() != RANDT_OK) {
    std::printf("no HIP device: the drive cannot run (there is no CPU fallback)\n");
    return 3;
  }
  LocalFuser fuser(ctx, size_poses, overlap);
  if (!slam_file.empty()) fuser.enableSlam();
  if (polar_az > 0) {
    if (polar_az * polar_bins != n_points) {
      std::fprintf(stderr, "--polar %d %d does not match %d points per scan\n", polar_az, polar_bins, n_points);
      return 2;
    }
    fuser.setPolar(polar_az, polar_bins);
  }
  std::FILE* out = std::fopen(argv[2], "w");
  if (!out) return 2;
  randt_pool_stats s0{}, s1{};
  std::chrono::steady_clock::time_point t0;
  for (int i = 0; i < n_scans; ++i) {
    if (i == warm) {
      randt_ctx_synchronize(ctx->get());
      randt_ctx_pool_stats(ctx->get(), &s0);
      t0 = std::chrono::steady_clock::now();
    }
    fuser.processScan(scans.data() + static_cast<size_t>(i) * n_points * stride, n_points, stride, ioff, clusters, 0.25 * i);
    if (!slam_file.empty()) {
      fuser.detectLoopClosures();
      if (i % 40 == 39) fuser.optimizePoseGraph();
    }
    const SE2d p = fuser.getTransform();
    std::fprintf(out, "%.17g %.17g %.17g %.17g\n", p.d[0], p.d[1], p.d[2], p.d[3]);
  }
  if (!slam_file.empty()) {
    std::FILE* g = std::fopen(slam_file.c_str(), "w");
    if (!g) return 2;
    for (const auto& kv : fuser.nodes()) std::fprintf(g, "node %.17g %.17g %.17g\n", kv.second.pos[0], kv.second.pos[1], kv.second.rot);
    for (const auto& l : fuser.loopLog()) std::fprintf(g, "loop %d %d %.17g %d\n", l.query, l.candidate, l.cs, l.accepted ? 1 : 0);
    for (const auto& e : fuser.edges()) std::fprintf(g, "edge %d %d %.17g %.17g %.17g\n", e.id_begin, e.id_end, e.trans.d[2], e.trans.d[3], e.trans.angle());
    std::fclose(g);
  }
  if (warm >= 0 && warm < n_scans) {
    randt_ctx_synchronize(ctx->get());
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    randt_ctx_pool_stats(ctx->get(), &s1);
    const double n = n_scans - warm;
    std::printf("{\"scans\": %d, \"warm_up_scans\": %d, \"ms_per_scan\": %.6f, \"scans_per_sec\": %.3f, \"device_allocs_per_scan\": %.4f, "
                "\"device_frees_per_scan\": %.4f, \"stream_syncs_per_scan\": %.4f, \"pool_hits_per_scan\": %.3f, \"pool_bytes\": %lld, "
                "\"pool_blocks\": %lld, \"insertion\": \"%s\", \"point_layout\": \"%s\", \"submaps_finished\": %d}\n",
                n_scans - warm, warm, el / n * 1e3, n / el, (s1.device_allocs - s0.device_allocs) / n, (s1.device_frees - s0.device_frees) / n,
                (s1.stream_syncs - s0.stream_syncs - 1) / n /* the closing synchronisation of this measurement */, (s1.pool_hits - s0.pool_hits) / n,
                static_cast<long long>(s1.pool_bytes), static_cast<long long>(s1.pool_blocks), clusters == 2 ? "one Map::insertCluster call per cluster" : (clusters ? "HierarchicalMap::addClusters (host clustering, the list in one call)" : "addScan"),
                xyzi8 ? "pcl::PointXYZI, 32 B" : "packed x y z I, 16 B", fuser.finishedSubmaps());
  }
  std::fclose(out);
  std::printf("drive of %d scans done: %d submaps finished, first error status %d\n", n_scans, fuser.finishedSubmaps(), first_error());
  return first_error() == RANDT_OK ? 0 : 1;
}
