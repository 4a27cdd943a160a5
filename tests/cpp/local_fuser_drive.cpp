// A C++ caller of the facade with the call pattern of the reference's front end -- LocalFuser::processScan /
// initializeNewSubmap (src/local_fuser/local_fuser.cpp:40-63, 99-300) and NDTSlam::radarCb's submap roll-over
// (src/ndt_slam/ndt_slam.cpp:211-223), data path only -- written against include/randt_facade.hpp exactly as a maintainer
// would write it against the reference's own Matcher / Map classes: Maps held by value in deques, the reference-signature
// Matcher::estimateTransformCeres(trans, trajectory, angle, stamp, fixed_ndts, moving_ndts), Map::mergeMapCell after
// Map::transformMap, Matcher::predictTransform.  (round-3 verdict, missing 5: the host side of north_star is C++.)
//
//   local_fuser_drive <scans.bin> <poses.txt> [submap_size_poses submap_overlap]
//   scans.bin: int32 n_scans, int32 n_points, float32 [n_scans][n_points][4] (x y z intensity), stamps = 0.25 s apart
//   poses.txt: one line per scan, the global pose [cos sin tx ty] with 17 significant digits
// tests/test_gpu_local_fuser_cpp.py runs it beside the Python harness (randt-slam_amd/odometry.py) on the same drive.
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
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
  bool submapComplete() const { return static_cast<int>(_trajectory.size()) >= submap_size_poses_; }
  int finishedSubmaps() const { return n_finished_submaps_; }

  // local_fuser.cpp:40-63
  void initializeNewSubmap(const SE2d& initial_transform) {
    _last_state = _trajectory.back();
    const SE2d old_submap_to_new_submap = mul(inv(current_global_transform_), initial_transform);  // :45, name and all
    _last_submap_transformed = _current_submap;                                                    // :44 (a copy)
    _last_submap_transformed.transformMap(old_submap_to_new_submap);                               // :46 (index grid left stale, like there)
    _next_maps_to_insert.clear();
    _map_window.clear();
    current_transform_ = SE2d();
    current_global_transform_ = initial_transform;
    _current_submap.clear();
    _trajectory.clear();
    ++n_finished_submaps_;
  }

  // local_fuser.cpp:99-300, data path only
  void processScan(const float* points, int n_points, double stamp) {
    Map scan_ndt;
    scan_ndt.initialize(ctx_, map_parameters_, 0.0, 0.0, 512);
    scan_ndt.addScan(points, n_points, 4, 3, preprocessor_parameters_);  // :102-105 clustering + NDT of the scan
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
      _map_window.push_back(scan_ndt);                         // :130
      std::deque<Map> fixed_ndts;                              // :128-136
      fixed_ndts.push_back(_current_submap);
      if (static_cast<int>(_trajectory.size()) < submap_overlap_ && n_finished_submaps_ > 0) fixed_ndts.push_back(_last_submap_transformed);
      ndt_matcher_.estimateTransformCeres(current_transform_, _trajectory, 0.0, stamp, fixed_ndts, _map_window);  // :139
      const int n = static_cast<int>(_trajectory.size());
      if (static_cast<int>(_map_window.size()) >= matcher_parameters_.smoothing_steps) _map_window.pop_front();  // :152-154
      if (n % insertion_step_ == 0) _next_maps_to_insert.push_back(scan_ndt);                                     // :155-161
      const int insertion_delay = matcher_parameters_.smoothing_steps + 1;                                       // ndt_slam.cpp:580
      if (n >= insertion_delay + insertion_step_ && (n - insertion_delay) % insertion_step_ == 0) {              // :164
        const SE2d smoothed = _trajectory.end()[-insertion_delay - 1].pose;                                       // :165-166
        Map kf = _next_maps_to_insert.front();
        _next_maps_to_insert.pop_front();
        kf.transformMap(smoothed);           // :177
        _current_submap.mergeMapCell(kf);    // :190
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
      Map first = scan_ndt;
      first.transformMap(current_transform_);  // :281
      _current_submap.mergeMapCell(first);     // :293
    }
  }

  std::shared_ptr<Context> ctx_;
  NDTMapParameters map_parameters_;                    // indoor preset
  RadarPreprocessorParameters preprocessor_parameters_;
  NDTMatcherParameters matcher_parameters_;
  Matcher ndt_matcher_;
  Map _current_submap, _last_submap_transformed;
  std::deque<Map> _map_window, _next_maps_to_insert;
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
  const int size_poses = argc > 3 ? std::atoi(argv[3]) : 135, overlap = argc > 4 ? std::atoi(argv[4]) : 20;

  auto ctx = std::make_shared<Context>(0);
  if (last_status() != RANDT_OK) {
    std::printf("no HIP device: the drive cannot run (there is no CPU fallback)\n");
    return 3;
  }
  LocalFuser fuser(ctx, size_poses, overlap);
  std::FILE* out = std::fopen(argv[2], "w");
  if (!out) return 2;
  for (int i = 0; i < n_scans; ++i) {
    fuser.processScan(scans.data() + static_cast<size_t>(i) * n_points * 4, n_points, 0.25 * i);
    const SE2d p = fuser.getTransform();
    std::fprintf(out, "%.17g %.17g %.17g %.17g\n", p.d[0], p.d[1], p.d[2], p.d[3]);
  }
  std::fclose(out);
  std::printf("drive of %d scans done: %d submaps finished, first error status %d\n", n_scans, fuser.finishedSubmaps(), first_error());
  return first_error() == RANDT_OK ? 0 : 1;
}
