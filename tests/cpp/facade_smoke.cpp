// Exercises include/randt_facade.hpp the way LocalFuser::detectLoopClosures uses the reference
// classes (local_fuser.cpp:329-339): build two NDT maps from point clouds, then
// Matcher::estimateLoopConstraint.  Exit 0: pose recovered; 3: no GPU (expected on a CPU-only box).
#include <cstdio>
#include <random>
#include <vector>

#include "randt_facade.hpp"

int main() {
  using namespace randt;
  std::shared_ptr<Context> ctx;
  try {
    ctx = std::make_shared<Context>(0);
  } catch (const std::exception& e) {
    std::printf("no device: %s\n", e.what());
    return 3;
  }
  // scene: 40 tight blobs on a ring; "submap" scan at identity, "query" scan seen from (0.3, -0.2, 0.1 rad)
  std::mt19937 rng(7);
  std::normal_distribution<float> noise(0.f, 0.03f);
  std::uniform_real_distribution<float> uni(20.f, 80.f);
  const SE2d truth(0.1, 0.3, -0.2);
  const SE2d inv = truth.inverse();
  std::vector<float> fixed_pts, moving_pts;
  for (int b = 0; b < 40; ++b) {
    const float ang = 6.2831853f * b / 40.f, rad = 5.f + 3.f * ((b * 7) % 5) / 5.f;
    const float cx = rad * std::cos(ang) + 0.25f, cy = rad * std::sin(ang) + 0.25f, I0 = uni(rng);
    for (int k = 0; k < 20; ++k) {
      const float x = cx + noise(rng), y = cy + noise(rng), in = I0 + 10.f * noise(rng);
      fixed_pts.insert(fixed_pts.end(), {x, y, 0.f, in});
      const float mx = (float)(inv.d[0] * x - inv.d[1] * y + inv.d[2]);
      const float my = (float)(inv.d[1] * x + inv.d[0] * y + inv.d[3]);
      moving_pts.insert(moving_pts.end(), {mx + noise(rng) * 0.1f, my + noise(rng) * 0.1f, 0.f, in});
    }
  }
  NDTMapParameters mp;
  RadarPreprocessorParameters rp;
  Map scan_a, scan_b, submap;
  scan_a.initialize(ctx, mp, 0.0, 0.0, 512);
  scan_b.initialize(ctx, mp, 0.0, 0.0, 512);
  submap.initialize(ctx, mp, 0.0, 0.0);
  scan_a.addScan(fixed_pts.data(), (int)fixed_pts.size() / 4, 4, 3, rp);
  scan_b.addScan(moving_pts.data(), (int)moving_pts.size() / 4, 4, 3, rp);
  submap.mergeMapCell(scan_a);  // first scan of a submap (local_fuser.cpp:293)
  std::printf("submap cells %u, scan cells %u\n", submap.get_n_cells(), scan_b.get_n_cells());
  Map copy = submap;            // value semantics like the reference
  if (copy.get_n_cells() != submap.get_n_cells()) return 1;

  Matcher matcher;
  NDTMatcherParameters prm;
  matcher.initialize(prm);
  SE2d trans(0.05, 0.15, -0.05);  // initial guess
  randt_result st{};
  const double cost = matcher.estimateLoopConstraint(trans, submap, scan_b, 2, true, 1.5, &st);
  std::printf("pose %.5f %.5f %.5f  cost %.4f  residuals %d  iterations %d\n", trans.d[2], trans.d[3], trans.angle(), cost,
              st.n_residuals, st.iterations);
  const bool ok = std::fabs(trans.d[2] - 0.3) < 0.02 && std::fabs(trans.d[3] + 0.2) < 0.02 && std::fabs(trans.angle() - 0.1) < 0.01;
  // empty moving map: warning + pose untouched
  Map empty;
  empty.initialize(ctx, mp, 0.0, 0.0, 16);
  SE2d keep(0.2, 1.0, 2.0);
  matcher.estimateLoopConstraint(keep, submap, empty, 2, true, 1.5);
  const bool kept = keep.d[2] == 1.0 && keep.d[3] == 2.0;
  return (ok && kept) ? 0 : 2;
}
