// Optional: the pair registration of Matcher::estimateLoopConstraint (src/ndt_registration/ndt_matcher.cpp:426-493 of the
// reference) driven by a REAL ceres::Solve, to measure the gap between true Ceres and the oracle's restatement of its
// trust-region loop.  Test infrastructure only; built only where find_package(Ceres) succeeds.
//
// What is wired exactly like the reference: one scalar residual block per correspondence,
// r = sqrt(d^T (R Sm R^T + Sf)^-1 d) with Jets (ceres_residuals.h:520-552), the shared LossFunctionWrapper around
// ScaledLoss(BarronLoss) (ndt_matcher.cpp:448,479-480; ceres_loss_functions.cpp:19-39), DENSE_QR + LEVENBERG_MARQUARDT,
// max_num_iterations, the GNC loop (:466-483).  Parameterisation: "manifold" = one 4-parameter block [cos, sin, tx, ty]
// with the SE(2) manifold (Plus(x, d) = x * exp(d), Sophus 1.22.10 restated here because Sophus is not assumed present);
// "ambient4" = the same block WITHOUT a manifold -- what the reference really optimises.
//
// Input (text, written by check.py): params, initial pose, correspondences with their 18 constants each.
// Output: final pose, cost, number of iterations per solve, per-iteration costs.
#include <ceres/ceres.h>

#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

namespace {

struct Pair {
  double mm[3], Sm[6], fm[3], Sf[6];  // mean xyz + upper covariance (xx xy xi yy yi ii) of the moving / fixed cell
};

// BarronLoss(a, alpha) with b = mu a^2 (ceres_loss_functions.h:27-35, .cpp:19-39)
class BarronLoss : public ceres::LossFunction {
 public:
  BarronLoss(double a, double alpha, double mu) : alpha_(alpha), b_(mu * a * a), c_(1.0 / b_) {
    factor_ = std::abs(alpha_ - 2.0);
    exponent_ = 0.5 * alpha_;
    pre_ = b_ * factor_ / alpha_;
    ts_ = 2.0 * c_ / factor_;
  }
  void Evaluate(double s, double rho[3]) const override {
    if (alpha_ >= 2.0) {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    } else if (std::abs(alpha_) <= 0.05) {
      const double sum = 1.0 + s * c_, inv = 1.0 / sum;
      rho[0] = b_ * std::log(sum);
      rho[1] = std::max(std::numeric_limits<double>::min(), inv);
      rho[2] = -c_ * (inv * inv);
    } else {
      const double u = s * ts_ + 1.0;
      rho[0] = pre_ * (std::pow(u, exponent_) - 1.0);
      rho[1] = pre_ * exponent_ * std::pow(u, exponent_ - 1.0) * ts_;
      rho[2] = pre_ * exponent_ * (exponent_ - 1.0) * std::pow(u, exponent_ - 2.0) * ts_ * ts_;
    }
  }

 private:
  double alpha_, b_, c_, factor_, exponent_, pre_, ts_;
};

// NDTFrameToMap{,Intensity}FactorResidualSE2 (ceres_residuals.h:454-552): D = 2 or 3
template <int D>
struct NdtResidual {
  explicit NdtResidual(const Pair& p) : p_(p) {}
  template <typename T>
  bool operator()(const T* const x, T* residual) const {
    const T theta = ceres::atan2(x[1], x[0]);
    const T c = ceres::cos(theta), s = ceres::sin(theta);
    const T R[3][3] = {{c, -s, T(0)}, {s, c, T(0)}, {T(0), T(0), T(1)}};
    const double* a = p_.Sm;
    const double* b = p_.Sf;
    const T Sm[3][3] = {{T(a[0]), T(a[1]), T(a[2])}, {T(a[1]), T(a[3]), T(a[4])}, {T(a[2]), T(a[4]), T(a[5])}};
    T C[3][3], RS[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) RS[i][j] = R[i][0] * Sm[0][j] + R[i][1] * Sm[1][j] + R[i][2] * Sm[2][j];
    const double Sf[3][3] = {{b[0], b[1], b[2]}, {b[1], b[3], b[4]}, {b[2], b[4], b[5]}};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) C[i][j] = RS[i][0] * R[j][0] + RS[i][1] * R[j][1] + RS[i][2] * R[j][2] + T(Sf[i][j]);
    T d[3] = {R[0][0] * T(p_.mm[0]) + R[0][1] * T(p_.mm[1]) + x[2] - T(p_.fm[0]),
              R[1][0] * T(p_.mm[0]) + R[1][1] * T(p_.mm[1]) + x[3] - T(p_.fm[1]), T(p_.mm[2]) - T(p_.fm[2])};
    T q;
    if (D == 2) {
      const T det = C[0][0] * C[1][1] - C[0][1] * C[1][0];
      q = (d[0] * (C[1][1] * d[0] - C[0][1] * d[1]) + d[1] * (-C[1][0] * d[0] + C[0][0] * d[1])) / det;
    } else {
      const T k00 = C[1][1] * C[2][2] - C[1][2] * C[2][1], k01 = C[0][2] * C[2][1] - C[0][1] * C[2][2], k02 = C[0][1] * C[1][2] - C[0][2] * C[1][1];
      const T k11 = C[0][0] * C[2][2] - C[0][2] * C[2][0], k12 = C[0][2] * C[1][0] - C[0][0] * C[1][2], k22 = C[0][0] * C[1][1] - C[0][1] * C[1][0];
      const T det = C[0][0] * k00 + C[1][0] * k01 + C[2][0] * k02;
      const T q0 = k00 * d[0] + k01 * d[1] + k02 * d[2], q1 = k01 * d[0] + k11 * d[1] + k12 * d[2], q2 = k02 * d[0] + k12 * d[1] + k22 * d[2];
      q = (d[0] * q0 + d[1] * q1 + d[2] * q2) / det;
    }
    residual[0] = ceres::sqrt(q);
    return true;
  }
  Pair p_;
};

// Sophus::Manifold<SE2> (sophus/ceres_manifold.hpp, se2.hpp, so2.hpp of 1.22.10): Plus(x, d) = x * exp(d),
// PlusJacobian = Dx_this_mul_exp_x_at_0
class SE2Manifold : public ceres::Manifold {
 public:
  int AmbientSize() const override { return 4; }
  int TangentSize() const override { return 3; }
  bool Plus(const double* x, const double* d, double* xp) const override {
    const double theta = d[2];
    double c = std::cos(theta), s = std::sin(theta);
    double len = std::sqrt(c * c + s * s);
    c /= len; s /= len;
    double sbt, omcbt;
    if (std::abs(theta) < 1e-10) {
      const double tsq = theta * theta;
      sbt = 1.0 - (1.0 / 6.0) * tsq;
      omcbt = 0.5 * theta - (1.0 / 24.0) * theta * tsq;
    } else {
      sbt = s / theta;
      omcbt = (1.0 - c) / theta;
    }
    const double ex = sbt * d[0] - omcbt * d[1], ey = omcbt * d[0] + sbt * d[1];
    double re = x[0] * c - x[1] * s, im = x[0] * s + x[1] * c;
    const double sq = re * re + im * im;
    if (sq != 1.0) {
      const double scale = 2.0 / (1.0 + sq);
      re *= scale; im *= scale;
    }
    len = std::sqrt(re * re + im * im);
    xp[0] = re / len; xp[1] = im / len;
    xp[2] = x[2] + (x[0] * ex - x[1] * ey);
    xp[3] = x[3] + (x[1] * ex + x[0] * ey);
    return true;
  }
  bool PlusJacobian(const double* x, double* J) const override {  // 4 x 3 row-major
    const double c = x[0], s = x[1];
    const double j[12] = {0, 0, -s, 0, 0, c, c, -s, 0, s, c, 0};
    for (int i = 0; i < 12; ++i) J[i] = j[i];
    return true;
  }
  bool Minus(const double* y, const double* x, double* d) const override {  // log(x^-1 y); not used by the minimiser loop
    const double c = x[0], s = -x[1];
    const double re = c * y[0] - s * y[1], im = c * y[1] + s * y[0];
    const double dx = y[2] - x[2], dy = y[3] - x[3];
    const double tx = c * dx - s * dy, ty = s * dx + c * dy;
    const double theta = std::atan2(im, re), half = 0.5 * theta;
    const double hbt = std::abs(re - 1.0) < 1e-10 ? 1.0 - theta * theta / 12.0 : -(half * im) / (re - 1.0);
    d[0] = hbt * tx + half * ty; d[1] = -half * tx + hbt * ty; d[2] = theta;
    return true;
  }
  bool MinusJacobian(const double* x, double* J) const override {
    const double c = x[0], s = x[1];
    const double j[12] = {0, 0, c, s, 0, 0, -s, c, -s, c, 0, 0};
    for (int i = 0; i < 12; ++i) J[i] = j[i];
    return true;
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "r");
  if (!f) return 2;
  double a, mu_scale, alpha, weight, div;
  int gnc_steps, max_it, d3, manifold, n;
  double x[4];
  if (std::fscanf(f, "%lf %lf %lf %lf %lf %d %d %d %d %d", &a, &mu_scale, &alpha, &weight, &div, &gnc_steps, &max_it, &d3, &manifold, &n) != 10) return 2;
  if (std::fscanf(f, "%lf %lf %lf %lf", &x[0], &x[1], &x[2], &x[3]) != 4) return 2;
  std::vector<Pair> pairs(n);
  for (auto& p : pairs) {
    double* v[4] = {p.mm, p.Sm, p.fm, p.Sf};
    const int len[4] = {3, 6, 3, 6};
    for (int k = 0; k < 4; ++k)
      for (int i = 0; i < len[k]; ++i)
        if (std::fscanf(f, "%lf", &v[k][i]) != 1) return 2;
  }
  std::fclose(f);

  ceres::Problem problem;
  auto* wrapper = new ceres::LossFunctionWrapper(new ceres::ScaledLoss(new BarronLoss(a, alpha, 1.0), weight, ceres::TAKE_OWNERSHIP), ceres::TAKE_OWNERSHIP);
  problem.AddParameterBlock(x, 4);
  if (manifold) problem.SetManifold(x, new SE2Manifold());
  for (const auto& p : pairs) {
    ceres::CostFunction* cf = d3 ? static_cast<ceres::CostFunction*>(new ceres::AutoDiffCostFunction<NdtResidual<3>, 1, 4>(new NdtResidual<3>(p)))
                                 : static_cast<ceres::CostFunction*>(new ceres::AutoDiffCostFunction<NdtResidual<2>, 1, 4>(new NdtResidual<2>(p)));
    problem.AddResidualBlock(cf, wrapper, x);
  }
  // max raw residual at the initial point -> gnc_mu (ndt_matcher.cpp:466-476)
  double raw_max = 0.0;
  {
    ceres::Problem::EvaluateOptions eo;
    eo.apply_loss_function = false;
    std::vector<double> res;
    problem.Evaluate(eo, nullptr, &res, nullptr, nullptr);
    for (double r : res) raw_max = std::max(raw_max, std::abs(r));
  }
  double gnc_mu = std::min(2.0 * raw_max * raw_max / (mu_scale * mu_scale), std::pow(div, gnc_steps - 1));
  ceres::Solver::Options opt;
  opt.max_num_iterations = max_it;
  opt.linear_solver_type = ceres::DENSE_QR;
  opt.trust_region_strategy_type = ceres::LEVENBERG_MARQUARDT;
  opt.num_threads = 1;
  opt.logging_type = ceres::SILENT;
  int solves = 0;
  do {
    gnc_mu = std::max(gnc_mu, 1.0);
    wrapper->Reset(new ceres::ScaledLoss(new BarronLoss(a, alpha, gnc_mu), weight, ceres::TAKE_OWNERSHIP), ceres::TAKE_OWNERSHIP);
    ceres::Solver::Summary summary;
    ceres::Solve(opt, &problem, &summary);
    std::printf("solve %d mu %.17g iterations %d termination %d final_cost %.17g\n", solves, gnc_mu, (int)summary.iterations.size(),
                (int)summary.termination_type, summary.final_cost);
    for (const auto& it : summary.iterations)
      std::printf("  it %d cost %.17g radius %.17g ok %d\n", it.iteration, it.cost, it.trust_region_radius, it.step_is_successful ? 1 : 0);
    ++solves;
    gnc_mu /= div;
  } while (gnc_mu > 1.0 / std::sqrt(div));
  std::printf("pose %.17g %.17g %.17g %.17g\n", x[0], x[1], x[2], x[3]);
  return 0;
}
