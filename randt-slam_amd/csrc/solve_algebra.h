// The Levenberg-Marquardt algebra between two residual passes of the pair solve, shared by every geometry of k_solve
// (solve.hip: one wavefront per registration, split mode; solve_tr.hip: one LANE per registration).
//
// Replaces Ceres 2.1.0's TrustRegionMinimizer / LevenbergMarquardtStrategy / DENSE_QR step on the 3- or 4-dimensional
// problem of Matcher::estimateLoopConstraint (src/ndt_registration/ndt_matcher.cpp:457-483) and
// Sophus::Manifold<SE2>::Plus (ceres_manifold.hpp, se2.hpp, so2.hpp).
//
// Every product-sum is an explicit fma / mul / add under `fp contract(off)`: the same statements are compiled into kernels
// of very different shape -- wave-uniform in the one-wavefront kernels, lane-divergent in the transposed one -- and the
// geometries must stay bit-identical (tests/test_gpu_config4.py).  Left to the compiler's own contraction the same
// expression fuses differently in different surroundings (seen in round 2 on the residual pass, solve_math.h).
// U = true: the values are the same in every lane and a branch on them goes through a ballot (a scalar branch);
// U = false: lane-private values, ordinary divergent branches.
#pragma once
#include "randt.h"
#include "solve_math.h"

namespace randt_lm {

using randt_solve::fast_rcp;
using randt_solve::fast_rsqrt;

template <bool U>
__device__ __forceinline__ bool cnd(bool c) {
  return U ? randt_solve::uni(c) : c;
}

// base sums of one pass: cost, g_b (tx, ty, theta), G upper (tt: 00 01 02 11 12 22)
struct Base {
  double v[10];
};

// RANDT_PARAM_VECTOR and RANDT_PARAM_ANALYTIC optimise the same (pos[2], rot) blocks; ANALYTIC only swaps the NDT functor's
// rotation Jacobian for the reference's hand-written one (solve_math.h, residual_sq_v)
__device__ __forceinline__ constexpr bool vec_like(int param) { return param == RANDT_PARAM_VECTOR || param == RANDT_PARAM_ANALYTIC; }

// packed lower triangle: (i, j), i >= j  ->  i (i + 1) / 2 + j
__device__ __forceinline__ constexpr int sym(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// ---------------------------------------------------------------- Sophus SE(2) pieces ----------
__device__ __forceinline__ void so2_normalize(double& c, double& s) {
#pragma clang fp contract(off)
  const double len = sqrt(fma(c, c, s * s));
  c = c / len;
  s = s / len;
}

// Sophus::Manifold<SE2>::Plus(T, delta) = T * exp(delta) (ceres_manifold.hpp, se2.hpp, so2.hpp)
template <bool U>
__device__ __forceinline__ void se2_plus(const double* x, const double* d, double* xp) {
#pragma clang fp contract(off)
  const double theta = d[2];
  double c = cos(theta), s = sin(theta);
  so2_normalize(c, s);
  double sbt, omcbt;
  if (cnd<U>(fabs(theta) < 1e-10)) {
    const double tsq = theta * theta;
    sbt = fma(-(1.0 / 6.0), tsq, 1.0);
    omcbt = fma(0.5, theta, -(((1.0 / 24.0) * theta) * tsq));
  } else {
    sbt = s / theta;
    omcbt = (1.0 - c) / theta;
  }
  const double ex = fma(sbt, d[0], -(omcbt * d[1]));
  const double ey = fma(omcbt, d[0], sbt * d[1]);
  double re = fma(x[0], c, -(x[1] * s));
  double im = fma(x[0], s, x[1] * c);
  const double sq = fma(re, re, im * im);
  if (cnd<U>(sq != 1.0)) {
    const double scale = 2.0 / (1.0 + sq);
    re *= scale;
    im *= scale;
  }
  so2_normalize(re, im);
  xp[0] = re;
  xp[1] = im;
  xp[2] = x[2] + fma(x[0], ex, -(x[1] * ey));
  xp[3] = x[3] + fma(x[1], ex, x[0] * ey);
}

template <int PARAM, bool U>
__device__ __forceinline__ void plus(const double* x, const double* d, double* xp) {
  if (PARAM == RANDT_PARAM_MANIFOLD) {
    se2_plus<U>(x, d, xp);
  } else if (PARAM == RANDT_PARAM_AMBIENT4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xp[i] = x[i] + d[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) xp[i] = x[i] + d[i];
    xp[3] = 0.0;
  }
}

// g = T g_b, H = T G T^T (packed lower, NT (NT + 1) / 2 entries) for the parameterisation at ambient point x.
// T is NT x 3 (solve.hip header); its zero / one entries are spelled out so that no multiplications by zero are left
// (the compiler may not fold 0 * x under IEEE rules).
template <int PARAM, int NT>
__device__ __forceinline__ void to_param(const Base& B, const double* x, double* g, double* H) {
#pragma clang fp contract(off)
  const double gb0 = B.v[1], gb1 = B.v[2], gb2 = B.v[3];
  const double G00 = B.v[4], G01 = B.v[5], G02 = B.v[6], G11 = B.v[7], G12 = B.v[8], G22 = B.v[9];
  if (vec_like(PARAM)) {
    g[0] = gb0; g[1] = gb1; g[2] = gb2;
    H[sym(0, 0)] = G00; H[sym(1, 0)] = G01; H[sym(1, 1)] = G11;
    H[sym(2, 0)] = G02; H[sym(2, 1)] = G12; H[sym(2, 2)] = G22;
    return;
  }
  const double cp = x[0], sp = x[1];
  const double in2 = fast_rcp(fma(cp, cp, sp * sp));
  const double a = -sp * in2, b = cp * in2;  // d theta / d c, d theta / d s  (theta = atan2(s, c))
  if (PARAM == RANDT_PARAM_AMBIENT4) {
    // rows of T: c -> (0, 0, a); s -> (0, 0, b); tx -> (1, 0, 0); ty -> (0, 1, 0)
    const double aG = a * G22, bG = b * G22;
    g[0] = a * gb2; g[1] = b * gb2; g[2] = gb0; g[NT - 1] = gb1;
    H[sym(0, 0)] = aG * a;
    H[sym(1, 0)] = bG * a; H[sym(1, 1)] = bG * b;
    H[sym(2, 0)] = a * G02; H[sym(2, 1)] = b * G02; H[sym(2, 2)] = G00;
    H[sym(NT - 1, 0)] = a * G12; H[sym(NT - 1, 1)] = b * G12; H[sym(NT - 1, 2)] = G01; H[sym(NT - 1, NT - 1)] = G11;
  } else {
    // ambient row times Sophus PlusJacobian [[0,0,-s],[0,0,c],[c,-s,0],[s,c,0]] (stored complex):
    // T = [[cp, sp, 0], [-sp, cp, 0], [0, 0, w]], w = a (-sp) + b cp
    const double w = fma(a, -sp, b * cp);
    g[0] = fma(cp, gb0, sp * gb1);
    g[1] = fma(cp, gb1, -(sp * gb0));
    g[2] = w * gb2;
    const double r00 = fma(cp, G00, sp * G01), r01 = fma(cp, G01, sp * G11);   // (R G2) rows
    const double r10 = fma(cp, G01, -(sp * G00)), r11 = fma(cp, G11, -(sp * G01));
    H[sym(0, 0)] = fma(r00, cp, r01 * sp);
    H[sym(1, 0)] = fma(r10, cp, r11 * sp);
    H[sym(1, 1)] = fma(r11, cp, -(r10 * sp));
    H[sym(2, 0)] = w * fma(cp, G02, sp * G12);
    H[sym(2, 1)] = w * fma(cp, G12, -(sp * G02));
    H[sym(2, 2)] = (w * G22) * w;
  }
}

// Jacobi scaling 1 / (1 + sqrt(H_ii)), fixed per solve; Newton-refined rsqrt / rcp (~1 ulp: the scaling is a change of
// variables, exact arithmetic does not see it)
__device__ __forceinline__ double jacobi_scale(double hii) {
#pragma clang fp contract(off)
  return fast_rcp(1.0 + (hii > 0.0 ? hii * fast_rsqrt(hii) : 0.0));
}

// Solve of the NT x NT SPD system A y = g by LDL^T (A packed lower, destroyed): NT reciprocals,
// no square roots.  Returns false if a pivot is not positive.
template <int NT>
__device__ __forceinline__ bool ldlt_solve(double* A, const double* g, double* y) {
#pragma clang fp contract(off)
  bool ok = true;
  double inv_d[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    double d = A[sym(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d = fma(-(A[sym(j, k)] * A[sym(j, k)]), A[sym(k, k)], d);
    if (!(d > 0.0)) ok = false;
    A[sym(j, j)] = d;
    inv_d[j] = fast_rcp(d);
#pragma unroll
    for (int i = j + 1; i < NT; ++i) {
      double a = A[sym(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) a = fma(-(A[sym(i, k)] * A[sym(j, k)]), A[sym(k, k)], a);
      A[sym(i, j)] = a * inv_d[j];
    }
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    double a = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) a = fma(-A[sym(i, k)], y[k], a);
    y[i] = a;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) y[i] *= inv_d[i];
#pragma unroll
  for (int i = NT - 1; i >= 0; --i) {
    double a = y[i];
#pragma unroll
    for (int k = i + 1; k < NT; ++k) a = fma(-A[sym(k, i)], y[k], a);
    y[i] = a;
  }
  return ok;
}

// LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled normal equations + the model cost change of the step:
// (Hs + diag / radius) y = gs, step = -y; model_cost_change = -(step.g + step^T H step / 2).  Returns "the linear
// solver succeeded" (positive pivots, finite step).
template <int NT>
__device__ __forceinline__ bool compute_step(const double* Hr, const double* gr, const double* dg, double radius, double* step, double& mcc) {
#pragma clang fp contract(off)
  constexpr int NS = NT * (NT + 1) / 2;
  double A[NS];
  const double inv_radius = fast_rcp(radius);
#pragma unroll
  for (int i = 0; i < NS; ++i) A[i] = Hr[i];
#pragma unroll
  for (int i = 0; i < NT; ++i) A[sym(i, i)] = fma(dg[i], inv_radius, A[sym(i, i)]);  // (sqrt(D^2 / radius))^2
  bool solved = ldlt_solve<NT>(A, gr, step);
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (!isfinite(step[i])) solved = false;
    step[i] = -step[i];
  }
  double m = 0.0;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    double hs = 0.0;
#pragma unroll
    for (int j = 0; j < NT; ++j) hs = fma(Hr[sym(i, j)], step[j], hs);
    m = fma(step[i], fma(0.5, hs, gr[i]), m);
  }
  mcc = -m;
  return solved;
}

template <int PARAM>
__device__ __forceinline__ double ambient_norm(const double* x) {
#pragma clang fp contract(off)
  double n = 0.0;
  constexpr int NA = vec_like(PARAM) ? 3 : 4;
#pragma unroll
  for (int i = 0; i < NA; ++i) n = fma(x[i], x[i], n);
  return n > 0.0 ? n * fast_rsqrt(n) : 0.0;  // sqrt to ~1 ulp without the IEEE sequence (only feeds the parameter-tolerance test)
}

// ||x - cand||^2 over the ambient coordinates (TrustRegionMinimizer::ParameterToleranceReached)
template <int PARAM>
__device__ __forceinline__ double step_norm_sq(const double* xc, const double* cc) {
#pragma clang fp contract(off)
  double sn2 = 0.0;
  constexpr int NA = vec_like(PARAM) ? 3 : 4;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const double d = xc[i] - cc[i];
    sn2 = fma(d, d, sn2);
  }
  return sn2;
}

// radius after a successful step: radius / max(1/3, 1 - (2 rho - 1)^3), capped (TrustRegionStepEvaluator / LM strategy)
__device__ __forceinline__ double radius_after_success(double radius, double rel, double rmax) {
#pragma clang fp contract(off)
  const double t = fma(2.0, rel, -1.0);
  radius = radius * fast_rcp(fmax(1.0 / 3.0, fma(-(t * t), t, 1.0)));
  return fmin(rmax, radius);
}

// Is ||x - Plus(x, -g)||_inf <= gtol (TrustRegionMinimizer::GradientToleranceReached)?  For the
// manifold the displacement is >= 0.4 max|g_i| (|omega| <= pi), so the exact Plus is only
// evaluated for tiny gradients.
template <int PARAM, int NT, bool U>
__device__ __forceinline__ bool gradient_converged(const double* x, const double* g, double gtol) {
#pragma clang fp contract(off)
  double gm = 0.0;
#pragma unroll
  for (int i = 0; i < NT; ++i) gm = fabs(g[i]) > gm ? fabs(g[i]) : gm;
  if (PARAM != RANDT_PARAM_MANIFOLD) return cnd<U>(gm <= gtol);
  if (cnd<U>(0.4 * gm > gtol && gm < 3.0)) return false;
  double neg[NT], xp[4];
#pragma unroll
  for (int i = 0; i < NT; ++i) neg[i] = -g[i];
  plus<PARAM, U>(x, neg, xp);
  double m = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double a = fabs(x[i] - xp[i]);
    m = a > m ? a : m;
  }
  return cnd<U>(m <= gtol);
}

// gnc_mu of the first GNC stage from the largest raw residual (ndt_matcher.cpp:466-476)
__device__ __forceinline__ double gnc_mu_start(double raw_max, double mu_scale, double mu_cap) {
#pragma clang fp contract(off)
  const double mu = 2.0 * (raw_max * raw_max) / (mu_scale * mu_scale);
  return fmin(mu, mu_cap);
}

}  // namespace randt_lm
