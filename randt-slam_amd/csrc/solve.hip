// Persistent registration solve kernel for gfx950: the whole GNC x Levenberg-Marquardt loop of one
// scan-to-submap registration runs inside ONE workgroup with no host round trips.
//
// Replaces (paths relative to /root/reference/ros/ndt_radar_slam/):
//   src/ndt_registration/ndt_matcher.cpp:217-246,457-492   residual-block construction, GNC loop, ceres::Solve
//   include/ndt_registration/ceres_residuals.h:421-552     NDTFrameToMap{,Intensity}FactorResidual{,SE2}
//   src/ndt_registration/ceres_loss_functions.cpp:19-39    BarronLoss (+ ceres::ScaledLoss)
//   Ceres 2.1.0 (un-vendored): TrustRegionMinimizer, LevenbergMarquardtStrategy, DENSE_QR, Corrector
//   Sophus 1.22.10 (un-vendored): SE2 exp / group product / Manifold<SE2>::Plus and PlusJacobian
//
// Layout: the frozen correspondence set is staged once into LDS as fp32 records of 9 floats
// (mean xyz + upper-triangular covariance; stride 9 words is odd => conflict-free ds_read_b32):
// moving cells [M][9] and, per correspondence slot, the fixed cell [M*k][9]; cast to fp64 in
// registers like the reference does (ndt_matcher.cpp:231).  Every LM iteration is one pass of the
// 256 lanes over the M*k slots (residual + tangent Jacobian + robust re-weighting in fp64), a
// fixed-order reduction of {cost, J^T r, upper(J^T J)} (wave __shfl_xor tree -> 4-way LDS combine,
// deterministic), and the 3x3 / 4x4 damped normal-equation solve + Ceres' accept/reject logic
// executed redundantly by all lanes (uniform control flow, no broadcast).  The candidate point is
// evaluated WITH its Jacobian so that an accepted step needs no second pass.
#include "randt_internal.h"

#include <float.h>

#define SOLVE_MAX_WAVES 4

namespace {

template <int NT>
struct Sums {
  double cost;
  double g[NT];
  double h[NT * (NT + 1) / 2];
};

// ---------------------------------------------------------------- Sophus SE(2) pieces ----------
__device__ __forceinline__ void so2_normalize(double& c, double& s) {
  const double len = sqrt(c * c + s * s);
  c = c / len;
  s = s / len;
}

// Sophus::Manifold<SE2>::Plus(T, delta) = T * exp(delta) (ceres_manifold.hpp, se2.hpp, so2.hpp)
__device__ __forceinline__ void se2_plus(const double* x, const double* d, double* xp) {
  const double theta = d[2];
  double c = cos(theta), s = sin(theta);
  so2_normalize(c, s);
  double sbt, omcbt;
  if (fabs(theta) < 1e-10) {
    const double tsq = theta * theta;
    sbt = 1.0 - (1.0 / 6.0) * tsq;
    omcbt = 0.5 * theta - (1.0 / 24.0) * theta * tsq;
  } else {
    sbt = s / theta;
    omcbt = (1.0 - c) / theta;
  }
  const double ex = sbt * d[0] - omcbt * d[1];
  const double ey = omcbt * d[0] + sbt * d[1];
  double re = x[0] * c - x[1] * s;
  double im = x[0] * s + x[1] * c;
  const double sq = re * re + im * im;
  if (sq != 1.0) {
    const double scale = 2.0 / (1.0 + sq);
    re *= scale;
    im *= scale;
  }
  so2_normalize(re, im);
  xp[0] = re;
  xp[1] = im;
  xp[2] = x[2] + (x[0] * ex - x[1] * ey);
  xp[3] = x[3] + (x[1] * ex + x[0] * ey);
}

template <int PARAM>
__device__ __forceinline__ void plus(const double* x, const double* d, double* xp) {
  if (PARAM == RANDT_PARAM_MANIFOLD) {
    se2_plus(x, d, xp);
  } else if (PARAM == RANDT_PARAM_AMBIENT4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xp[i] = x[i] + d[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) xp[i] = x[i] + d[i];
    xp[3] = 0.0;
  }
}

// ---------------------------------------------------------------- loss -------------------------
struct Loss {
  double b, c, factor, exponent, pre, ts, alpha, weight, sqrt_w;
  int mode;  // 0: identity (alpha >= 2), 1: log (|alpha| <= 0.05), 2: alpha == -2 closed form, 3: general pow
};

// BarronLoss ctor (ceres_loss_functions.h:27-35)
__device__ __forceinline__ Loss make_loss(double a, double alpha, double mu, double weight) {
  Loss L;
  L.alpha = alpha;
  L.b = mu * a * a;
  L.c = 1 / L.b;
  L.factor = fabs(alpha - 2.0);
  L.exponent = 0.5 * alpha;
  L.pre = L.b * L.factor / alpha;
  L.ts = 2 * L.c / L.factor;
  L.weight = weight;
  L.sqrt_w = sqrt(weight);
  L.mode = alpha >= 2.0 ? 0 : (fabs(alpha) <= 0.05 ? 1 : (alpha == -2.0 ? 2 : 3));
  return L;
}

// BarronLoss::Evaluate (ceres_loss_functions.cpp:19-39) x ScaledLoss
__device__ __forceinline__ void loss_eval(const Loss& L, double s, double& r0, double& r1, double& r2) {
  if (L.mode == 2) {
    // alpha = -2: exponent -1 -> pow(u,-1) = 1/u, pow(u,-2), pow(u,-3)
    const double u = s * L.ts + 1.0;
    const double iu = 1.0 / u;
    r0 = L.pre * (iu - 1.);
    r1 = L.pre * L.exponent * (iu * iu) * L.ts;
    r2 = L.pre * L.exponent * (L.exponent - 1) * (iu * iu * iu) * L.ts * L.ts;
  } else if (L.mode == 0) {
    r0 = s;
    r1 = 1;
    r2 = 0;
  } else if (L.mode == 1) {
    const double sum = 1.0 + s * L.c;
    const double inv = 1.0 / sum;
    r0 = L.b * log(sum);
    r1 = inv > DBL_MIN ? inv : DBL_MIN;
    r2 = -L.c * (inv * inv);
  } else {
    const double u = s * L.ts + 1.0;
    r0 = L.pre * (pow(u, L.exponent) - 1.);
    r1 = L.pre * L.exponent * pow(u, L.exponent - 1.) * L.ts;
    r2 = L.pre * L.exponent * (L.exponent - 1) * pow(u, L.exponent - 2.) * L.ts * L.ts;
  }
  r0 *= L.weight;
  r1 *= L.weight;
  r2 *= L.weight;
}

// ---------------------------------------------------------------- residual ---------------------
// One D2D residual r = sqrt(d^T (R Sm R^T + Sf)^-1 d) and its Jacobian row in the chosen
// parameterisation.  mv/fv: 9 floats (mean xyz, cov xx xy xi yy yi ii).
// c, s = cos/sin of theta = atan2 of the stored complex; cp, sp = the stored complex itself.
template <int D, int PARAM, int NT>
__device__ __forceinline__ double residual(const float* mv, const float* fv, double c, double s, double cp, double sp,
                                           double n2, double tx, double ty, double* J, bool want_jac) {
  const double m0 = mv[0], m1 = mv[1];
  const double a = mv[3], b = mv[4], dd = mv[6];
  const double F0 = fv[3], F1 = fv[4], F3 = fv[6];
  const double RS00 = c * a - s * b, RS01 = c * b - s * dd;
  const double RS10 = s * a + c * b, RS11 = s * b + c * dd;
  const double C00 = (RS00 * c - RS01 * s) + F0;
  const double C01 = (RS00 * s + RS01 * c) + F1;
  const double C11 = (RS10 * s + RS11 * c) + F3;
  const double d0 = (c * m0 - s * m1) + tx - fv[0];
  const double d1 = (s * m0 + c * m1) + ty - fv[1];
  double q0, q1, q2 = 0.0, ssq;
  double cc = 0.0, e = 0.0;
  if (D == 3) {
    cc = mv[5];
    e = mv[7];
    const double f = mv[8];
    const double C02 = (c * cc - s * e) + fv[5];
    const double C12 = (s * cc + c * e) + fv[7];
    const double C22 = f + fv[8];
    const double d2 = (double)mv[2] - (double)fv[2];
    const double k00 = C11 * C22 - C12 * C12;
    const double k01 = C12 * C02 - C01 * C22;
    const double k02 = C01 * C12 - C11 * C02;
    const double det = C00 * k00 + C01 * k01 + C02 * k02;
    const double id = 1.0 / det;
    const double k11 = C00 * C22 - C02 * C02;
    const double k12 = C02 * C01 - C00 * C12;
    const double k22 = C00 * C11 - C01 * C01;
    q0 = (k00 * d0 + k01 * d1 + k02 * d2) * id;
    q1 = (k01 * d0 + k11 * d1 + k12 * d2) * id;
    q2 = (k02 * d0 + k12 * d1 + k22 * d2) * id;
    ssq = d0 * q0 + d1 * q1 + d2 * q2;
  } else {
    const double det = C00 * C11 - C01 * C01;
    const double id = 1.0 / det;
    q0 = (C11 * d0 - C01 * d1) * id;
    q1 = (-C01 * d0 + C00 * d1) * id;
    ssq = d0 * q0 + d1 * q1;
  }
  if (!want_jac) return sqrt(ssq);
  {
    if (!(ssq > 0.0)) {
      // autodiff of sqrt(0) is singular in the reference (ceres_residuals.h:545): zero row instead
#pragma unroll
      for (int i = 0; i < NT; ++i) J[i] = 0.0;
      return sqrt(ssq);
    }
    // r = ssq * rsqrt(ssq), 1/r = rsqrt(ssq): one transcendental instead of sqrt + divide
    const double ir = rsqrt(ssq);
    const double r = ssq * ir;
    const double u0 = c * q0 + s * q1, u1 = -s * q0 + c * q1;
    double Su0 = a * u0 + b * u1, Su1 = b * u0 + dd * u1;
    if (D == 3) {
      Su0 += cc * q2;
      Su1 += e * q2;
    }
    const double dth = ((u1 * m0 - u0 * m1) - (u1 * Su0 - u0 * Su1)) * ir;
    const double dtx = q0 * ir, dty = q1 * ir;
    if (PARAM == RANDT_PARAM_MANIFOLD) {
      const double dc = dth * (-sp / n2), ds = dth * (cp / n2);
      J[0] = dtx * cp + dty * sp;
      J[1] = -dtx * sp + dty * cp;
      J[2] = dc * (-sp) + ds * cp;
    } else if (PARAM == RANDT_PARAM_AMBIENT4) {
      J[0] = dth * (-sp / n2);
      J[1] = dth * (cp / n2);
      J[2] = dtx;
      J[3] = dty;
    } else {
      J[0] = dtx;
      J[1] = dty;
      J[2] = dth;
    }
    return r;
  }
}

// ---------------------------------------------------------------- reductions -------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// Where a pass finds the frozen correspondence set.  STAGE = true: LDS copies (mov [M][9], fix
// [M*k][9], stride 9 words => conflict-free ds_read_b32).  STAGE = false: the 48-byte cell records
// are read in place from L1/L2 (the first 9 floats of a record are mean xyz + covariance).
struct Stage {
  const float* mov;
  const float* fix;
  const int* valid;    // [M*k] compact fixed index or -1
  int n_slots, k;
};

// Pass over all correspondence slots at ambient point x.  MODE 0: max raw residual; MODE 1: cost,
// J^T r, J^T J with loss + corrector (Ceres residual_block.cc / corrector.cc).
// Returns false if any residual was non-finite.
template <int D, int PARAM, int NT, int MODE, int BLOCK, bool STAGE>
__device__ __forceinline__ bool eval_pass(const Stage& S, const double* x, const Loss& L, Sums<NT>& out, double& raw_max,
                                          double (*red)[24]) {
  constexpr int SOLVE_WAVES = BLOCK / 64;
  double cp, sp, tx, ty, c, s, n2;
  if (PARAM == RANDT_PARAM_VECTOR) {
    c = cos(x[2]);
    s = sin(x[2]);
    cp = c;
    sp = s;
    n2 = 1.0;
    tx = x[0];
    ty = x[1];
  } else {
    cp = x[0];
    sp = x[1];
    tx = x[2];
    ty = x[3];
    n2 = cp * cp + sp * sp;
    // R = AngleAxis(atan2(sp, cp)): cos/sin of the angle == normalised complex
    const double inv = 1.0 / sqrt(n2);
    c = cp * inv;
    s = sp * inv;
  }
  constexpr int NH = NT * (NT + 1) / 2;
  constexpr int NA = 1 + NT + NH;
  double acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = 0.0;
  double mx = -DBL_MAX;
  int bad = 0;
  for (int slot = threadIdx.x; slot < S.n_slots; slot += BLOCK) {
    const int ci = S.valid[slot];
    if (ci < 0) continue;
    const float* mv = STAGE ? S.mov + (slot / S.k) * 9 : S.mov + (size_t)(slot / S.k) * 12;
    const float* fv = STAGE ? S.fix + slot * 9 : S.fix + (size_t)ci * 12;
    double J[NT];
    const double r = residual<D, PARAM, NT>(mv, fv, c, s, cp, sp, n2, tx, ty, J, MODE == 1);
    if (!isfinite(r)) bad = 1;
    if (MODE == 0) {
      mx = r > mx ? r : mx;
    } else {
      const double sq = r * r;
      double r0, r1, r2;
      double rs, jscale;
      if (L.mode == 2) {
        // alpha = -2: rho' = w / u^2 > 0, rho'' < 0 always => corrector is sqrt(rho') = sqrt(w) / u
        const double iu = 1.0 / (sq * L.ts + 1.0);
        acc[0] += 0.5 * (L.weight * (L.pre * (iu - 1.)));
        rs = jscale = L.sqrt_w * iu;
      } else {
      loss_eval(L, sq, r0, r1, r2);
      acc[0] += 0.5 * r0;
      const double sqrt_rho1 = sqrt(r1);
      if (sq == 0.0 || r2 <= 0.0) {
        rs = sqrt_rho1;
        jscale = sqrt_rho1;
      } else {
        const double Dc = 1.0 + 2.0 * sq * r2 / r1;
        const double al = 1.0 - sqrt(Dc);
        rs = sqrt_rho1 / (1 - al);
        jscale = sqrt_rho1 * (1.0 - al);  // J - (alpha/sq) r r^T J for a scalar residual
      }
      }
      const double wr = rs * r;
      double wJ[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) wJ[i] = jscale * J[i];
      int hidx = 0;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        acc[1 + i] += wJ[i] * wr;
#pragma unroll
        for (int j = i; j < NT; ++j) acc[1 + NT + (hidx++)] += wJ[i] * wJ[j];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (SOLVE_WAVES == 1) {
    // one wavefront owns the registration: pure register reduction, no LDS, no barrier
    const double b1 = wave_max((double)bad);
    if (MODE == 0) {
      raw_max = wave_max(mx);
      return b1 == 0.0;
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i] = wave_sum(acc[i]);
      out.cost = acc[0];
#pragma unroll
      for (int i = 0; i < NT; ++i) out.g[i] = acc[1 + i];
#pragma unroll
      for (int i = 0; i < NH; ++i) out.h[i] = acc[1 + NT + i];
      return b1 == 0.0 && isfinite(acc[0]);
    }
  }
  __syncthreads();  // red[] reuse
  if (MODE == 0) {
    mx = wave_max(mx);
    const double b = wave_max((double)bad);
    if (lane == 0) {
      red[wave][0] = mx;
      red[wave][1] = b;
    }
    __syncthreads();
    double m = red[0][0], bb = red[0][1];
#pragma unroll
    for (int w = 1; w < SOLVE_WAVES; ++w) {
      m = red[w][0] > m ? red[w][0] : m;
      bb = red[w][1] > bb ? red[w][1] : bb;
    }
    raw_max = m;
    return bb == 0.0;
  } else {
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = wave_sum(acc[i]);
    const double b = wave_max((double)bad);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) red[wave][i] = acc[i];
      red[wave][NA] = b;
    }
    __syncthreads();
    double tot[NA];
    double bb = 0.0;
#pragma unroll
    for (int i = 0; i < NA; ++i) tot[i] = 0.0;
#pragma unroll
    for (int w = 0; w < SOLVE_WAVES; ++w) {
#pragma unroll
      for (int i = 0; i < NA; ++i) tot[i] += red[w][i];
      bb = red[w][NA] > bb ? red[w][NA] : bb;
    }
    out.cost = tot[0];
#pragma unroll
    for (int i = 0; i < NT; ++i) out.g[i] = tot[1 + i];
#pragma unroll
    for (int i = 0; i < NH; ++i) out.h[i] = tot[1 + NT + i];
    return bb == 0.0 && isfinite(tot[0]);
  }
}

// upper-triangular packed index
template <int NT>
__device__ __forceinline__ constexpr int hix(int i, int j) {
  return i <= j ? (i * NT - i * (i - 1) / 2 + (j - i)) : (j * NT - j * (j - 1) / 2 + (i - j));
}

// Solve of the NT x NT SPD system A y = g by LDL^T (A full, row-major, destroyed): NT reciprocals,
// no square roots.  Returns false if a pivot is not positive.
template <int NT>
__device__ __forceinline__ bool chol_solve(double* A, const double* g, double* y) {
  bool ok = true;
  double inv_d[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    double d = A[j * NT + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j * NT + k] * A[j * NT + k] * A[k * NT + k];
    if (!(d > 0.0)) ok = false;
    A[j * NT + j] = d;
    inv_d[j] = 1.0 / d;
#pragma unroll
    for (int i = j + 1; i < NT; ++i) {
      double a = A[i * NT + j];
#pragma unroll
      for (int k = 0; k < j; ++k) a -= A[i * NT + k] * A[j * NT + k] * A[k * NT + k];
      A[i * NT + j] = a * inv_d[j];
    }
  }
  // L z = g ; D w = z ; L^T y = w
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    double a = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) a -= A[i * NT + k] * y[k];
    y[i] = a;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) y[i] *= inv_d[i];
#pragma unroll
  for (int i = NT - 1; i >= 0; --i) {
    double a = y[i];
#pragma unroll
    for (int k = i + 1; k < NT; ++k) a -= A[k * NT + i] * y[k];
    y[i] = a;
  }
  return ok;
}

template <int PARAM>
__device__ __forceinline__ double ambient_norm(const double* x) {
  double n = 0.0;
  constexpr int NA = PARAM == RANDT_PARAM_VECTOR ? 3 : 4;
#pragma unroll
  for (int i = 0; i < NA; ++i) n += x[i] * x[i];
  return sqrt(n);
}

// ||x - Plus(x, -g)||_inf (TrustRegionMinimizer::EvaluateGradientAndJacobian)
template <int PARAM, int NT>
__device__ __forceinline__ double grad_max_norm(const double* x, const double* g) {
  double neg[NT], xp[4];
#pragma unroll
  for (int i = 0; i < NT; ++i) neg[i] = -g[i];
  plus<PARAM>(x, neg, xp);
  double m = 0.0;
  constexpr int NA = PARAM == RANDT_PARAM_VECTOR ? 3 : 4;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const double a = fabs(x[i] - xp[i]);
    m = a > m ? a : m;
  }
  return m;
}

__device__ __forceinline__ void trace_push(double* tr, int max_len, double cost, double radius, int flag) {
  if (tr && threadIdx.x == 0) {
    const int n = (int)tr[0];
    if (3 * (n + 1) + 1 <= max_len) {
      tr[1 + 3 * n + 0] = cost;
      tr[1 + 3 * n + 1] = radius;
      tr[1 + 3 * n + 2] = (double)flag;
      tr[0] = (double)(n + 1);
    }
  }
}

template <int D, int PARAM, int BLOCK, bool STAGE>
__global__ __launch_bounds__(BLOCK) void k_solve(MapView fixed, const int32_t* __restrict__ fixed_idx, MapView moving,
                                                       int moving_first, const int32_t* __restrict__ corr, SolveParams P,
                                                       double* __restrict__ pose4, randt_result* __restrict__ results,
                                                       double* trace, int trace_len) {
  constexpr int NT = PARAM == RANDT_PARAM_AMBIENT4 ? 4 : 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[BLOCK / 64][24];
  __shared__ int s_count;

  const int tid = threadIdx.x;
  const int pair = blockIdx.x;
  const int fmap = fixed_idx ? fixed_idx[pair] : 0;
  const int mmap = moving_first + pair;
  const int k = P.k;
  int M = moving.counts[mmap];
  M = M > moving.cap ? moving.cap : M;
  const int n_slots = M * k;

  const randt_cell* mcells = moving.cells + (size_t)mmap * moving.cap;
  const randt_cell* fcells = fixed.cells + (size_t)fmap * fixed.cap;
  const int32_t* pc = corr + (size_t)pair * moving.cap * k;
  Stage S;
  S.n_slots = n_slots;
  S.k = k;
  int n_res;
  if (STAGE) {
    // ---- stage the frozen correspondence set in LDS (addNDTFactor, ndt_matcher.cpp:217-246)
    float* lmov = reinterpret_cast<float*>(smem);
    float* lfix = lmov + (size_t)moving.cap * 9;
    int* lvalid = reinterpret_cast<int*>(lfix + (size_t)moving.cap * k * 9);
    if (tid == 0) s_count = 0;
    __syncthreads();
    for (int i = tid; i < M; i += BLOCK) {
      const float4* q = reinterpret_cast<const float4*>(mcells + i);
      const float4 a = q[0], b = q[1], c = q[2];
      float* o = lmov + i * 9;
      o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; o[8] = c.x;
    }
    int local = 0;
    for (int sidx = tid; sidx < n_slots; sidx += BLOCK) {
      int ci = pc[sidx];
      if (ci >= fixed.cap) ci = -1;
      lvalid[sidx] = ci;
      if (ci >= 0) {
        const float4* q = reinterpret_cast<const float4*>(fcells + ci);
        const float4 a = q[0], b = q[1], c = q[2];
        float* o = lfix + sidx * 9;
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; o[8] = c.x;
        ++local;
      }
    }
    if (local) atomicAdd(&s_count, local);
    __syncthreads();
    n_res = s_count;
    S.mov = lmov;
    S.fix = lfix;
    S.valid = lvalid;
  } else {
    // ---- read the 48-byte records in place (L1/L2 resident after the first pass)
    int local = 0;
    for (int sidx = tid; sidx < n_slots; sidx += BLOCK) {
      const int ci = pc[sidx];
      local += (ci >= 0 && ci < fixed.cap) ? 1 : 0;
    }
    if (BLOCK == 64) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
      n_res = local;
    } else {
      if (tid == 0) s_count = 0;
      __syncthreads();
      if (local) atomicAdd(&s_count, local);
      __syncthreads();
      n_res = s_count;
    }
    S.mov = reinterpret_cast<const float*>(mcells);
    S.fix = reinterpret_cast<const float*>(fcells);
    S.valid = pc;
  }

  double* tr = trace ? trace + (size_t)pair * trace_len : nullptr;
  if (tr && tid == 0) tr[0] = 0.0;

  randt_result res;
  res.cost = res.final_cost = res.initial_cost = res.mu0 = 0.0;
  res.n_residuals = n_res;
  res.iterations = res.gnc_solves = res.n_evals = 0;
  res.termination = RANDT_TERM_NONE;
  res.status = 0;
  res.reserved[0] = res.reserved[1] = 0;

  double x[4], best[4];
  {
    const double p0 = pose4[4 * (size_t)pair + 0], p1 = pose4[4 * (size_t)pair + 1];
    const double p2 = pose4[4 * (size_t)pair + 2], p3 = pose4[4 * (size_t)pair + 3];
    if (PARAM == RANDT_PARAM_VECTOR) {
      x[0] = p2; x[1] = p3; x[2] = atan2(p1, p0); x[3] = 0.0;  // trans.log()(2), ndt_matcher.cpp:439
    } else {
      x[0] = p0; x[1] = p1; x[2] = p2; x[3] = p3;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) best[i] = x[i];

  if (n_res == 0) {
    // "WARNING: NO RESIDUALS ADDED!" (ndt_matcher.cpp:454-456): pose unchanged
    res.status = 1;
    if (tid == 0) results[pair] = res;
    return;
  }

  // ---- raw residuals at the initial point -> gnc_mu (ndt_matcher.cpp:466-476)
  Loss L = make_loss(P.loss_a, P.alpha, 1.0, P.weight);
  Sums<NT> cur, cnd;
  double raw_max = 0.0;
  bool ok = eval_pass<D, PARAM, NT, 0, BLOCK, STAGE>(S, x, L, cur, raw_max, red);
  res.n_evals++;
  double gnc_mu = 2.0 * (raw_max * raw_max) / (P.mu_scale * P.mu_scale);
  gnc_mu = fmin(gnc_mu, pow(P.gnc_div, (double)(P.gnc_steps - 1)));
  res.mu0 = gnc_mu;
  int term = RANDT_TERM_FAILURE;
  double summary_min = 0.0;
  if (!ok) res.status = 2;

  if (ok) {
    do {
      gnc_mu = fmax(gnc_mu, 1.0);
      L = make_loss(P.loss_a, P.alpha, gnc_mu, P.weight);
      // ================= one ceres::Solve (TrustRegionMinimizer::Minimize) =================
      double sigma[NT], diag[NT], step[NT], delta[NT], cand[4];
      double radius = P.r0, decrease = 2.0;
      bool reuse = false, step_ok = true;
      int num_invalid = 0, iteration = 0;
      double minimum_cost = DBL_MAX;
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = best[i];
      double x_norm = ambient_norm<PARAM>(x);
      bool e_ok = eval_pass<D, PARAM, NT, 1, BLOCK, STAGE>(S, x, L, cur, raw_max, red);
      res.n_evals++;
      res.iterations++;
      if (!e_ok) {
        term = RANDT_TERM_FAILURE;
        res.status = 2;
        res.gnc_solves++;
        break;
      }
      if (res.gnc_solves == 0) res.initial_cost = cur.cost;
      summary_min = cur.cost;
#pragma unroll
      for (int i = 0; i < NT; ++i) sigma[i] = 1.0 / (1.0 + sqrt(cur.h[hix<NT>(i, i)]));
      double gmax = grad_max_norm<PARAM, NT>(x, cur.g);
      trace_push(tr, trace_len, cur.cost, radius, 0);

      for (;;) {
        // ---- FinalizeIterationAndCheckIfMinimizerCanContinue
        if (step_ok && cur.cost < minimum_cost) {
          minimum_cost = cur.cost;
#pragma unroll
          for (int i = 0; i < 4; ++i) best[i] = x[i];
        }
        if (iteration >= P.max_it) { term = RANDT_TERM_NO_CONVERGENCE; break; }
        if (step_ok && gmax <= P.gtol) { term = RANDT_TERM_CONVERGENCE_GRADIENT; break; }
        if (radius <= P.rmin) { term = RANDT_TERM_CONVERGENCE_RADIUS; break; }
        ++iteration;
        res.iterations++;

        // ---- LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled normal equations
        double A[NT * NT], gs[NT], Hs[NT * NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          gs[i] = cur.g[i] * sigma[i];
#pragma unroll
          for (int j = 0; j < NT; ++j) Hs[i * NT + j] = cur.h[hix<NT>(i, j)] * sigma[i] * sigma[j];
        }
        if (!reuse) {
#pragma unroll
          for (int i = 0; i < NT; ++i) diag[i] = fmin(fmax(Hs[i * NT + i], P.dmin), P.dmax);
        }
        const double inv_radius = 1.0 / radius;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
          for (int j = 0; j < NT; ++j) A[i * NT + j] = Hs[i * NT + j];
          A[i * NT + i] += diag[i] * inv_radius;  // (sqrt(D^2/radius))^2
        }
        bool solved = chol_solve<NT>(A, gs, step);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          if (!isfinite(step[i])) solved = false;
          step[i] = -step[i];
        }
        reuse = true;
        // model_cost_change = -(J step)^T (r + J step / 2) = -(step.g + step^T H step / 2)
        double mcc = 0.0;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          double hs = 0.0;
#pragma unroll
          for (int j = 0; j < NT; ++j) hs += Hs[i * NT + j] * step[j];
          mcc += step[i] * (gs[i] + 0.5 * hs);
        }
        mcc = -mcc;
        const bool valid = solved && mcc > 0.0;
        if (!valid) {
          // ---- HandleInvalidStep
          if (++num_invalid >= P.max_invalid) { term = RANDT_TERM_FAILURE; break; }
          radius = radius / decrease;
          decrease *= 2.0;
          reuse = true;
          step_ok = false;
          summary_min = fmin(summary_min, cur.cost);
          trace_push(tr, trace_len, cur.cost, radius, 3);
          continue;
        }
        num_invalid = 0;
#pragma unroll
        for (int i = 0; i < NT; ++i) delta[i] = step[i] * sigma[i];
        plus<PARAM>(x, delta, cand);

        // ---- candidate cost (+ speculative gradient / J^T J)
        bool c_ok = eval_pass<D, PARAM, NT, 1, BLOCK, STAGE>(S, cand, L, cnd, raw_max, red);
        res.n_evals++;
        const double cand_cost = c_ok ? cnd.cost : DBL_MAX;

        // ---- ParameterToleranceReached / FunctionToleranceReached (before accept/reject)
        double sn = 0.0;
        {
          constexpr int NAmb = PARAM == RANDT_PARAM_VECTOR ? 3 : 4;
#pragma unroll
          for (int i = 0; i < NAmb; ++i) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
          sn = sqrt(sn);
        }
        if (sn <= P.ptol * (x_norm + P.ptol)) { term = RANDT_TERM_CONVERGENCE_PARAMETER; break; }
        const double cost_change = cur.cost - cand_cost;
        if (fabs(cost_change) <= P.ftol * cur.cost) { term = RANDT_TERM_CONVERGENCE_FUNCTION; break; }

        const double rel = c_ok ? cost_change / mcc : -DBL_MAX;
        if (rel > P.min_rel) {
          // ---- HandleSuccessfulStep
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = cand[i];
          x_norm = ambient_norm<PARAM>(x);
          cur = cnd;
          gmax = grad_max_norm<PARAM, NT>(x, cur.g);
          step_ok = true;
          const double t = 2.0 * rel - 1.0;
          radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
          radius = fmin(P.rmax, radius);
          decrease = 2.0;
          reuse = false;
          summary_min = fmin(summary_min, cur.cost);
          trace_push(tr, trace_len, cur.cost, radius, 1);
        } else {
          step_ok = false;
          radius = radius / decrease;
          decrease *= 2.0;
          reuse = true;
          summary_min = fmin(summary_min, cand_cost);
          trace_push(tr, trace_len, cand_cost, radius, 2);
        }
      }
      res.gnc_solves++;
      gnc_mu /= P.gnc_div;
    } while (gnc_mu > 1.0 / sqrt(P.gnc_div));
  }

  res.termination = term;
  res.final_cost = summary_min;
  res.cost = summary_min / (double)n_res;
  if (tid == 0) {
    double* po = pose4 + 4 * (size_t)pair;
    if (PARAM == RANDT_PARAM_VECTOR) {
      double c = cos(best[2]), s = sin(best[2]);  // Sophus::SE2d(rot, pos), ndt_matcher.cpp:486
      so2_normalize(c, s);
      po[0] = c; po[1] = s; po[2] = best[0]; po[3] = best[1];
    } else {
      po[0] = best[0]; po[1] = best[1]; po[2] = best[2]; po[3] = best[3];
    }
    results[pair] = res;
  }
}

template <int D, int PARAM, int BLOCK, bool STAGE>
int launch_cfg(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving, int moving_first,
               int n_pairs, const int32_t* d_corr, const SolveParams& P, size_t lds, double* d_pose4,
               randt_result* d_results) {
  if (STAGE)
    RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve<D, PARAM, BLOCK, STAGE>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_solve<D, PARAM, BLOCK, STAGE>), dim3(n_pairs), dim3(BLOCK), STAGE ? lds : 0, ctx->stream, fixed,
                     d_fixed_idx, moving, moving_first, d_corr, P, d_pose4, d_results, ctx->d_trace, ctx->trace_len);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

// Geometry: `block` threads cooperate on one registration (64 = one wavefront, no barriers);
// stage = 1 copies the correspondence set into LDS first.
template <int D, int PARAM>
int launch_one(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving, int moving_first,
               int n_pairs, const int32_t* d_corr, const SolveParams& P, size_t lds, double* d_pose4,
               randt_result* d_results, int block, int stage) {
#define RANDT_CFG(B, S) \
  return launch_cfg<D, PARAM, B, S>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_corr, P, lds, d_pose4, d_results)
  if (stage) {
    if (block == 64) RANDT_CFG(64, true);
    if (block == 128) RANDT_CFG(128, true);
    RANDT_CFG(256, true);
  } else {
    if (block == 64) RANDT_CFG(64, false);
    if (block == 128) RANDT_CFG(128, false);
    RANDT_CFG(256, false);
  }
#undef RANDT_CFG
}

}  // namespace

int launch_solve(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving,
                 int moving_first, int n_pairs, const int32_t* d_corr, const randt_matcher_params* mp,
                 double* d_pose4, randt_result* d_results) {
  if (n_pairs <= 0) return RANDT_OK;
  SolveParams P;
  P.loss_a = mp->loss_scale;
  P.mu_scale = mp->mu_scale;
  P.alpha = mp->loss_alpha;
  P.weight = mp->loss_weight;
  P.gnc_div = mp->gnc_divisor;
  P.ftol = mp->function_tolerance;
  P.gtol = mp->gradient_tolerance;
  P.ptol = mp->parameter_tolerance;
  P.r0 = mp->initial_radius;
  P.rmax = mp->max_radius;
  P.rmin = mp->min_radius;
  P.min_rel = mp->min_relative_decrease;
  P.dmin = mp->min_lm_diagonal;
  P.dmax = mp->max_lm_diagonal;
  P.gnc_steps = mp->gnc_steps;
  P.max_it = mp->max_iterations;
  P.k = mp->n_neighbours;
  P.max_invalid = mp->max_consecutive_invalid_steps;
  if (P.k <= 0) return randt_set_error(ctx, RANDT_ERR_INVALID, "n_neighbours must be > 0", hipSuccess);
  const size_t lds = (size_t)moving.cap * 9 * 4 + (size_t)moving.cap * P.k * 9 * 4 + (size_t)moving.cap * P.k * 4;
  int block = ctx->solve_block, stage = ctx->solve_stage;
  if (stage && lds + 2048 > (size_t)ctx->lds_limit) stage = 0;  // too big for LDS: read cells in place
  const int d3 = mp->use_intensity ? 1 : 0;
#define RANDT_DISPATCH(DD, PP) \
  return launch_one<DD, PP>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_corr, P, lds, d_pose4, d_results, block, stage)
  switch (mp->parameterization) {
    case RANDT_PARAM_MANIFOLD:
      if (d3) RANDT_DISPATCH(3, RANDT_PARAM_MANIFOLD); else RANDT_DISPATCH(2, RANDT_PARAM_MANIFOLD);
    case RANDT_PARAM_AMBIENT4:
      if (d3) RANDT_DISPATCH(3, RANDT_PARAM_AMBIENT4); else RANDT_DISPATCH(2, RANDT_PARAM_AMBIENT4);
    case RANDT_PARAM_VECTOR:
      if (d3) RANDT_DISPATCH(3, RANDT_PARAM_VECTOR); else RANDT_DISPATCH(2, RANDT_PARAM_VECTOR);
    default:
      return randt_set_error(ctx, RANDT_ERR_INVALID, "unknown parameterization", hipSuccess);
  }
#undef RANDT_DISPATCH
}
