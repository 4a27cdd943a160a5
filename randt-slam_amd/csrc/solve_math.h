// fp64 device math shared by the pair solve (solve.hip) and the fixed-lag window solve (window.hip):
// Newton reciprocals, the Barron / scaled loss, the D2D residual with its base Jacobian, wavefront reductions.
// Reference formulas: include/ndt_registration/ceres_residuals.h:421-552, src/ndt_registration/ceres_loss_functions.cpp:19-39.
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>

namespace randt_solve {

// 1/x and 1/sqrt(x) to ~1 ulp: hardware seed (2^-23 relative) + two Newton-Raphson steps.  A correctly
// rounded fp64 division costs ~25 instructions on gfx950, these 5 / 9.
__device__ __forceinline__ double fast_rcp(double x) {
#pragma clang fp contract(off)
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
#pragma clang fp contract(off)
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-x * y, y, 1.0);
  return fma(0.5 * y, e, y);
}

// The solver state is the same in every lane, but it lives in VGPRs (fp64 has no scalar unit), so the compiler
// must treat a branch on it as divergent: exec-mask bookkeeping and select/copy merges at every join.  uni()
// turns such a condition into a scalar one (one compare into an SGPR pair + s_cmp): real scalar branches.
__device__ __forceinline__ bool uni(bool c) { return __ballot(c) != 0ull; }

// ---------------------------------------------------------------- loss -------------------------
struct Loss {
  double b, c, factor, exponent, pre, ts, alpha, weight, sqrt_w, half_w_pre;
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
  L.half_w_pre = 0.5 * weight * L.pre;
  L.mode = alpha >= 2.0 ? 0 : (fabs(alpha) <= 0.05 ? 1 : (alpha == -2.0 ? 2 : 3));
  return L;
}

// The same for the shipped shape alpha = -2, only what the closed-form corrector reads: factor = 4, exponent = -1,
// pre = -2 b, ts = 1 / (2 b): one division instead of four and a square root (once per GNC step, but on every lane).
__device__ __forceinline__ Loss make_loss_am2(double a, double mu, double weight) {
  Loss L;
  L.alpha = -2.0;
  L.b = mu * a * a;
  L.c = L.factor = L.exponent = L.pre = L.sqrt_w = 0.0;
  L.ts = 0.5 * (1.0 / L.b);  // 2 c / factor with c = 1 / b correctly rounded, like make_loss (the long Oxford GNC schedule is sensitive to its last bit)
  L.weight = weight;
  L.half_w_pre = -weight * L.b;
  L.mode = 2;
  return L;
}

// BarronLoss::Evaluate (ceres_loss_functions.cpp:19-39) x ScaledLoss, general branches
__device__ __forceinline__ void loss_eval(const Loss& L, double s, double& r0, double& r1, double& r2) {
  if (L.mode == 0) {
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
// One D2D residual: ssq = d^T (R Sm R^T + Sf)^-1 d and, if WANT_JAC,
// jb = r * d r / d (tx, ty, theta) with r = sqrt(ssq) (A.2 in the global frame): the 1/r factor is
// folded into the accumulation by the caller, so no square root is taken per residual.
// mv/fv: first 9 floats of a cell record (mean xyz, cov xx xy xi yy yi ii); c, s = cos/sin(theta).
// Rotation of one pass, with the products the covariance rotation needs: R S R^T is linear in (c^2, cs, s^2),
// so the 2 x 2 block of C = R Sm R^T + Sf costs nine fused multiply-adds instead of seventeen operations.
struct Rot {
  double c, s, c2, s2, cs, cs2, c2ms2;
};
__device__ __forceinline__ Rot make_rot(double c, double s) {
#pragma clang fp contract(off)
  Rot R;
  R.c = c;
  R.s = s;
  R.c2 = c * c;
  R.s2 = s * s;
  R.cs = c * s;
  R.cs2 = 2.0 * R.cs;
  R.c2ms2 = R.c2 - R.s2;
  return R;
}

// ANALYTIC: the rotation entry of jb is the one of the reference's hand-written functors (ceres_residuals.h:244-246, 294-296:
// u^T J m + (Sm u)^T (R J q) -- equal to the derivative only at theta = 0, reproduced for
// use_analytic_expressions_for_optimization: true, RANDT_PARAM_ANALYTIC).
template <int D, bool WANT_JAC, bool ANALYTIC = false>
__device__ __forceinline__ double residual_sq_v(const float4 ma, const float4 mb, const float4 mc4, const float4 fa, const float4 fb,
                                                const float4 fc4, const Rot& R, double tx, double ty, double* jb) {
#pragma clang fp contract(off)
  const float mv[9] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w, mc4.x};
  const float fv[9] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w, fc4.x};
  const double c = R.c, s = R.s;
  const double m0 = mv[0], m1 = mv[1];
  const double a = mv[3], b = mv[4], dd = mv[6];
  const double C00 = fma(R.c2, a, fma(-R.cs2, b, fma(R.s2, dd, (double)fv[3])));
  const double C11 = fma(R.s2, a, fma(R.cs2, b, fma(R.c2, dd, (double)fv[6])));
  const double C01 = fma(R.cs, a - dd, fma(R.c2ms2, b, (double)fv[4]));
  const double d0 = fma(c, m0, fma(-s, m1, tx - (double)fv[0]));
  const double d1 = fma(s, m0, fma(c, m1, ty - (double)fv[1]));
  double q0, q1, q2 = 0.0, ssq;
  double cc = 0.0, e = 0.0;
  // Every product-sum below is an explicit fma / mul / add under `fp contract(off)`: the same residual is evaluated by
  // different wavefronts in different kernels (k_solve's split mode, the window solve) and must round identically everywhere
  // -- the compiler's own contraction picks different fusions for the same expression in different surroundings.
  if (D == 3) {
    cc = mv[5];
    e = mv[7];
    const double C02 = fma(c, cc, fma(-s, e, (double)fv[5]));
    const double C12 = fma(s, cc, fma(c, e, (double)fv[7]));
    const double C22 = (double)mv[8] + (double)fv[8];
    const double d2 = (double)mv[2] - (double)fv[2];
    const double k00 = fma(C11, C22, -(C12 * C12));
    const double k01 = fma(C12, C02, -(C01 * C22));
    const double k02 = fma(C01, C12, -(C11 * C02));
    const double det = fma(C02, k02, fma(C01, k01, C00 * k00));
    const double id = fast_rcp(det);
    const double k11 = fma(C00, C22, -(C02 * C02));
    const double k12 = fma(C02, C01, -(C00 * C12));
    const double k22 = fma(C00, C11, -(C01 * C01));
    q0 = fma(k02, d2, fma(k01, d1, k00 * d0)) * id;
    q1 = fma(k12, d2, fma(k11, d1, k01 * d0)) * id;
    q2 = fma(k22, d2, fma(k12, d1, k02 * d0)) * id;
    ssq = fma(d2, q2, fma(d1, q1, d0 * q0));
  } else {
    const double det = fma(C00, C11, -(C01 * C01));
    const double id = fast_rcp(det);
    q0 = fma(C11, d0, -(C01 * d1)) * id;
    q1 = fma(C00, d1, -(C01 * d0)) * id;
    ssq = fma(d1, q1, d0 * q0);
  }
  if (WANT_JAC) {
    const double u0 = fma(s, q1, c * q0), u1 = fma(c, q1, -(s * q0));
    double Su0 = fma(b, u1, a * u0), Su1 = fma(dd, u1, b * u0);
    if (D == 3) {
      Su0 = fma(cc, q2, Su0);
      Su1 = fma(e, q2, Su1);
    }
    jb[0] = q0;
    jb[1] = q1;
    if (ANALYTIC) {
      const double w0 = -fma(c, q1, s * q0), w1 = fma(c, q0, -(s * q1));  // R J q
      jb[2] = fma(u1, m0, -(u0 * m1)) + fma(Su0, w0, Su1 * w1);
    } else {
      jb[2] = fma(u1, m0, -(u0 * m1)) - fma(u1, Su0, -(u0 * Su1));
    }
  }
  return ssq;
}
template <int D, bool WANT_JAC, bool ANALYTIC = false>
__device__ __forceinline__ double residual_sq(const float4* mrec, const float4* frec, const Rot& R, double tx, double ty, double* jb) {
  // 48-byte records as three 16-byte loads each (global: dwordx4, LDS: ds_read_b128, conflict-free at stride 48)
  const float4 ma = mrec[0], mb = mrec[1], mc4 = mrec[2];
  const float4 fa = frec[0], fb = frec[1], fc4 = frec[2];
  return residual_sq_v<D, WANT_JAC, ANALYTIC>(ma, mb, mc4, fa, fb, fc4, R, tx, ty, jb);
}

// Loss + Ceres corrector (residual_block.cc / corrector.cc) of one scalar residual r = sqrt(sq) with base
// Jacobian row jb / r, added to the ten base sums acc = {cost, J^T r (3), upper J^T J (6)}.
// With r = sqrt(sq) and the true Jacobian row J = jb / r: the corrected residual is rs * r and the corrected row
// js * J, so  J^T r += (js rs) jb  and  J^T J += (js^2 / sq) jb jb^T  -- no square root.
// sq == 0: autodiff of sqrt(0) is singular in the reference (ceres_residuals.h:545); zero row instead (branch-free: jb is
// exactly zero when sq is, so clamping the divisor suffices; 1e-280 keeps weight / clamp finite for any sane weight).
// Two halves, so that a residual can be evaluated by one wavefront and accumulated by another (k_solve's split mode)
// with the SAME roundings as when one lane does both: residual_terms yields (jr, h, c), accumulate_terms adds them with
// explicit fused multiply-adds (nothing is left to the compiler's contraction choices).
//   jr = js rs, h = js^2 / sq, c = the cost term's variable part (alpha = -2: 1/u - 1, added as half_w_pre * c; else rho / 2)
template <bool AM2>
__device__ __forceinline__ void residual_terms(const Loss& L, double sq, double& jr, double& h, double& c) {
#pragma clang fp contract(off)
  if (AM2) {
    // alpha = -2: rho' = w / u^2 > 0, rho'' < 0 always => corrector is sqrt(rho') = sqrt(w) / u, rs = js.
    // ONE reciprocal serves 1 / u and 1 / sq: rP = 1 / (u^2 sq)  ->  h = w rP,  js rs = w / u^2 = h sq,  1 / u = rP u sq.
    const double u = fma(sq, L.ts, 1.0);
    const double sqc = fmax(sq, 1e-280);
    const double rP = fast_rcp((u * u) * sqc);
    h = L.weight * rP;
    jr = h * sqc;
    c = fma(rP * u, sqc, -1.0);  // 1 / u - 1
  } else {
    double rs, js;  // residual and Jacobian scale of the corrector (equal for a scalar residual when rho'' <= 0)
    double r0, r1, r2;
    loss_eval(L, sq, r0, r1, r2);
    c = 0.5 * r0;
    const double sqrt_rho1 = sqrt(r1);
    if (sq == 0.0 || r2 <= 0.0) {
      rs = js = sqrt_rho1;
    } else {
      const double Dc = 1.0 + 2.0 * sq * r2 / r1;
      const double al = 1.0 - sqrt(Dc);
      rs = sqrt_rho1 / (1 - al);
      js = sqrt_rho1 * (1.0 - al);  // J - (alpha/sq) r r^T J for a scalar residual
    }
    jr = js * rs;
    h = js * js * fast_rcp(fmax(sq, 1e-280));
  }
}
template <bool AM2>
__device__ __forceinline__ void accumulate_terms(const Loss& L, double jr, double h, const double* jb, double c, double* acc) {
#pragma clang fp contract(off)
  acc[0] = AM2 ? fma(L.half_w_pre, c, acc[0]) : acc[0] + c;
  const double h0 = h * jb[0], h1 = h * jb[1], h2 = h * jb[2];
  acc[1] = fma(jr, jb[0], acc[1]);
  acc[2] = fma(jr, jb[1], acc[2]);
  acc[3] = fma(jr, jb[2], acc[3]);
  acc[4] = fma(h0, jb[0], acc[4]);
  acc[5] = fma(h0, jb[1], acc[5]);
  acc[6] = fma(h0, jb[2], acc[6]);
  acc[7] = fma(h1, jb[1], acc[7]);
  acc[8] = fma(h1, jb[2], acc[8]);
  acc[9] = fma(h2, jb[2], acc[9]);
}
template <bool AM2>
__device__ __forceinline__ void accumulate_residual(const Loss& L, double sq, const double* jb, double* acc) {
  double jr, h, c;
  residual_terms<AM2>(L, sq, jr, h, c);
  accumulate_terms<AM2>(L, jr, h, jb, c, acc);
}

// ---------------------------------------------------------------- reductions -------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  // every control used here is a full permutation inside a row, so all lanes are written: no "old" value to set up
  // (update_dpp with old = 0 costs a v_mov per half in front of every DPP move)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Sum over the 64 lanes, result in every lane; fixed association order.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror -> every lane holds its row's (16-lane) sum
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
// The ten base sums at once: instead of ten independent butterflies (23 instructions each) the values are
// folded pairwise with the gfx950 lane-swap instructions -- after the 32-lane swap one register carries two
// values' half-sums, after the 16-lane (row) swap four values' quarter-sums -- so only three registers go
// through the four in-row DPP steps.  ~80 instructions instead of ~230; fixed association order.
__device__ __forceinline__ double swap_add32(double a, double b) {
  // v_permlane32_swap: lanes [32,63] of a <-> lanes [0,31] of b; a + b = {a over lane pairs | b over lane pairs}
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {
  // v_permlane16_swap: odd rows of a <-> even rows of b
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror -> every lane holds its row's (16-lane) sum
  return v;
}
__device__ __forceinline__ void wave_sum10(double* v) {
  const double w0 = swap_add32(v[0], v[1]), w1 = swap_add32(v[2], v[3]), w2 = swap_add32(v[4], v[5]);
  const double w3 = swap_add32(v[6], v[7]), w4 = swap_add32(v[8], v[9]);
  // rows of u0: v0 v2 v1 v3; u1: v4 v6 v5 v7; u2: v8 - v9 -
  const double u0 = row_sum(swap_add16(w0, w1)), u1 = row_sum(swap_add16(w2, w3)), u2 = row_sum(swap_add16(w4, 0.0));
  v[0] = readlane_f64(u0, 0);
  v[2] = readlane_f64(u0, 16);
  v[1] = readlane_f64(u0, 32);
  v[3] = readlane_f64(u0, 48);
  v[4] = readlane_f64(u1, 0);
  v[6] = readlane_f64(u1, 16);
  v[5] = readlane_f64(u1, 32);
  v[7] = readlane_f64(u1, 48);
  v[8] = readlane_f64(u2, 0);
  v[9] = readlane_f64(u2, 32);
}
// Maximum over the 64 lanes, result in every lane: four in-row DPP steps and four row broadcasts (no LDS crossbar).
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
// Does any lane raise the flag?  One ballot.
__device__ __forceinline__ double wave_any(bool flag) { return __ballot(flag) != 0ull ? 1.0 : 0.0; }


}  // namespace randt_solve
