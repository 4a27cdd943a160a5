// Device math shared by the two fixed-lag window kernels (window.hip: the tuned <= 3-state kernel; window_gen.hip: the
// general one for smoothing_steps > 3): Sophus SE(2) pieces, the motion / IMU factors of both parameterisations with their
// analytic Jacobians, the state record layout.  Reference formulas: include/ndt_registration/ceres_residuals.h:25-83,
// 307-370, 554-679; Sophus 1.22.10 se2.hpp / so2.hpp (un-vendored).
#pragma once
#include "randt_internal.h"
#include "solve_math.h"

namespace randt_window {
using namespace randt_solve;

// ---------------------------------------------------------------- SE(2) (Sophus 1.22.10) -------
__device__ __forceinline__ void so2_normalize(double& c, double& s) {
  const double inv = fast_rsqrt(c * c + s * s);  // Newton-refined reciprocal square root (~1 ulp) instead of sqrt + 2 divisions
  c = c * inv;
  s = s * inv;
}
// Small-argument sin / cos / atan2 for the serial chains of this kernel (motion factor, Plus): ONE lane pays ~16 cycles per
// dependent fp64 operation and ocml's sincos / atan2 are ~40 operations deep.  Taylor polynomials in Estrin form (depth 6),
// error <= 2.2e-16 relative on |x| <= 0.5 (sin, cos) and |t| <= 0.125 (atan) -- the rotation between two radar scans;
// ocml beyond (uniform branch).  The reference calls libm here; all three are 1-ulp functions.
__device__ __forceinline__ double estrin8(double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7,
                                          double z, double z2, double z4) {
  return fma(z4, fma(z2, fma(c7, z, c6), fma(c5, z, c4)), fma(z2, fma(c3, z, c2), fma(c1, z, c0)));
}
__device__ __forceinline__ void sincos_small(double x, double* sp, double* cp) {
  if (uni(!(fabs(x) <= 0.5))) {
    sincos(x, sp, cp);
    return;
  }
  const double z = x * x, z2 = z * z, z4 = z2 * z2;
  const double ps = estrin8(-0.16666666666666666, 0.008333333333333333, -0.0001984126984126984, 2.7557319223985893e-06,
                            -2.505210838544172e-08, 1.6059043836821613e-10, -7.647163731819816e-13, 2.8114572543455206e-15, z, z2, z4);
  const double pc = estrin8(-0.5, 0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05, -2.755731922398589e-07,
                            2.08767569878681e-09, -1.1470745597729725e-11, 4.779477332387385e-14, z, z2, z4);
  *sp = fma(x * z, ps, x);
  *cp = fma(z, pc, 1.0);
}
__device__ __forceinline__ double atan2_small(double y, double x) {
  if (uni(!(x > 0.0 && fabs(y) <= 0.125 * x))) return atan2(y, x);
  const double t = y * fast_rcp(x);
  const double z = t * t, z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
  const double p = fma(z8, fma(0.047619047619047616, z, -0.05263157894736842),
                       estrin8(-0.3333333333333333, 0.2, -0.14285714285714285, 0.1111111111111111, -0.09090909090909091,
                               0.07692307692307693, -0.06666666666666667, 0.058823529411764705, z, z2, z4));
  return fma(t * z, p, t);
}
// raw_sc (nullable): sin / cos of theta as sincos() returned them (before the SO2 normalisation)
__device__ __forceinline__ void se2_exp(const double* xi, double* out, double* raw_sc = nullptr) {
  const double theta = xi[2];
  double c, s;
  sincos_small(theta, &s, &c);
  if (raw_sc) {
    raw_sc[0] = s;
    raw_sc[1] = c;
  }
  so2_normalize(c, s);
  double sbt, omcbt;
  if (fabs(theta) < 1e-10) {
    const double tsq = theta * theta;
    sbt = 1.0 - (1.0 / 6.0) * tsq;
    omcbt = 0.5 * theta - (1.0 / 24.0) * theta * tsq;
  } else {
    const double it = fast_rcp(theta);
    sbt = s * it;
    omcbt = (1.0 - c) * it;
  }
  out[0] = c;
  out[1] = s;
  out[2] = sbt * xi[0] - omcbt * xi[1];
  out[3] = omcbt * xi[0] + sbt * xi[1];
}
__device__ __forceinline__ void se2_mul(const double* a, const double* b, double* out) {
  double re = a[0] * b[0] - a[1] * b[1];
  double im = a[0] * b[1] + a[1] * b[0];
  const double sq = re * re + im * im;
  if (sq != 1.0) {
    const double scale = 2.0 * fast_rcp(1.0 + sq);
    re *= scale;
    im *= scale;
  }
  so2_normalize(re, im);
  const double tx = a[2] + (a[0] * b[2] - a[1] * b[3]);
  const double ty = a[3] + (a[1] * b[2] + a[0] * b[3]);
  out[0] = re;
  out[1] = im;
  out[2] = tx;
  out[3] = ty;
}
__device__ __forceinline__ void se2_inv(const double* a, double* out) {
  const double c = a[0], s = -a[1];
  const double tx = -a[2], ty = -a[3];
  out[0] = c;
  out[1] = s;
  out[2] = c * tx - s * ty;
  out[3] = s * tx + c * ty;
}
__device__ __forceinline__ void se2_log(const double* p, double* xi) {
  const double theta = atan2_small(p[1], p[0]);
  const double half = 0.5 * theta;
  const double rm1 = p[0] - 1.0;
  double hbt;
  if (fabs(rm1) < 1e-10) {
    hbt = 1.0 - (1.0 / 12) * theta * theta;
  } else {
    hbt = -(half * p[1]) * fast_rcp(rm1);
  }
  xi[0] = hbt * p[2] + half * p[3];
  xi[1] = -half * p[2] + hbt * p[3];
  xi[2] = theta;
}

// state record in LDS: [0..3] pose, [4,5] lin_vel, [6] rot_vel, [7,8] lin_acc, [9] imu_bias, [10] rot, [11] unused.
// Vector parameterisation (optimize_on_manifold: false; parameter blocks pos[2], rot[1]): the parameters are [2], [3], [10],
// and [0], [1] = cos / sin of [10] are kept up to date by Plus, so that the NDT pass reads one layout in both modes.
#define ST_STRIDE 12

// MotionModelFactorSE2 (ceres_residuals.h:621-679): UNWEIGHTED residual r[8] and Jacobian Ju[8][16]
// w.r.t. tangent [X0: pose3 v2 w1 a2 | X1: pose3 v2 w1 a2] (right perturbations).
__device__ void motion_factor(const double* x0, const double* x1, double raw_dt, double* r, double* Ju /* LDS 8x16 */) {
  const double dt = raw_dt > 0.2 ? raw_dt : 0.2;  // predictSE2 clamp (:73)
  const double xi[3] = {x0[4] * dt + 0.5 * dt * x0[7], x0[5] * dt + 0.5 * dt * x0[8], x0[6] * dt};
  double e[4], pred[4], pinv[4], E[4], lg[3], sc_w[2];
  se2_exp(xi, e, sc_w);  // sin / cos of xi[2] are needed again for d exp / d xi below
  se2_mul(x0, e, pred);
  se2_inv(pred, pinv);
  se2_mul(pinv, x1, E);
  se2_log(E, lg);
#if defined(RANDT_TIMING) && defined(WIN_FACTOR_WAVE)
  if (threadIdx.x == 64 * WIN_FACTOR_WAVE) atomicAdd((unsigned long long*)&g_randt_win_timing[15], (unsigned long long)wall_clock64());
#endif
  r[0] = lg[0];
  r[1] = lg[1];
  r[2] = lg[2];
  r[3] = x1[4] - (x0[4] + dt * x0[7]);
  r[4] = x1[5] - (x0[5] + dt * x0[8]);
  r[5] = x1[6] - x0[6];
  r[6] = x1[7] - x0[7];
  r[7] = x1[8] - x0[8];
  // (Ju was zeroed by the whole wavefront in factors_unweighted)
  const double phi = lg[2];
  const double cE = E[0], sE = E[1], tEx = E[2], tEy = E[3];
  double h, dh;  // Vinv(phi) = [[h, phi/2], [-phi/2, h]]
  if (fabs(E[0] - 1.0) < 1e-10) {
    h = 1.0 - phi * phi / 12.0;
    dh = -phi / 6.0;
  } else {
    // cot(phi/2) = (1 + cos phi) / sin phi and 1 / sin^2(phi/2) = 2 (1 + cos phi) / sin^2 phi from the unit complex
    // (cE, sE) of E itself (phi = atan2(sE, cE)): no second sincos; 1 + cos phi has no cancellation for |phi| < pi
    const double half = 0.5 * phi;
    const double isn = fast_rcp(sE), cot = (1.0 + cE) * isn;
    h = half * cot;
    dh = 0.5 * cot - half * (cot * isn);
  }
  const double Vi00 = h, Vi01 = 0.5 * phi, Vi10 = -0.5 * phi, Vi11 = h;
  const double dVt0 = dh * tEx + 0.5 * tEy, dVt1 = -0.5 * tEx + dh * tEy;
  // X1 <- X1 exp(d1): dphi = dw1, dt_E = R_E dv1
  Ju[0 * 16 + 8] = Vi00 * cE + Vi01 * sE;
  Ju[0 * 16 + 9] = Vi00 * (-sE) + Vi01 * cE;
  Ju[1 * 16 + 8] = Vi10 * cE + Vi11 * sE;
  Ju[1 * 16 + 9] = Vi10 * (-sE) + Vi11 * cE;
  Ju[0 * 16 + 10] = dVt0;
  Ju[1 * 16 + 10] = dVt1;
  Ju[2 * 16 + 10] = 1.0;
  // pred = X0 exp(xi)
  double a, b, da, db;
  if (fabs(xi[2]) < 1e-10) {
    const double w = xi[2];
    a = 1.0 - w * w / 6.0;
    b = 0.5 * w - w * w * w / 24.0;
    da = -w / 3.0;
    db = 0.5 - w * w / 8.0;
  } else {
    const double w = xi[2];
    const double s = sc_w[0], c = sc_w[1];
    const double iw = fast_rcp(w), iw2 = iw * iw;
    a = s * iw;
    b = (1.0 - c) * iw;
    da = (w * c - s) * iw2;
    db = (w * s - (1.0 - c)) * iw2;
  }
  const double c0 = x0[0], s0 = x0[1];
  const double Vx = a * xi[0] - b * xi[1], Vy = b * xi[0] + a * xi[1];
  const double dVx = da * xi[0] - db * xi[1], dVy = db * xi[0] + da * xi[1];
  double dtp[6][2], dth[6];
  dtp[0][0] = c0;  dtp[0][1] = s0;  dth[0] = 0;
  dtp[1][0] = -s0; dtp[1][1] = c0;  dth[1] = 0;
  dtp[2][0] = c0 * (-Vy) - s0 * Vx;
  dtp[2][1] = s0 * (-Vy) + c0 * Vx;
  dth[2] = 1;
  dtp[3][0] = c0 * a - s0 * b;     dtp[3][1] = s0 * a + c0 * b;   dth[3] = 0;
  dtp[4][0] = c0 * (-b) - s0 * a;  dtp[4][1] = s0 * (-b) + c0 * a; dth[4] = 0;
  dtp[5][0] = c0 * dVx - s0 * dVy;
  dtp[5][1] = s0 * dVx + c0 * dVy;
  dth[5] = 1;
  const double cp = pred[0], sp = pred[1];
  double G[6][3];
#pragma unroll
  for (int g = 0; g < 6; ++g) {
    const double ex = -(cp * dtp[g][0] + sp * dtp[g][1]) - (-tEy) * dth[g];
    const double ey = -(-sp * dtp[g][0] + cp * dtp[g][1]) - (tEx)*dth[g];
    const double dphi = -dth[g];
    G[g][0] = Vi00 * ex + Vi01 * ey + dVt0 * dphi;
    G[g][1] = Vi10 * ex + Vi11 * ey + dVt1 * dphi;
    G[g][2] = dphi;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Ju[i * 16 + 0] = G[0][i];
    Ju[i * 16 + 1] = G[1][i];
    Ju[i * 16 + 2] = G[2][i];
    Ju[i * 16 + 3] = G[3][i] * dt;
    Ju[i * 16 + 4] = G[4][i] * dt;
    Ju[i * 16 + 5] = G[5][i] * dt;
    Ju[i * 16 + 6] = G[3][i] * 0.5 * dt;
    Ju[i * 16 + 7] = G[4][i] * 0.5 * dt;
  }
  Ju[3 * 16 + 3] = -1; Ju[3 * 16 + 6] = -dt; Ju[3 * 16 + 11] = 1;
  Ju[4 * 16 + 4] = -1; Ju[4 * 16 + 7] = -dt; Ju[4 * 16 + 12] = 1;
  Ju[5 * 16 + 5] = -1; Ju[5 * 16 + 13] = 1;
  Ju[6 * 16 + 6] = -1; Ju[6 * 16 + 14] = 1;
  Ju[7 * 16 + 7] = -1; Ju[7 * 16 + 15] = 1;
}

// RotationalResidualSE2 (ceres_residuals.h:338-370): r[2], J[2][8] w.r.t. [X0 pose3, X1 pose3, b0, b1]
__device__ void imu_factor(const double* x0, const double* x1, double raw_dt, double imu_rot, double w, double wb, double* r,
                           double* J /* LDS 2x8 */) {
  const double screw[3] = {0.0, 0.0, x1[9] * raw_dt};
  double e[4], M1[4], inv0[4], E[4], lg[3];
  se2_exp(screw, e);
  se2_mul(x1, e, M1);
  se2_inv(x0, inv0);
  se2_mul(inv0, M1, E);
  se2_log(E, lg);
  r[0] = w * (imu_rot - lg[2]);
  r[1] = wb * (x1[9] - x0[9]);
  for (int i = 0; i < 16; ++i) J[i] = 0.0;
  J[2] = w;
  J[3 + 2] = -w;
  J[7] = -w * raw_dt;
  J[8 + 6] = -wb;
  J[8 + 7] = wb;
}

// NormalizeAngle (include/ndt_registration/state_manifold.h:17-23)
__device__ __forceinline__ double normalize_angle(double a) {
  const double two_pi = 2.0 * 3.14159265358979323846;
  return a - two_pi * floor((a + 3.14159265358979323846) / two_pi);
}

// MotionModelFactor (ceres_residuals.h:554-619) on the (pos, rot) blocks, with predict() (:25-55): UNWEIGHTED residual r[8]
// and Jacobian Ju[8][16] w.r.t. [X0: pos2 rot1 v2 w1 a2 | X1: the same] (what Ceres' autodiff yields; NormalizeAngle has
// derivative 1).  (Ju was zeroed by the whole wavefront once per kernel: a factor always writes the same entries.)
__device__ void motion_factor_vec(const double* x0, const double* x1, double raw_dt, double* r, double* Ju /* LDS 8x16 */) {
  const double dt = raw_dt > 0.2 ? raw_dt : 0.2;
  const double mid = normalize_angle(x0[10] + 0.5 * dt * x0[6]);
  double new_rot = x0[10];
  new_rot += dt * x0[6];
  new_rot = normalize_angle(new_rot);
  double sy, cy;
  sincos(mid, &sy, &cy);
  const double half_dt2 = 0.5 * dt * dt;
  const double delta_x = x0[4] * dt + 0.5 * x0[7] * dt * dt;
  const double delta_y = x0[5] * dt + 0.5 * x0[8] * dt * dt;
  const double dxr = cy * delta_x - sy * delta_y, dyr = sy * delta_x + cy * delta_y;
  r[0] = x1[2] - (x0[2] + dxr);
  r[1] = x1[3] - (x0[3] + dyr);
  r[2] = normalize_angle(x1[10] - new_rot);
  r[3] = x1[4] - (x0[4] + dt * x0[7]);
  r[4] = x1[5] - (x0[5] + dt * x0[8]);
  r[5] = x1[6] - x0[6];
  r[6] = x1[7] - x0[7];
  r[7] = x1[8] - x0[8];
  Ju[0 * 16 + 0] = -1; Ju[1 * 16 + 1] = -1;
  Ju[0 * 16 + 2] = dyr;               Ju[1 * 16 + 2] = -dxr;
  Ju[0 * 16 + 3] = -cy * dt;          Ju[0 * 16 + 4] = sy * dt;
  Ju[1 * 16 + 3] = -sy * dt;          Ju[1 * 16 + 4] = -cy * dt;
  Ju[0 * 16 + 5] = dyr * 0.5 * dt;    Ju[1 * 16 + 5] = -dxr * 0.5 * dt;
  Ju[0 * 16 + 6] = -cy * half_dt2;    Ju[0 * 16 + 7] = sy * half_dt2;
  Ju[1 * 16 + 6] = -sy * half_dt2;    Ju[1 * 16 + 7] = -cy * half_dt2;
  Ju[0 * 16 + 8] = 1; Ju[1 * 16 + 9] = 1;
  Ju[2 * 16 + 2] = -1; Ju[2 * 16 + 5] = -dt; Ju[2 * 16 + 10] = 1;
  Ju[3 * 16 + 3] = -1; Ju[3 * 16 + 6] = -dt; Ju[3 * 16 + 11] = 1;
  Ju[4 * 16 + 4] = -1; Ju[4 * 16 + 7] = -dt; Ju[4 * 16 + 12] = 1;
  Ju[5 * 16 + 5] = -1; Ju[5 * 16 + 13] = 1;
  Ju[6 * 16 + 6] = -1; Ju[6 * 16 + 14] = 1;
  Ju[7 * 16 + 7] = -1; Ju[7 * 16 + 15] = 1;
}

// RotationalResidual (ceres_residuals.h:307-336) on rot0, rot1, bias0, bias1; J as imu_factor's
__device__ void imu_factor_vec(const double* x0, const double* x1, double raw_dt, double imu_rot, double w, double wb, double* r,
                               double* J /* LDS 2x8 */) {
  r[0] = w * (imu_rot - normalize_angle(x1[10] - x0[10] + x1[9] * raw_dt));
  r[1] = wb * (x1[9] - x0[9]);
  for (int i = 0; i < 16; ++i) J[i] = 0.0;
  J[2] = w;
  J[3 + 2] = -w;
  J[7] = -w * raw_dt;
  J[8 + 6] = -wb;
  J[8 + 7] = wb;
}


// Entry (a, b) of T G T^T and entry a of T g_b for the manifold pose block of a state
// (T rows: [cp, sp, 0], [-sp, cp, 0], [0, 0, kappa], see solve.hip::to_param).
__device__ __forceinline__ void pose_T(const double* xp, double T[3][3], int vec) {
  if (vec) {  // (pos, rot) blocks: the base Jacobian w.r.t. (tx, ty, theta) IS the block's Jacobian
    T[0][0] = 1; T[0][1] = 0; T[0][2] = 0;
    T[1][0] = 0; T[1][1] = 1; T[1][2] = 0;
    T[2][0] = 0; T[2][1] = 0; T[2][2] = 1;
    return;
  }
  const double cp = xp[0], sp = xp[1];
  const double n2 = cp * cp + sp * sp;
  const double a = -sp / n2, b = cp / n2;
  T[0][0] = cp;  T[0][1] = sp; T[0][2] = 0;
  T[1][0] = -sp; T[1][1] = cp; T[1][2] = 0;
  T[2][0] = 0;   T[2][1] = 0;  T[2][2] = a * (-sp) + b * cp;
}


// ---------------------------------------------------------------- state buffers in LDS -------
// SH: the kernel's LDS image; needs xs[2][S + 1][ST_STRIDE], off_tan[][5], off_amb[][5].
// Plus for every variable block: xs[dst] = Plus(xs[src], sign * vec), one lane per state.
template <class SH>
__device__ void plus_states(const WinDesc& W, SH& sh, int src, int dst, const double* vec, double sign, int lane) {
  if (lane <= W.S) {
    const int j = lane;
    const double* x = sh.xs[src][j];
    double* y = sh.xs[dst][j];
    if (sh.off_tan[j][0] >= 0 && W.vec) {  // plain addition on pos and rot; cos / sin follow
      y[2] = x[2] + sign * vec[sh.off_tan[j][0]];
      y[3] = x[3] + sign * vec[sh.off_tan[j][0] + 1];
      const double rot = x[10] + sign * vec[sh.off_tan[j][0] + 2];
      y[10] = rot;
      double sr, cr;
      sincos(rot, &sr, &cr);
      y[0] = cr;
      y[1] = sr;
    } else if (sh.off_tan[j][0] >= 0) {
      const double d[3] = {sign * vec[sh.off_tan[j][0]], sign * vec[sh.off_tan[j][0] + 1], sign * vec[sh.off_tan[j][0] + 2]};
      double e[4];
      se2_exp(d, e);
      se2_mul(x, e, y);
      y[10] = x[10];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = x[i];
      y[10] = x[10];
    }
    y[4] = x[4] + (sh.off_tan[j][1] >= 0 ? sign * vec[sh.off_tan[j][1]] : 0.0);
    y[5] = x[5] + (sh.off_tan[j][1] >= 0 ? sign * vec[sh.off_tan[j][1] + 1] : 0.0);
    y[6] = x[6] + (sh.off_tan[j][2] >= 0 ? sign * vec[sh.off_tan[j][2]] : 0.0);
    y[7] = x[7] + (sh.off_tan[j][3] >= 0 ? sign * vec[sh.off_tan[j][3]] : 0.0);
    y[8] = x[8] + (sh.off_tan[j][3] >= 0 ? sign * vec[sh.off_tan[j][3] + 1] : 0.0);
    y[9] = x[9] + (sh.off_tan[j][4] >= 0 ? sign * vec[sh.off_tan[j][4]] : 0.0);
  }
}

// squared norm over the VARIABLE ambient blocks of (xs[a] - xs[b]) or of xs[a] (b < 0); wave 0, all lanes get it
template <class SH>
__device__ double ambient_sq(const WinDesc& W, const SH& sh, int a, int b, int lane) {
  double v = 0.0;
  if (lane <= W.S) {
    const int j = lane;
    const int lo[5] = {0, 4, 6, 7, 9}, sz[5] = {4, 2, 1, 2, 1};
#pragma unroll
    for (int blk = 0; blk < 5; ++blk)
      if (sh.off_amb[j][blk] >= 0) {
        if (blk == 0 && W.vec) {  // ambient elements of the (pos, rot) blocks
          const int el[3] = {2, 3, 10};
          for (int e = 0; e < 3; ++e) {
            const double d = sh.xs[a][j][el[e]] - (b >= 0 ? sh.xs[b][j][el[e]] : 0.0);
            v += d * d;
          }
          continue;
        }
        for (int e = 0; e < sz[blk]; ++e) {
          const double d = sh.xs[a][j][lo[blk] + e] - (b >= 0 ? sh.xs[b][j][lo[blk] + e] : 0.0);
          v += d * d;
        }
      }
  }
  return wave_sum(v);
}

}  // namespace randt_window
