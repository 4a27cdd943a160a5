// General fixed-lag window solve for gfx950: Matcher::estimateTransformCeres with smoothing_steps > 3.
//
// The shipped configurations all smooth over three scans (config/parameters_*.yaml: smoothing_steps: 3) and window.hip is
// built around that size (<= 32 tangent dimensions, <= 6 NDT terms, state blocks in 16-lane DPP rows).  The reference itself
// takes any lag (src/ndt_registration/ndt_matcher.cpp:343 smoothing_steps_iter = min(trajectory.size() - 1, smoothing_steps);
// the parameter is read without a bound, src/ndt_slam/ndt_slam.cpp:576), so this kernel covers the rest: 4..7 optimised
// states, <= 2 fixed maps (<= 14 NDT terms), <= 68 tangent dimensions.  Same problem wiring, factors (window_math.h), loss,
// GNC loop and Ceres 2.1.0 trust-region decisions as window.hip -- the control flow below is that kernel's, statement for
// statement -- with data structures sized by the lag instead of hand-placed:
//   * one 512-thread workgroup per window.  Wavefronts 0..6 stream the NDT terms (term q on wavefront q mod 7: cell
//     records straight from device memory, ten fp64 base sums per term); wavefront 7 evaluates the motion / IMU factors, one
//     lane per factor, at the same time;
//   * sqrtI is applied to the factor blocks in place by all threads (no diagonal special case);
//   * J^T J is block tridiagonal (a motion factor couples state f with f + 1, an NDT term touches one pose block): it is
//     kept as a band of half-width 17 (two 9-dimensional state blocks), 35 doubles per row, with its Jacobi-scaled copy;
//   * the damped solve is a banded Cholesky factorisation in LDS by ONE wavefront (right-looking: per pivot 17 column
//     entries and the 153 entries of the trailing triangle, three per lane; the right-hand side rides along as forward
//     substitution, the back substitution is column-oriented too), no workgroup barriers inside.  As in window.hip the
//     wavefronts 0..5 solve the running radius and the radii the next five rejections lead to when a rejection chain is due
//     (the reference's initial radius of 1e4 is rejected five times at the start of every GNC stage);
//   * candidate points are evaluated with their Jacobians, so an accepted step costs one pass.
// Not tuned beyond that: ~12 us per damped solve against window.hip's ~1; it exists so that no lag the reference accepts
// is refused.  RANDT_WINDOW_GENERAL=1 routes the three-state windows here as well (tests: the two kernels agree).
#include <float.h>

#include "randt_internal.h"
#include "solve_math.h"
#include "window_math.h"

#define GEN_BLOCK 512
#define GEN_WAVES 8
#define GEN_NDT_WAVES 7
#define GEN_FACTOR_WAVE 7
// The sizes below are this translation unit's; window_gen_big.hip compiles the same source a second time for 8..12
// optimised states (GEN_LEVELS 1: the rejection chain's radii are then solved one after the other -- the LDS that the
// six workspaces of the 4..7 instantiation take goes into the longer band).
#ifndef GEN_SMAX
#define GEN_SMAX 7
#define GEN_NMAX 72
#define GEN_TMAX 14
#define GEN_LEVELS 6
#define GEN_LAUNCHER launch_solve_window_gen
#define GEN_LIMIT_TEXT "window too large for the device solver (<= 7 optimised states, <= 2 fixed maps)"
#endif
#define GEN_HB 17                 // half bandwidth: columns of two adjacent 9-dimensional state blocks are at most 17 apart
#define GEN_BW (2 * GEN_HB + 1)   // stored band: column b of row a at [a][b - a + GEN_HB]
static_assert(GEN_SMAX * 16 + GEN_SMAX * 8 <= GEN_BLOCK && GEN_NMAX <= 128 && GEN_SMAX + 1 <= RANDT_WIN_MAX_STATES && GEN_TMAX <= RANDT_WIN_MAX_TERMS,
              "thread partitions of gen_weight / gen_assemble and the descriptor's arrays");

namespace {
using namespace randt_solve;
using namespace randt_window;

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct GShared {
  double xs[2][GEN_SMAX + 1][ST_STRIDE];  // states: buffer p = current, 1 - p = candidate
  double Ju[2][GEN_SMAX][128];            // motion Jacobians (8 x 16), weighted in place by gen_weight
  double ru[2][GEN_SMAX][8];              // unweighted / weighted motion residuals
  double rw[2][GEN_SMAX][8];
  double J2[2][GEN_SMAX][16];             // IMU factors
  double r2[2][GEN_SMAX][2];
  double tsum[2][GEN_TMAX][10];           // ten base sums of NDT term q at the point in buffer [.]
  double HB[GEN_NMAX][GEN_BW];            // J^T J, band
  double HS[GEN_NMAX][GEN_BW];            // Jacobi-scaled
  double LW[GEN_LEVELS][GEN_NMAX][GEN_HB + 1];  // Cholesky workspace of level q: row i, column j = i - 17 + c at [i][c]
  double yw[GEN_LEVELS][GEN_NMAX];        // right-hand side of level q
  double g[GEN_NMAX], gs[GEN_NMAX], sigma[GEN_NMAX], diag[GEN_NMAX];
  double step[GEN_LEVELS][GEN_NMAX], delta[GEN_LEVELS][GEN_NMAX];
  double solved[GEN_LEVELS];
  double red[GEN_WAVES][4];
  double scal[8];  // 0 mcc, 1 sn2, 2 x_norm, 4 gconv, 7 raw max
  int lcol[GEN_SMAX][GEN_NMAX];   // tangent column -> local column of motion factor f (-1 none)
  int lcol2[GEN_SMAX][GEN_NMAX];  // ... of IMU factor f
  int pose_of[GEN_NMAX];          // tangent column -> state whose pose block holds it (-1 none)
  int state_of[GEN_NMAX];         // tangent column -> state
  int off_tan[RANDT_WIN_MAX_STATES][5], off_amb[RANDT_WIN_MAX_STATES][5];
  int term_first[GEN_SMAX + 2];   // NDT terms of state j: [term_first[j], term_first[j + 1]) (built in state order, api.hip)
  Loss loss;
};

// base sum i of state j at the point in buffer buf: its terms in term order
__device__ __forceinline__ double state_sum(const GShared& sh, int buf, int j, int i) {
  double a = 0.0;
  for (int q = sh.term_first[j]; q < sh.term_first[j + 1]; ++q) a += sh.tsum[buf][q][i];
  return a;
}

// Motion / IMU factors at xs[buf]: one lane per factor, the rest of the wavefront clears the blocks first (the factors
// write the structurally non-zero entries only and gen_weight overwrites the blocks with sqrtI * J).
__device__ void gen_factors(const WinDesc& W, GShared& sh, int buf) {
  const int lane = threadIdx.x & 63;
  for (int e = lane; e < W.S * 128; e += 64) (&sh.Ju[buf][0][0])[e] = 0.0;
  wave_fence();
  if (lane < W.S) {
    const int f = lane;  // factor between states f and f + 1
    double r[8];
    if (W.vec) motion_factor_vec(sh.xs[buf][f], sh.xs[buf][f + 1], W.raw_dt[f + 1], r, sh.Ju[buf][f]);
    else motion_factor(sh.xs[buf][f], sh.xs[buf][f + 1], W.raw_dt[f + 1], r, sh.Ju[buf][f]);
#pragma unroll
    for (int i = 0; i < 8; ++i) sh.ru[buf][f][i] = r[i];
    if (W.use_imu) {
      double r2[2];
      if (W.vec) imu_factor_vec(sh.xs[buf][f], sh.xs[buf][f + 1], W.raw_dt[f + 1], W.imu[f], W.w_imu, W.w_bias, r2, sh.J2[buf][f]);
      else imu_factor(sh.xs[buf][f], sh.xs[buf][f + 1], W.raw_dt[f + 1], W.imu[f], W.w_imu, W.w_bias, r2, sh.J2[buf][f]);
      sh.r2[buf][f][0] = r2[0];
      sh.r2[buf][f][1] = r2[1];
    }
  }
}

// One pass over every NDT term at the states in xs[buf].  MODE 0: max raw residual -> *raw_out; MODE 1: ten base sums per
// term -> tsum[buf], while the factor wavefront evaluates the motion / IMU factors of the same point (and, for a candidate,
// ||x_candidate - x||^2 -> scal[1]).  Returns false when a residual was not finite.
template <int D, int MODE, bool AM2, bool ANALYTIC>
__device__ bool gen_pass(const MapView& fixed, const MapView& moving, const WinDesc& W, const int32_t* __restrict__ corr, GShared& sh,
                         int buf, double* raw_out, int step_from) {
  const Loss L = sh.loss;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double mx = -DBL_MAX;
  int bad = 0;
  if (MODE == 1 && wave == GEN_FACTOR_WAVE) {
    gen_factors(W, sh, buf);
    if (step_from >= 0) {
      const double sn2 = ambient_sq(W, sh, step_from, buf, lane);
      if (lane == 0) sh.scal[1] = sn2;
    }
  }
  if (wave < GEN_NDT_WAVES) {
    const int k = W.k;
    const unsigned kmagic = k > 1 ? (unsigned)((0x100000000ull + (unsigned)k - 1) / (unsigned)k) : 0u;
    for (int q = wave; q < W.n_terms; q += GEN_NDT_WAVES) {
      const int mmap = W.term_moving[q], fmap = W.term_fixed[q];
      int M = moving.counts[mmap];
      M = M > moving.cap ? moving.cap : M;
      const int n_slots = M * k;
      const float4* mov = reinterpret_cast<const float4*>(moving.cells + (size_t)mmap * moving.cap);
      const float4* fix = reinterpret_cast<const float4*>(fixed.cells + (size_t)fmap * fixed.cap);
      const int32_t* pc = corr + (size_t)q * moving.cap * k;
      const double* xp = sh.xs[buf][W.term_state[q]];
      const double inv = fast_rsqrt(xp[0] * xp[0] + xp[1] * xp[1]);
      const double c = xp[0] * inv, s = xp[1] * inv, tx = xp[2], ty = xp[3];
      const Rot rot = make_rot(c, s);
      double a10[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) a10[i] = 0.0;
      for (int s0 = 0; s0 < n_slots; s0 += 64) {
        const int slot = s0 + lane;
        const int ci = slot < n_slots ? pc[slot] : -1;
        if (!(ci >= 0 && ci < fixed.cap)) continue;
        const unsigned mi = k == 1 ? (unsigned)slot : __umulhi((unsigned)slot, kmagic);  // slot / k
        const float4* mv = mov + (size_t)mi * 3;
        const float4* fv = fix + (size_t)ci * 3;
        float4 mrec[3], frec[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          mrec[t] = mv[t];
          frec[t] = fv[t];
        }
        double jb[3];
        const double sq = residual_sq<D, MODE == 1, ANALYTIC>(mrec, frec, rot, tx, ty, jb);
        if (!(MODE == 1 && AM2) && !isfinite(sq)) bad = 1;  // closed-form loss: a non-finite residual shows in the cost sum
        if (MODE == 0) mx = sq > mx ? sq : mx;
        else accumulate_residual<AM2>(L, sq, jb, a10);
      }
      if (MODE == 1) {
        wave_sum10(a10);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < 10; ++i) sh.tsum[buf][q][i] = a10[i];
        }
      }
    }
  }
  double badf = wave_any(bad != 0);
  if (MODE == 0) mx = wave_max(mx);
  if (lane == 0) {
    sh.red[wave][0] = badf;
    sh.red[wave][1] = mx;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < GEN_WAVES; ++w) badf = sh.red[w][0] > badf ? sh.red[w][0] : badf;
  if (MODE == 0) {
#pragma unroll
    for (int w = 0; w < GEN_WAVES; ++w) mx = sh.red[w][1] > mx ? sh.red[w][1] : mx;
    if (threadIdx.x == 0) *raw_out = mx > 0.0 ? sqrt(mx) : 0.0;
    __syncthreads();
  }
  return uni(badf == 0.0);
}

// residuals_map.applyOnTheLeft(sqrtI_) (ceres_residuals.h:676) on the blocks gen_factors left at xs[buf], in place, one
// thread per Jacobian column / residual.  Returns the factors' 1/2 |r|^2 (identical in every thread).
__device__ double gen_weight(const WinDesc& W, GShared& sh, int buf) {
  const int tid = threadIdx.x;
  if (tid < W.S * 16) {
    const int f = tid >> 4, c = tid & 15;
    double col[8], out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) col[k] = sh.Ju[buf][f][k * 16 + c];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) a += W.sqrtI[i * 8 + k] * col[k];
      out[i] = a;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sh.Ju[buf][f][i * 16 + c] = out[i];
  } else if (tid >= GEN_SMAX * 16 && tid < GEN_SMAX * 16 + W.S * 8) {
    const int f = (tid - GEN_SMAX * 16) >> 3, i = (tid - GEN_SMAX * 16) & 7;
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += W.sqrtI[i * 8 + k] * sh.ru[buf][f][k];
    sh.rw[buf][f][i] = a;
  }
  __syncthreads();
  double cost = 0.0;
  for (int f = 0; f < W.S; ++f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) cost += 0.5 * sh.rw[buf][f][i] * sh.rw[buf][f][i];
    if (W.use_imu) cost += 0.5 * sh.r2[buf][f][0] * sh.r2[buf][f][0] + 0.5 * sh.r2[buf][f][1] * sh.r2[buf][f][1];
  }
  return cost;
}

// J^T J (band) and J^T r at xs[buf] from the weighted factor blocks and the per-state NDT base sums; first: the Jacobi
// scaling of this ceres::Solve is taken from this point (trust_region_minimizer.cc: jacobian_scaling_ is computed at
// iteration zero only).  Leaves HS / gs / diag for the new point.
__device__ void gen_assemble(const WinDesc& W, GShared& sh, int buf, bool first, double dmin, double dmax) {
  const int n = W.n_tan, tid = threadIdx.x;
  for (int e = tid; e < n * (GEN_HB + 1); e += GEN_BLOCK) {
    const int a = e / (GEN_HB + 1), d = e - a * (GEN_HB + 1), b = a + d;
    if (b >= n) continue;
    double h = 0.0;
    const int sa = sh.state_of[a], sb = sh.state_of[b];
    for (int f = (sb > 0 ? sb - 1 : 0); f <= sa && f < W.S; ++f) {  // factors that hold both columns: {sa - 1, sa} & {sb - 1, sb}
      const int la = sh.lcol[f][a], lb = sh.lcol[f][b];
      if (la >= 0 && lb >= 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) h += sh.Ju[buf][f][i * 16 + la] * sh.Ju[buf][f][i * 16 + lb];
      }
      if (W.use_imu) {
        const int ma = sh.lcol2[f][a], mb = sh.lcol2[f][b];
        if (ma >= 0 && mb >= 0) h += sh.J2[buf][f][ma] * sh.J2[buf][f][mb] + sh.J2[buf][f][8 + ma] * sh.J2[buf][f][8 + mb];
      }
    }
    const int j = sh.pose_of[a];
    if (j >= 0 && sh.pose_of[b] == j) {
      double T[3][3];
      pose_T(sh.xs[buf][j], T, W.vec);
      double B[10];
#pragma unroll
      for (int i = 4; i < 10; ++i) B[i] = state_sum(sh, buf, j, i);
      const double G[3][3] = {{B[4], B[5], B[6]}, {B[5], B[7], B[8]}, {B[6], B[8], B[9]}};
      const int ia = a - sh.off_tan[j][0], ib = b - sh.off_tan[j][0];
      double v = 0.0;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) v += T[ia][p] * G[p][q] * T[ib][q];
      h += v;
    }
    sh.HB[a][GEN_HB + d] = h;
    sh.HB[b][GEN_HB - d] = h;
  }
  if (tid >= GEN_BLOCK - 128 && tid < GEN_BLOCK - 128 + n) {  // the gradient
    const int a = tid - (GEN_BLOCK - 128);
    double g = 0.0;
    const int sa = sh.state_of[a];
    for (int f = (sa > 0 ? sa - 1 : 0); f <= sa && f < W.S; ++f) {
      const int la = sh.lcol[f][a];
      if (la >= 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g += sh.Ju[buf][f][i * 16 + la] * sh.rw[buf][f][i];
      }
      if (W.use_imu) {
        const int ma = sh.lcol2[f][a];
        if (ma >= 0) g += sh.J2[buf][f][ma] * sh.r2[buf][f][0] + sh.J2[buf][f][8 + ma] * sh.r2[buf][f][1];
      }
    }
    const int j = sh.pose_of[a];
    if (j >= 0) {
      double T[3][3];
      pose_T(sh.xs[buf][j], T, W.vec);
      const int ia = a - sh.off_tan[j][0];
      g += T[ia][0] * state_sum(sh, buf, j, 1) + T[ia][1] * state_sum(sh, buf, j, 2) + T[ia][2] * state_sum(sh, buf, j, 3);
    }
    sh.g[a] = g;
  }
  __syncthreads();
  if (first) {
    if (tid < n) sh.sigma[tid] = 1.0 / (1.0 + sqrt(sh.HB[tid][GEN_HB]));
    __syncthreads();
  }
  for (int e = tid; e < n * GEN_BW; e += GEN_BLOCK) {
    const int a = e / GEN_BW, c = e - a * GEN_BW, b = a + c - GEN_HB;
    if (b < 0 || b >= n) continue;
    const double hs = sh.HB[a][c] * sh.sigma[a] * sh.sigma[b];
    sh.HS[a][c] = hs;
    if (c == GEN_HB) {
      sh.diag[a] = fmin(fmax(hs, dmin), dmax);  // LM diagonal: kept across rejected steps (the same values)
      sh.gs[a] = sh.g[a] * sh.sigma[a];
    }
  }
  __syncthreads();
}

// (t, u), 1 <= u <= t <= 17, of trailing-triangle entry e (row-major over the triangle)
__device__ __forceinline__ void tri_tu(int e, int& t, int& u) {
  int tt = 1;
  while (tt * (tt + 1) / 2 <= e) ++tt;
  t = tt;
  u = e - tt * (tt - 1) / 2 + 1;
}

// The damped solve (H_s + D / radius) y = g_s by a banded Cholesky factorisation, one wavefront, level q (its own
// workspace): step = -y, delta = step * sigma, solved[q] = pivots positive && step finite
// (LevenbergMarquardtStrategy::ComputeStep; DENSE_QR on [J; sqrt(D / radius)] solves the same normal equations).
__device__ void gen_band_solve(GShared& sh, int n, int lane, double inv_radius, int q) {
  double (*L)[GEN_HB + 1] = sh.LW[q];
  double* y = sh.yw[q];
  for (int e = lane; e < n * (GEN_HB + 1); e += 64) {
    const int i = e / (GEN_HB + 1), c = e - i * (GEN_HB + 1);
    double v = sh.HS[i][c];
    if (c == GEN_HB) v += sh.diag[i] * inv_radius;  // (sqrt(D / radius))^2
    L[i][c] = (i - GEN_HB + c >= 0) ? v : 0.0;
  }
  for (int i = lane; i < n; i += 64) y[i] = sh.gs[i];
  // this lane's entries of the trailing triangle
  int tt[3], uu[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int e = lane + 64 * s;
    tt[s] = uu[s] = 0;
    if (e < GEN_HB * (GEN_HB + 1) / 2) tri_tu(e, tt[s], uu[s]);
  }
  wave_fence();
  double okf = 1.0;
  for (int j = 0; j < n; ++j) {
    const double pj = L[j][GEN_HB];
    if (!(pj > 0.0)) okf = 0.0;
    const double rd = fast_rsqrt(pj);  // 1 / l_jj
    const int t = lane + 1;            // lanes 0..16: row j + t of column j
    const bool act = lane < GEN_HB && j + t < n;
    const double lt = act ? L[j + t][GEN_HB - t] * rd : 0.0;
    const double zj = y[j] * rd;  // forward substitution rides along
    if (act) {
      L[j + t][GEN_HB - t] = lt;
      y[j + t] = fma(-lt, zj, y[j + t]);
    }
    if (lane == 0) {
      L[j][GEN_HB] = rd;  // the reciprocal is what the substitutions need
      y[j] = zj;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const double a = __shfl(lt, tt[s] - 1), b = __shfl(lt, uu[s] - 1);
      if (tt[s] > 0 && j + tt[s] < n) L[j + tt[s]][GEN_HB - tt[s] + uu[s]] = fma(-a, b, L[j + tt[s]][GEN_HB - tt[s] + uu[s]]);
    }
    wave_fence();
  }
  // L^T x = z, column-oriented: x_j = z_j / l_jj, then z_{j - t} -= l_{j, j - t} x_j
  for (int j = n - 1; j >= 0; --j) {
    const double xj = y[j] * L[j][GEN_HB];
    const int t = lane + 1;
    if (lane < GEN_HB && j - t >= 0) y[j - t] = fma(-L[j][GEN_HB - t], xj, y[j - t]);
    if (lane == 0) y[j] = xj;
    wave_fence();
  }
  double fin = 1.0;
  for (int i = lane; i < n; i += 64) {
    const double st = -y[i];
    if (!isfinite(st)) fin = 0.0;
    sh.step[q][i] = st;
    sh.delta[q][i] = st * sh.sigma[i];
  }
  fin = 1.0 - wave_any(fin == 0.0);
  if (lane == 0) sh.solved[q] = (okf != 0.0 && fin != 0.0) ? 1.0 : 0.0;
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

template <int D, bool AM2, bool ANALYTIC>
__global__ __launch_bounds__(GEN_BLOCK) void k_solve_window_gen(MapView fixed, MapView moving, const WinDesc* __restrict__ Wp,
                                                                const int32_t* __restrict__ corr, SolveParams P,
                                                                double* __restrict__ states, randt_result* __restrict__ result,
                                                                double* trace, int trace_len, int corr_stride, int state_stride) {
  __shared__ GShared sh;
  // one workgroup per window of a batch (window.hip, k_solve_window)
  Wp += blockIdx.x;
  corr += (size_t)blockIdx.x * corr_stride;
  states += (size_t)blockIdx.x * state_stride;
  result += blockIdx.x;
  if (trace) trace += (size_t)blockIdx.x * trace_len;
  const WinDesc& W = *Wp;  // indexed dynamically (state j, term q): read from device memory on demand
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = W.n_tan, S = W.S;

  // ---- load states, build the column maps, clear the bands (entries outside the matrix stay zero)
  for (int e = tid; e < (S + 1) * ST_STRIDE; e += GEN_BLOCK) {
    sh.xs[0][e / ST_STRIDE][e % ST_STRIDE] = states[e];
    sh.xs[1][e / ST_STRIDE][e % ST_STRIDE] = states[e];
  }
  for (int e = tid; e < GEN_SMAX * GEN_NMAX; e += GEN_BLOCK) {
    sh.lcol[e / GEN_NMAX][e % GEN_NMAX] = -1;
    sh.lcol2[e / GEN_NMAX][e % GEN_NMAX] = -1;
  }
  for (int e = tid; e < GEN_NMAX * GEN_BW; e += GEN_BLOCK) {
    (&sh.HB[0][0])[e] = 0.0;
    (&sh.HS[0][0])[e] = 0.0;
  }
  if (tid < GEN_NMAX) {
    sh.pose_of[tid] = -1;
    sh.state_of[tid] = 0;
  }
  if (tid < RANDT_WIN_MAX_STATES * 5) {
    sh.off_tan[tid / 5][tid % 5] = W.off_tan[tid / 5][tid % 5];
    sh.off_amb[tid / 5][tid % 5] = W.off_amb[tid / 5][tid % 5];
  }
  __syncthreads();
  if (tid == 0) {
    const int sz[5] = {3, 2, 1, 2, 1}, lbase[4] = {0, 3, 5, 6};
    for (int f = 0; f < S; ++f)
      for (int side = 0; side < 2; ++side) {
        const int j = f + side;
        for (int blk = 0; blk < 4; ++blk)
          if (W.off_tan[j][blk] >= 0)
            for (int e = 0; e < sz[blk]; ++e) sh.lcol[f][W.off_tan[j][blk] + e] = side * 8 + lbase[blk] + e;
        if (W.off_tan[j][0] >= 0)
          for (int e = 0; e < 3; ++e) sh.lcol2[f][W.off_tan[j][0] + e] = side * 3 + e;
        if (W.off_tan[j][4] >= 0) sh.lcol2[f][W.off_tan[j][4]] = 6 + side;
      }
    for (int j = 0; j <= S; ++j)
      for (int blk = 0; blk < 5; ++blk)
        if (W.off_tan[j][blk] >= 0)
          for (int e = 0; e < sz[blk]; ++e) {
            sh.state_of[W.off_tan[j][blk] + e] = j;
            if (blk == 0) sh.pose_of[W.off_tan[j][blk] + e] = j;
          }
    int q = 0;
    for (int j = 0; j <= S + 1; ++j) {  // terms are listed in state order (randt_register_window)
      while (q < W.n_terms && W.term_state[q] < j) ++q;
      sh.term_first[j] = q;
    }
  }
  __syncthreads();

  // number of NDT residual blocks and of moving cells
  int n_res = 0;
  for (int t = 0; t < W.n_terms; ++t) {
    int M = moving.counts[W.term_moving[t]];
    M = M > moving.cap ? moving.cap : M;
    const int32_t* pc = corr + (size_t)t * moving.cap * W.k;
    for (int s = tid; s < M * W.k; s += GEN_BLOCK) {
      const int ci = pc[s];
      n_res += (ci >= 0 && ci < fixed.cap) ? 1 : 0;
    }
  }
  {
    const double v = wave_sum((double)n_res);
    if (lane == 0) sh.red[wave][2] = v;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < GEN_WAVES; ++w) tot += sh.red[w][2];
    n_res = (int)tot;
    __syncthreads();
  }
  int n_cells = 0;  // sum over optimised states (ndt_matcher.cpp:367)
  {
    int last = -1;
    for (int t = 0; t < W.n_terms; ++t)
      if (W.term_state[t] != last) {
        const int M = moving.counts[W.term_moving[t]];
        n_cells += M > moving.cap ? moving.cap : M;
        last = W.term_state[t];
      }
  }

  double* tr = trace;
  if (tr && tid == 0) tr[0] = 0.0;
  randt_result res;
  res.cost = res.final_cost = res.initial_cost = res.mu0 = 0.0;
  res.n_residuals = n_res;
  res.iterations = res.gnc_solves = res.n_evals = 0;
  res.termination = RANDT_TERM_NONE;
  res.status = 0;
  res.reserved[0] = res.reserved[1] = 0;

  // ---- raw NDT residuals -> gnc_mu (ndt_matcher.cpp:382-389)
  const double weight = n_cells > 0 ? W.ndt_weight / (double)(n_cells * W.k) : 0.0;
  if (tid == 0) sh.loss = AM2 ? make_loss_am2(P.loss_a, 1.0, weight) : make_loss(P.loss_a, P.alpha, 1.0, weight);
  __syncthreads();
  double raw_max = 0.0;
  bool ok = true;
  if (n_res > 0) {
    ok = gen_pass<D, 0, AM2, ANALYTIC>(fixed, moving, W, corr, sh, 0, &sh.scal[7], -1);
    raw_max = sh.scal[7];
    res.n_evals++;
    __syncthreads();
  }
  double gnc_mu = 2.0 * (raw_max * raw_max) / (P.mu_scale * P.mu_scale);
  gnc_mu = fmin(gnc_mu, P.mu_cap);
  res.mu0 = gnc_mu;
  int term = RANDT_TERM_FAILURE;
  double summary_min = 0.0;
  if (!ok) res.status = 2;
  int p = 0;  // current state buffer

  if (ok) {
    do {
      gnc_mu = fmax(gnc_mu, 1.0);
      __syncthreads();  // every pass of the previous stage has read its loss
      if (tid == 0) sh.loss = AM2 ? make_loss_am2(P.loss_a, gnc_mu, weight) : make_loss(P.loss_a, P.alpha, gnc_mu, weight);
      __syncthreads();
      // ================= one ceres::Solve =================
      double radius = P.r0, decrease = 2.0;
      bool step_ok = true;
      int lvl = 0, n_lvl = 0;  // damped solves in stock: level lvl of n_lvl is the running radius
      int num_invalid = 0, iteration = 0;
      double minimum_cost = DBL_MAX;
      const bool e_ok = gen_pass<D, 1, AM2, ANALYTIC>(fixed, moving, W, corr, sh, p, nullptr, -1);
      double cost = gen_weight(W, sh, p);
      res.n_evals++;
      res.iterations++;
      for (int j = 1; j <= S; ++j) cost += state_sum(sh, p, j, 0);
      if (uni(!e_ok || !isfinite(cost))) {
        term = RANDT_TERM_FAILURE;
        res.status = 2;
        res.gnc_solves++;
        break;
      }
      if (res.gnc_solves == 0) res.initial_cost = cost;
      summary_min = cost;
      gen_assemble(W, sh, p, true, P.dmin, P.dmax);
      bool fresh = true;  // a new point: gradient test and ||x|| are due
      double x_norm = 0.0;
      trace_push(tr, trace_len, cost, radius, 0);

      for (;;) {
        // Wavefront 7: gradient test and ||x|| of a freshly accepted point, while wavefronts 0.. take the damped solve; the
        // stopping tests are evaluated after the barrier that publishes both (a stop discards the solve and takes back the
        // iteration count, so the decision sequence is the reference's: max iterations, gradient, radius).
        if (fresh && wave == GEN_FACTOR_WAVE) {
          // gradient tolerance: ||x - Plus(x, -g)||_inf <= gtol
          double gm = 0.0;
          for (int i = lane; i < n; i += 64) gm = fmax(gm, fabs(sh.g[i]));
          gm = wave_max(gm);
          double gconv = 0.0;
          // the displacement of Plus(x, -g) is >= 0.4 max|g_i| (|omega| <= pi): exact test only for tiny gradients
          if (uni(!(0.4 * gm > P.gtol && gm < 3.0))) {
            plus_states(W, sh, p, 1 - p, sh.g, -1.0, lane);  // the candidate buffer is dead until the step below is taken
            wave_fence();
            double m = 0.0;
            if (lane <= S) {
              for (int e = W.vec ? 2 : 0; e < ST_STRIDE - 1; ++e) {  // vector form: [0], [1] are cos / sin of the parameter [10]
                const double d = fabs(sh.xs[p][lane][e] - sh.xs[1 - p][lane][e]);
                m = d > m ? d : m;
              }
            }
            m = wave_max(m);
            gconv = m <= P.gtol ? 1.0 : 0.0;
          }
          const double xn = ambient_sq(W, sh, p, -1, lane);
          if (lane == 0) {
            sh.scal[4] = gconv;
            sh.scal[2] = sqrt(xn);
          }
        }
        fresh = false;
        // ---- FinalizeIterationAndCheckIfMinimizerCanContinue (accepted steps are monotone: buffer p is the best point)
        if (step_ok && uni(cost < minimum_cost)) minimum_cost = cost;
        if (iteration >= P.max_it) { term = RANDT_TERM_NO_CONVERGENCE; break; }
        ++iteration;
        res.iterations++;

        // ---- LevenbergMarquardtStrategy::ComputeStep, for this radius and -- when a rejection chain is due: first iteration
        // of a stage, or behind a rejection -- for the radii the next rejections lead to, one wavefront each
        if (lvl >= n_lvl) {
          lvl = 0;
          n_lvl = (iteration <= 1 || !step_ok) ? GEN_LEVELS : 1;
          if (wave < n_lvl) {
            double rr = radius, dc = decrease;  // the reference's update, replayed: radius /= decrease; decrease *= 2
            for (int q = 0; q < wave; ++q) {
              rr = rr / dc;
              dc *= 2.0;
            }
            gen_band_solve(sh, n, lane, fast_rcp(rr), wave);
          }
        }
        const int cs = lvl;
        __syncthreads();  // publishes: step / delta / solved, gradient test and ||x||
        x_norm = sh.scal[2];
        if (step_ok && uni(sh.scal[4] != 0.0)) { res.iterations--; term = RANDT_TERM_CONVERGENCE_GRADIENT; break; }
        if (uni(radius <= P.rmin)) { res.iterations--; term = RANDT_TERM_CONVERGENCE_RADIUS; break; }
        if (wave == 1) {
          // model_cost_change = -(step.gs + step^T Hs step / 2)
          double t = 0.0;
          for (int a = lane; a < n; a += 64) {
            double hs = 0.0;
            for (int c = 0; c < GEN_BW; ++c) {
              const int b = a + c - GEN_HB;
              if (b >= 0 && b < n) hs += sh.HS[a][c] * sh.step[cs][b];
            }
            t += sh.step[cs][a] * (sh.gs[a] + 0.5 * hs);
          }
          const double mcc = -wave_sum(t);
          if (lane == 0) sh.scal[0] = mcc;
        } else if (wave == 0) {
          plus_states(W, sh, p, 1 - p, sh.delta[cs], 1.0, lane);
        }
        __syncthreads();
        const double mcc = sh.scal[0];
        const bool valid = uni(sh.solved[cs] != 0.0 && mcc > 0.0);
        if (!valid) {
          // ---- HandleInvalidStep
          if (++num_invalid >= P.max_invalid) { term = RANDT_TERM_FAILURE; break; }
          radius = radius / decrease;
          decrease *= 2.0;
          step_ok = false;
          ++lvl;  // an invalid step shrinks the radius like a rejection: the next level is that radius
          summary_min = fmin(summary_min, cost);
          trace_push(tr, trace_len, cost, radius, 3);
          __syncthreads();
          continue;
        }
        num_invalid = 0;

        // ---- candidate: factors + NDT terms with Jacobians
        const bool c_ok = gen_pass<D, 1, AM2, ANALYTIC>(fixed, moving, W, corr, sh, 1 - p, nullptr, p);
        const double sn2 = sh.scal[1];
        double cand_cost = gen_weight(W, sh, 1 - p);
        res.n_evals++;
        for (int j = 1; j <= S; ++j) cand_cost += state_sum(sh, 1 - p, j, 0);
        const bool cfin = uni(c_ok && isfinite(cand_cost));
        if (!cfin) cand_cost = DBL_MAX;

        // ---- ParameterToleranceReached / FunctionToleranceReached
        const double ptol_abs = P.ptol * (x_norm + P.ptol);
        if (uni(sn2 <= ptol_abs * ptol_abs)) { term = RANDT_TERM_CONVERGENCE_PARAMETER; break; }
        const double cost_change = cost - cand_cost;
        if (uni(fabs(cost_change) <= P.ftol * cost)) { term = RANDT_TERM_CONVERGENCE_FUNCTION; break; }
        const double rel = cfin ? cost_change / mcc : -DBL_MAX;
        if (uni(rel > P.min_rel)) {
          // ---- HandleSuccessfulStep: the candidate buffer becomes current
          p = 1 - p;
          cost = cand_cost;
          gen_assemble(W, sh, p, false, P.dmin, P.dmax);
          fresh = true;
          step_ok = true;
          const double t = 2.0 * rel - 1.0;
          radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
          radius = fmin(P.rmax, radius);
          decrease = 2.0;
          n_lvl = 0;  // new point, new normal equations
          summary_min = fmin(summary_min, cost);
          trace_push(tr, trace_len, cost, radius, 1);
        } else {
          step_ok = false;
          radius = radius / decrease;
          decrease *= 2.0;
          ++lvl;  // the step for exactly this radius may already be solved
          summary_min = fmin(summary_min, cand_cost);
          trace_push(tr, trace_len, cand_cost, radius, 2);
          __syncthreads();
        }
      }
      res.gnc_solves++;
      gnc_mu /= P.gnc_div;
    } while (uni(gnc_mu > P.mu_stop));
  }
  __syncthreads();
  for (int e = tid; e < (S + 1) * ST_STRIDE; e += GEN_BLOCK) states[e] = sh.xs[p][e / ST_STRIDE][e % ST_STRIDE];
  res.termination = term;
  res.final_cost = summary_min;
  res.cost = n_res > 0 ? summary_min / (double)n_res : 0.0;
  if (tid == 0) result[0] = res;
}

}  // namespace

int GEN_LAUNCHER(randt_ctx* ctx, const MapView& fixed, const MapView& moving, const WinDesc& desc, const WinDesc* d_desc,
                 const int32_t* d_corr, const SolveParams& P, double* d_states, randt_result* d_result, int n_windows, int corr_stride,
                 int state_stride) {
#if GEN_SMAX == 7
  if (desc.S > GEN_SMAX) return launch_solve_window_gen_big(ctx, fixed, moving, desc, d_desc, d_corr, P, d_states, d_result, n_windows, corr_stride, state_stride);
#endif
  if (desc.n_tan > GEN_NMAX || desc.S > GEN_SMAX || desc.n_terms > GEN_TMAX) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, GEN_LIMIT_TEXT, hipSuccess);
#define RANDT_GEN_LAUNCH(DD, AA, NN)                                                                                             \
  hipLaunchKernelGGL((k_solve_window_gen<DD, AA, NN>), dim3(n_windows), dim3(GEN_BLOCK), 0, ctx->stream, fixed, moving, d_desc, d_corr, P, \
                     d_states, d_result, ctx->d_trace, ctx->trace_len, corr_stride, state_stride)
  const bool am2 = P.alpha == -2.0;
  if (desc.pad_) {  // RANDT_PARAM_ANALYTIC
    if (desc.d3) {
      if (am2) RANDT_GEN_LAUNCH(3, true, true); else RANDT_GEN_LAUNCH(3, false, true);
    } else {
      if (am2) RANDT_GEN_LAUNCH(2, true, true); else RANDT_GEN_LAUNCH(2, false, true);
    }
  } else if (desc.d3) {
    if (am2) RANDT_GEN_LAUNCH(3, true, false); else RANDT_GEN_LAUNCH(3, false, false);
  } else {
    if (am2) RANDT_GEN_LAUNCH(2, true, false); else RANDT_GEN_LAUNCH(2, false, false);
  }
#undef RANDT_GEN_LAUNCH
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
