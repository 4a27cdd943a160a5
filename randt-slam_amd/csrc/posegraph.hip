// f-4: 2-D pose-graph optimisation on the device -- the GPU side of GlobalFuser::optimizePoseGraph
// (src/global_fuser/global_fuser.cpp:13-105) with PoseGraph2dErrorTerm (include/global_fuser/
// pose_graph_2d_error_term.h:33-80), optional ceres::HuberLoss, and the Ceres 2.1.0 trust-region
// Levenberg-Marquardt loop driven from the host, one small read-back per iteration.
//
// The reference hands the damped normal equations to SuiteSparse (SPARSE_NORMAL_CHOLESKY, :54-57).  Here the same
// system is factorised exactly, with an elimination order chosen for the shape a SLAM graph has -- a long odometry
// chain plus a few loop closures:
//   * "separator" poses = variable poses touched by an edge that is not chain-adjacent (|i - j| != 1) to another
//     variable pose; everything else is "interior".  Interior poses, in id order, form a block-tridiagonal system T
//     (3x3 blocks; the chain simply breaks where a separator sits between two interiors).
//   * T = L L^T is block diagonal over the chain SEGMENTS between separators: one lane per segment factorises it
//     (k_pg_factor) and one lane per (segment, right-hand side) solves T W = [C | g_int] (k_pg_chain_solve) -- a separator
//     couples only to the two interiors next to it, so a segment has just seven right-hand sides and W is [3 n_int][8].
//     Long runs are cut by extra separators every 128 poses so that the serial recurrences stay short;
//   * the Schur complement S - C^T W on the separators is a dense SPD system (k_pg_schur): right-looking blocked Cholesky
//     over the whole device -- per 32-column panel a diagonal-block factor (k_pg_potrf), a panel solve (k_pg_trsm) and a
//     tiled trailing update (k_pg_syrk); the right-hand side rides along as an extra row; k_pg_trsv back-substitutes and
//     k_pg_backsub recovers the interiors from W.
// That is block Cholesky in a nested-dissection order -- no Woodbury-style cancellation when loop closures carry
// weights of 4e4 (parameters_indoor.yaml:10).  Everything is fp64 and deterministic (gathers through host-built
// incidence lists and fixed-order reductions; no floating-point atomics).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "randt_internal.h"
#include "solve_math.h"

namespace {

using randt_solve::fast_rcp;
using randt_solve::fast_rsqrt;

struct PgDev {
  int n_poses, n_edges, n_int, n_sep, ns;
  const int32_t *ia, *ib;
  const double *meas, *sqi;
  const int32_t *inc_off, *inc_ent;  // per pose: incident (edge << 1 | side) entries, ascending edge order
  const int32_t *is_var, *int_of, *sep_of;
  const int32_t *int_pose, *sep_pose;
  int n_seg;
  const int32_t *seg_first, *seg_len;  // maximal runs of chain-adjacent interiors
  const int32_t *bndL, *bndR;  // interior m: separator bounding its segment on the left / right, or -1
  const int32_t *nbL, *nbR;    // separator q: interior index of pose-1 / pose+1, or -1
  double *x, *cand;
  double *r[2], *Ja[2], *Jb[2], *cost_e[2];
  double *Ds;        // [n_poses][6] Jacobi-scaled diagonal block (xx, xy, xt, yy, yt, tt)
  double *gs;        // [n_poses][3] scaled gradient
  double *sigma;     // [n_poses][3] Jacobi scaling, fixed per solve
  double *diagonal;  // [n_poses][3] clamp(diag(J_s^T J_s)), refreshed on successful steps
  double *E, *CL, *CR;  // per interior: H[m+1, m], H[m, sep(pose-1)], H[m, sep(pose+1)], scaled, row-major 3x3
  double *Lf, *Ff;      // per interior: l00 l10 l11 l20 l21 l22 1/l00 1/l11 1/l22 ; L[m+1, m] row-major
  double *W, *S0, *Sw, *xsep, *step;
  double* Linv;  // [ceil(ns / PG_NB)][PG_NB][PG_NB] inverses of the factor's diagonal blocks
  double *p_gabs, *p_xsq, *p_sn, *p_mcc;  // per-pose / per-edge partials
  double *scal;                            // [8]: cost cur, cost cand, mcc sum, step norm^2, x norm^2, grad max
  int32_t* flags;                          // [0] factor ok, [1] dense ok, [2] step finite
};

__device__ __forceinline__ double pg_normalize_angle(double a) {
  const double two_pi = 2.0 * M_PI;
  return a - two_pi * floor((a + M_PI) / two_pi);  // state_manifold.h:17-23
}

// One thread per residual block: residual, both 3x3 Jacobians (autodiff written out), loss correction.
__global__ __launch_bounds__(256) void k_pg_edges(PgDev d, const double* __restrict__ x, int b, int robust, double huber_a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.n_edges) return;
  const double* pa = x + 3 * (size_t)d.ia[e];
  const double* pb = x + 3 * (size_t)d.ib[e];
  const double* ms = d.meas + 3 * (size_t)e;
  const double* sq = d.sqi + 9 * (size_t)e;
  double s, c;
  sincos(pa[2], &s, &c);
  const double dx = pb[0] - pa[0], dy = pb[1] - pa[1];
  double ev[3];
  ev[0] = (c * dx + s * dy) - ms[0];
  ev[1] = (-s * dx + c * dy) - ms[1];
  ev[2] = pg_normalize_angle((pb[2] - pa[2]) - ms[2]);
  const double A[9] = {-c, -s, -s * dx + c * dy, s, -c, -c * dx - s * dy, 0.0, 0.0, -1.0};
  const double B[9] = {c, s, 0.0, -s, c, 0.0, 0.0, 0.0, 1.0};
  double r[3], Ja[9], Jb[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    r[i] = sq[i * 3 + 0] * ev[0] + sq[i * 3 + 1] * ev[1] + sq[i * 3 + 2] * ev[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Ja[i * 3 + j] = sq[i * 3 + 0] * A[0 + j] + sq[i * 3 + 1] * A[3 + j] + sq[i * 3 + 2] * A[6 + j];
      Jb[i * 3 + j] = sq[i * 3 + 0] * B[0 + j] + sq[i * 3 + 1] * B[3 + j] + sq[i * 3 + 2] * B[6 + j];
    }
  }
  const double sn = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double rho0 = sn, rho1 = 1.0;
  if (robust) {  // ceres::HuberLoss(a): b = a^2; rho'' <= 0 so the corrector is a plain sqrt(rho') scaling
    const double bb = huber_a * huber_a;
    if (sn > bb) {
      const double rr = sqrt(sn);
      rho0 = 2.0 * huber_a * rr - bb;
      rho1 = fmax(DBL_MIN, huber_a / rr);
    }
  }
  const double w = sqrt(rho1);
  d.cost_e[b][e] = 0.5 * rho0;
#pragma unroll
  for (int i = 0; i < 3; ++i) d.r[b][3 * (size_t)e + i] = r[i] * w;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    d.Ja[b][9 * (size_t)e + i] = Ja[i] * w;
    d.Jb[b][9 * (size_t)e + i] = Jb[i] * w;
  }
}

// Fixed-order block reductions: block k reduces one (array, length) pair into scal[slot]; op 0 = sum, 1 = max.
struct PgReduce {
  const double* in[4];
  int n[4], slot[4], op[4];
};
__global__ __launch_bounds__(1024) void k_pg_reduce(PgReduce R, double* __restrict__ scal) {
  __shared__ double sh[1024];
  const int k = blockIdx.x, tid = threadIdx.x;
  const double* in = R.in[k];
  const int n = R.n[k];
  const bool mx = R.op[k] == 1;
  double acc = 0.0;
  for (int i = tid; i < n; i += 1024) acc = mx ? fmax(acc, in[i]) : acc + in[i];
  sh[tid] = acc;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) sh[tid] = mx ? fmax(sh[tid], sh[tid + off]) : sh[tid] + sh[tid + off];
    __syncthreads();
  }
  if (tid == 0) scal[R.slot[k]] = sh[0];
}

// One thread per pose: gather J^T J diagonal block and J^T r over the pose's incident edges (list order), Jacobi
// scaling on the first linearisation of the solve, |g|_max and |x|^2 partials.
__global__ __launch_bounds__(256) void k_pg_pose_diag(PgDev d, int b, int first) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= d.n_poses) return;
  if (!d.is_var[v]) {
    d.p_gabs[v] = 0.0;
    d.p_xsq[v] = 0.0;
    return;
  }
  double D[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (int k = d.inc_off[v]; k < d.inc_off[v + 1]; ++k) {
    const int ent = d.inc_ent[k];
    const int e = ent >> 1;
    const double* J = ((ent & 1) ? d.Jb[b] : d.Ja[b]) + 9 * (size_t)e;
    const double* r = d.r[b] + 3 * (size_t)e;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double j0 = J[q * 3 + 0], j1 = J[q * 3 + 1], j2 = J[q * 3 + 2], rq = r[q];
      D[0] += j0 * j0; D[1] += j0 * j1; D[2] += j0 * j2;
      D[3] += j1 * j1; D[4] += j1 * j2; D[5] += j2 * j2;
      g[0] += j0 * rq; g[1] += j1 * rq; g[2] += j2 * rq;
    }
  }
  double sg[3];
  if (first) {
    sg[0] = 1.0 / (1.0 + sqrt(D[0]));
    sg[1] = 1.0 / (1.0 + sqrt(D[3]));
    sg[2] = 1.0 / (1.0 + sqrt(D[5]));
#pragma unroll
    for (int i = 0; i < 3; ++i) d.sigma[3 * (size_t)v + i] = sg[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) sg[i] = d.sigma[3 * (size_t)v + i];
  }
  d.p_gabs[v] = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  const double* xv = d.x + 3 * (size_t)v;
  d.p_xsq[v] = xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2];
  double* Ds = d.Ds + 6 * (size_t)v;
  Ds[0] = D[0] * sg[0] * sg[0]; Ds[1] = D[1] * sg[0] * sg[1]; Ds[2] = D[2] * sg[0] * sg[2];
  Ds[3] = D[3] * sg[1] * sg[1]; Ds[4] = D[4] * sg[1] * sg[2]; Ds[5] = D[5] * sg[2] * sg[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) d.gs[3 * (size_t)v + i] = g[i] * sg[i];
}

// One thread per pose: off-diagonal blocks H[v, u] = sum_e J_v^T J_u (scaled) routed to where the elimination needs
// them -- interior/interior chain neighbours (E), interior/separator couplings (CL, CR), separator/separator (S0).
// A separator thread owns its block row of S0, so no two threads write the same word.  S0 is zeroed beforehand.
__global__ __launch_bounds__(256) void k_pg_pose_offdiag(PgDev d, int b) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= d.n_poses || !d.is_var[v]) return;
  const int m = d.int_of[v], q = d.sep_of[v];
  double E[9], CL[9], CR[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) E[i] = CL[i] = CR[i] = 0.0;
  const double* sv = d.sigma + 3 * (size_t)v;
  if (q >= 0) {
    const double* Ds = d.Ds + 6 * (size_t)v;
    double* S = d.S0 + (size_t)(3 * q) * d.ns + 3 * q;
    S[0] = Ds[0]; S[1] = Ds[1]; S[2] = Ds[2];
    S[d.ns + 0] = Ds[1]; S[d.ns + 1] = Ds[3]; S[d.ns + 2] = Ds[4];
    S[2 * (size_t)d.ns + 0] = Ds[2]; S[2 * (size_t)d.ns + 1] = Ds[4]; S[2 * (size_t)d.ns + 2] = Ds[5];
  }
  for (int k = d.inc_off[v]; k < d.inc_off[v + 1]; ++k) {
    const int ent = d.inc_ent[k];
    const int e = ent >> 1, side = ent & 1;
    const int u = side ? d.ia[e] : d.ib[e];
    if (!d.is_var[u]) continue;
    const double* Jv = (side ? d.Jb[b] : d.Ja[b]) + 9 * (size_t)e;
    const double* Ju = (side ? d.Ja[b] : d.Jb[b]) + 9 * (size_t)e;
    const double* su = d.sigma + 3 * (size_t)u;
    double B[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        B[i * 3 + j] = (Jv[0 + i] * Ju[0 + j] + Jv[3 + i] * Ju[3 + j] + Jv[6 + i] * Ju[6 + j]) * (sv[i] * su[j]);
    if (m >= 0) {
      if (u == v + 1) {
        if (d.int_of[u] >= 0) {
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) E[j * 3 + i] += B[i * 3 + j];  // stored as H[m+1, m]
        } else {
#pragma unroll
          for (int i = 0; i < 9; ++i) CR[i] += B[i];
        }
      } else if (u == v - 1 && d.sep_of[u] >= 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) CL[i] += B[i];
      }
    } else {
      const int q2 = d.sep_of[u];
      if (q2 >= 0) {
        double* S = d.S0 + (size_t)(3 * q) * d.ns + 3 * q2;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) S[(size_t)i * d.ns + j] += B[i * 3 + j];
      }
    }
  }
  if (m >= 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      d.E[9 * (size_t)m + i] = E[i];
      d.CL[9 * (size_t)m + i] = CL[i];
      d.CR[9 * (size_t)m + i] = CR[i];
    }
  }
}

// LevenbergMarquardtStrategy: diagonal = clamp(diag(J_s^T J_s), min, max), refreshed only after a successful step.
__global__ __launch_bounds__(256) void k_pg_lm_diagonal(PgDev d, double dmin, double dmax) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= d.n_poses || !d.is_var[v]) return;
  const double* Ds = d.Ds + 6 * (size_t)v;
  d.diagonal[3 * (size_t)v + 0] = fmin(fmax(Ds[0], dmin), dmax);
  d.diagonal[3 * (size_t)v + 1] = fmin(fmax(Ds[3], dmin), dmax);
  d.diagonal[3 * (size_t)v + 2] = fmin(fmax(Ds[5], dmin), dmax);
}

// Block-tridiagonal Cholesky, one lane per chain segment (the chain breaks at every separator): a strictly serial
// recurrence of 3x3 blocks with the next block's operands loaded ahead of the arithmetic that depends on the previous one.
__global__ __launch_bounds__(64) void k_pg_factor(PgDev d, double inv_radius) {
  const int sgi = blockIdx.x * 64 + threadIdx.x;
  if (sgi >= d.n_seg) return;
  const int m0 = d.seg_first[sgi], n = d.seg_len[sgi];
  double F[9];
  bool prev = false;
  int ok = 1;
  double a[6], dg[3], En[9];
  auto load = [&](int m) {
    const int v = d.int_pose[m];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = d.Ds[6 * (size_t)v + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) dg[i] = d.diagonal[3 * (size_t)v + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) En[i] = d.E[9 * (size_t)m + i];
  };
  load(m0);
  for (int t = 0; t < n; ++t) {
    const int m = m0 + t;
    double a00 = a[0] + dg[0] * inv_radius, a10 = a[1], a20 = a[2];
    double a11 = a[3] + dg[1] * inv_radius, a21 = a[4], a22 = a[5] + dg[2] * inv_radius;
    double Ec[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Ec[i] = En[i];
    if (t + 1 < n) load(m + 1);
    if (prev) {
      a00 -= F[0] * F[0] + F[1] * F[1] + F[2] * F[2];
      a10 -= F[3] * F[0] + F[4] * F[1] + F[5] * F[2];
      a11 -= F[3] * F[3] + F[4] * F[4] + F[5] * F[5];
      a20 -= F[6] * F[0] + F[7] * F[1] + F[8] * F[2];
      a21 -= F[6] * F[3] + F[7] * F[4] + F[8] * F[5];
      a22 -= F[6] * F[6] + F[7] * F[7] + F[8] * F[8];
    }
    if (!(a00 > 0.0)) ok = 0;
    const double l00 = sqrt(a00), i00 = 1.0 / l00;
    const double l10 = a10 * i00, l20 = a20 * i00;
    const double t11 = a11 - l10 * l10;
    if (!(t11 > 0.0)) ok = 0;
    const double l11 = sqrt(t11), i11 = 1.0 / l11;
    const double l21 = (a21 - l20 * l10) * i11;
    const double t22 = a22 - l20 * l20 - l21 * l21;
    if (!(t22 > 0.0)) ok = 0;
    const double l22 = sqrt(t22), i22 = 1.0 / l22;
    double* Lf = d.Lf + 9 * (size_t)m;
    Lf[0] = l00; Lf[1] = l10; Lf[2] = l11; Lf[3] = l20; Lf[4] = l21; Lf[5] = l22; Lf[6] = i00; Lf[7] = i11; Lf[8] = i22;
    double* Ff = d.Ff + 9 * (size_t)m;
    if (t + 1 < n) {  // inside a segment consecutive interiors are chain-adjacent by construction
#pragma unroll
      for (int i = 0; i < 3; ++i) {  // row i of F solves F L^T = E
        const double x0 = Ec[i * 3 + 0] * i00;
        const double x1 = (Ec[i * 3 + 1] - x0 * l10) * i11;
        const double x2 = (Ec[i * 3 + 2] - x0 * l20 - x1 * l21) * i22;
        F[i * 3 + 0] = x0; F[i * 3 + 1] = x1; F[i * 3 + 2] = x2;
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) Ff[i] = F[i];
      prev = true;
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) Ff[i] = 0.0;
    }
  }
  if (!ok) d.flags[0] = 0;
}

// T W = [C | g_int], segment by segment.  T is block diagonal over segments and a separator couples only to the last
// interior of the segment on its left and the first interior of the segment on its right, so each segment has just seven
// right-hand sides: three for its left-bounding separator, three for its right-bounding one, and the gradient.
// One lane per (segment, right-hand side); W is [3 n_int][8].
__global__ __launch_bounds__(64) void k_pg_chain_solve(PgDev d) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int sgi = t >> 3, col = t & 7;
  if (sgi >= d.n_seg || col == 7) return;
  const int m0 = d.seg_first[sgi], n = d.seg_len[sgi], mlast = m0 + n - 1;
  const bool left = col < 3, right = col >= 3 && col < 6;
  const bool empty = (left && d.bndL[m0] < 0) || (right && d.bndR[m0] < 0);
  double y0 = 0.0, y1 = 0.0, y2 = 0.0;
  for (int m = m0; m <= mlast; ++m) {
    double* w = d.W + (size_t)(3 * m) * 8 + col;
    if (empty) {
      w[0] = w[8] = w[16] = 0.0;
      continue;
    }
    double b0 = 0.0, b1 = 0.0, b2 = 0.0;
    if (col == 6) {
      const int v = d.int_pose[m];
      b0 = d.gs[3 * (size_t)v + 0]; b1 = d.gs[3 * (size_t)v + 1]; b2 = d.gs[3 * (size_t)v + 2];
    } else if (left && m == m0) {
      const double* C = d.CL + 9 * (size_t)m;
      b0 = C[0 + col]; b1 = C[3 + col]; b2 = C[6 + col];
    } else if (right && m == mlast) {
      const double* C = d.CR + 9 * (size_t)m;
      b0 = C[0 + col - 3]; b1 = C[3 + col - 3]; b2 = C[6 + col - 3];
    }
    if (m > m0) {
      const double* F = d.Ff + 9 * (size_t)(m - 1);
      b0 -= F[0] * y0 + F[1] * y1 + F[2] * y2;
      b1 -= F[3] * y0 + F[4] * y1 + F[5] * y2;
      b2 -= F[6] * y0 + F[7] * y1 + F[8] * y2;
    }
    const double* L = d.Lf + 9 * (size_t)m;
    y0 = b0 * L[6];
    y1 = (b1 - L[1] * y0) * L[7];
    y2 = (b2 - L[3] * y0 - L[4] * y1) * L[8];
    w[0] = y0; w[8] = y1; w[16] = y2;
  }
  if (empty) return;
  double x0 = 0.0, x1 = 0.0, x2 = 0.0;
  for (int m = mlast; m >= m0; --m) {
    double* w = d.W + (size_t)(3 * m) * 8 + col;
    double t0 = w[0], t1 = w[8], t2 = w[16];
    if (m < mlast) {  // (L^T)[m, m+1] = F_m^T
      const double* F = d.Ff + 9 * (size_t)m;
      t0 -= F[0] * x0 + F[3] * x1 + F[6] * x2;
      t1 -= F[1] * x0 + F[4] * x1 + F[7] * x2;
      t2 -= F[2] * x0 + F[5] * x1 + F[8] * x2;
    }
    const double* L = d.Lf + 9 * (size_t)m;
    x2 = t2 * L[8];
    x1 = (t1 - L[4] * x2) * L[7];
    x0 = (t0 - L[1] * x1 - L[3] * x2) * L[6];
    w[0] = x0; w[8] = x1; w[16] = x2;
  }
}

// (C^T W) entry for separator scalar (q, jr) against the column of separator scalar (qc, jc), or the gradient column if
// qc < 0: only q's two interior chain neighbours contribute, and only if qc bounds their segment.
__device__ __forceinline__ double pg_ctw(const PgDev& d, int q, int jr, int qc, int jc) {
  double acc = 0.0;
  const int mL = d.nbL[q], mR = d.nbR[q];
  if (mL >= 0) {  // interior at pose-1: last of its segment, coupled through its CR block; q is that segment's right bound
    const int col = qc < 0 ? 6 : (qc == q ? 3 + jc : (qc == d.bndL[mL] ? jc : -1));
    if (col >= 0) {
      const double* C = d.CR + 9 * (size_t)mL;
      const double* w = d.W + (size_t)(3 * mL) * 8 + col;
      acc += C[0 + jr] * w[0] + C[3 + jr] * w[8] + C[6 + jr] * w[16];
    }
  }
  if (mR >= 0) {  // interior at pose+1: first of its segment, coupled through its CL block; q is that segment's left bound
    const int col = qc < 0 ? 6 : (qc == q ? jc : (qc == d.bndR[mR] ? 3 + jc : -1));
    if (col >= 0) {
      const double* C = d.CL + 9 * (size_t)mR;
      const double* w = d.W + (size_t)(3 * mR) * 8 + col;
      acc += C[0 + jr] * w[0] + C[3 + jr] * w[8] + C[6 + jr] * w[16];
    }
  }
  return acc;
}

// Schur complement on the separators, lower triangle, with the right-hand side stored as an extra ROW ns so that the
// factorisation's panel solves and trailing updates forward-substitute it for free.
__global__ __launch_bounds__(256) void k_pg_schur(PgDev d, double inv_radius) {
  const int ns = d.ns;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)(ns + 1) * ns) return;
  const int r = (int)(idx / (unsigned)ns), c = (int)(idx - (size_t)r * ns);
  if (r < ns) {
    if (c > r) return;
    double val = d.S0[(size_t)r * ns + c];
    if (r == c) val += d.diagonal[3 * (size_t)d.sep_pose[r / 3] + (r % 3)] * inv_radius;
    if (d.n_int > 0) val -= pg_ctw(d, r / 3, r % 3, c / 3, c % 3);
    d.Sw[(size_t)r * ns + c] = val;
  } else {
    double val = d.gs[3 * (size_t)d.sep_pose[c / 3] + (c % 3)];
    if (d.n_int > 0) val -= pg_ctw(d, c / 3, c % 3, -1, 0);
    d.Sw[(size_t)ns * ns + c] = val;
  }
}

// ---- dense Cholesky of the Schur complement: right-looking, panels of PG_NB columns, three launches per panel so
// that the O(ns^3) trailing update is spread over the whole device.  Row ns carries the right-hand side.
#define PG_NB 32

// (1) factor the diagonal block in LDS (one workgroup, thread (i, k)) and invert the factor alongside: the elimination
// runs in its square-root-free form (one barrier per column: a column is only read, never written, in its own step) and
// the same row operations applied to the identity give the inverse of the unit factor; both are scaled by D^-1/2 at the end.
// Writes L over the block and L^-1 (row-major, zero-padded to PG_NB) into Linv[k0 / PG_NB].
__global__ __launch_bounds__(PG_NB* PG_NB) void k_pg_potrf(PgDev d, int k0) {
  __shared__ double a[PG_NB][PG_NB + 1];
  __shared__ double m[PG_NB][PG_NB + 1];
  const int n = d.ns, nb = min(PG_NB, n - k0);
  const int i = threadIdx.x / PG_NB, k = threadIdx.x % PG_NB;
  double* A = d.Sw;
  a[i][k] = (i < nb && k <= i) ? A[(size_t)(k0 + i) * n + k0 + k] : (i == k ? 1.0 : 0.0);
  m[i][k] = i == k ? 1.0 : 0.0;
  __syncthreads();
  int ok = 1;
  for (int j = 0; j < PG_NB; ++j) {
    const double piv = a[j][j];
    if (!(piv > 0.0)) ok = 0;
    if (i > j) {
      const double f = a[i][j] * fast_rcp(piv);
      if (k > j && k <= i) a[i][k] -= f * a[k][j];
      if (k <= j) m[i][k] -= f * m[j][k];
    }
    __syncthreads();
  }
  const double sk = 1.0 / sqrt(a[k][k]), si = 1.0 / sqrt(a[i][i]);
  const double lik = k < i ? a[i][k] * sk : (k == i ? sqrt(a[i][i]) : 0.0);  // L = (unit factor) D^1/2
  const double vik = k <= i ? m[i][k] * si : 0.0;                           // L^-1 = D^-1/2 (unit factor)^-1
  if (i < nb && k <= i) A[(size_t)(k0 + i) * n + k0 + k] = lik;
  d.Linv[(size_t)(k0 / PG_NB) * PG_NB * PG_NB + threadIdx.x] = (i < nb && k < nb) ? vik : 0.0;
  if (threadIdx.x == 0 && !ok) d.flags[1] = 0;
}

// (2) panel solve: every row below the diagonal block (the right-hand-side row included) becomes row * L^-T -- with the
// inverse at hand a small matrix product: 64 rows per workgroup, thread = (row, 8 of the 32 columns).
__global__ __launch_bounds__(256) void k_pg_trsm(PgDev d, int k0) {
  __shared__ double v[PG_NB][PG_NB + 1];
  __shared__ double bs[PG_NB][64];
  const int n = d.ns, nb = min(PG_NB, n - k0), tid = threadIdx.x;
  double* A = d.Sw;
  const double* Li = d.Linv + (size_t)(k0 / PG_NB) * PG_NB * PG_NB;
  for (int t = tid; t < PG_NB * PG_NB; t += 256) v[t / PG_NB][t % PG_NB] = Li[t];
  const int r0 = k0 + nb + blockIdx.x * 64;
  for (int t = tid; t < 64 * PG_NB; t += 256) {
    const int rr = t / PG_NB, q = t % PG_NB;
    bs[q][rr] = (r0 + rr <= n && q < nb) ? A[(size_t)(r0 + rr) * n + k0 + q] : 0.0;
  }
  __syncthreads();
  const int rr = tid & 63, pg = tid >> 6;
  double acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.0;
  for (int q = 0; q < PG_NB; ++q) {
    const double bq = bs[q][rr];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += bq * v[pg * 8 + u][q];  // v[p][q] = 0 for q > p
  }
  if (r0 + rr <= n) {
    double* ar = A + (size_t)(r0 + rr) * n + k0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (pg * 8 + u < nb) ar[pg * 8 + u] = acc[u];
  }
}

// (3) trailing update A[i][j] -= sum_p P[i][p] P[j][p] on 64 x 64 tiles of the lower triangle (4 x 4 outputs per thread)
__global__ __launch_bounds__(256) void k_pg_syrk(PgDev d, int k0) {
  __shared__ double pi[64][PG_NB + 1], pj[64][PG_NB + 1];
  const int n = d.ns, nb = min(PG_NB, n - k0), k1 = k0 + nb;
  // linear tile index -> (bi >= bj)
  int bi = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) ++bi;
  while (bi * (bi + 1) / 2 > (int)blockIdx.x) --bi;
  const int bj = (int)blockIdx.x - bi * (bi + 1) / 2;
  const int r0 = k1 + 64 * bi, c0 = k1 + 64 * bj;
  double* A = d.Sw;
  for (int t = threadIdx.x; t < 64 * PG_NB; t += 256) {
    const int rr = t / PG_NB, p = t % PG_NB;
    pi[rr][p] = (r0 + rr <= n && p < nb) ? A[(size_t)(r0 + rr) * n + k0 + p] : 0.0;
    pj[rr][p] = (c0 + rr < n && p < nb) ? A[(size_t)(c0 + rr) * n + k0 + p] : 0.0;
  }
  __syncthreads();
  const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
#pragma unroll 8
  for (int p = 0; p < PG_NB; ++p) {
    double av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) av[u] = pi[ty * 4 + u][p];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = pj[tx + 16 * v][p];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] += av[u] * bv[v];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = r0 + ty * 4 + u;
    if (row > n) continue;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int colg = c0 + tx + 16 * v;
      if (colg >= n || (row < n && colg > row)) continue;
      A[(size_t)row * n + colg] -= acc[u][v];
    }
  }
}

// L^T z = y (y = row ns after the factorisation), blocks of PG_NB from the bottom: z_blk = L_blk^-T y_blk is a product
// with the stored inverse, then every remaining y_i takes its update from that block's rows.  The next block's inverse is
// fetched while the update runs.
__global__ __launch_bounds__(1024) void k_pg_trsv(PgDev d) {
  __shared__ double v[PG_NB][PG_NB + 1];
  __shared__ double yb[PG_NB], z[PG_NB];
  const int n = d.ns, tid = threadIdx.x;
  double* A = d.Sw;
  double* y = A + (size_t)n * n;
  const int nblk = (n + PG_NB - 1) / PG_NB;
  double vnext = d.Linv[(size_t)(nblk - 1) * PG_NB * PG_NB + tid];
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int k0 = blk * PG_NB, nb = min(PG_NB, n - k0);
    v[tid / PG_NB][tid % PG_NB] = vnext;
    if (tid < PG_NB) yb[tid] = tid < nb ? y[k0 + tid] : 0.0;
    __syncthreads();
    if (blk > 0) vnext = d.Linv[(size_t)(blk - 1) * PG_NB * PG_NB + tid];
    if (tid < PG_NB) {  // z_j = sum_{i >= j} Linv[i][j] y_i
      double acc = 0.0;
      for (int i = tid; i < PG_NB; ++i) acc += v[i][tid] * yb[i];
      z[tid] = acc;
      if (tid < nb) d.xsep[k0 + tid] = acc;
    }
    __syncthreads();
    for (int i = tid; i < k0; i += 1024) {
      double lv[PG_NB];
#pragma unroll
      for (int p = 0; p < PG_NB; ++p) lv[p] = A[(size_t)min(k0 + p, n) * n + i];  // all loads in flight; z is zero past nb
      double acc = y[i];
#pragma unroll
      for (int p = 0; p < PG_NB; ++p) acc -= lv[p] * z[p];
      y[i] = acc;
    }
    __syncthreads();
  }
}

// Interior unknowns z = W[:, g] - W[:, left] z_left - W[:, right] z_right, one thread per interior pose; separators copy
// z_sep.  step = -z.
__global__ __launch_bounds__(256) void k_pg_backsub(PgDev d) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < d.n_int) {
    const int bl = d.bndL[t], br = d.bndR[t];
    double zl[3] = {0, 0, 0}, zr[3] = {0, 0, 0};
    if (bl >= 0)
#pragma unroll
      for (int j = 0; j < 3; ++j) zl[j] = d.xsep[3 * bl + j];
    if (br >= 0)
#pragma unroll
      for (int j = 0; j < 3; ++j) zr[j] = d.xsep[3 * br + j];
    const int v = d.int_pose[t];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double* w = d.W + (size_t)(3 * t + i) * 8;
      const double z = w[6] - (w[0] * zl[0] + w[1] * zl[1] + w[2] * zl[2]) - (w[3] * zr[0] + w[4] * zr[1] + w[5] * zr[2]);
      d.step[3 * (size_t)v + i] = -z;
    }
  } else if (t < d.n_int + d.n_sep) {
    const int q = t - d.n_int;
#pragma unroll
    for (int i = 0; i < 3; ++i) d.step[3 * (size_t)d.sep_pose[q] + i] = -d.xsep[3 * q + i];
  }
}

// candidate = x + step .* sigma (Euclidean blocks), |delta|^2 partials, finiteness flag
__global__ __launch_bounds__(256) void k_pg_candidate(PgDev d) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= d.n_poses) return;
  double sn = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double xv = d.x[3 * (size_t)v + i];
    double dl = 0.0;
    if (d.is_var[v]) {
      const double st = d.step[3 * (size_t)v + i];
      if (!isfinite(st)) d.flags[2] = 0;
      dl = st * d.sigma[3 * (size_t)v + i];
    }
    const double cv = xv + dl;
    d.cand[3 * (size_t)v + i] = cv;
    sn += (xv - cv) * (xv - cv);
  }
  d.p_sn[v] = sn;
}

// model_cost_change = -(J step)^T (r + J step / 2), per residual block
__global__ __launch_bounds__(256) void k_pg_mcc(PgDev d, int b) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.n_edges) return;
  const int a = d.ia[e], bb = d.ib[e];
  double da[3] = {0, 0, 0}, db[3] = {0, 0, 0};
  if (d.is_var[a])
#pragma unroll
    for (int i = 0; i < 3; ++i) da[i] = d.step[3 * (size_t)a + i] * d.sigma[3 * (size_t)a + i];
  if (d.is_var[bb])
#pragma unroll
    for (int i = 0; i < 3; ++i) db[i] = d.step[3 * (size_t)bb + i] * d.sigma[3 * (size_t)bb + i];
  const double* Ja = d.Ja[b] + 9 * (size_t)e;
  const double* Jb = d.Jb[b] + 9 * (size_t)e;
  const double* r = d.r[b] + 3 * (size_t)e;
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double mr = Ja[i * 3 + 0] * da[0] + Ja[i * 3 + 1] * da[1] + Ja[i * 3 + 2] * da[2] + Jb[i * 3 + 0] * db[0] +
                      Jb[i * 3 + 1] * db[1] + Jb[i * 3 + 2] * db[2];
    acc += mr * (r[i] + mr / 2.0);
  }
  d.p_mcc[e] = acc;
}

struct Carver {  // bump allocator over one device block
  char* base = nullptr;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += sizeof(T) * (n > 0 ? n : 1);
    return p;
  }
};

inline int grid_for(size_t n, int block) { return (int)((n + block - 1) / block); }

}  // namespace

void randt_pg_params_default(randt_pg_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->use_robust_loss = 0;
  p->loss_scale = 60.0;
  p->max_iterations = 200000;  // global_fuser.cpp:52
  p->max_consecutive_invalid_steps = 5;
  p->function_tolerance = 1e-6;
  p->gradient_tolerance = 1e-10;
  p->parameter_tolerance = 1e-8;
  p->initial_radius = 1e4;
  p->max_radius = 1e16;
  p->min_radius = 1e-32;
  p->min_relative_decrease = 1e-3;
  p->min_lm_diagonal = 1e-6;
  p->max_lm_diagonal = 1e32;
}

int randt_pose_graph_optimize(randt_ctx* ctx, int n_poses, double* h_poses, int n_edges, const int32_t* h_id_begin,
                                         const int32_t* h_id_end, const double* h_meas, const double* h_sqrt_info,
                                         int max_update_index, const randt_pg_params* opt, randt_pg_result* out) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !opt || n_poses <= 0 || n_edges < 0 || !h_poses) return RANDT_ERR_INVALID;
  if (n_edges > 0 && (!h_id_begin || !h_id_end || !h_meas || !h_sqrt_info)) return RANDT_ERR_INVALID;
  if (out) {
    memset(out, 0, sizeof(*out));
    out->n_loop_closures = n_edges + 1 - n_poses;  // global_fuser.cpp:26
  }
  // ---- the residual blocks the reference adds (global_fuser.cpp:31-46)
  std::vector<int32_t> ia, ib;
  std::vector<double> meas, sqi;
  std::vector<int32_t> is_var(n_poses, 0);
  for (int e = 0; e < n_edges; ++e) {
    const int a = h_id_begin[e], b = h_id_end[e];
    if (!(a + 1 == b || b <= max_update_index)) continue;
    if (a < 0 || b < 0 || a >= n_poses || b >= n_poses || a == b)
      return randt_set_error(ctx, RANDT_ERR_INVALID, "pose graph: edge endpoints out of range or identical", hipSuccess);
    ia.push_back(a);
    ib.push_back(b);
    meas.insert(meas.end(), h_meas + 3 * (size_t)e, h_meas + 3 * (size_t)e + 3);
    sqi.insert(sqi.end(), h_sqrt_info + 9 * (size_t)e, h_sqrt_info + 9 * (size_t)e + 9);
    is_var[a] = is_var[b] = 1;
  }
  is_var[0] = 0;  // poses.begin(): SetParameterBlockConstant (:48-49)
  const int nu = (int)ia.size();
  int n_var = 0;
  for (int i = 0; i < n_poses; ++i) n_var += is_var[i];
  if (out) out->n_residual_blocks = nu;
  if (nu == 0 || n_var == 0) return RANDT_OK;

  // ---- elimination order: separators last.  Besides the poses loop closures touch, every seg_cap-th pose of a long
  // uninterrupted run becomes a separator too: the chain recurrences are serial, so shorter segments = more lanes busy.
  std::vector<int32_t> is_sep(n_poses, 0);
  for (int e = 0; e < nu; ++e) {
    const int a = ia[e], b = ib[e];
    if (is_var[a] && is_var[b] && std::abs(a - b) != 1) is_sep[a] = is_sep[b] = 1;
  }
  {
    int seg_cap = 128;
    if (const char* env = getenv("RANDT_PG_SEGMENT")) seg_cap = std::max(2, atoi(env));
    int run = 0;
    for (int i = 0; i < n_poses; ++i) {
      if (!is_var[i] || is_sep[i]) { run = 0; continue; }
      if (++run > seg_cap) { is_sep[i] = 1; run = 0; }
    }
  }
  std::vector<int32_t> int_of(n_poses, -1), sep_of(n_poses, -1), int_pose, sep_pose;
  for (int i = 0; i < n_poses; ++i) {
    if (!is_var[i]) continue;
    if (is_sep[i]) {
      sep_of[i] = (int)sep_pose.size();
      sep_pose.push_back(i);
    } else {
      int_of[i] = (int)int_pose.size();
      int_pose.push_back(i);
    }
  }
  const int n_int = (int)int_pose.size(), n_sep = (int)sep_pose.size();
  const int ns = 3 * n_sep;
  if (out) out->n_separator_poses = n_sep;
  if (n_sep > RANDT_PG_MAX_SEPARATORS)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "pose graph: too many loop-closure poses for the dense Schur complement", hipSuccess);
  std::vector<int32_t> seg_first, seg_len, bndL(n_int > 0 ? n_int : 1, -1), bndR(n_int > 0 ? n_int : 1, -1);
  std::vector<int32_t> nbL(n_sep > 0 ? n_sep : 1, -1), nbR(n_sep > 0 ? n_sep : 1, -1);
  for (int m = 0; m < n_int;) {
    int len = 1;
    while (m + len < n_int && int_pose[m + len] == int_pose[m + len - 1] + 1) ++len;
    const int vf = int_pose[m], vl = int_pose[m + len - 1];
    const int ql = (vf - 1 >= 0) ? sep_of[vf - 1] : -1, qr = (vl + 1 < n_poses) ? sep_of[vl + 1] : -1;
    for (int t = 0; t < len; ++t) {
      bndL[m + t] = ql;
      bndR[m + t] = qr;
    }
    if (ql >= 0) nbR[ql] = m;
    if (qr >= 0) nbL[qr] = m + len - 1;
    seg_first.push_back(m);
    seg_len.push_back(len);
    m += len;
  }
  const int n_seg = (int)seg_first.size();
  // incidence lists (ascending edge order per pose)
  std::vector<int32_t> inc_off(n_poses + 1, 0), inc_ent(2 * (size_t)nu);
  for (int e = 0; e < nu; ++e) {
    inc_off[ia[e] + 1]++;
    inc_off[ib[e] + 1]++;
  }
  for (int i = 0; i < n_poses; ++i) inc_off[i + 1] += inc_off[i];
  {
    std::vector<int32_t> fill(inc_off.begin(), inc_off.end() - 1);
    for (int e = 0; e < nu; ++e) {
      inc_ent[fill[ia[e]]++] = (e << 1) | 0;
      inc_ent[fill[ib[e]]++] = (e << 1) | 1;
    }
  }

  // ---- device block
  PgDev d{};
  d.n_poses = n_poses; d.n_edges = nu; d.n_int = n_int; d.n_sep = n_sep; d.ns = ns; d.n_seg = n_seg;
  int32_t *d_ia, *d_ib, *d_inc_off, *d_inc_ent, *d_is_var, *d_int_of, *d_sep_of, *d_int_pose, *d_sep_pose, *d_seg_first, *d_seg_len, *d_bndL, *d_bndR,
      *d_nbL, *d_nbR;
  double *d_meas, *d_sqi;
  Carver cv;
  auto carve = [&]() {
    d_ia = cv.take<int32_t>(nu); d_ib = cv.take<int32_t>(nu);
    d_meas = cv.take<double>(3 * (size_t)nu); d_sqi = cv.take<double>(9 * (size_t)nu);
    d_inc_off = cv.take<int32_t>(n_poses + 1); d_inc_ent = cv.take<int32_t>(2 * (size_t)nu);
    d_is_var = cv.take<int32_t>(n_poses); d_int_of = cv.take<int32_t>(n_poses); d_sep_of = cv.take<int32_t>(n_poses);
    d_int_pose = cv.take<int32_t>(n_int); d_sep_pose = cv.take<int32_t>(n_sep);
    d_seg_first = cv.take<int32_t>(n_seg); d_seg_len = cv.take<int32_t>(n_seg);
    d_bndL = cv.take<int32_t>(n_int); d_bndR = cv.take<int32_t>(n_int);
    d_nbL = cv.take<int32_t>(n_sep); d_nbR = cv.take<int32_t>(n_sep);
    d.x = cv.take<double>(3 * (size_t)n_poses); d.cand = cv.take<double>(3 * (size_t)n_poses);
    for (int b = 0; b < 2; ++b) {
      d.r[b] = cv.take<double>(3 * (size_t)nu); d.Ja[b] = cv.take<double>(9 * (size_t)nu);
      d.Jb[b] = cv.take<double>(9 * (size_t)nu); d.cost_e[b] = cv.take<double>(nu);
    }
    d.Ds = cv.take<double>(6 * (size_t)n_poses); d.gs = cv.take<double>(3 * (size_t)n_poses);
    d.sigma = cv.take<double>(3 * (size_t)n_poses); d.diagonal = cv.take<double>(3 * (size_t)n_poses);
    d.E = cv.take<double>(9 * (size_t)n_int); d.CL = cv.take<double>(9 * (size_t)n_int); d.CR = cv.take<double>(9 * (size_t)n_int);
    d.Lf = cv.take<double>(9 * (size_t)n_int); d.Ff = cv.take<double>(9 * (size_t)n_int);
    d.W = cv.take<double>(3 * (size_t)n_int * 8);
    d.S0 = cv.take<double>((size_t)ns * ns); d.Sw = cv.take<double>((size_t)(ns + 1) * ns);
    d.xsep = cv.take<double>(ns); d.step = cv.take<double>(3 * (size_t)n_poses);
    d.Linv = cv.take<double>((size_t)((ns + PG_NB - 1) / PG_NB) * PG_NB * PG_NB);
    d.p_gabs = cv.take<double>(n_poses); d.p_xsq = cv.take<double>(n_poses); d.p_sn = cv.take<double>(n_poses);
    d.p_mcc = cv.take<double>(nu);
    d.scal = cv.take<double>(8); d.flags = cv.take<int32_t>(4);
  };
  carve();  // sizing pass
  const size_t bytes = cv.off + 256;
  char* blk = nullptr;
  size_t blk_bytes = 0;
  RANDT_HIP_CHECK(ctx, randt_dev_alloc(ctx, reinterpret_cast<void**>(&blk), bytes, &blk_bytes));  // the context's storage pool (api.hip)
  cv = Carver{blk, 0};
  carve();
  d.ia = d_ia; d.ib = d_ib; d.meas = d_meas; d.sqi = d_sqi; d.inc_off = d_inc_off; d.inc_ent = d_inc_ent;
  d.is_var = d_is_var; d.int_of = d_int_of; d.sep_of = d_sep_of; d.int_pose = d_int_pose; d.sep_pose = d_sep_pose;
  d.seg_first = d_seg_first; d.seg_len = d_seg_len; d.bndL = d_bndL; d.bndR = d_bndR; d.nbL = d_nbL; d.nbR = d_nbR;

  hipStream_t st = ctx->stream;
  int rc = RANDT_OK;
#define PG_TRY(call)                                                                \
  do {                                                                              \
    hipError_t e__ = (call);                                                        \
    if (e__ != hipSuccess && rc == RANDT_OK) rc = randt_set_error(ctx, RANDT_ERR_HIP, #call, e__); \
  } while (0)
#define PG_UP(dst, vec, T) \
  if (!(vec).empty()) PG_TRY(hipMemcpyAsync((dst), (vec).data(), sizeof(T) * (vec).size(), hipMemcpyHostToDevice, st))
  PG_UP(d_ia, ia, int32_t); PG_UP(d_ib, ib, int32_t); PG_UP(d_meas, meas, double); PG_UP(d_sqi, sqi, double);
  PG_UP(d_inc_off, inc_off, int32_t); PG_UP(d_inc_ent, inc_ent, int32_t); PG_UP(d_is_var, is_var, int32_t);
  PG_UP(d_int_of, int_of, int32_t); PG_UP(d_sep_of, sep_of, int32_t); PG_UP(d_int_pose, int_pose, int32_t);
  PG_UP(d_sep_pose, sep_pose, int32_t);
  if (n_int > 0) {
    PG_UP(d_seg_first, seg_first, int32_t); PG_UP(d_seg_len, seg_len, int32_t); PG_UP(d_bndL, bndL, int32_t); PG_UP(d_bndR, bndR, int32_t);
  }
  if (n_sep > 0) {
    PG_UP(d_nbL, nbL, int32_t); PG_UP(d_nbR, nbR, int32_t);
  }
#undef PG_UP
  PG_TRY(hipMemcpyAsync(d.x, h_poses, sizeof(double) * 3 * (size_t)n_poses, hipMemcpyHostToDevice, st));
  PG_TRY(hipMemsetAsync(d.step, 0, sizeof(double) * 3 * (size_t)n_poses, st));
  // the host arrays above must outlive the copies
  PG_TRY(hipStreamSynchronize(st));

  const int robust = opt->use_robust_loss ? 1 : 0;
  double h_scal[8];
  int32_t h_flags[4];
  int cur = 0;
  auto eval_edges = [&](const double* xp, int b, int slot) {
    hipLaunchKernelGGL(k_pg_edges, dim3(grid_for(nu, 256)), dim3(256), 0, st, d, xp, b, robust, opt->loss_scale);
    PgReduce R{};
    R.in[0] = d.cost_e[b]; R.n[0] = nu; R.slot[0] = slot; R.op[0] = 0;
    hipLaunchKernelGGL(k_pg_reduce, dim3(1), dim3(1024), 0, st, R, d.scal);
  };
  auto linearize = [&](int b, int first) {
    hipLaunchKernelGGL(k_pg_pose_diag, dim3(grid_for(n_poses, 256)), dim3(256), 0, st, d, b, first);
    if (ns > 0) PG_TRY(hipMemsetAsync(d.S0, 0, sizeof(double) * (size_t)ns * ns, st));
    hipLaunchKernelGGL(k_pg_pose_offdiag, dim3(grid_for(n_poses, 256)), dim3(256), 0, st, d, b);
    PgReduce R{};
    R.in[0] = d.p_xsq; R.n[0] = n_poses; R.slot[0] = 4; R.op[0] = 0;
    R.in[1] = d.p_gabs; R.n[1] = n_poses; R.slot[1] = 5; R.op[1] = 1;
    hipLaunchKernelGGL(k_pg_reduce, dim3(2), dim3(1024), 0, st, R, d.scal);
  };
  auto read_back = [&]() {
    PG_TRY(hipMemcpyAsync(h_scal, d.scal, sizeof(h_scal), hipMemcpyDeviceToHost, st));
    PG_TRY(hipMemcpyAsync(h_flags, d.flags, sizeof(h_flags), hipMemcpyDeviceToHost, st));
    PG_TRY(hipStreamSynchronize(st));
    PG_TRY(hipGetLastError());
  };

  // ---- TrustRegionMinimizer::Minimize (host control flow)
  int term = RANDT_TERM_FAILURE;
  double radius = opt->initial_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false, step_successful = true;
  int num_invalid = 0, iteration = 0, n_it = 1;
  double minimum_cost = DBL_MAX, summary_min_cost = 0.0;
  std::vector<double> best(h_poses, h_poses + 3 * (size_t)n_poses), h_x(3 * (size_t)n_poses);

  eval_edges(d.x, cur, 0);
  linearize(cur, 1);
  read_back();
  double x_cost = h_scal[0], x_norm = std::sqrt(h_scal[4]), grad_max_norm = h_scal[5];
  if (out) out->initial_cost = x_cost;
  summary_min_cost = x_cost;

  while (rc == RANDT_OK) {
    if (step_successful && x_cost < minimum_cost) {
      minimum_cost = x_cost;
      PG_TRY(hipMemcpyAsync(best.data(), d.x, sizeof(double) * 3 * (size_t)n_poses, hipMemcpyDeviceToHost, st));
      PG_TRY(hipStreamSynchronize(st));
    }
    if (iteration >= opt->max_iterations) { term = RANDT_TERM_NO_CONVERGENCE; break; }
    if (step_successful && grad_max_norm <= opt->gradient_tolerance) { term = RANDT_TERM_CONVERGENCE_GRADIENT; break; }
    if (radius <= opt->min_radius) { term = RANDT_TERM_CONVERGENCE_RADIUS; break; }
    ++iteration;
    ++n_it;

    if (!reuse_diagonal)
      hipLaunchKernelGGL(k_pg_lm_diagonal, dim3(grid_for(n_poses, 256)), dim3(256), 0, st, d, opt->min_lm_diagonal, opt->max_lm_diagonal);
    const double inv_radius = 1.0 / radius;
    PG_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d.flags), 1, 4, st));
    if (n_int > 0) {
      hipLaunchKernelGGL(k_pg_factor, dim3(grid_for(n_seg, 64)), dim3(64), 0, st, d, inv_radius);
      hipLaunchKernelGGL(k_pg_chain_solve, dim3(grid_for((size_t)n_seg * 8, 64)), dim3(64), 0, st, d);
    }
    if (ns > 0) {
      hipLaunchKernelGGL(k_pg_schur, dim3(grid_for((size_t)(ns + 1) * ns, 256)), dim3(256), 0, st, d, inv_radius);
      for (int k0 = 0; k0 < ns; k0 += PG_NB) {
        const int nb = std::min(PG_NB, ns - k0), below = ns - (k0 + nb) + 1;  // rows under the block, RHS row included
        hipLaunchKernelGGL(k_pg_potrf, dim3(1), dim3(PG_NB * PG_NB), 0, st, d, k0);
        hipLaunchKernelGGL(k_pg_trsm, dim3(grid_for(below, 64)), dim3(256), 0, st, d, k0);
        const int tiles = (below + 63) / 64;
        hipLaunchKernelGGL(k_pg_syrk, dim3(tiles * (tiles + 1) / 2), dim3(256), 0, st, d, k0);
      }
      hipLaunchKernelGGL(k_pg_trsv, dim3(1), dim3(1024), 0, st, d);
    }
    hipLaunchKernelGGL(k_pg_backsub, dim3(grid_for(n_int + n_sep, 256)), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_pg_candidate, dim3(grid_for(n_poses, 256)), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_pg_mcc, dim3(grid_for(nu, 256)), dim3(256), 0, st, d, cur);
    {
      PgReduce R{};
      R.in[0] = d.p_mcc; R.n[0] = nu; R.slot[0] = 2; R.op[0] = 0;
      R.in[1] = d.p_sn; R.n[1] = n_poses; R.slot[1] = 3; R.op[1] = 0;
      hipLaunchKernelGGL(k_pg_reduce, dim3(2), dim3(1024), 0, st, R, d.scal);
    }
    eval_edges(d.cand, cur ^ 1, 1);  // residuals AND Jacobians at the candidate: reused if the step is accepted
    read_back();
    if (rc != RANDT_OK) break;
    reuse_diagonal = true;
    if (getenv("RANDT_PG_DEBUG"))
      fprintf(stderr, "[pg] it %d radius %.3e flags %d %d %d cost %.9e cand %.9e mcc %.6e step2 %.3e n_int %d n_sep %d n_seg %d\n", iteration,
              radius, h_flags[0], h_flags[1], h_flags[2], x_cost, h_scal[1], -h_scal[2], h_scal[3], n_int, n_sep, n_seg);
    const bool solved = h_flags[0] && h_flags[1] && h_flags[2];
    const double model_cost_change = -h_scal[2];
    if (!(solved && model_cost_change > 0.0)) {
      if (++num_invalid >= opt->max_consecutive_invalid_steps) { term = RANDT_TERM_FAILURE; break; }
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      step_successful = false;
      summary_min_cost = std::min(summary_min_cost, x_cost);
      continue;
    }
    num_invalid = 0;
    const double cand_cost = h_scal[1];
    const double step_norm = std::sqrt(h_scal[3]);
    if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { term = RANDT_TERM_CONVERGENCE_PARAMETER; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= opt->function_tolerance * x_cost) { term = RANDT_TERM_CONVERGENCE_FUNCTION; break; }
    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > opt->min_relative_decrease) {
      std::swap(d.x, d.cand);
      cur ^= 1;
      x_cost = cand_cost;
      linearize(cur, 0);
      read_back();
      x_norm = std::sqrt(h_scal[4]);
      grad_max_norm = h_scal[5];
      step_successful = true;
      radius = radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::fmin(opt->max_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
      summary_min_cost = std::min(summary_min_cost, x_cost);
    } else {
      step_successful = false;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      summary_min_cost = std::min(summary_min_cost, cand_cost);
    }
  }
#undef PG_TRY
  (void)randt_sync(ctx);
  randt_dev_release(ctx, blk, blk_bytes);
  if (rc != RANDT_OK) return rc;
  memcpy(h_poses, best.data(), sizeof(double) * 3 * (size_t)n_poses);
  if (out) {
    out->final_cost = summary_min_cost;
    out->iterations = n_it;
    out->termination = term;
  }
  return RANDT_OK;
}
