// The residual pass of the pair solve (one wavefront, or BLOCK / 64 wavefronts, over a frozen correspondence set), shared by
// solve.hip (one wavefront per registration, split mode) and solve_tr.hip (trip wavefronts of the transposed geometry).
// Reference: include/ndt_registration/ceres_residuals.h:421-552 (functor), Ceres residual_block.cc / corrector.cc (loss).
#pragma once
#include "randt_internal.h"
#include "solve_algebra.h"
#include "solve_math.h"

#include <float.h>

// Issue priority inside the solve (s_setprio): the residual trips are throughput work, the reduction and the solver algebra
// behind them one dependent fp64 chain per wavefront.  RANDT_SOLVE_PRIO_ALG > 0 raises the chain's priority over the trips of
// the wavefronts it shares the SIMD with; RANDT_SOLVE_PRIO_PRO is the priority of the prologue (correspondence count and
// compaction: global round trips).
#ifndef RANDT_SOLVE_PRIO_ALG
#define RANDT_SOLVE_PRIO_ALG 0
#endif
#ifndef RANDT_SOLVE_PRIO_PRO
#define RANDT_SOLVE_PRIO_PRO 3
#endif
#define RANDT_PRIO_TRIPS() do { if (RANDT_SOLVE_PRIO_ALG > 0) __builtin_amdgcn_s_setprio(0); } while (0)
#define RANDT_PRIO_CHAIN() do { if (RANDT_SOLVE_PRIO_ALG > 0) __builtin_amdgcn_s_setprio(RANDT_SOLVE_PRIO_ALG); } while (0)

namespace randt_pass {
using namespace randt_solve;

__device__ __forceinline__ double sgpr_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
using randt_lm::Base;
using randt_lm::vec_like;

// valid correspondences of one registration, compacted once into LDS: (moving index << 21) | fixed index
constexpr int PAIR_CAP = 1024;
constexpr int PAIR_SHIFT = 21;
constexpr unsigned PAIR_MASK = (1u << PAIR_SHIFT) - 1u;

struct Stage {
  const float4* mov;      // moving cell records (3 x float4 each)
  const float4* fix;      // fixed cell records
  const int* corr;        // [M*k] compact fixed index or -1
  const unsigned* pairs;  // LDS: compacted valid correspondences, or nullptr (then the raw slots are walked)
  int n_pairs;
  int pack16;             // pairs hold (3 * moving index) << 16 | 3 * fixed index (both record offsets in float4 units < 2^16)
  int n_slots, k, fixed_cap;
  unsigned kmagic;        // ceil(2^32 / k): slot / k == umulhi(slot, kmagic) for slot < 2^32 / k (k >= 2)
};

// cos / sin of the rotation and the translation of ambient point x (what a pass evaluates the residuals at)
template <int D, int PARAM>
__device__ __forceinline__ void pass_pose(const double* x, double& c, double& s, double& tx, double& ty) {
#pragma clang fp contract(off)
  if (vec_like(PARAM)) {
    c = cos(x[2]);
    s = sin(x[2]);
    tx = x[0];
    ty = x[1];
  } else {
    // R = AngleAxis(atan2(sp, cp)): cos/sin of the angle == normalised stored complex
    const double inv = fast_rsqrt(fma(x[0], x[0], x[1] * x[1]));
    c = x[0] * inv;
    s = x[1] * inv;
    tx = x[2];
    ty = x[3];
  }
}

// Pass over all correspondence slots at ambient point x.  MODE 0: max raw residual (out.v[0]);
// MODE 1: the ten base sums with loss + corrector (Ceres residual_block.cc / corrector.cc).
// Returns false if any residual was non-finite.  red: [2][BLOCK/64][12] LDS, parity alternates per call.
template <int D, int PARAM, int MODE, int BLOCK, bool AM2>
__device__ __forceinline__ bool eval_pass(const Stage& S, const double* x, const Loss& L, Base& out, double* red, int& parity, int tid) {
  constexpr int WAVES = BLOCK / 64;
  double c, s, tx, ty;
  pass_pose<D, PARAM>(x, c, s, tx, ty);
  // the evaluation point is the same in every lane: its rotation products, the translation and the loss constants live in
  // SGPRs (a VALU instruction takes one scalar operand) instead of 2 x 12 vector registers per lane -- the four-per-SIMD
  // kernels then need 122 registers and no scratch (128 with seven spilled dwords before; same speed, round 4)
  c = sgpr_f64(c); s = sgpr_f64(s); tx = sgpr_f64(tx); ty = sgpr_f64(ty);
  Rot rot = make_rot(c, s);
  rot.c2 = sgpr_f64(rot.c2); rot.s2 = sgpr_f64(rot.s2); rot.cs = sgpr_f64(rot.cs); rot.cs2 = sgpr_f64(rot.cs2); rot.c2ms2 = sgpr_f64(rot.c2ms2);
  Loss Lu = L;
  Lu.ts = sgpr_f64(L.ts); Lu.weight = sgpr_f64(L.weight); Lu.half_w_pre = sgpr_f64(L.half_w_pre);
  RANDT_PRIO_TRIPS();
  double acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.0;
  double mx = -DBL_MAX;
  int bad = 0;
  // one residual; mv / fv = moving / fixed cell record
  auto one_rec = [&](const float4* mv, const float4* fv) {
    double jb[3];
    const double sq = residual_sq<D, MODE == 1, PARAM == RANDT_PARAM_ANALYTIC>(mv, fv, rot, tx, ty, jb);
    // closed-form loss: a non-finite residual makes the cost sum non-finite (u or 1 / (u^2 s) is NaN / 0 x inf), which the
    // caller tests after the reduction -- no per-residual class test in the hot loop
    if (!(MODE == 1 && AM2) && !isfinite(sq)) bad = 1;
    if (MODE == 0) {
      mx = sq > mx ? sq : mx;
    } else {
      accumulate_residual<AM2>(Lu, sq, jb, acc);
    }
  };
  auto one = [&](unsigned mi, unsigned ci) { one_rec(S.mov + (size_t)mi * 3, S.fix + (size_t)ci * 3); };
  if (S.n_pairs > 0 && S.pack16) {  // scalar: the dense list in LDS, record BYTE offsets two shifts / masks away
    // (32-bit offsets against the uniform table bases: the loads take the scalar-base addressing form, no 64-bit adds)
    const char* mb = reinterpret_cast<const char*>(S.mov);
    const char* fb = reinterpret_cast<const char*>(S.fix);
    for (int e = tid; e < S.n_pairs; e += BLOCK) {
      const unsigned u = S.pairs[e];
      one_rec(reinterpret_cast<const float4*>(mb + ((u >> 12) & 0xffff0u)), reinterpret_cast<const float4*>(fb + ((u << 4) & 0xffff0u)));
    }
  } else if (S.n_pairs > 0) {
    for (int e = tid; e < S.n_pairs; e += BLOCK) {
      const unsigned u = S.pairs[e];
      one(u >> PAIR_SHIFT, u & PAIR_MASK);
    }
  } else {
    for (int slot = tid; slot < S.n_slots; slot += BLOCK) {
      const int cr = S.corr[slot];
      if (cr < 0 || cr >= S.fixed_cap) continue;
      one(S.k == 1 ? (unsigned)slot : __umulhi((unsigned)slot, S.kmagic) /* slot / k */, (unsigned)cr);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RANDT_PRIO_CHAIN();
  double badf = wave_any(bad != 0);
  if (MODE == 0) {
    mx = wave_max(mx);
    if (WAVES > 1) {
      double* r = red + parity * (WAVES * 12);
      parity ^= 1;
      if (lane == 0) {
        r[wave * 12 + 0] = mx;
        r[wave * 12 + 1] = badf;
      }
      __syncthreads();
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        mx = r[w * 12] > mx ? r[w * 12] : mx;
        badf = r[w * 12 + 1] > badf ? r[w * 12 + 1] : badf;
      }
    }
    out.v[0] = mx > 0.0 ? sqrt(mx) : 0.0;  // max raw residual
    return uni(badf == 0.0);
  }
  wave_sum10(acc);
  if (WAVES > 1) {
    double* r = red + parity * (WAVES * 12);
    parity ^= 1;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 10; ++i) r[wave * 12 + i] = acc[i];
      r[wave * 12 + 10] = badf;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
#pragma unroll
      for (int i = 0; i < 10; ++i) acc[i] += r[w * 12 + i];
      badf = r[w * 12 + 10] > badf ? r[w * 12 + 10] : badf;
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) out.v[i] = acc[i];
  return uni(badf == 0.0 && isfinite(acc[0]));
}

}  // namespace randt_pass
