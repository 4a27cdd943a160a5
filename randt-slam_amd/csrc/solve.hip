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
// Geometry: BLOCK = 64 / 128 threads cooperate on one registration (default 64 = a single wavefront
// with no barrier at all; 128 = two wavefronts on two SIMDs).  The frozen
// correspondence set is compacted once into an LDS index list and its 48-byte cell records (first
// 9 floats = mean xyz + upper-triangular covariance) are read in place: they stay L1/L2 resident
// across the ~30 passes of a registration and are cast to fp64 in registers like the reference
// does (ndt_matcher.cpp:231).  (Staging the records themselves in LDS was measured slower: the LDS
// it takes costs the co-running kernels more than the solve gains.)
// Every LM iteration is one pass over the valid correspondences: fp64 residual, its (un-normalised) Jacobian
// with respect to (tx, ty, theta), loss + Ceres corrector -- no square root, no IEEE division per residual
// (solve_math.h) -- and TEN accumulators {cost, J^T r (3), upper J^T J (6)} reduced in a fixed order
// (lane-swap folding -> DPP row reduction -> readlane; LDS combine across wavefronts): deterministic,
// run-to-run bit-identical.  The Jacobian of the actual
// parameterisation (SE(2) tangent / ambient [c,s,tx,ty] / vector) is a per-evaluation-constant
// linear map T of that base Jacobian, so H = T G T^T and g = T g_b are formed after the reduction.
// Jacobi scaling, LM diagonal clamp, damped normal equations (packed LDL^T), model-cost change, Plus,
// both convergence tests, accept/reject and radius update run redundantly on all lanes; every branch on
// that uniform state goes through a ballot (uni()) so that the compiler emits scalar branches.  The
// candidate is evaluated WITH its Jacobian so an accepted step needs no second pass.  The template
// parameter AM2 selects the closed-form loss of the shipped Barron shape (-2), which keeps pow() out of
// the kernel.
#include "solve_pass.h"

using namespace randt_solve;
using namespace randt_pass;
using namespace randt_lm;

namespace {

// ---------------------------------------------------------------- split mode (small batches) ---
// A batch that cannot fill the chip with one wavefront per registration (one 512-registration loop-closure burst is 512
// wavefronts on 1024 SIMDs; an 8-GPU split of it leaves 64 per GPU) gives every registration W wavefronts on the SIMDs of
// one CU.  Wavefront 0 runs the whole solver exactly as in the one-wavefront kernel; for every pass it publishes the
// evaluation point, and wavefront w evaluates residual trip w (correspondences [64 w, 64 w + 64)) and hands the per-lane
// TERMS (jr, h, jb, c -- solve_math.h) back through LDS.  Wavefront 0 adds them to its accumulators in trip order with
// the same fused multiply-adds a lone wavefront would have executed, so the ten sums -- and everything after them -- are
// bit-identical to the one-wavefront kernel; trips beyond W (n_res > 64 W) it evaluates itself.  A pass then costs one
// trip + the combine instead of ceil(n_res / 64) trips.  The helpers keep their two cell records in registers for the
// whole solve (the association is frozen) and sleep at the workgroup barrier while wavefront 0 does the solver algebra.
#ifdef RANDT_SPLIT_TIMING  // developer probe (tools/ab_build.sh): where wavefront 0 of workgroup 0 spends a split-mode solve
__device__ long long g_randt_split_timing[10];
__shared__ long long s_split_tim[10];
extern "C" int randt_debug_split_timing(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_randt_split_timing), sizeof(long long) * 10) == hipSuccess ? 0 : 2;
}
#define ST(slot) do { if ((threadIdx.x) == 0) { const long long n_ = clock64(); s_split_tim[slot] += n_ - s_split_tim[9]; s_split_tim[9] = n_; } } while (0)
#else
#define ST(slot) do {} while (0)
#endif
constexpr int SPLIT_MAXW = 8;          // wavefronts per registration (workgroup of <= 512 threads)
constexpr int SPLIT_PAIR_CAP = 2048;   // compacted correspondences per registration in split mode (launcher guarantees M k <= this)
struct SplitReq {
  double x[4];
  Loss L;
  int mode;  // 0: raw-residual maximum, 1: terms of the ten sums, -1: the solve is over
  int pad;
};

__device__ __forceinline__ void pair_records(const Stage& S, int e, const float4*& mv, const float4*& fv) {
  const unsigned u = S.pairs[e];
  if (S.pack16) {
    mv = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.mov) + ((u >> 12) & 0xffff0u));
    fv = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.fix) + ((u << 4) & 0xffff0u));
  } else {
    mv = S.mov + (size_t)(u >> PAIR_SHIFT) * 3;
    fv = S.fix + (size_t)(u & PAIR_MASK) * 3;
  }
}

// wavefront w >= 1 of a split-mode registration: serve trip w of every pass until the solve is over
template <int D, int PARAM, bool AM2>
__device__ __forceinline__ void split_helper(const Stage& S, const SplitReq* req, double* mine /* [64][6] */, int* bad_flag, int w, int lane) {
  const int e = w * 64 + lane;
  const bool active = e < S.n_pairs;
  float4 ma, mb, mc4, fa, fb, fc4;
  ma = mb = mc4 = fa = fb = fc4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    const float4 *mv, *fv;
    pair_records(S, e, mv, fv);
    ma = mv[0]; mb = mv[1]; mc4 = mv[2];
    fa = fv[0]; fb = fv[1]; fc4 = fv[2];
  }
  double2* out = reinterpret_cast<double2*>(mine + lane * 6);
  for (;;) {
    __syncthreads();  // A: the request is visible
    const int mode = req->mode;
    if (mode < 0) return;
    double c, s, tx, ty;
    pass_pose<D, PARAM>(req->x, c, s, tx, ty);
    const Rot rot = make_rot(c, s);
    double o0 = mode == 0 ? -DBL_MAX : 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0, o4 = 0.0, o5 = 0.0;
    int bad = 0;
    if (active) {
      double jb[3];
      const double sq = residual_sq_v<D, true, PARAM == RANDT_PARAM_ANALYTIC>(ma, mb, mc4, fa, fb, fc4, rot, tx, ty, jb);
      if (!(mode == 1 && AM2) && !isfinite(sq)) bad = 1;
      if (mode == 0) {
        o0 = sq;
      } else {
        Loss L;
        if (AM2) {  // the closed form reads three members
          L.ts = req->L.ts;
          L.weight = req->L.weight;
          L.half_w_pre = req->L.half_w_pre;
        } else {
          L = req->L;
        }
        double jr, h, cterm;
        residual_terms<AM2>(L, sq, jr, h, cterm);
        o0 = jr; o1 = h; o2 = jb[0]; o3 = jb[1]; o4 = jb[2]; o5 = cterm;
      }
    }
    out[0] = make_double2(o0, o1);
    out[1] = make_double2(o2, o3);
    out[2] = make_double2(o4, o5);
    const bool anybad = __ballot(bad != 0) != 0ull;
    if (lane == 0) *bad_flag = anybad ? 1 : 0;
    __syncthreads();  // B: the terms are visible
  }
}

// wavefront 0's pass in split mode (same contract as eval_pass with one wavefront)
template <int D, int PARAM, int MODE, bool AM2>
__device__ __forceinline__ bool eval_pass_split(const Stage& S, const double* x, const Loss& L, Base& out, SplitReq* req, const double* part,
                                                const int* bad_flags, int W, int lane) {
  ST(4);  // solver algebra since the previous pass
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) req->x[i] = x[i];
    req->L = L;
    req->mode = MODE;
  }
  __syncthreads();  // A
  ST(0);
  double c, s, tx, ty;
  pass_pose<D, PARAM>(x, c, s, tx, ty);
  const Rot rot = make_rot(c, s);
  double acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.0;
  double mx = -DBL_MAX;
  int bad = 0;
  auto one_rec = [&](const float4* mv, const float4* fv) {
    double jb[3];
    const double sq = residual_sq<D, MODE == 1, PARAM == RANDT_PARAM_ANALYTIC>(mv, fv, rot, tx, ty, jb);
    if (!(MODE == 1 && AM2) && !isfinite(sq)) bad = 1;
    if (MODE == 0) {
      mx = sq > mx ? sq : mx;
    } else {
      accumulate_residual<AM2>(L, sq, jb, acc);
    }
  };
  if (lane < S.n_pairs) {  // trip 0
    const float4 *mv, *fv;
    pair_records(S, lane, mv, fv);
    one_rec(mv, fv);
  }
  ST(1);
  __syncthreads();  // B
  ST(2);
  const int T = (S.n_pairs + 63) >> 6;
  const int Tp = T < W ? T : W;
  for (int w = 1; w < Tp; ++w) {  // trips 1 .. W-1 from the helpers, in trip order
    const double2* p = reinterpret_cast<const double2*>(part + ((size_t)(w - 1) * 64 + lane) * 6);
    const double2 a = p[0], b = p[1], d = p[2];
    if (MODE == 0) {
      mx = a.x > mx ? a.x : mx;
    } else {
      const double jb[3] = {b.x, b.y, d.x};
      accumulate_terms<AM2>(L, a.x, a.y, jb, d.y, acc);
    }
    if (bad_flags[w] != 0) bad = 1;
  }
  for (int t = W; t < T; ++t) {  // what the helpers do not cover
    const int e = t * 64 + lane;
    if (e < S.n_pairs) {
      const float4 *mv, *fv;
      pair_records(S, e, mv, fv);
      one_rec(mv, fv);
    }
  }
  const double badf = wave_any(bad != 0);
  if (MODE == 0) {
    mx = wave_max(mx);
    out.v[0] = mx > 0.0 ? sqrt(mx) : 0.0;
    return uni(badf == 0.0);
  }
  wave_sum10(acc);
#pragma unroll
  for (int i = 0; i < 10; ++i) out.v[i] = acc[i];
  ST(3);
  return uni(badf == 0.0 && isfinite(acc[0]));
}

__device__ __forceinline__ void trace_push(double* tr, int max_len, int tid, double cost, double radius, int flag) {
  if (tr && tid == 0) {  // thread 0 of the registration (RPB > 1: lane 0 of its wavefront)
    const int n = (int)tr[0];
    if (3 * (n + 1) + 1 <= max_len) {
      tr[1 + 3 * n + 0] = cost;
      tr[1 + 3 * n + 1] = radius;
      tr[1 + 3 * n + 2] = (double)flag;
      tr[0] = (double)(n + 1);
    }
  }
}

// AM2: the Barron shape is exactly -2 (the reference's shipped configurations): closed-form loss, no pow()
// in the kernel -- 30 fewer VGPRs and a third of the code.
// RPB > 1 (BLOCK = 64 only): RPB independent registrations per workgroup, one per wavefront -- nothing is shared between
// them (no workgroup barrier); the point is placement: the dispatcher spreads the wavefronts of ONE workgroup over the four
// SIMDs of a CU, which it does not do for single-wavefront workgroups arriving from many queues.
// Wavefronts per SIMD.  The closed-form-loss kernels (AM2) are pinned to THREE (<= 168 registers): with the cold solver
// state in LDS they need ~190, and squeezed to 168 the compiler parks ~20 dwords in scratch, one access of which sits in
// the residual loop -- measured -12 % per chip-filling solve launch against two wavefronts per SIMD (a lone wavefront is
// VALU-active about half the time; the third fills the gaps).  Four per SIMD (128 registers) spilled 70 dwords and lost then; since
// the solver state moves in bulk (136 registers at three) it is 9 dwords, and since the short kernels run at raised priority in
// register-capped instantiations the one-wavefront kernels are pinned to FOUR: the 512-registration launches of
// 16 streams 27.8 -> 25.9 us each, the pipelined region 11.05 -> 11.43 M registrations/s (it was +-0 before those changes).
// The general-alpha kernels carry pow() and stay at two, BLOCK = 128 at three.  RANDT_SOLVE_WPE: experiment knob (tools/ab_build.sh).
#ifdef RANDT_SOLVE_WPE
#define RANDT_SOLVE_OCC(AM2, SPLIT) __attribute__((amdgpu_waves_per_eu(RANDT_SOLVE_WPE, RANDT_SOLVE_WPE)))
#else
#ifndef RANDT_SPLIT_WPE
#define RANDT_SPLIT_WPE 4  // split mode: wavefront 0's pass holds ONE residual trip -> 127 registers (five spilled dwords), four per SIMD
#endif
#define RANDT_SOLVE_WPE_OF(AM2, SPLIT, BLOCK) ((SPLIT) ? RANDT_SPLIT_WPE : ((AM2) ? ((BLOCK) == 64 ? 4 : 3) : 2))
#define RANDT_SOLVE_OCC(AM2, SPLIT) \
  __attribute__((amdgpu_waves_per_eu(RANDT_SOLVE_WPE_OF(AM2, SPLIT, BLOCK), RANDT_SOLVE_WPE_OF(AM2, SPLIT, BLOCK))))
#endif
#ifdef RANDT_SOLVE_NUM_VGPR  // experiment knob: cap the one-wavefront closed-form kernels below their 128-register slot
#define RANDT_SOLVE_VGPR_CAP __attribute__((amdgpu_num_vgpr(RANDT_SOLVE_NUM_VGPR)))
#undef RANDT_SOLVE_OCC
#define RANDT_SOLVE_OCC(AM2, SPLIT)
#else
#define RANDT_SOLVE_VGPR_CAP
#endif
template <int D, int PARAM, int BLOCK, bool AM2, int RPB, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 64 * SPLIT_MAXW : BLOCK* RPB) RANDT_SOLVE_OCC(AM2, SPLIT) RANDT_SOLVE_VGPR_CAP void k_solve(MapView fixed, const int32_t* __restrict__ fixed_idx, MapView moving,
                                                      int moving_first, const int32_t* __restrict__ corr, SolveParams P,
                                                      double* __restrict__ pose4, randt_result* __restrict__ results,
                                                      double* trace, int trace_len, int n_total, const int32_t* __restrict__ order) {
  static_assert(RPB == 1 || BLOCK == 64, "several registrations per workgroup: one wavefront each");
  static_assert(!SPLIT || (BLOCK == 64 && RPB == 1), "split mode: wavefront 0 is the one-wavefront solver, the others serve residual trips");
  constexpr int NT = PARAM == RANDT_PARAM_AMBIENT4 ? 4 : 3;
  constexpr int WAVES = BLOCK / 64;
  constexpr int PCAP = SPLIT ? SPLIT_PAIR_CAP : PAIR_CAP;
  __shared__ double red_all[RPB][2 * WAVES * 12];
  __shared__ int s_count_all[RPB][WAVES];
  __shared__ unsigned s_pairs_all[RPB][PCAP];
  // split mode: request block, the helpers' per-lane terms (six doubles per lane per helper), their bad-residual flags
  __shared__ SplitReq s_req;
  __shared__ __attribute__((aligned(16))) double s_part[SPLIT ? (SPLIT_MAXW - 1) * 64 * 6 : 2];
  __shared__ int s_badflag[SPLIT_MAXW];
  // Solver state that is not touched while a residual pass runs lives in LDS (one copy per registration, written by one
  // lane, read back with uniform addresses): best point, Jacobi scaling and its products, LM diagonal, scaled gradient /
  // J^T J.  That takes ~70 registers out of the pass and lets a third wavefront share the SIMD.
  __shared__ double cold_all[RPB * WAVES][56];  // one copy per WAVEFRONT: the two wavefronts of BLOCK = 128 run the solver redundantly and are only synchronised inside a pass

  if (RANDT_SOLVE_PRIO_PRO > 0) __builtin_amdgcn_s_setprio(RANDT_SOLVE_PRIO_PRO);
  const int sub = RPB > 1 ? (int)(threadIdx.x >> 6) : 0;
  const int tid = (RPB > 1 || SPLIT) ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const int split_wave = SPLIT ? (int)(threadIdx.x >> 6) : 0;   // split mode: every wavefront runs the prologue below for itself
  const int split_W = SPLIT ? (int)(blockDim.x >> 6) : 1;
  const int slot = blockIdx.x * RPB + sub;
  if (slot >= n_total) return;  // RPB > 1: a whole wavefront leaves; there is no workgroup barrier below in that mode
  // order (nullable, RPB > 1): the registrations of the batch sorted by descending size (k_solve_order), so that the four that
  // share a workgroup -- its LDS and its place in the dispatcher's queue are held until the slowest ends -- are of one length class
  const int pair = (RPB > 1 && order) ? order[slot] : slot;
  double* red = red_all[sub];
  int* s_count = s_count_all[sub];
  unsigned* s_pairs = s_pairs_all[sub];
  double* const cold = cold_all[(RPB > 1 || SPLIT) ? sub : (int)(threadIdx.x >> 6)];
  const bool w0 = (threadIdx.x & 63) == 0;  // the lane that writes the LDS-resident state
#define RANDT_COLD_SET(ref, val) do { const double v__ = (val); if (w0) (ref) = v__; } while (0)
#define RANDT_COLD_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
  const int fmap = fixed_idx ? fixed_idx[pair] : 0;
  const int mmap = moving_first + pair;
  const int k = P.k;
  int M = moving.counts[mmap];
  M = M > moving.cap ? moving.cap : M;

  Stage S;
  {
    // the table bases are the same in every lane of the wavefront: say so (scalar element offsets from the kernel-argument
    // pointers, which keeps them global-address-space pointers), so that record loads can use base + 32-bit offset addressing
    auto uniform_off = [](size_t v) {
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
      return (size_t)(((unsigned long long)hi << 32) | lo);
    };
    S.mov = reinterpret_cast<const float4*>(moving.cells + uniform_off((size_t)mmap * moving.cap));
    S.fix = reinterpret_cast<const float4*>(fixed.cells + uniform_off((size_t)fmap * fixed.cap));
  }
  S.corr = corr + (size_t)pair * moving.cap * k;
  S.n_slots = M * k;
  S.k = k;
  S.fixed_cap = fixed.cap;
  S.pairs = nullptr;
  S.n_pairs = 0;
  S.pack16 = (3 * fixed.cap <= 65536 && 3 * M <= 65536) ? 1 : 0;
  S.kmagic = k > 1 ? (unsigned)((0x100000000ull + (unsigned)k - 1) / (unsigned)k) : 0u;

  // number of residual blocks (addNDTFactor, ndt_matcher.cpp:217-246)
  int n_res = 0;
  for (int sidx = tid; sidx < S.n_slots; sidx += BLOCK) {
    const int ci = S.corr[sidx];
    n_res += (ci >= 0 && ci < fixed.cap) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n_res += __shfl_xor(n_res, off, 64);
  if (WAVES > 1) {
    if ((tid & 63) == 0) s_count[tid >> 6] = n_res;
    __syncthreads();
    n_res = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) n_res += s_count[w];
  }
  n_res = __builtin_amdgcn_readfirstlane(n_res);  // same in every lane: keep it (and what hangs off it) scalar
  int parity = 0;

  // Compact the valid correspondences once (ascending slot order): every pass then walks a dense list,
  // ceil(n_res / BLOCK) trips per lane instead of ceil(M k / BLOCK), with no per-pass index arithmetic.
  if (n_res > 0 && n_res <= PCAP && fixed.cap <= (int)PAIR_MASK + 1 && M <= (1 << (32 - PAIR_SHIFT))) {
    if (tid < 64 && split_wave == 0) {
      int n_out = 0;
      for (int base = 0; base < S.n_slots; base += 64) {
        const int slot = base + tid;
        const int cr = slot < S.n_slots ? S.corr[slot] : -1;
        const bool valid = cr >= 0 && cr < fixed.cap;
        const unsigned long long mask = __ballot(valid);
        if (valid) {
          const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          const unsigned mi = k == 1 ? (unsigned)slot : __umulhi((unsigned)slot, S.kmagic);
          s_pairs[n_out + rank] = S.pack16 ? ((3u * mi) << 16) | (3u * (unsigned)cr) : (mi << PAIR_SHIFT) | (unsigned)cr;
        }
        n_out += __popcll(mask);
      }
    }
    if (WAVES > 1 || SPLIT) {
      __syncthreads();
    } else {  // the list is written and read by the same wavefront
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    S.pairs = s_pairs;
    S.n_pairs = n_res;
  }

  double* tr = trace ? trace + (size_t)pair * trace_len : nullptr;
  if (tr && tid == 0 && split_wave == 0) tr[0] = 0.0;

  randt_result res;
  res.cost = res.final_cost = res.initial_cost = res.mu0 = 0.0;
  res.n_residuals = n_res;
  res.iterations = res.gnc_solves = res.n_evals = 0;
  res.termination = RANDT_TERM_NONE;
  res.status = 0;
  res.reserved[0] = res.reserved[1] = 0;

  double* const best = cold;     // [4]
  double* const x = cold + 36;   // [4] current point
  double* const cand = cold + 40;  // [4] candidate point
  double& minimum_cost = cold[44];
  double& summary_min = cold[45];
  double& x_norm = cold[46];
  {
    const double p0 = pose4[4 * (size_t)pair + 0], p1 = pose4[4 * (size_t)pair + 1];
    const double p2 = pose4[4 * (size_t)pair + 2], p3 = pose4[4 * (size_t)pair + 3];
    double x0[4];
    if (vec_like(PARAM)) {
      x0[0] = p2; x0[1] = p3; x0[2] = atan2(p1, p0); x0[3] = 0.0;  // trans.log()(2), ndt_matcher.cpp:439
    } else {
      x0[0] = p0; x0[1] = p1; x0[2] = p2; x0[3] = p3;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      RANDT_COLD_SET(x[i], x0[i]);
      RANDT_COLD_SET(best[i], x0[i]);
    }
    RANDT_COLD_SET(summary_min, 0.0);
  }
  RANDT_COLD_SYNC();

  if (n_res == 0) {
    // "WARNING: NO RESIDUALS ADDED!" (ndt_matcher.cpp:454-456): pose unchanged
    res.status = 1;
    if (tid == 0 && split_wave == 0) results[pair] = res;
    return;
  }
  if (SPLIT && S.n_pairs != n_res) {  // cannot happen (launch_one only takes split mode when M k <= SPLIT_PAIR_CAP etc.): fail loudly
    res.status = 2;
    res.termination = RANDT_TERM_FAILURE;
    if (tid == 0 && split_wave == 0) results[pair] = res;
    return;
  }
#ifdef RANDT_SPLIT_TIMING
  if (SPLIT && threadIdx.x == 0) {
    for (int i = 0; i < 9; ++i) s_split_tim[i] = 0;
    s_split_tim[9] = clock64();
  }
#endif
  if (SPLIT && split_wave != 0) {
    // the helpers serve one residual trip per pass and sleep; wavefront 0 is the registration's critical path -- and shares its
    // SIMD with helpers of the CU's other registration: it keeps the raised priority for the whole solve, the helpers give it up
    // (one 512-registration batch alone: 112.0 -> 104.4 us)
    __builtin_amdgcn_s_setprio(0);
    split_helper<D, PARAM, AM2>(S, &s_req, s_part + (size_t)(split_wave - 1) * 64 * 6, &s_badflag[split_wave], split_wave, tid);
    return;
  }
#define RANDT_EVAL(MODE, XP, OUT)                                                                                     \
  (SPLIT ? eval_pass_split<D, PARAM, MODE, AM2>(S, XP, L, OUT, &s_req, s_part, s_badflag, split_W, tid) \
         : eval_pass<D, PARAM, MODE, BLOCK, AM2>(S, XP, L, OUT, red, parity, tid))

  if (SPLIT) __builtin_amdgcn_s_setprio(3);  // wavefront 0 of a split-mode registration (see the helpers' branch above)
  else if (RANDT_SOLVE_PRIO_PRO > 0 && RANDT_SOLVE_PRIO_ALG != RANDT_SOLVE_PRIO_PRO) __builtin_amdgcn_s_setprio(RANDT_SOLVE_PRIO_ALG);
  // ---- raw residuals at the initial point -> gnc_mu (ndt_matcher.cpp:466-476)
  Loss L = AM2 ? make_loss_am2(P.loss_a, 1.0, P.weight) : make_loss(P.loss_a, P.alpha, 1.0, P.weight);
  Base cur, cnd;
  bool ok = RANDT_EVAL(0, x, cur);
  const double raw_max = cur.v[0];
  res.n_evals++;
  double gnc_mu = gnc_mu_start(raw_max, P.mu_scale, P.mu_cap);
  res.mu0 = gnc_mu;
  int term = RANDT_TERM_FAILURE;
  if (!ok) res.status = 2;

  if (ok) {
    do {
      gnc_mu = fmax(gnc_mu, 1.0);
      L = AM2 ? make_loss_am2(P.loss_a, gnc_mu, P.weight) : make_loss(P.loss_a, P.alpha, gnc_mu, P.weight);
      // ================= one ceres::Solve (TrustRegionMinimizer::Minimize) =================
      // The LDS-resident state is moved in BULK: one group of reads in front of the step computation, one behind the pass,
      // writes without a dependent read behind them -- a lone wavefront (split mode) otherwise pays an LDS round trip per
      // statement group.  The arithmetic is statement for statement the reference's.
      double step[NT], delta[NT];
      constexpr int NS = NT * (NT + 1) / 2;
      double* const sigma = cold + 4;  // [NT]
      double* const diag = cold + 8;   // [NT]  LM diagonal of the current point (kept across rejected steps: reuse_diagonal)
      double* const gs = cold + 12;    // [NT]  Jacobi-scaled gradient / J^T J (packed lower) at the current point
      double* const Hs = cold + 16;    // [NS]
      double* const ss = cold + 26;    // [NS]
      double g[NT], H[NS];
      double radius = P.r0, decrease = 2.0;
      bool step_ok = true;
      int num_invalid = 0, iteration = 0;
      double x0[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) x0[i] = best[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) RANDT_COLD_SET(x[i], x0[i]);
      RANDT_COLD_SET(x_norm, ambient_norm<PARAM>(x0));
      const bool e_ok = RANDT_EVAL(1, x0, cur);
      asm volatile("" ::: "memory");
      res.n_evals++;
      res.iterations++;
      if (!e_ok) {
        term = RANDT_TERM_FAILURE;
        res.status = 2;
        res.gnc_solves++;
        break;
      }
      double cost = cur.v[0];
      if (res.gnc_solves == 0) res.initial_cost = cost;
      bool gconv;
      {
        double xr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[i] = x[i];
        to_param<PARAM, NT>(cur, xr, g, H);
        double sg[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {  // jacobi scaling 1 / (1 + sqrt(H_ii)), fixed per solve; Newton-refined rsqrt / rcp (~1 ulp:
          const double hii = H[sym(i, i)];  // the scaling is a change of variables, exact arithmetic does not see it)
          sg[i] = jacobi_scale(hii);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          RANDT_COLD_SET(sigma[i], sg[i]);
          RANDT_COLD_SET(gs[i], g[i] * sg[i]);
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            const double sij = sg[i] * sg[j];
            const double hs = H[sym(i, j)] * sij;
            RANDT_COLD_SET(ss[sym(i, j)], sij);
            RANDT_COLD_SET(Hs[sym(i, j)], hs);
            if (i == j) RANDT_COLD_SET(diag[i], fmin(fmax(hs, P.dmin), P.dmax));  // LevenbergMarquardtStrategy: clamp of the scaled J^T J diagonal
          }
        }
        gconv = gradient_converged<PARAM, NT, true>(xr, g, P.gtol);
        // FinalizeIterationAndCheckIfMinimizerCanContinue of iteration zero: the starting point is the best one so far
        RANDT_COLD_SET(minimum_cost, cost);
#pragma unroll
        for (int i = 0; i < 4; ++i) RANDT_COLD_SET(best[i], xr[i]);
        RANDT_COLD_SET(summary_min, cost);
        RANDT_COLD_SYNC();
      }
      trace_push(tr, trace_len, tid, cost, radius, 0);

      for (;;) {
        // ---- FinalizeIterationAndCheckIfMinimizerCanContinue (the copy of an improved point to the user parameters happens
        // where the point is accepted: nothing can leave the loop between there and here)
        if (iteration >= P.max_it) { term = RANDT_TERM_NO_CONVERGENCE; break; }
        if (step_ok && gconv) { term = RANDT_TERM_CONVERGENCE_GRADIENT; break; }
        if (uni(radius <= P.rmin)) { term = RANDT_TERM_CONVERGENCE_RADIUS; break; }
        ++iteration;
        res.iterations++;

        // ---- LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled normal equations
        double Hr[NS], gr[NT], dg[NT], sg[NT], xr[4];
#pragma unroll
        for (int i = 0; i < NS; ++i) Hr[i] = Hs[i];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          gr[i] = gs[i];
          dg[i] = diag[i];
          sg[i] = sigma[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[i] = x[i];
        double mcc;
        const bool solved = compute_step<NT>(Hr, gr, dg, radius, step, mcc);  // + model_cost_change = -(step.g + step^T H step / 2)
        const bool valid = uni(solved && mcc > 0.0);
        if (!valid) {
          // ---- HandleInvalidStep
          if (++num_invalid >= P.max_invalid) { term = RANDT_TERM_FAILURE; break; }
          radius = radius * fast_rcp(decrease);  // decrease is a power of two: exact
          decrease *= 2.0;
          step_ok = false;
          RANDT_COLD_SET(summary_min, fmin(summary_min, cost));
          RANDT_COLD_SYNC();
          trace_push(tr, trace_len, tid, cost, radius, 3);
          continue;
        }
        num_invalid = 0;
#pragma unroll
        for (int i = 0; i < NT; ++i) delta[i] = step[i] * sg[i];
        double cnew[4];
        plus<PARAM, true>(xr, delta, cnew);
#pragma unroll
        for (int i = 0; i < 4; ++i) RANDT_COLD_SET(cand[i], cnew[i]);
        RANDT_COLD_SYNC();

        // ---- candidate cost (+ speculative gradient / J^T J); the pass takes the point from registers
        const bool c_ok = RANDT_EVAL(1, cnew, cnd);
        asm volatile("" ::: "memory");  // the LDS-resident state is re-read after the pass, not carried through it in registers
        res.n_evals++;
        const double cand_cost = c_ok ? cnd.v[0] : DBL_MAX;

        // ---- behind the pass: one group of reads
        double xc[4], cc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xc[i] = x[i];
          cc[i] = cand[i];
        }
        const double xn_cur = x_norm, smin = summary_min, mincost = minimum_cost;
        double sgn[NT], ssn[NS];
#pragma unroll
        for (int i = 0; i < NT; ++i) sgn[i] = sigma[i];
#pragma unroll
        for (int i = 0; i < NS; ++i) ssn[i] = ss[i];

        // ---- ParameterToleranceReached / FunctionToleranceReached (before accept/reject)
        const double sn2 = step_norm_sq<PARAM>(xc, cc);
        const double ptol_abs = P.ptol * (xn_cur + P.ptol);
        if (uni(sn2 <= ptol_abs * ptol_abs)) { term = RANDT_TERM_CONVERGENCE_PARAMETER; break; }
        const double cost_change = cost - cand_cost;
        if (uni(fabs(cost_change) <= P.ftol * cost)) { term = RANDT_TERM_CONVERGENCE_FUNCTION; break; }

        const double rel = c_ok ? cost_change * fast_rcp(mcc) : -DBL_MAX;
        if (uni(rel > P.min_rel)) {
          // ---- HandleSuccessfulStep
#pragma unroll
          for (int i = 0; i < 4; ++i) RANDT_COLD_SET(x[i], cc[i]);
          RANDT_COLD_SET(x_norm, ambient_norm<PARAM>(cc));
          cost = cand_cost;
          to_param<PARAM, NT>(cnd, cc, g, H);
#pragma unroll
          for (int i = 0; i < NT; ++i) RANDT_COLD_SET(gs[i], g[i] * sgn[i]);
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
              const double hs = H[sym(i, j)] * ssn[sym(i, j)];
              RANDT_COLD_SET(Hs[sym(i, j)], hs);
              if (i == j) RANDT_COLD_SET(diag[i], fmin(fmax(hs, P.dmin), P.dmax));  // the LM diagonal of the new point
            }
          gconv = gradient_converged<PARAM, NT, true>(cc, g, P.gtol);
          step_ok = true;
          radius = radius_after_success(radius, rel, P.rmax);
          decrease = 2.0;
          // FinalizeIteration...: x is copied to the user parameters when the accepted point lowers the minimum cost
          if (uni(cost < mincost)) {
            RANDT_COLD_SET(minimum_cost, cost);
#pragma unroll
            for (int i = 0; i < 4; ++i) RANDT_COLD_SET(best[i], cc[i]);
          }
          RANDT_COLD_SET(summary_min, fmin(smin, cost));
          RANDT_COLD_SYNC();
          trace_push(tr, trace_len, tid, cost, radius, 1);
        } else {
          step_ok = false;
          radius = radius * fast_rcp(decrease);  // decrease is a power of two: exact
          decrease *= 2.0;
          RANDT_COLD_SET(summary_min, fmin(smin, cand_cost));
          RANDT_COLD_SYNC();
          trace_push(tr, trace_len, tid, cand_cost, radius, 2);
        }
      }
      res.gnc_solves++;
      gnc_mu /= P.gnc_div;
    } while (uni(gnc_mu > P.mu_stop));
  }

#undef RANDT_EVAL
  if (SPLIT) {  // the helpers leave at their next barrier
    if (tid == 0) s_req.mode = -1;
    __syncthreads();
  }
#ifdef RANDT_SPLIT_TIMING
  if (SPLIT && blockIdx.x == 0 && threadIdx.x == 0) {
    ST(4);
    for (int i = 0; i < 9; ++i) g_randt_split_timing[i] = s_split_tim[i];
    g_randt_split_timing[8] = res.n_evals;
  }
#endif
  res.termination = term;
  res.final_cost = summary_min;
  res.cost = summary_min / (double)n_res;
  if (tid == 0) {
    double* po = pose4 + 4 * (size_t)pair;
    if (vec_like(PARAM)) {
      double c = cos(best[2]), s = sin(best[2]);  // Sophus::SE2d(rot, pos), ndt_matcher.cpp:486
      so2_normalize(c, s);
      po[0] = c; po[1] = s; po[2] = best[0]; po[3] = best[1];
    } else {
      po[0] = best[0]; po[1] = best[1]; po[2] = best[2]; po[3] = best[3];
    }
    results[pair] = res;
  }
}

// f-3: problem.Evaluate(apply_loss_function = true) at many poses for ONE frozen correspondence set
// (Matcher::estimateTransformGlobalBNB, ndt_matcher.cpp:561-576): one wavefront per pose, cost only.
template <int D>
__global__ __launch_bounds__(64) void k_eval_cost(MapView fixed, int fmap, MapView moving, int mmap, const int32_t* __restrict__ corr,
                                                  int k, double scale, double alpha, const double* __restrict__ poses4,
                                                  double* __restrict__ cost, int32_t* __restrict__ n_res_out) {
  const int p = blockIdx.x, lane = threadIdx.x;
  int M = moving.counts[mmap];
  M = M > moving.cap ? moving.cap : M;
  const float4* mov = reinterpret_cast<const float4*>(moving.cells + (size_t)mmap * moving.cap);
  const float4* fix = reinterpret_cast<const float4*>(fixed.cells + (size_t)fmap * fixed.cap);
  const Loss L = make_loss(scale, alpha, 1.0, 1.0);  // BarronLoss(scale, alpha): b = a^2, no ScaledLoss (:517)
  const double* x = poses4 + 4 * (size_t)p;
  const double inv = rsqrt(x[0] * x[0] + x[1] * x[1]);
  const double c = x[0] * inv, s = x[1] * inv, tx = x[2], ty = x[3];
  const Rot rot = make_rot(c, s);
  double acc = 0.0;
  int n = 0;
  for (int slot = lane; slot < M * k; slot += 64) {
    const int ci = corr[slot];
    if (ci < 0 || ci >= fixed.cap) continue;
    double jb[3];
    const double sq = residual_sq<D, false>(mov + (size_t)(slot / k) * 3, fix + (size_t)ci * 3, rot, tx, ty, jb);
    ++n;
    if (L.mode == 2) {
      const double iu = 1.0 / (sq * L.ts + 1.0);
      acc += L.half_w_pre * (iu - 1.);
    } else {
      double r0, r1, r2;
      loss_eval(L, sq, r0, r1, r2);
      acc += 0.5 * r0;
    }
  }
  acc = wave_sum(acc);
  const double nn = wave_sum((double)n);
  if (lane == 0) {
    cost[p] = acc;
    if (n_res_out && p == 0) n_res_out[0] = (int)nn;
  }
}

// Registrations of a batch in descending order of their residual trips, ceil(cells x k / 64) -- what a one-wavefront solve's
// duration is made of besides its (unpredictable) number of passes: round 5, tools/solve_length_probe.py: the trips explain
// 78 % of the variance of passes x trips over the config-4 batch, the passes themselves cannot be predicted (R^2 <= 0.06).
// A counting sort over 64 keys by ONE workgroup; the order inside a key is whatever the atomics give (a placement, not a result).
__global__ __launch_bounds__(1024) void k_solve_order(MapView moving, int moving_first, int n, int k, int32_t* __restrict__ order) {
  __shared__ int hist[64], base[64];
  const int tid = threadIdx.x;
  if (tid < 64) hist[tid] = 0;
  __syncthreads();
  auto key_of = [&](int i) {
    int M = moving.counts[moving_first + i];
    M = M > moving.cap ? moving.cap : M;
    const int trips = (M * k + 63) >> 6;
    return 63 - (trips > 63 ? 63 : trips);  // ascending key = descending size
  };
  for (int i = tid; i < n; i += 1024) atomicAdd(&hist[key_of(i)], 1);
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int b = 0; b < 64; ++b) {
      base[b] = acc;
      acc += hist[b];
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) order[atomicAdd(&base[key_of(i)], 1)] = i;
}

template <int D, int PARAM, int BLOCK, bool AM2, int RPB = 1>
int launch_cfg(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving, int moving_first,
               int n_pairs, const int32_t* d_corr, const SolveParams& P, double* d_pose4, randt_result* d_results) {
  const int32_t* d_order = nullptr;
  // measured (round 5, profiles/experiments/r05_solve_length_grouping.md): a launch that fills the chip several times over gains
  // 3.8 % (8192 registrations: 595 -> 572 us); the 512-registration launches of the 16-stream pipeline gain nothing -- wavefronts
  // that end early give their ISSUE slots to their SIMD neighbours anyway, only LDS and wave slots wait for the slowest -- so the
  // sort is taken from eight registrations per compute unit upwards (RANDT_SOLVE_GROUP = 0 / 1: never / always)
  const bool grouped = ctx->solve_group < 0 ? n_pairs >= 8 * ctx->n_cus : ctx->solve_group > 0;
  if (RPB > 1 && grouped && n_pairs >= 2 * RPB) {
    const size_t need = sizeof(int32_t) * (size_t)n_pairs;
    if (need > ctx->order_ws_bytes) {
      if (ctx->order_ws) {
        RANDT_HIP_CHECK(ctx, randt_sync(ctx));
        RANDT_HIP_CHECK(ctx, randt_hip_free(ctx, ctx->order_ws));
        ctx->order_ws = nullptr;
        ctx->order_ws_bytes = 0;
      }
      RANDT_HIP_CHECK(ctx, randt_hip_malloc(ctx, &ctx->order_ws, need + need / 2 + 256));
      ctx->order_ws_bytes = need + need / 2 + 256;
    }
    hipLaunchKernelGGL(k_solve_order, dim3(1), dim3(1024), 0, ctx->stream, moving, moving_first, n_pairs, P.k, static_cast<int32_t*>(ctx->order_ws));
    d_order = static_cast<const int32_t*>(ctx->order_ws);
  }
  hipLaunchKernelGGL((k_solve<D, PARAM, BLOCK, AM2, RPB>), dim3((n_pairs + RPB - 1) / RPB), dim3(BLOCK * RPB), 0, ctx->stream, fixed,
                     d_fixed_idx, moving, moving_first, d_corr, P, d_pose4, d_results, ctx->d_trace, ctx->trace_len, n_pairs, d_order);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

// Wavefronts per registration for a batch of n_pairs (0 = the one-wavefront kernel).  The split-mode kernels hold four
// wavefronts per SIMD (16 slots per CU).  Only multiples of four are taken: a workgroup's wavefronts are dealt round-robin
// over the CU's four SIMDs, so W = 8 packs two workgroups per CU as 4-4-4-4 whatever their start SIMD, whereas W = 6
// (2-2-1-1) may not fit beside a resident one and then waits for a second round (measured on 512 registrations: W = 6 at
// three wavefronts per SIMD 200 us, W = 4 133 us).  W = 8 while two registrations per CU cover the batch (512 on MI355X:
// 117 us against 148 us for one wavefront each; 64 registrations: 97 against 128), W = 4 up to three per CU; beyond that
// every SIMD has a registration of its own and splitting only adds barriers.  RANDT_SOLVE_SPLIT = 0 / 2..8 overrides
// (experiments: tools/split_probe.py).
int split_width(randt_ctx* ctx, const MapView& fixed, const MapView& moving, int n_pairs, int k) {
  if ((long long)moving.cap * k > SPLIT_PAIR_CAP || fixed.cap > (int)PAIR_MASK + 1 || moving.cap > (1 << (32 - PAIR_SHIFT))) return 0;
  if (ctx->solve_split >= 0) return ctx->solve_split < 2 ? 0 : (ctx->solve_split > SPLIT_MAXW ? SPLIT_MAXW : ctx->solve_split);
  if (n_pairs > 3 * ctx->n_cus) return 0;
  if (randt_throughput_placement(ctx)) return 0;  // the caller says, or the library sees, that other batches share the chip
  return n_pairs <= 2 * ctx->n_cus ? 8 : 4;
}

template <int D, int PARAM>
int launch_one(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving, int moving_first,
               int n_pairs, const int32_t* d_corr, const SolveParams& P, double* d_pose4, randt_result* d_results, int block) {
#define RANDT_CFG(BB, AA) \
  return launch_cfg<D, PARAM, BB, AA>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_corr, P, d_pose4, d_results)
  const bool am2 = P.alpha == -2.0;
  if (block == 64 && am2) {
    const int W = split_width(ctx, fixed, moving, n_pairs, P.k);
    ctx->last_placement = W >= 2 ? W : 0;
    randt_note_enqueue(ctx);
    if (W >= 2) {
      hipLaunchKernelGGL((k_solve<D, PARAM, 64, true, 1, true>), dim3(n_pairs), dim3(64 * W), 0, ctx->stream, fixed, d_fixed_idx, moving,
                         moving_first, d_corr, P, d_pose4, d_results, ctx->d_trace, ctx->trace_len, n_pairs, nullptr);
      RANDT_HIP_CHECK(ctx, hipGetLastError());
      return RANDT_OK;
    }
  }
  if (block == 64 && am2 && ctx->solve_rpb > 1) {
    if (ctx->solve_rpb == 2)
      return launch_cfg<D, PARAM, 64, true, 2>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_corr, P, d_pose4, d_results);
    if (ctx->solve_rpb == 8)
      return launch_cfg<D, PARAM, 64, true, 8>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_corr, P, d_pose4, d_results);
    return launch_cfg<D, PARAM, 64, true, 4>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_corr, P, d_pose4, d_results);
  }
  if (block == 64) {
    if (am2) RANDT_CFG(64, true); else RANDT_CFG(64, false);
  }
  if (am2) RANDT_CFG(128, true); else RANDT_CFG(128, false);
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
  P.mu_cap = pow(mp->gnc_divisor, (double)(mp->gnc_steps - 1));
  P.mu_stop = 1.0 / sqrt(mp->gnc_divisor);
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
  const int block = ctx->solve_block;
  const int d3 = mp->use_intensity ? 1 : 0;
#define RANDT_DISPATCH(DD, PP) \
  return launch_one<DD, PP>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_corr, P, d_pose4, d_results, block)
  switch (mp->parameterization) {
    case RANDT_PARAM_MANIFOLD:
      if (d3) RANDT_DISPATCH(3, RANDT_PARAM_MANIFOLD); else RANDT_DISPATCH(2, RANDT_PARAM_MANIFOLD);
    case RANDT_PARAM_AMBIENT4:
      if (d3) RANDT_DISPATCH(3, RANDT_PARAM_AMBIENT4); else RANDT_DISPATCH(2, RANDT_PARAM_AMBIENT4);
    case RANDT_PARAM_VECTOR:
      if (d3) RANDT_DISPATCH(3, RANDT_PARAM_VECTOR); else RANDT_DISPATCH(2, RANDT_PARAM_VECTOR);
    case RANDT_PARAM_ANALYTIC:
      if (d3) RANDT_DISPATCH(3, RANDT_PARAM_ANALYTIC); else RANDT_DISPATCH(2, RANDT_PARAM_ANALYTIC);
    default:
      return randt_set_error(ctx, RANDT_ERR_INVALID, "unknown parameterization", hipSuccess);
  }
#undef RANDT_DISPATCH
}

int launch_eval_cost(randt_ctx* ctx, const MapView& fixed, int fmap, const MapView& moving, int mmap, const int32_t* d_corr, int k,
                     int use_intensity, double scale, double alpha, const double* d_poses4, int n_poses, double* d_cost,
                     int32_t* d_n_res) {
  if (n_poses <= 0) return RANDT_OK;
  if (use_intensity)
    hipLaunchKernelGGL(k_eval_cost<3>, dim3(n_poses), dim3(64), 0, ctx->stream, fixed, fmap, moving, mmap, d_corr, k, scale, alpha,
                       d_poses4, d_cost, d_n_res);
  else
    hipLaunchKernelGGL(k_eval_cost<2>, dim3(n_poses), dim3(64), 0, ctx->stream, fixed, fmap, moving, mmap, d_corr, k, scale, alpha,
                       d_poses4, d_cost, d_n_res);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
