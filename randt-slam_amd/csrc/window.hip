// Fixed-lag window solve for gfx950: Matcher::estimateTransformCeres on the device.
//
// Replaces (paths relative to /root/reference/ros/ndt_radar_slam/):
//   src/ndt_registration/ndt_matcher.cpp:322-424          estimateTransformCeres (problem wiring, GNC loop, ceres::Solve)
//   include/ndt_registration/ceres_residuals.h:62-83      predictSE2
//   include/ndt_registration/ceres_residuals.h:621-679    MotionModelFactorSE2
//   include/ndt_registration/ceres_residuals.h:338-370    RotationalResidualSE2
//   include/ndt_registration/ceres_residuals.h:520-552    NDTFrameToMapIntensityFactorResidualSE2 (and the 2-D form)
//   Ceres 2.1.0 trust-region LM / Sophus 1.22.10 SE(2) manifold (un-vendored), as in solve.hip.
//
// One 512-thread workgroup owns one window problem (<= 3 optimised states, <= 2 fixed maps, 21-32
// tangent dimensions) and runs the whole GNC x LM loop without host round trips:
//   * NDT terms: six wavefronts share the (state, fixed map) terms (a term is split over 6 / 3 / 2 of them when there are
//     1 / 2 / 3 terms); the cell records of a lane's first three 64-slot trips are staged in LDS once per solve (the
//     association is frozen; the kernel declares 141 KB of the CU's 160 KB), ten fp64 base sums per wavefront (one lane-swap
//     reduction), combined per state in share order.  Wavefronts w and w + 4 share a SIMD: the heavy shares sit on 0, 1, 3,
//     the light ones on their partners 4, 5, 7, and SIMD 2 holds the factor wavefront (6) and the spare one (2: step norm);
//   * motion / IMU factors: one lane per factor evaluates residual + analytic Jacobian (right perturbations, verified
//     against finite differences in tests/test_oracle_window.py) with small-argument sin / cos / atan2; a diagonal
//     square-root information (every shipped configuration) is applied by the assembly itself, a full one by all threads;
//   * J^T J / J^T r: one thread per entry of the upper triangle gathers the factor blocks and the per-state 3x3 NDT blocks
//     (T G T^T with Sophus' PlusJacobian) into LDS, mirrored on the way out, with the scaled copy and the LM diagonal;
//   * the damped solve: banded block Gauss-Jordan, the band [state b, state b + 1] in one 16-lane DPP row, pivot rows
//     broadcast by v_fmac_f64's row_newbcast modifier (pivot_group: one assembly block per pivot step with the next
//     pivot's reciprocal woven in).  Rejections come in chains -- the reference keeps Ceres' initial radius of 1e4, which
//     costs five rejected candidates at the start of EVERY GNC stage -- so wavefronts 0..6 solve the running radius and the
//     next six at once when such a chain is due (band_solve levels); windows with a 9-dimensional state block (IMU bias)
//     take the dense register Gauss-Jordan (gj_dense_solve);
//   * model-cost change (wavefront 1), Plus on every manifold block (wavefront 0), convergence tests / accept-reject /
//     radius update redundantly by every lane from broadcast scalars (uniform control flow).
// The candidate point is evaluated with its Jacobians so that an accepted step costs one pass.
#include <float.h>

#include "randt_internal.h"
#include "solve_math.h"

#ifdef RANDT_TIMING
__device__ long long g_randt_win_timing[16];
extern "C" int randt_debug_win_timing(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_randt_win_timing), sizeof(long long) * 16);
}
#define WT_DECL long long wt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long wt_last = wall_clock64();
#define WT(slot) do { const long long now_ = wall_clock64(); wt_acc[slot] += now_ - wt_last; wt_last = now_; } while (0)
#define WT_FLUSH do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 13; ++i_) g_randt_win_timing[i_] = wt_acc[i_]; } while (0)
#else
#define WT_DECL
#define WT(slot) do {} while (0)
#define WT_FLUSH do {} while (0)
#endif

#define WIN_BLOCK 512
#define WIN_WAVES 8
#define WIN_NDT_WAVES 6  // wavefronts that stream NDT slots during a pass
// Roles of the eight wavefronts in a pass.  Wavefronts w and w + 4 share a SIMD, and with three terms a term's first share
// has three 64-slot trips, its second two: the heavy shares go to wavefronts 0, 1, 3, their SIMD partners 4, 5, 7 take the
// light ones, and SIMD 2 holds the factor wavefront (6) next to the one (2) that only takes the step norm.
#define WIN_FACTOR_WAVE 6
#define WIN_SPARE_WAVE 2
__device__ __forceinline__ int ndt_share_of_wave(int wave) {  // 0..5 = share (term = share / wpt, part = share % wpt), -1: none
#ifdef RANDT_WIN_PLAIN_ROLES
  return wave == 7 ? 5 : (wave == 2 ? -1 : (wave < 2 ? wave : (wave < 6 ? wave - 1 : -1)));  // 0 1 . 2 3 4 F 5
#else
  return wave == 0 ? 0 : (wave == 4 ? 1 : (wave == 1 ? 2 : (wave == 5 ? 3 : (wave == 3 ? 4 : (wave == 7 ? 5 : -1)))));
#endif
}
#define WIN_NMAX 32  // tangent dimensions
#define WIN_SMAX 3   // optimised states
#ifndef WIN_LEVELS
#define WIN_LEVELS 7  // damped solves taken at once, one wavefront each (<= 8): the running radius and the next six a rejection chain
                      // visits -- chains of five or six rejections and the accepted step behind them (5 / 6 / 7 / 8 measured: 7)
#endif

#include "window_math.h"

using namespace randt_solve;
using namespace randt_window;

namespace {

// (SE(2) pieces, motion / IMU factors, state record layout: window_math.h)

// ---------------------------------------------------------------- loss (as in solve.hip) -------
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}


// ---------------------------------------------------------------- banded block elimination ------
// The damped normal equations of the window are BLOCK TRIDIAGONAL in the state order (motion / IMU factors couple
// neighbouring states only, NDT terms are unary).  With every state's tangent block padded to 8 rows, block step b works on
// the 16 x 16 band [state b, state b + 1] inside ONE 16-lane DPP row (lane = row, register = column): the pivot row is
// broadcast by the DPP row_newbcast modifier of v_fmac_f64 itself (the only DPP control the 64-bit ALU accepts on
// gfx90a+), so an update costs ONE instruction instead of two v_readlane + one FMA.  DPP row b of the wavefront holds block
// step b (all four load their originals at once); the Schur-updated rows of state b + 1 move on through LDS.
template <int L>
__device__ __forceinline__ double bcast16(double v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + L, 0xf, 0xf, true);  // v_mov_b64_dpp row_newbcast:L
}
// One pivot step as ONE assembly block.  Updates: d += (lane JJ of d's 16-lane row) * m for the NC live columns behind the
// pivot and the right-hand side -- v_fmac_f64 with the DPP row_newbcast modifier, which the compiler cannot emit (its DPP
// combiner leaves 64-bit operations alone).  LA (look-ahead): the reciprocal of the NEXT pivot (hardware seed + two Newton
// steps on its broadcast value, out in rpn) is woven between the updates, right behind the update of its column, so the
// dependent rcp -> fma -> fma -> fma -> fma chain runs in the shadow of the other columns' updates instead of in front of
// them.  Inline assembly is opaque to the compiler's hazard recogniser; the block keeps the gfx9 rules itself: 2 wait
// states between a VALU write and a DPP read of the same register (updates write distinct registers; a leading s_nop
// covers whatever the compiler placed in front), 1 wait state behind v_rcp_f64.  tools/check_dpp_hazards.py verifies
// the built object.  (The case table is generated: operands 0 .. NC = C[JJ+1 .. JJ+NC], B.)
template <int JJ, int NC, bool LA>
__device__ __forceinline__ void pivot_group(double (&C)[16], double& B, double nf, double& rpn) {
  [[maybe_unused]] double t0, t1 = 0.0, t2;
  if constexpr (LA == true && NC == 1) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\ts_nop 0\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]" : "+v"(C[JJ + 1]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 2) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\ts_nop 0\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 3) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\ts_nop 0\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 4) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 5) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 6) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64 %[t1], %[t1], %[t2]" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 7) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 8) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 9) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 10) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 11) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 12) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %12, %12, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(C[JJ + 12]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 13) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %12, %12, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %13, %13, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(C[JJ + 12]), "+v"(C[JJ + 13]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 14) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %12, %12, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %13, %13, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %14, %14, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(C[JJ + 12]), "+v"(C[JJ + 13]), "+v"(C[JJ + 14]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == true && NC == 15) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %[t0], %0 row_newbcast:%[l1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %[t1], %[t0]\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %[t2], -%[t0], %[t1], 1.0\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %12, %12, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %[t1], %[t1], %[t2]\n\tv_fmac_f64_dpp %13, %13, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %14, %14, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %15, %15, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(C[JJ + 12]), "+v"(C[JJ + 13]), "+v"(C[JJ + 14]), "+v"(C[JJ + 15]), "+v"(B), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) : [m] "v"(nf), [l] "n"(JJ), [l1] "n"(JJ + 1));
  if constexpr (LA == false && NC == 0) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 1) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 2) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 3) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 4) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 5) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 6) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 7) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 8) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 9) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 10) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 11) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 12) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %12, %12, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(C[JJ + 12]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 13) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %12, %12, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %13, %13, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(C[JJ + 12]), "+v"(C[JJ + 13]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 14) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %12, %12, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %13, %13, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %14, %14, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(C[JJ + 12]), "+v"(C[JJ + 13]), "+v"(C[JJ + 14]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA == false && NC == 15) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %8, %8, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %9, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %10, %10, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %11, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %12, %12, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %13, %13, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %14, %14, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %15, %15, %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf" : "+v"(C[JJ + 1]), "+v"(C[JJ + 2]), "+v"(C[JJ + 3]), "+v"(C[JJ + 4]), "+v"(C[JJ + 5]), "+v"(C[JJ + 6]), "+v"(C[JJ + 7]), "+v"(C[JJ + 8]), "+v"(C[JJ + 9]), "+v"(C[JJ + 10]), "+v"(C[JJ + 11]), "+v"(C[JJ + 12]), "+v"(C[JJ + 13]), "+v"(C[JJ + 14]), "+v"(C[JJ + 15]), "+v"(B) : [m] "v"(nf), [l] "n"(JJ));
  if constexpr (LA) rpn = t1;
}
// Gauss-Jordan step on pivot JJ of the band's first block: every other lane of the row (rows above, below and of the next
// state) subtracts its multiple of the pivot row from columns JJ+1 .. W-1 and the right-hand side.  rpb: reciprocal of the
// pivot, same in every lane of the row (in); of the next pivot (out, LA).
template <int JJ, int W, bool LA>
__device__ __forceinline__ void band_pivot(double (&C)[16], double& B, double& RD, int l, double& rpb) {
  const bool me = l == JJ;
  RD = me ? rpb : RD;
  const double cm = me ? 0.0 : C[JJ];
  const double nf = -(cm * rpb);
  pivot_group<JJ, W - 1 - JJ, LA>(C, B, nf, rpb);
}
// sz (uniform): tangent dimensions of the block's state; the padding rows behind them are identity rows nobody couples to,
// their pivot steps are skipped (constant-velocity windows: 3 + 6 + 6 + 6 of 4 x 8)
// SZ: tangent dimensions of the block's state, compile-time for the shipped layouts (constant velocity 3 | 6 6 6, constant
// acceleration 5 | 8 8 8): straight-line code -- with run-time skips the register allocator shuffled the live columns
// through copies at every join.  The padding rows behind SZ are identity rows nobody couples to; their pivot steps are skipped.
template <int W, int SZ>
__device__ __forceinline__ void band_block_n(double (&C)[16], double& B, double& RD, int l) {
  double rpb = fast_rcp(bcast16<0>(C[0]));
  band_pivot<0, W, (SZ > 1)>(C, B, RD, l, rpb);
  if constexpr (SZ > 1) band_pivot<1, W, (SZ > 2)>(C, B, RD, l, rpb);
  if constexpr (SZ > 2) band_pivot<2, W, (SZ > 3)>(C, B, RD, l, rpb);
  if constexpr (SZ > 3) band_pivot<3, W, (SZ > 4)>(C, B, RD, l, rpb);
  if constexpr (SZ > 4) band_pivot<4, W, (SZ > 5)>(C, B, RD, l, rpb);
  if constexpr (SZ > 5) band_pivot<5, W, (SZ > 6)>(C, B, RD, l, rpb);
  if constexpr (SZ > 6) band_pivot<6, W, (SZ > 7)>(C, B, RD, l, rpb);
  if constexpr (SZ > 7) band_pivot<7, W, false>(C, B, RD, l, rpb);
}
template <int W>
__device__ __forceinline__ void band_block(double (&C)[16], double& B, double& RD, int l, int sz) {
  switch (sz) {  // uniform
    case 3: band_block_n<W, 3>(C, B, RD, l); break;
    case 5: band_block_n<W, 5>(C, B, RD, l); break;
    case 6: band_block_n<W, 6>(C, B, RD, l); break;
    case 7: band_block_n<W, 7>(C, B, RD, l); break;
    default: band_block_n<W, 8>(C, B, RD, l); break;  // any other size <= 8: identity padding pivots are harmless
  }
}

struct Shared {
  double xs[2][WIN_SMAX + 1][ST_STRIDE];  // states: buffer p = current, 1-p = candidate
  double Ju[2][WIN_SMAX][128];            // unweighted motion Jacobians at current / candidate (zeroed ONCE: a factor always
                                          // writes the same entries)
  double ru[2][WIN_SMAX][8];
  double Jf[2][WIN_SMAX][128];            // weighted motion Jacobians (only when the square-root information is not diagonal)
  double rf[2][WIN_SMAX][8];
  double fc[2][WIN_SMAX];                 // 1/2 |weighted residual|^2 of factor f (diagonal case: summed by the factor's own lane)
  unsigned short tri[WIN_NMAX * (WIN_NMAX + 1) / 2];  // upper-triangle entry e -> a | b << 8 (a <= b)
  int sw_first[WIN_SMAX + 1], sw_cnt[WIN_SMAX + 1];   // shares (ndt_share_of_wave) that stream the NDT terms of state j: [first, first + cnt)
  double d2[8];                           // squared diagonal of the square-root information
  int sq_diag;                            // it IS diagonal (shipped configurations): weighting folded into the assembly
  double J2[2][WIN_SMAX][16];             // IMU factors
  double r2[2][WIN_SMAX][2];
  double H[WIN_NMAX * WIN_NMAX];
  double Hs[WIN_NMAX * WIN_NMAX];
  double g[WIN_NMAX], gs[WIN_NMAX], sigma[WIN_NMAX], diag[WIN_NMAX];
  double step[WIN_LEVELS][WIN_NMAX], delta[WIN_LEVELS][WIN_NMAX];  // level q: the step for the radius after q rejections (band_solve)
  double solved[WIN_LEVELS];                    // the solve succeeded (pivots positive, step finite)
  double red[2][WIN_WAVES][34];
  double scal[8];  // 0 mcc, 1 sn2, 2 x_norm, 3 solved, 4 gconv
  int lcol[WIN_SMAX][WIN_NMAX];   // tangent column -> local column of motion factor f (-1 none)
  int lcol2[WIN_SMAX][WIN_NMAX];  // ... of IMU factor f
  int pose_of[WIN_NMAX];          // tangent column -> state whose pose block holds it (-1 none)
  int wave_state[WIN_WAVES];      // state of the NDT term share s streams (-1: none)
  // banded block solve (band_solve below): per-lane LDS byte offsets of the 16 band columns + right-hand side, the lane's
  // tangent row (-1: padding), damped diagonal, constants 0 / 1 for padding entries, hand-over buffers between block steps
  int boff[18][64];      // bit 0 set: the entry is the row's damped diagonal, in dd[level]
  double dd[WIN_LEVELS][WIN_NMAX];
  double cst[2];
  double xfer[WIN_LEVELS][8][10];
  int bsize[WIN_SMAX + 1];  // tangent dimensions of state b
  int off_tan[RANDT_WIN_MAX_STATES][5], off_amb[RANDT_WIN_MAX_STATES][5];  // WinDesc's, for the lane-indexed readers (Plus, step norms, assembly)
  int band_ok;
  // The cell records of every lane's first three NDT trips, staged once per solve (the association is frozen): a lone
  // workgroup has nothing to hide the L2 latency of the record gathers behind, and they sit on the pass's critical path
  // three times per pass.  [share][trip][group][lane]: moving record floats 0-3, 4-7, fixed 0-3, 4-7, (moving 8, fixed 8, -, -).
  // gfx950 lets one workgroup declare all 160 KiB of the CU's LDS.
  float4 recs[WIN_NDT_WAVES][3][5][64];
  Loss loss;  // robust loss of the running GNC step (uniform; in LDS so that it does not occupy ~20 registers across the solve)
};

// Unweighted motion / IMU factors at xs[buf], one lane per factor (called by ONE wavefront).
__device__ void factors_unweighted(const WinDesc& W, Shared& sh, int buf) {
  const int lane = threadIdx.x & 63;
#ifdef RANDT_TIMING
  if (threadIdx.x == 64 * WIN_FACTOR_WAVE) atomicAdd((unsigned long long*)&g_randt_win_timing[15], (unsigned long long)(-wall_clock64()));
#endif
  if (lane < W.S) {
    const int f = lane;  // factor between states f and f+1
    double r[8];
    if (W.vec) motion_factor_vec(sh.xs[buf][f], sh.xs[buf][f + 1], W.raw_dt[f + 1], r, sh.Ju[buf][f]);
    else motion_factor(sh.xs[buf][f], sh.xs[buf][f + 1], W.raw_dt[f + 1], r, sh.Ju[buf][f]);
#pragma unroll
    for (int i = 0; i < 8; ++i) sh.ru[buf][f][i] = r[i];
    double c = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) c += 0.5 * sh.d2[i] * (r[i] * r[i]);
    if (W.use_imu) {
      double r2[2];
      if (W.vec) imu_factor_vec(sh.xs[buf][f], sh.xs[buf][f + 1], W.raw_dt[f + 1], W.imu[f], W.w_imu, W.w_bias, r2, sh.J2[buf][f]);
      else imu_factor(sh.xs[buf][f], sh.xs[buf][f + 1], W.raw_dt[f + 1], W.imu[f], W.w_imu, W.w_bias, r2, sh.J2[buf][f]);
      sh.r2[buf][f][0] = r2[0];
      sh.r2[buf][f][1] = r2[1];
      c += 0.5 * r2[0] * r2[0] + 0.5 * r2[1] * r2[1];
    }
    sh.fc[buf][f] = c;
  }
}

// What a wavefront needs to stream its share of ONE NDT term, fetched once per kernel (the window descriptor lives in
// device memory: re-reading term -> map -> cell count through three dependent global loads cost every pass ~1.5 us).
struct TermShare {
  const float4* mov;
  const float4* fix;
  const int32_t* pc;
  int state, n_slots, first, stride, k, active;
  unsigned kmagic;
  int wpt, share;
  int ci0[4];  // correspondences of this lane's first three trips (the association is frozen for the whole solve)
};
#ifndef WIN_TRIP_GROUP
#define WIN_TRIP_GROUP 3  // trips whose record reads are in flight together (all three staged ones: +0.7 % over 1, measured with the records in LDS)
#endif

// Base sum i of state jj from the per-wavefront partial sums of a pass (wavefront order: fixed association).
__device__ __forceinline__ double state_sum(const Shared& sh, const double* r, int jj, int i) {
  double a = 0.0;
  const int w0 = sh.sw_first[jj], w1 = w0 + sh.sw_cnt[jj];
  for (int w = w0; w < w1; ++w) a += r[w * 10 + i];
  return a;
}

// NDT pass over every term at the states in xs[buf].  MODE 0: max raw residual -> out[0];
// MODE 1: ten base sums per wavefront -> rsum[w * 10 ..] (state_sum() combines them per state).
// In MODE 1 wavefront 6 evaluates the motion / IMU factors of the same point while wavefronts 0-5
// stream the NDT slots (the factors are a ~2000-instruction serial chain: hidden behind the pass).
template <int D, int MODE, bool AM2, bool ANALYTIC>
__device__ bool ndt_pass(const MapView& fixed, const MapView& moving, const WinDesc& W, const TermShare& T,
                         const Shared& sh, int buf, const Loss& Lsh, double* out, int& parity, Shared& shw, const double*& rsum,
                         int step_from = -1) {
  const Loss L = Lsh;  // LDS -> registers for the duration of the pass
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool factor_wave = MODE == 1 && wave == WIN_FACTOR_WAVE;
#ifdef RANDT_TIMING
  const long long wt_t0 = wall_clock64();
#endif
  if (factor_wave) factors_unweighted(W, shw, buf);
  // candidate evaluation: the idle last wavefront takes ||x_candidate - x||^2 (parameter-tolerance test) off wavefront 0
  if (MODE == 1 && step_from >= 0 && wave == WIN_SPARE_WAVE) {
    const double sn2 = ambient_sq(W, sh, step_from, buf, lane);
    if (lane == 0) shw.scal[1] = sn2;
  }
#ifdef RANDT_TIMING
  if (factor_wave && lane == 0) atomicAdd((unsigned long long*)&g_randt_win_timing[13], (unsigned long long)(wall_clock64() - wt_t0));
#endif
  // Six NDT wavefronts share the <= 6 terms: with 1 / 2 / 3 terms every term is split over 6 / 3 / 2 wavefronts (a
  // wavefront takes every wpt-th 64-slot trip of its term), otherwise one wavefront per term.  Each wavefront reduces its
  // own ten base sums; the per-state combine below adds the parts in wavefront order.
  double* r = shw.red[parity][0];  // [8 wavefronts][10] | [96 + wave] bad | [104 + wave] max
  rsum = r;
  parity ^= 1;
  double mx = -DBL_MAX;
  int bad = 0;
  if (T.active) {
    {
      const double* xp = sh.xs[buf][T.state];
      const double inv = fast_rsqrt(xp[0] * xp[0] + xp[1] * xp[1]);
      const double c = xp[0] * inv, s = xp[1] * inv, tx = xp[2], ty = xp[3];
      const Rot rot = make_rot(c, s);
      const float4* mov = T.mov;
      const float4* fix = T.fix;
      const int32_t* pc = T.pc;
      const int n_slots = T.n_slots;
      const unsigned kmagic = T.kmagic;
      double a10[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) a10[i] = 0.0;
      // A lone workgroup has nothing to hide memory latency behind: a trip used to be index load -> record loads ->
      // arithmetic, two exposed L2 round trips each.  The correspondences of a lane's first three trips are kept in
      // registers for the whole solve (the association is frozen); WIN_TRIP_GROUP > 1 additionally puts the record loads of
      // several trips in flight together (measured: the registers that costs lose more elsewhere -- the NDT wavefronts,
      // ~2.1 us per pass, are the pass's critical path, the factor wavefront needs ~1.7 us; software-pipelining the
      // next trip's records behind the current residual spilled and lost 2 %).
      for (int s0 = T.first, grp = 0; s0 < n_slots; s0 += WIN_TRIP_GROUP * T.stride, ++grp) {
        int ci[WIN_TRIP_GROUP];
        bool val[WIN_TRIP_GROUP];
#pragma unroll
        for (int t = 0; t < WIN_TRIP_GROUP; ++t) {
          const int slot = s0 + t * T.stride + lane;
          const int tix = grp * WIN_TRIP_GROUP + t;  // trip number of this wavefront
          ci[t] = tix < 3 ? (tix == 0 ? T.ci0[0] : (tix == 1 ? T.ci0[1] : T.ci0[2])) : (slot < n_slots ? pc[slot] : -1);
          val[t] = ci[t] >= 0 && ci[t] < fixed.cap;
        }
        float4 mrec[WIN_TRIP_GROUP][3], frec[WIN_TRIP_GROUP][3];
#pragma unroll
        for (int t = 0; t < WIN_TRIP_GROUP; ++t) {
          const int slot = s0 + t * T.stride + lane;
          const unsigned mi = T.k == 1 ? (unsigned)slot : __umulhi((unsigned)slot, kmagic);  // slot / k
          const int tix2 = grp * WIN_TRIP_GROUP + t;
          if (tix2 < 3) {  // staged in LDS at the start of the solve (uniform branch)
            const float4 (*rc)[64] = sh.recs[T.share][tix2];
            mrec[t][0] = rc[0][lane];
            mrec[t][1] = rc[1][lane];
            frec[t][0] = rc[2][lane];
            frec[t][1] = rc[3][lane];
            const float4 tail = rc[4][lane];
            mrec[t][2] = make_float4(tail.x, 0.f, 0.f, 0.f);
            frec[t][2] = make_float4(tail.y, 0.f, 0.f, 0.f);
          } else {
            const float4* mv = mov + (size_t)(val[t] ? mi : 0u) * 3;
            const float4* fv = fix + (size_t)(val[t] ? ci[t] : 0) * 3;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              mrec[t][q] = mv[q];
              frec[t][q] = fv[q];
            }
          }
        }
#pragma unroll
        for (int t = 0; t < WIN_TRIP_GROUP; ++t) {
          if (!val[t]) continue;
          double jb[3];
          const double sq = residual_sq<D, MODE == 1, ANALYTIC>(mrec[t], frec[t], rot, tx, ty, jb);
          // closed-form loss: a non-finite residual makes the cost sum non-finite (u or 1 / (u^2 s) is NaN / 0 x inf), which the
          // caller tests after the reduction -- no per-residual class test in the hot loop
          if (!(MODE == 1 && AM2) && !isfinite(sq)) bad = 1;
          if (MODE == 0) mx = sq > mx ? sq : mx;
          else accumulate_residual<AM2>(L, sq, jb, a10);
        }
      }
      if (MODE == 1) {
        wave_sum10(a10);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < 10; ++i) r[T.share * 10 + i] = a10[i];
        }
      }
    }
  }
#ifdef RANDT_TIMING
  if (MODE == 1 && wave == 0 && lane == 0) atomicAdd((unsigned long long*)&g_randt_win_timing[14], (unsigned long long)(wall_clock64() - wt_t0));
#endif
  double badf = wave_any(bad != 0);
  if (MODE == 0) mx = wave_max(mx);
  if (lane == 0) {
    r[96 + wave] = badf;
    r[104 + wave] = mx;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < WIN_WAVES; ++w) badf = r[96 + w] > badf ? r[96 + w] : badf;
  if (MODE == 0) {
#pragma unroll
    for (int w = 0; w < WIN_WAVES; ++w) mx = r[104 + w] > mx ? r[104 + w] : mx;
    if (threadIdx.x == 0) out[0] = mx > 0.0 ? sqrt(mx) : 0.0;
    __syncthreads();
    return uni(badf == 0.0);
  }
  return uni(badf == 0.0);
}

// Weighting half of the factor evaluation (the unweighted residuals / Jacobians were produced by
// factors_unweighted on the factor wavefront during the NDT pass, whose barrier published them):
// residuals_map.applyOnTheLeft(sqrtI_).  Returns sum of 1/2 r^2 over the factor residuals (identical in every thread).
// Diagonal square-root information (every shipped configuration): nothing is materialised -- the assembly applies
// d_i^2 itself -- and there is no barrier; otherwise one thread per entry forms sqrtI * J, one barrier.
__device__ double factors_weight(const WinDesc& W, Shared& sh, int buf) {
  const int tid = threadIdx.x;
  double cost = 0.0;
  if (sh.sq_diag) {
    for (int f = 0; f < W.S; ++f) cost += sh.fc[buf][f];
    return cost;
  }
  for (int e = tid; e < W.S * 128; e += WIN_BLOCK) {
    const int f = e >> 7, i = (e & 127) >> 4, c = e & 15;
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += W.sqrtI[i * 8 + k] * sh.Ju[buf][f][k * 16 + c];
    sh.Jf[buf][f][i * 16 + c] = a;
  }
  if (tid < W.S * 8) {
    const int f = tid >> 3, i = tid & 7;
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += W.sqrtI[i * 8 + k] * sh.ru[buf][f][k];
    sh.rf[buf][f][i] = a;
  }
  __syncthreads();
  for (int f = 0; f < W.S; ++f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) cost += 0.5 * sh.rf[buf][f][i] * sh.rf[buf][f][i];
    if (W.use_imu) cost += 0.5 * sh.r2[buf][f][0] * sh.r2[buf][f][0] + 0.5 * sh.r2[buf][f][1] * sh.r2[buf][f][1];
  }
  return cost;
}

// J^T J and J^T r at xs[buf] from the factor blocks in LDS and the per-state NDT base sums.
// have_sigma: the Jacobi scaling of this solve is known (every assembly but the first of a solve): the scaled copies
// Hs / gs are written in the same pass instead of by wavefront 0 afterwards.
__device__ void assemble(const WinDesc& W, Shared& sh, int buf, const double* rsum /* per-wavefront base sums of the pass */,
                         bool have_sigma, double dmin, double dmax) {
  const int n = W.n_tan, tid = threadIdx.x;
  const bool dg = sh.sq_diag != 0;
  const double(*J)[128] = dg ? sh.Ju[buf] : sh.Jf[buf];
  const double(*rr)[8] = dg ? sh.ru[buf] : sh.rf[buf];
  // upper triangle only (n (n + 1) / 2 <= 528 entries: one round), mirrored on the way out
  for (int e = tid; e < n * (n + 1) / 2; e += WIN_BLOCK) {
    const int ab = sh.tri[e], a = ab & 255, b = ab >> 8;
    double h = 0.0;
    for (int f = 0; f < W.S; ++f) {
      const int la = sh.lcol[f][a], lb = sh.lcol[f][b];
      if (la >= 0 && lb >= 0) {
        if (dg) {
#pragma unroll
          for (int i = 0; i < 8; ++i) h += sh.d2[i] * (J[f][i * 16 + la] * J[f][i * 16 + lb]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) h += J[f][i * 16 + la] * J[f][i * 16 + lb];
        }
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
      for (int i = 4; i < 10; ++i) B[i] = state_sum(sh, rsum, j, i);
      const double G[3][3] = {{B[4], B[5], B[6]}, {B[5], B[7], B[8]}, {B[6], B[8], B[9]}};
      const int ia = a - sh.off_tan[j][0], ib = b - sh.off_tan[j][0];
      double v = 0.0;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) v += T[ia][p] * G[p][q] * T[ib][q];
      h += v;
    }
    sh.H[a * n + b] = h;
    sh.H[b * n + a] = h;
    if (have_sigma) {
      const double hs = h * sh.sigma[a] * sh.sigma[b];
      sh.Hs[a * n + b] = hs;
      sh.Hs[b * n + a] = hs;
      if (a == b) sh.diag[a] = fmin(fmax(hs, dmin), dmax);  // LM diagonal of the new point
    }
  }
  if (tid >= WIN_BLOCK - n) {  // the gradient on the block's last threads (idle above unless n = 32)
    const int a = tid - (WIN_BLOCK - n);
    double g = 0.0;
    for (int f = 0; f < W.S; ++f) {
      const int la = sh.lcol[f][a];
      if (la >= 0) {
        if (dg) {
#pragma unroll
          for (int i = 0; i < 8; ++i) g += sh.d2[i] * (J[f][i * 16 + la] * rr[f][i]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) g += J[f][i * 16 + la] * rr[f][i];
        }
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
      g += T[ia][0] * state_sum(sh, rsum, j, 1) + T[ia][1] * state_sum(sh, rsum, j, 2) + T[ia][2] * state_sum(sh, rsum, j, 3);
    }
    sh.g[a] = g;
    if (have_sigma) sh.gs[a] = g * sh.sigma[a];
  }
  __syncthreads();
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

// The damped solve (H_s + D / radius) y = g_s by the banded block Gauss-Jordan (band_pivot above), one wavefront per LEVEL:
// level 0 is the radius of the running iteration, level q the radius q rejections later (radius / decrease, decrease doubling).
// A rejected -- or invalid -- step changes nothing but the radius, and with Ceres' default initial radius of 1e4 more than half
// of all iterations of these windows are rejections (the first five of every GNC step): wavefronts 0..6 are idle during the
// solve anyway, so the next radii cost (almost) no time and the following rejected iterations start from a finished step.  Own scratch per level; results in step[q] / delta[q] / solved[q].
__device__ __forceinline__ void band_solve(Shared& sh, int n, int S, int lane, double inv_radius, int Q) {
  if (lane < n) sh.dd[Q][lane] = sh.Hs[lane * n + lane] + sh.diag[lane] * inv_radius;
  wave_fence();
  double okf = 1.0;
  {
            // ---- banded block Gauss-Jordan with DPP broadcasts (see band_pivot above)
            const char* const sbase = reinterpret_cast<const char*>(&sh);
            int boffv[18];
#pragma unroll
            for (int k = 0; k < 18; ++k) boffv[k] = sh.boff[k][lane];
            {
              const int dshift = Q * (int)sizeof(sh.dd[0]) - 1;  // flagged entries: this level's damped diagonal
#pragma unroll
              for (int k = 0; k < 16; ++k) boffv[k] += (boffv[k] & 1) ? dshift : 0;
            }
            double C[16], B;
#pragma unroll
            for (int k = 0; k < 16; ++k) C[k] = *reinterpret_cast<const double*>(sbase + boffv[k]);
            B = *reinterpret_cast<const double*>(sbase + boffv[16]);
            const int trow = boffv[17];
            const int l = lane & 15, br = lane >> 4;
            double RD = 0.0, X = 0.0;

#pragma nounroll
            for (int b = 0; b <= S; ++b) {
              const int bsz = __builtin_amdgcn_readfirstlane(sh.bsize[b]);
              if (br == b) {
                if (b > 0 && l < 8) {  // rows of state b as block step b - 1 left them
#pragma unroll
                  for (int k = 0; k < 8; ++k) C[k] = sh.xfer[Q][l][k];
                  B = sh.xfer[Q][l][8];
                }
                asm volatile("s_nop 4");
                if (b < S) {
                  band_block<16>(C, B, RD, l, bsz);
                  if (l >= 8) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) sh.xfer[Q][l - 8][k] = C[8 + k];
                    sh.xfer[Q][l - 8][8] = B;
                  }
                } else {
                  band_block<8>(C, B, RD, l, bsz);  // last state: nothing behind it
                }
              }
              wave_fence();
            }
            // rows of state b: diagonal in their own columns now; x_b = (B - U x_{b+1}) / d, last state first.  The next
            // state's solution sits in lanes 0..7 of the next DPP row: eight uniform values, fetched with v_readlane
            // (no LDS round trip on this dependent chain).
#pragma nounroll
            for (int b = S; b >= 0; --b) {
              double a0 = B, a1 = 0.0;
              if (b < S) {
                const int src = 16 * (b + 1);
#pragma unroll
                for (int m = 0; m < 8; m += 2) {
                  a0 = fma(-C[8 + m], readlane_f64(X, src + m), a0);
                  a1 = fma(-C[9 + m], readlane_f64(X, src + m + 1), a1);
                }
              }
              const double xb = (a0 + a1) * RD;
              X = br == b ? xb : X;
            }
            const bool real = trow >= 0;
            const double st = -X;
            okf = wave_any(real && !(RD > 0.0)) != 0.0 ? 0.0 : 1.0;
            const double fin = 1.0 - wave_any(real && !isfinite(st));
            if (real) {
              sh.step[Q][trow] = st;
              sh.delta[Q][trow] = st * sh.sigma[trow];
            }
            if (lane == 0) sh.solved[Q] = (okf != 0.0 && fin != 0.0) ? 1.0 : 0.0;
  }
}

// Dense fallback of the damped solve (windows whose state blocks exceed 8 tangent dimensions, i.e. with the IMU bias):
// Gauss-Jordan elimination entirely in registers.  Lane i keeps row i of A and gs_i; after every step all rows are shifted
// left by one column, so the pivot column is ALWAYS register 0 and no register is indexed dynamically: step j broadcasts
// lane j's row with v_readlane (the lane select may be a loop variable), every other lane subtracts its multiple of it, and
// lane j parks its pivot.  Eliminating above the pivot as well leaves a diagonal system -- no back substitution, no LDS round
// trips, no barriers; the j loop stays rolled.  Kept out of line: its 32 row registers must not set the kernel's budget.
__device__ __noinline__ void gj_dense_solve(Shared& sh, int n, double inv_radius, int lane) {
  double okf = 1.0;
  {
    // row i of the damped matrix straight into registers: Hs is symmetric, so lane i reads COLUMN i
    // (consecutive lanes -> consecutive words, no bank conflicts); (sqrt(D^2 / radius))^2 on the diagonal
    double row[WIN_NMAX], b = lane < n ? sh.gs[lane] : 0.0, dg = 1.0;
    const double damp = lane < n ? sh.diag[lane] * inv_radius : 0.0;
#pragma clang loop unroll(full)
    for (int k = 0; k < WIN_NMAX; ++k) {
      row[k] = (lane < n && k < n) ? sh.Hs[k * n + lane] : 0.0;
      if (k == lane) row[k] += damp;
    }
    // at step j only the first n - j columns are still live: eight rolled loops of width 32, 28, ... 4
    int j = 0;
#define RANDT_GJ_STEPS(WIDTH)                                                                          \
  for (; j < n && n - j > (WIDTH) - 4; ++j) {                                                         \
    const double pj = readlane_f64(row[0], j);                                                        \
    if (!(pj > 0.0)) okf = 0.0;                                                                       \
    if (lane == j) dg = pj;                                                                           \
    const double f = lane != j ? row[0] * fast_rcp(pj) : 0.0; /* lane j: f = 0, its row only shifts */ \
    b = fma(-f, readlane_f64(b, j), b);                                                               \
    _Pragma("clang loop unroll(full)") for (int k = 1; k < (WIDTH); ++k)                              \
        row[k - 1] = fma(-f, readlane_f64(row[k], j), row[k]);                                        \
    row[(WIDTH) - 1] = 0.0;                                                                           \
  }
    RANDT_GJ_STEPS(32)
    RANDT_GJ_STEPS(28)
    RANDT_GJ_STEPS(24)
    RANDT_GJ_STEPS(20)
    RANDT_GJ_STEPS(16)
    RANDT_GJ_STEPS(12)
    RANDT_GJ_STEPS(8)
    RANDT_GJ_STEPS(4)
#undef RANDT_GJ_STEPS
    // scaled step (negated: Ceres solves for -step), its finiteness, the unscaled delta
    const double st = lane < n ? -(b * fast_rcp(dg)) : 0.0;
    const double fin = 1.0 - wave_any(!isfinite(st));
    if (lane < n) {
      sh.step[0][lane] = st;
      sh.delta[0][lane] = st * sh.sigma[lane];
    }
    if (lane == 0) sh.solved[0] = (okf != 0.0 && fin != 0.0) ? 1.0 : 0.0;
  }
}

// AM2: Barron shape exactly -2 (the shipped configurations): closed-form loss, no pow() in the kernel
template <int D, bool AM2, bool ANALYTIC>
__global__ __launch_bounds__(WIN_BLOCK) void k_solve_window(MapView fixed, MapView moving, const WinDesc* __restrict__ Wp,
                                                            const int32_t* __restrict__ corr, SolveParams P,
                                                            double* __restrict__ states, randt_result* __restrict__ result,
                                                            double* trace, int trace_len, int corr_stride, int state_stride) {
  __shared__ Shared sh;
  // one workgroup per window of a batch (randt_register_window_batch): descriptors, correspondence tables, states, results and
  // traces of window w lie w strides behind the first window's (a single window: blockIdx.x = 0)
  Wp += blockIdx.x;
  corr += (size_t)blockIdx.x * corr_stride;
  states += (size_t)blockIdx.x * state_stride;
  result += blockIdx.x;
  if (trace) trace += (size_t)blockIdx.x * trace_len;
  // dynamically indexed (state j, term t): read from device memory on demand instead of pinning ~200 SGPRs
  const WinDesc& W = *Wp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = W.n_tan, S = W.S;
  // entry e = lane + 64 t of an n x n matrix is (ent_i0 + t ent_di + carries, ...): one division per kernel
  const int nn = n > 0 ? n : 1;
  const int ent_i0 = lane / nn, ent_j0 = lane % nn, ent_di = 64 / nn, ent_dj = 64 % nn;
  int parity = 0;

  // ---- load states, build column maps
  for (int e = tid; e < (S + 1) * ST_STRIDE; e += WIN_BLOCK) {
    sh.xs[0][e / ST_STRIDE][e % ST_STRIDE] = states[e];
    sh.xs[1][e / ST_STRIDE][e % ST_STRIDE] = states[e];
  }
  for (int e = tid; e < WIN_SMAX * WIN_NMAX; e += WIN_BLOCK) {
    sh.lcol[e / WIN_NMAX][e % WIN_NMAX] = -1;
    sh.lcol2[e / WIN_NMAX][e % WIN_NMAX] = -1;
  }
  if (tid < WIN_NMAX) sh.pose_of[tid] = -1;
  if (tid < RANDT_WIN_MAX_STATES * 5) {
    sh.off_tan[tid / 5][tid % 5] = W.off_tan[tid / 5][tid % 5];
    sh.off_amb[tid / 5][tid % 5] = W.off_amb[tid / 5][tid % 5];
  }
  __syncthreads();
  if (tid == 0) {
    const int sz[4] = {3, 2, 1, 2}, lbase[4] = {0, 3, 5, 6};
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
    for (int j = 1; j <= S; ++j)
      if (W.off_tan[j][0] >= 0)
        for (int e = 0; e < 3; ++e) sh.pose_of[W.off_tan[j][0] + e] = j;
  }
  if (tid < 64) {
    // band layout of the block solve: DPP row br = block step br, lane l of it = row l of the band [state br, state br + 1]
    auto bstart = [&](int j) { return j > S ? n : (W.off_tan[j][0] >= 0 ? W.off_tan[j][0] : W.off_tan[j][1]); };
    auto tix = [&](int blk, int m) {
      if (blk > S) return -1;
      const int st = bstart(blk);
      return m < bstart(blk + 1) - st ? st + m : -1;
    };
    auto off_of = [&](const double* q) { return (int)(reinterpret_cast<const char*>(q) - reinterpret_cast<const char*>(&sh)); };
    const int br = tid >> 4, l = tid & 15;
    const int rb = br + (l >> 3), rm = l & 7, trow = tix(rb, rm);
    for (int k = 0; k < 16; ++k) {
      const int cb = br + (k >> 3), cm = k & 7, tc = tix(cb, cm);
      int off;
      if (trow >= 0 && tc >= 0) off = trow == tc ? (off_of(&sh.dd[0][trow]) | 1) : off_of(&sh.Hs[tc * n + trow]);
      else off = off_of(&sh.cst[(rb == cb && rm == cm) ? 1 : 0]);
      sh.boff[k][tid] = off;
    }
    sh.boff[16][tid] = trow >= 0 ? off_of(&sh.gs[trow]) : off_of(&sh.cst[0]);
    sh.boff[17][tid] = (l < 8 && br <= S) ? trow : -1;
    if (tid == 0) {
      int okb = 1;
      for (int j = 0; j <= S; ++j) {
        okb &= (bstart(j + 1) - bstart(j) <= 8) ? 1 : 0;
        sh.bsize[j] = bstart(j + 1) - bstart(j);
      }
#ifdef RANDT_WIN_NO_BAND
      okb = 0;
#endif
      sh.band_ok = okb;
      sh.cst[0] = 0.0;
      sh.cst[1] = 1.0;
      int dgl = 1;
      for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 8; ++k)
          if (i != k && W.sqrtI[i * 8 + k] != 0.0) dgl = 0;
#ifdef RANDT_WIN_NO_DIAG
      dgl = 0;
#endif
      sh.sq_diag = dgl;
      for (int i = 0; i < 8; ++i) sh.d2[i] = W.sqrtI[i * 8 + i] * W.sqrtI[i * 8 + i];
    }
  }
  for (int e = tid; e < 2 * WIN_SMAX * 128; e += WIN_BLOCK) (&sh.Ju[0][0][0])[e] = 0.0;  // once: see Shared::Ju
  for (int e = tid; e < n * (n + 1) / 2; e += WIN_BLOCK) {
    // row a of the upper triangle starts at a n - a (a - 1) / 2
    int a = 0;
    while (a + 1 < n && (a + 1) * n - (a + 1) * a / 2 <= e) ++a;
    sh.tri[e] = (unsigned short)(a | ((a + (e - (a * n - a * (a - 1) / 2))) << 8));
  }
  __syncthreads();

  // this wavefront's share of the NDT terms (fixed for the whole solve)
  TermShare T;
  {
    T.wpt = W.n_terms <= 3 && W.n_terms > 0 ? WIN_NDT_WAVES / W.n_terms : 1;
    T.share = ndt_share_of_wave(wave);
    const int t = T.share >= 0 ? T.share / T.wpt : -1;
    T.active = (t >= 0 && t < W.n_terms) ? 1 : 0;
    T.k = W.k;
    T.kmagic = W.k > 1 ? (unsigned)((0x100000000ull + (unsigned)W.k - 1) / (unsigned)W.k) : 0u;
    T.first = 64 * (T.share >= 0 ? T.share % T.wpt : 0);
    T.stride = 64 * T.wpt;
    T.state = 0;
    T.n_slots = 0;
    T.mov = T.fix = nullptr;
    T.pc = nullptr;
    if (T.active) {
      const int mmap = W.term_moving[t], fmap = W.term_fixed[t];
      int M = moving.counts[mmap];
      M = M > moving.cap ? moving.cap : M;
      T.state = W.term_state[t];
      T.n_slots = M * W.k;
      T.mov = reinterpret_cast<const float4*>(moving.cells + (size_t)mmap * moving.cap);
      T.fix = reinterpret_cast<const float4*>(fixed.cells + (size_t)fmap * fixed.cap);
      T.pc = corr + (size_t)t * moving.cap * W.k;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int slot = T.first + q * T.stride + lane;
      T.ci0[q] = (T.active && slot < T.n_slots) ? T.pc[slot] : -1;
      if (T.share >= 0 && T.ci0[q] >= 0 && T.ci0[q] < fixed.cap) {
        const unsigned mi = T.k == 1 ? (unsigned)slot : __umulhi((unsigned)slot, T.kmagic);  // slot / k
        const float4* mv = T.mov + (size_t)mi * 3;
        const float4* fv = T.fix + (size_t)T.ci0[q] * 3;
        const float4 m0 = mv[0], m1 = mv[1], m2 = mv[2], f0 = fv[0], f1 = fv[1], f2 = fv[2];
        sh.recs[T.share][q][0][lane] = m0;
        sh.recs[T.share][q][1][lane] = m1;
        sh.recs[T.share][q][2][lane] = f0;
        sh.recs[T.share][q][3][lane] = f1;
        sh.recs[T.share][q][4][lane] = make_float4(m2.x, f2.x, 0.f, 0.f);
      }
    }
    if (lane == 0 && T.share >= 0) sh.wave_state[T.share] = T.active ? T.state : -1;
  }
  __syncthreads();
  if (tid <= WIN_SMAX) {
    int first = 0, cnt = 0;
    for (int w = WIN_NDT_WAVES - 1; w >= 0; --w)
      if (sh.wave_state[w] == tid && tid >= 1) {
        first = w;
        ++cnt;
      }
    sh.sw_first[tid] = first;
    sh.sw_cnt[tid] = cnt;
  }
  __syncthreads();
  const bool band_ok = __builtin_amdgcn_readfirstlane(sh.band_ok) != 0;

  // number of NDT residual blocks and of moving cells
  int n_res = 0;
  for (int t = 0; t < W.n_terms; ++t) {
    int M = moving.counts[W.term_moving[t]];
    M = M > moving.cap ? moving.cap : M;
    const int32_t* pc = corr + (size_t)t * moving.cap * W.k;
    for (int s = tid; s < M * W.k; s += WIN_BLOCK) {
      const int ci = pc[s];
      n_res += (ci >= 0 && ci < fixed.cap) ? 1 : 0;
    }
  }
  {
    double v = wave_sum((double)n_res);
    if (lane == 0) sh.red[0][wave][33] = v;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < WIN_WAVES; ++w) tot += sh.red[0][w][33];
    n_res = (int)tot;
    __syncthreads();
  }
  int n_cells = 0;  // sum over optimised states (ndt_matcher.cpp:367)
  {
    int last = -1;
    for (int t = 0; t < W.n_terms; ++t)
      if (W.term_state[t] != last) {
        int M = moving.counts[W.term_moving[t]];
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
  double raw_max = 0.0;
  bool ok = true;
  if (n_res > 0) {
    const double* unused;
    ok = ndt_pass<D, 0, AM2, ANALYTIC>(fixed, moving, W, T, sh, 0, sh.loss, &sh.scal[7], parity, sh, unused);
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
  WT_DECL
  WT(0);

  if (ok) {
    do {
      gnc_mu = fmax(gnc_mu, 1.0);
      __syncthreads();  // every pass of the previous step has read its loss
      if (tid == 0) sh.loss = AM2 ? make_loss_am2(P.loss_a, gnc_mu, weight) : make_loss(P.loss_a, P.alpha, gnc_mu, weight);
      __syncthreads();
      // ================= one ceres::Solve =================
      double radius = P.r0, decrease = 2.0;
      bool step_ok = true;
      int lvl = 0, n_lvl = 0;  // damped solves in stock: level lvl of n_lvl is the running radius
      int num_invalid = 0, iteration = 0;
      double minimum_cost = DBL_MAX;
      const double* rs_cur;  // per-wavefront NDT base sums at the current point
      bool e_ok = ndt_pass<D, 1, AM2, ANALYTIC>(fixed, moving, W, T, sh, p, sh.loss, nullptr, parity, sh, rs_cur);
      WT(1);
      double fcost = factors_weight(W, sh, p);
      WT(2);
      res.n_evals++;
      res.iterations++;
      double cost = fcost;
#pragma unroll
      for (int j = 0; j < WIN_SMAX; ++j) cost += state_sum(sh, rs_cur, j + 1, 0);
      if (uni(!e_ok || !isfinite(cost))) {
        term = RANDT_TERM_FAILURE;
        res.status = 2;
        res.gnc_solves++;
        break;
      }
      if (res.gnc_solves == 0) res.initial_cost = cost;
      summary_min = cost;
      assemble(W, sh, p, rs_cur, false, P.dmin, P.dmax);
      WT(3);
      bool first = true, need_scale = true;
      double x_norm = 0.0;
      trace_push(tr, trace_len, cost, radius, 0);

      for (;;) {
        // ---- Two wavefronts share the serial part of an iteration.  Wavefront 0: Jacobi scaling (first assembly of a solve
        // only; later assemblies write Hs / gs themselves), the damped solve, Plus and the step norm.  Wavefront 1, at the
        // same time: the gradient test and ||x|| of a freshly accepted point, then -- once the step is published --
        // the model cost change.  The gradient / radius stopping tests are therefore evaluated AFTER the solve has been
        // started (their operands arrive with the barrier that publishes the step); a stop discards that work and takes
        // back the iteration count, so the decision sequence is the reference's (max iterations, gradient, radius).
        if (need_scale && wave == 1) {
          // gradient tolerance: ||x - Plus(x, -g)||_inf <= gtol
          double gm = lane < n ? fabs(sh.g[lane]) : 0.0;
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
        if (need_scale && first && wave == 0) {
          if (lane < n) sh.sigma[lane] = 1.0 / (1.0 + sqrt(sh.H[lane * n + lane]));
          wave_fence();
          // all 64 lanes over the n^2 entries; (row, column) of entry e advance incrementally (no division)
          int ei = ent_i0, ej = ent_j0;
          for (int e = lane; e < n * n; e += 64) {
            sh.Hs[e] = sh.H[e] * sh.sigma[ei] * sh.sigma[ej];
            ei += ent_di;
            ej += ent_dj;
            if (ej >= n) {
              ej -= n;
              ++ei;
            }
          }
          if (lane < n) sh.gs[lane] = sh.g[lane] * sh.sigma[lane];
          wave_fence();
          // LM diagonal: a function of the scaled J^T J alone (Ceres keeps it across rejected steps -- the same values)
          if (lane < n) sh.diag[lane] = fmin(fmax(sh.Hs[lane * n + lane], P.dmin), P.dmax);
          wave_fence();
        }
        if (need_scale && first) __syncthreads();  // first assembly of a solve: wavefront 0 scaled it; wavefronts 1..3 solve from it too
        if (need_scale) {
          need_scale = false;
          first = false;
          WT(4);
        }
        // ---- FinalizeIterationAndCheckIfMinimizerCanContinue
        // (Ceres copies x to the user parameters after successful steps that lower the minimum cost;
        //  accepted steps are monotone here, so the current buffer p always is that point.)
        if (step_ok && uni(cost < minimum_cost)) minimum_cost = cost;
        if (iteration >= P.max_it) { term = RANDT_TERM_NO_CONVERGENCE; break; }
        ++iteration;
        res.iterations++;

        // ---- LevenbergMarquardtStrategy::ComputeStep: damped normal equations -- for this radius and, at no extra time, for
        // the radii the next rejections lead to (band_solve).  Skipped while such a level is still in stock.
        if (lvl >= n_lvl) {
          lvl = 0;
          // Several wavefronts solving at once run measurably slower than one alone (config 3: always all levels 2.5 % worse
          // than this policy), so the stock is only laid in where rejections come in chains: at the first iteration of a solve
          // (Ceres' initial radius 1e4 is shrunk five times before the first step of every GNC step is accepted) and after a
          // rejection; behind an accepted step the next one is usually accepted too and the wavefront solves alone.
          n_lvl = (band_ok && (iteration <= 1 || !step_ok)) ? WIN_LEVELS : 1;
#ifdef RANDT_WIN_NO_SPEC
          n_lvl = 1;
#endif
          if (wave < n_lvl) {
            double rr = radius, dc = decrease;  // the reference's update, replayed: radius /= decrease; decrease *= 2
            for (int q = 0; q < wave; ++q) {
              rr = rr / dc;
              dc *= 2.0;
            }
            const double inv_radius = fast_rcp(rr);
            WT(8);
#ifndef RANDT_WIN_REPEAT_SOLVE
#define RANDT_WIN_REPEAT_SOLVE 1  // > 1: cost probe (the solve is idempotent: same inputs, same outputs)
#endif
#pragma nounroll
            for (int rpt = 0; rpt < RANDT_WIN_REPEAT_SOLVE; ++rpt)
            if (band_ok) {
              band_solve(sh, n, S, lane, inv_radius, wave);
            } else {
              gj_dense_solve(sh, n, inv_radius, lane);
            }
            WT(9);
          }
        }
        const int cs = lvl;
        __syncthreads();  // publishes: step / delta / scal[3] (wave 0), gradient test and ||x|| (wave 1), Hs / gs of a first assembly
        x_norm = sh.scal[2];
        if (step_ok && uni(sh.scal[4] != 0.0)) { res.iterations--; term = RANDT_TERM_CONVERGENCE_GRADIENT; break; }
        if (uni(radius <= P.rmin)) { res.iterations--; term = RANDT_TERM_CONVERGENCE_RADIUS; break; }
        if (wave == 1) {
          // model_cost_change = -(step.gs + step^T Hs step / 2)
          double t = 0.0;
          if (lane < n) {
            double hs = 0.0;
#pragma unroll 8
            for (int b = 0; b < n; ++b) hs += sh.Hs[b * n + lane] * sh.step[cs][b];  // symmetric: column read, conflict-free
            t = sh.step[cs][lane] * (sh.gs[lane] + 0.5 * hs);
          }
          const double mcc = -wave_sum(t);
          if (lane == 0) sh.scal[0] = mcc;
        } else if (wave == 0) {
          WT(10);
          plus_states(W, sh, p, 1 - p, sh.delta[cs], 1.0, lane);
          WT(11);
        }
        __syncthreads();
        WT(5);
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

        // ---- candidate: factors + NDT terms with Jacobians (speculative)
        const double* rs_cand;
        const bool c_ok = ndt_pass<D, 1, AM2, ANALYTIC>(fixed, moving, W, T, sh, 1 - p, sh.loss, nullptr, parity, sh, rs_cand, p);
        const double sn2 = sh.scal[1];
        WT(1);
        const double cf = factors_weight(W, sh, 1 - p);
        WT(2);
        res.n_evals++;
        double cand_cost = cf;
#pragma unroll
        for (int j = 0; j < WIN_SMAX; ++j) cand_cost += state_sum(sh, rs_cand, j + 1, 0);
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
          WT(6);
          assemble(W, sh, p, rs_cand, true, P.dmin, P.dmax);
          WT(3);
          need_scale = true;
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
  for (int e = tid; e < (S + 1) * ST_STRIDE; e += WIN_BLOCK) states[e] = sh.xs[p][e / ST_STRIDE][e % ST_STRIDE];
  res.termination = term;
  res.final_cost = summary_min;
  res.cost = n_res > 0 ? summary_min / (double)n_res : 0.0;
  WT(7);
  WT_FLUSH;
  if (tid == 0) result[0] = res;
}

}  // namespace

int launch_solve_window(randt_ctx* ctx, const MapView& fixed, const MapView& moving, const WinDesc& desc, const WinDesc* d_desc,
                        const int32_t* d_corr, const randt_matcher_params* mp, double* d_states, randt_result* d_result, int n_windows,
                        int corr_stride, int state_stride) {
  randt_note_enqueue(ctx);  // (RANDT_SOLVE_AUTO of the process's other contexts: this one has work in flight)
  if (n_windows <= 0) return RANDT_OK;
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
  // more than three optimised states (no shipped configuration): the general kernel
  if (desc.n_tan > WIN_NMAX || desc.S > WIN_SMAX || desc.n_terms > 6 || ctx->window_general)
    return launch_solve_window_gen(ctx, fixed, moving, desc, d_desc, d_corr, P, d_states, d_result, n_windows, corr_stride, state_stride);
#define RANDT_WIN_LAUNCH(DD, AA, NN)                                                                                       \
  hipLaunchKernelGGL((k_solve_window<DD, AA, NN>), dim3(n_windows), dim3(WIN_BLOCK), 0, ctx->stream, fixed, moving, d_desc, d_corr, P, \
                     d_states, d_result, ctx->d_trace, ctx->trace_len, corr_stride, state_stride)
  const bool am2 = P.alpha == -2.0;
  if (desc.pad_) {  // RANDT_PARAM_ANALYTIC: the reference's hand-written NDT functor (never set by a shipped configuration)
    if (desc.d3) {
      if (am2) RANDT_WIN_LAUNCH(3, true, true); else RANDT_WIN_LAUNCH(3, false, true);
    } else {
      if (am2) RANDT_WIN_LAUNCH(2, true, true); else RANDT_WIN_LAUNCH(2, false, true);
    }
  } else if (desc.d3) {
    if (am2) RANDT_WIN_LAUNCH(3, true, false); else RANDT_WIN_LAUNCH(3, false, false);
  } else {
    if (am2) RANDT_WIN_LAUNCH(2, true, false); else RANDT_WIN_LAUNCH(2, false, false);
  }
#undef RANDT_WIN_LAUNCH
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
