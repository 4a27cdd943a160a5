// Radar scan filter for gfx950 -- SURVEY row f-1, the HBM-streaming stage in front of the NDT build
// (compiled with -ffp-contract=off: the emitted points must equal the reference's bit for bit).
//
// Replaces RadarPreprocessor::filterScan (src/radar_preprocessing/radar_preprocessor.cpp:45-125):
// per azimuth the strongest return inside (min_range, max_range), then the run of monotonically
// decreasing intensity around it, thresholded by min_intensity, transformed sensor -> base.
//
//   k_filter_rows   one 256-thread WORKGROUP per azimuth row (round 5; rounds 3 / 4: history below).  Per row:
//                   1. the row of the raw polar scan (16 B / point, 19.2 MB for 400 x 3000) is read exactly once with
//                      coalesced dwordx4 loads, twelve per lane = the whole row in flight at once; the arg-max (strict '>',
//                      first index wins) stays in registers, is combined by shuffles and, over the four wavefronts, through
//                      LDS behind ONE barrier; the same pass checks that the cloud really is organised azimuth after azimuth
//                      (the reference detects azimuth changes with |atan2 - current| > 1e-4 while walking the cloud
//                      sequentially): a cross / dot test flags, a flagged row is walked again with the exact atan2f;
//                   2. the row's detection is expanded towards / away from the sensor by wavefront 0 while the
//                      row is cache-hot (expand_both below); the same loads count the run's points that pass the output
//                      thresholds and -- for runs of <= FILT_STAGE kept points that end within 32 bins on both sides:
//                      nearly all -- produce the output points themselves (the "stage"); a 32-byte row record.
//   k_filter_emit   one workgroup per scan: replays the reference's detection-list semantics (last azimuth never
//                   flushed, the very first boundary pushes index 0), block-scans the kept counts over the rows and
//                   emits the points and the per-azimuth peaks in the reference's order: staged rows are copied (record
//                   and stage arrive in one round trip), the others re-read their run, eight independent loads at a time.
// Measured (16 scans per launch, 307 MB, rocprofv3 averages; a bare read of the same bytes takes 44 .. 45 us when consecutive
// launches meet their lines in the 256 MB Infinity Cache and 48.4 .. 49.5 us per 307 MB of a 1.2 GB buffer, i.e. from HBM:
// tools/hbm_stream_probe.hip; 51.5 us as launches of 16 scans over distinct inputs):
//   rounds 3 / early 4  a PERSISTENT workgroup walking rows, barrier between arg-max and expansion: 58.2 + 7.6 (emission) us; with
//                       expansion and per-point tests compiled out still 53.6: it paid for the barrier, behind which four wavefronts
//                       have nothing in flight; next row in a second register set (191 registers, two per CU) 93;
//                       non-temporal row loads 100; 96 / 80 registers (spills) 84 / 104; emission behind a ticket per row 130;
//                       next row's loads issued before the barrier (+ a fifth wavefront for the expansion) 54 .. 69
//   round 4             wavefront per row 54.8 -> ring of loads 53.2 -> exact test out of the loop, scalar row base, one
//                       dwordx4 per point 51.4 over one input, 56.3 .. 56.6 over distinct inputs; emission 7.6 -> 5.6 us with the
//                       stage.  ONE scan per launch: 11.97 + 5.22 us = 0.14 of 8 TB/s (400 wavefronts, 47 dependent rounds each)
//   round 5 (this file) a workgroup per ROW (not persistent: three wavefronts retire at the barrier, the dispatcher refills):
//                       52.6 + 5.6 us over distinct inputs, ONE scan 7.9 + 4.4 us (profiles/experiments/r05_filter_workgroup_per_row.md;
//                       the emission folded into the row kernel behind a per-scan ticket: +12 .. +290 us, priced there)
#include "randt_internal.h"

#include <math.h>
#include <string.h>

#pragma clang fp contract(off)

#ifndef FILT_BLOCK
#define FILT_BLOCK 256
#endif
#define FILT_WAVES (FILT_BLOCK / 64)
#define FILT_STAGE 4     // kept points per row that the row kernel hands to the emission ready-made (2 float4 each)
#define FILT_EBLOCK 512  // emission kernel: all azimuth rows of a 400-row scan in one round
#ifndef FILT_UNROLL
#define FILT_UNROLL 12
#endif
// loads in flight per lane: rows of up to 12 * 256 = 3072 bins are fetched in one go

namespace {

// std::hypot(float, float) as glibc evaluates it: double sqrt, one rounding.
__device__ __forceinline__ float hypot_f(float x, float y) { return (float)sqrt((double)x * (double)x + (double)y * (double)y); }

// what the row workgroups leave for the emitting workgroup (32 bytes)
struct RowRec {
  int32_t m;        // cloud index of the row's detection, -1: none
  int32_t closer;   // first / last cloud index of the expanded run
  int32_t further;
  int32_t kept;     // points of the run that pass the range / intensity thresholds
  float angle;      // atan2 of the row's first point (the reference's current_angle)
  float maxi;       // peak intensity
  float peak_range; // hypot of the detection (the emission's per-azimuth peak record)
  int32_t bad;      // bit 0: an azimuth change inside the row (the scan is not organised azimuth after azimuth);
                    // bit 1: all of the row's kept points are in the stage
};

struct FilterArgs {
  const float* raw;   // [n_scans][n_az][n_bins][stride]
  int n_scans, n_az, n_bins, stride, ioff;
  float min_d, max_d, min_i, thr;
  double lo2, hi2;    // range test on the squared distance: min_d < hypot(x, y) < max_d  <=>  lo2 <= x^2 + y^2 <= hi2
  float T[12];
  float* out_pts;     // [n_scans][pitch_out][4]
  float* out_polar;   // nullable [n_scans][pitch_out][2]
  float* peaks;       // nullable [n_scans][n_az][3]
  int32_t* out_counts;
  int32_t* peak_counts;  // nullable
  int32_t* status;       // [n_scans] 0 ok, 1 not azimuth-organised, 2 output overflow
  int pitch_out;
  int out_vec;        // out_pts is 16-byte aligned
  // scratch
  RowRec* rows;       // [n_scans][n_az]
  float4* stage;      // [n_scans][n_az][FILT_STAGE][2]: {x' y' z' I}, {atan2 range - -} of a row's first kept points
};

// PACKED: 16-byte x y z I records (one dwordx4 load per point), known at compile time so that a row's loads are
// issued back to back with no branch between them.
template <bool PACKED>
__device__ __forceinline__ void fetch(const FilterArgs& A, const float* base, long long i, float& x, float& y, float& in) {
  if (PACKED) {
    const float4 p = reinterpret_cast<const float4*>(base)[i];
    x = p.x;
    y = p.y;
    in = A.ioff == 3 ? p.w : (A.ioff == 2 ? p.z : (A.ioff == 1 ? p.y : p.x));
  } else {
    const float* p = base + (size_t)i * A.stride;
    x = p[0];
    y = p[1];
    in = p[A.ioff];
  }
}

__device__ __forceinline__ bool keep_point(const FilterArgs& A, float x, float y, float in, float& dist) {
  dist = hypot_f(x, y);
  return (double)dist > (double)A.min_d && (double)dist < (double)A.max_d && (double)in > (double)A.min_i;
}

template <int NW>
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long* scratch, unsigned long long* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  __syncthreads();
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const unsigned long long s = scratch[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

// the output point of a kept return: pcl::transformPointCloud with initial_transform_radar_baselink_ (:124)
__device__ __forceinline__ float4 to_base(const FilterArgs& A, float x, float y, float z, float in) {
  float4 o;
  o.x = ((A.T[0] * x + A.T[1] * y) + A.T[2] * z) + A.T[3];
  o.y = ((A.T[4] * x + A.T[5] * y) + A.T[6] * z) + A.T[7];
  o.z = ((A.T[8] * x + A.T[9] * y) + A.T[10] * z) + A.T[11];
  o.w = in;
  return o;
}
template <bool PACKED>
__device__ __forceinline__ float fetch_z(const FilterArgs& A, const float* base, long long i) {
  return PACKED ? reinterpret_cast<const float4*>(base)[i].z : (A.stride > 2 ? base[(size_t)i * A.stride + 2] : 0.f);
}
template <bool PACKED>
__global__ __launch_bounds__(FILT_EBLOCK) void k_filter_emit(FilterArgs A) {
  __shared__ unsigned long long scratch[FILT_EBLOCK / 64];
  const int scan = blockIdx.x, tid = threadIdx.x;
  const float* base = A.raw + (size_t)scan * A.n_az * A.n_bins * A.stride;
  const RowRec* rows = A.rows + (size_t)scan * A.n_az;
  int bad = 0;
  int n_det = 0, n_out = 0;
  float* out = A.out_pts + (size_t)scan * A.pitch_out * 4;
  float* pol = A.out_polar ? A.out_polar + (size_t)scan * A.pitch_out * 2 : nullptr;
  float* pk = A.peaks ? A.peaks + (size_t)scan * A.n_az * 3 : nullptr;
  // the last azimuth is never flushed (the push happens when the NEXT azimuth starts): its record says "none"
  const int n_rows = A.n_az - 1;
  for (int r0 = 0; r0 < A.n_az; r0 += FILT_EBLOCK) {
    const int r = r0 + tid;
    RowRec rec;
    rec.m = -1;
    rec.closer = 0;
    rec.further = -1;
    rec.kept = 0;
    rec.angle = rec.maxi = rec.peak_range = 0.f;
    rec.bad = 0;
    float4 st[2 * FILT_STAGE];  // the row's staged points, fetched in the same round trip as its record
#pragma unroll
    for (int k = 0; k < 2 * FILT_STAGE; ++k) st[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n_rows) {
      rec = rows[r];
      const float4* sp = A.stage + ((size_t)scan * A.n_az + r) * (2 * FILT_STAGE);
#pragma unroll
      for (int k = 0; k < 2 * FILT_STAGE; ++k)
        if (pol || !(k & 1)) st[k] = sp[k];
    }
    // (the same round trip serves the organisation check: an azimuth change inside a row -- k_filter_rows --, and consecutive
    // azimuths must differ by more than the reference's 1e-4 rad threshold; the last row's record only takes part here)
    if (r < A.n_az) {
      const RowRec cur = r < n_rows ? rec : rows[r];
      if (cur.bad & 1) bad = 1;
      if (r > 0 && !(fabsf(cur.angle - rows[r - 1].angle) > 0.0001)) bad = 1;
    }
    const bool det = rec.m >= 0;
    // ONE block scan for both running counts: detections in the upper, kept points in the lower 32 bits (neither can carry:
    // a scan holds < 2^30 points)
    unsigned long long tot;
    const unsigned long long at = block_excl_scan<FILT_EBLOCK / 64>(det ? ((1ull << 32) | (unsigned long long)(unsigned)rec.kept) : 0ull, scratch, &tot);
    const int tot_det = (int)(tot >> 32), tot_kept = (int)(unsigned)tot;
    const int det_at = n_det + (int)(at >> 32);
    int out_at = n_out + (int)(unsigned)at;
    if (det) {
      if (pk) {
        pk[3 * det_at + 0] = rec.angle;
        pk[3 * det_at + 1] = rec.peak_range;
        pk[3 * det_at + 2] = rec.maxi;
      }
      auto put = [&](const float4& o, float ang, float dist) {
        if (out_at < A.pitch_out) {
          if (A.out_vec) {  // 16-byte aligned output: one store per point
            reinterpret_cast<float4*>(out)[out_at] = o;
          } else {
            float* op = out + (size_t)out_at * 4;
            op[0] = o.x, op[1] = o.y, op[2] = o.z, op[3] = o.w;
          }
          if (pol) {
            pol[2 * (size_t)out_at + 0] = ang;
            pol[2 * (size_t)out_at + 1] = dist;
          }
        }
        ++out_at;
      };
      if (rec.bad & 2) {
        // the row kernel left the run's kept points ready-made
#pragma unroll
        for (int k = 0; k < FILT_STAGE; ++k)
          if (k < rec.kept) put(st[2 * k], st[2 * k + 1].x, st[2 * k + 1].y);
      } else {
        // the run, eight bins per round trip (the loads of a round are independent; the stores are not)
        for (long long j0 = rec.closer; j0 <= rec.further; j0 += 8) {
          float x[8], y[8], in[8], z[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const long long j = j0 + u;
            x[u] = y[u] = in[u] = z[u] = 0.f;
            if (j <= rec.further) {
              fetch<PACKED>(A, base, j, x[u], y[u], in[u]);
              z[u] = fetch_z<PACKED>(A, base, j);
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            float dist;
            if (j0 + u <= rec.further && keep_point(A, x[u], y[u], in[u], dist)) put(to_base(A, x[u], y[u], z[u], in[u]), pol ? atan2f(y[u], x[u]) : 0.f, dist);
          }
        }
      }
    }
    n_det += tot_det;
    n_out += tot_kept;
  }
  bad = __syncthreads_or(bad);
  if (tid == 0) {
    A.out_counts[scan] = n_out <= A.pitch_out ? n_out : A.pitch_out;
    if (A.peak_counts) A.peak_counts[scan] = n_det;
    A.status[scan] = n_out > A.pitch_out ? 2 : (bad ? 1 : 0);  // written, not max-ed: no clearing launch in front of the two kernels
  }
}

// The run expansion (:80-108) in direction DIR = -1 (towards the sensor) / +1 (away from it): the first d >= 0 at which
//   (m + DIR (d + 1) leaves the cloud)  or  hypot(a) - hypot(b) > thr  or  I(a) <= I(b)  or  hypot(a) < min_range
// with a = m + DIR d, b = a + DIR; the run ends at that a.  Sequential in the reference; here both directions in ONE
// wavefront: lanes 0..31 walk inwards, lanes 32..63 outwards, 32 values of d per step and direction, one lane per bin
// evaluates the stopping rule and the lowest stopping lane of a half ends its walk (lanes beyond it evaluate bins the
// reference never touches: discarded); the loads of both directions share a round trip (runs are a handful of bins: 2
// kept points per azimuth in the Oxford-shaped scans).
// kept: how many bins of the walked parts (d = 0 .. stop inwards, d = 1 .. stop outwards, so that the detection itself
// is counted once) pass the output thresholds (:110-118) -- the same loads serve both questions.
template <bool PACKED>
__device__ __forceinline__ bool expand_both(const FilterArgs& A, const float* base, long long n, long long m, int lane, long long& closer, long long& further,
                                            int& kept, float& range_m, float4* stage) {
  const int half = lane >> 5, l = lane & 31;
  const long long dir = half ? 1 : -1;
  bool done_c = false, done_f = false, staged = false;
  closer = further = m;
  kept = 0;
  range_m = 0.f;
  for (long long d0 = 0; !(done_c && done_f); d0 += 32) {
    const long long a = m + dir * (d0 + l), b = a + dir;
    const bool live = half ? !done_f : !done_c;
    const bool a_in = live && a >= 0 && a <= n - 1, b_in = b >= 0 && b <= n - 1;
    bool stop = true, keep = false;
    float ha = 0.f, ax = 0.f, ay = 0.f, ai = 0.f;
    if (a_in) {
      float bx = 0.f, by = 0.f, bi = 0.f;
      fetch<PACKED>(A, base, a, ax, ay, ai);
      if (b_in) fetch<PACKED>(A, base, b, bx, by, bi);
      keep = keep_point(A, ax, ay, ai, ha) && (half == 0 || d0 + l > 0);  // the detection itself is counted once (inward half)
      if (b_in) stop = ((double)(ha - hypot_f(bx, by)) > (double)A.thr) || (ai <= bi) || ((double)ha < (double)A.min_d);
    }
    if (d0 == 0) range_m = __shfl(ha, 0, 64);  // hypot of the detection: the per-azimuth peak record
    const unsigned long long mask = __ballot(stop), kmask = __ballot(keep);
    const unsigned int mc = (unsigned int)mask, mf = (unsigned int)(mask >> 32), kc = (unsigned int)kmask, kf = (unsigned int)(kmask >> 32);
    unsigned int vc = 0, vf = 0;  // the kept bins of this step that belong to the run
    if (!done_c) {
      if (mc) {
        const int first = __ffs((int)mc) - 1;
        vc = kc & (first == 31 ? ~0u : ((1u << (first + 1)) - 1u));
        closer = m - (d0 + first);
        done_c = true;
      } else {
        vc = kc;
      }
    }
    if (!done_f) {
      if (mf) {
        const int first = __ffs((int)mf) - 1;
        vf = kf & (first == 31 ? ~0u : ((1u << (first + 1)) - 1u));
        further = m + (d0 + first);
        done_f = true;
      } else {
        vf = kf;
      }
    }
    kept += __popc(vc) + __popc(vf);
    // A run that ends within the first step on both sides (nearly all do) and keeps <= FILT_STAGE points hands them to the
    // emission ready-made, in cloud order (inward lanes: higher lane = lower index): the emission then has no second,
    // dependent look at the raw scan.  Same expressions as the emission's own path (to_base, atan2f, keep_point's range).
    if (d0 == 0 && done_c && done_f && stage && kept <= FILT_STAGE) {
      staged = true;
      const bool mine = half ? ((vf >> l) & 1u) : ((vc >> l) & 1u);
      if (mine) {
        const int rank = half ? __popc(vc) + __popc(vf & ((1u << l) - 1u)) : __popc(l == 31 ? 0u : (vc >> (l + 1)));
        stage[2 * rank] = to_base(A, ax, ay, fetch_z<PACKED>(A, base, a), ai);
        if (A.out_polar) stage[2 * rank + 1] = make_float4(atan2f(ay, ax), ha, 0.f, 0.f);
      }
    }
  }
  return staged;
}

// One 256-thread WORKGROUP per azimuth row (round 5).  The whole row -- up to FILT_UNROLL x 256 = 3072 bins, 48 KB -- is in
// flight at once, twelve dwordx4 loads per lane issued back to back; every lane keeps the arg-max of its own bins, a wavefront
// reduces with shuffles, the four partial results meet in LDS behind ONE barrier, and wavefront 0 alone expands the run while
// the row is cache-hot (expand_both) and writes the row record; the other three retire at the barrier.
// Why not the wavefront-per-row kernel of round 4 any more (history at the top of the file): a launch of ONE scan has 400 rows
// = 400 wavefronts for 1024 SIMDs, each walking its row through a ring of twelve loads -- 47 dependent rounds, 12 us, 0.14 of
// the HBM rate (round-4 verdict).  Split four ways the row needs ONE round trip, and tools/filter_ticket_probe.hip measured the
// organisation ahead at every launch size: 6.8 against 9.7 us per scan launched alone, 50.5 against 60.8 us per 16 scans over
// distinct inputs (a bare 16-scan read: 51.5).  The same probe priced the remaining idea -- folding the emission into the scan's
// last-arriving workgroup behind a device-scope ticket -- at +70 .. +290 us per 16 scans (one __threadfence per workgroup walks
// the L2): the emission stays a launch of its own.
#ifndef FILT_WPE
#define FILT_WPE 4
#endif
#define FILT_OCC __attribute__((amdgpu_waves_per_eu(FILT_WPE, FILT_WPE)))
// I3: the intensity is the fourth float of a packed record (x y z I, the reference's PointXYZI) -- known at compile time so
// that the visiting code has no branch on ioff
template <bool PACKED, bool I3>
__global__ __launch_bounds__(FILT_BLOCK) FILT_OCC void k_filter_rows(FilterArgs A) {
  __shared__ float s_best[FILT_WAVES];
  __shared__ int s_idx[FILT_WAVES], s_flag[FILT_WAVES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long g = blockIdx.x;  // scan * n_az + row
  const int scan = (int)(g / A.n_az), row = (int)(g - (long long)scan * A.n_az);
  const float* base = A.raw + (size_t)scan * A.n_az * A.n_bins * A.stride;
  const long long n = (long long)A.n_az * A.n_bins, r0 = (long long)row * A.n_bins;
  const unsigned rec = PACKED ? 16u : 4u * (unsigned)A.stride;          // bytes per point
  const char* rb = reinterpret_cast<const char*>(base + r0 * A.stride);  // uniform; lane offsets within a row fit 32 bits

  float best_i = 0.f;  // max_intensity starts at 0: only intensity > 0 can win
  int best_idx = 0x7fffffff;
  // the row's first point: its angle is the reference's current_angle for this azimuth (a uniform address: one scalar load)
  const float x0 = reinterpret_cast<const float*>(rb)[0], y0 = reinterpret_cast<const float*>(rb)[1];
  // The kernel must stay on the HBM roofline, so the two per-point tests are restated without transcendental
  // work: (a) range: hypot() in float after a double sqrt is monotone in d2 = x^2 + y^2, so the launcher
  // bisects the two double thresholds once; (b) organisation: a point whose direction is within 4e-5 rad of
  // the row's first point (|cross| <= 4e-5 dot, dot > 0) cannot differ from it by 1e-4 in atan2f; only other points
  // (none in an organised scan) and rows next to the +-pi cut take the exact atan2f comparison (below).
  // near_cut: |atan2f(y0, x0)| >= 3.14, decided without the atan2f (a superset: tan(pi - 3.14) = 0.00159)
  const bool near_cut = x0 < 0.f && fabsf(y0) <= 0.002f * fabsf(x0);
  bool suspect = false;
  auto visit = [&](int b, float px, float py, float pin) {
    const float cross = x0 * py - y0 * px, dot = x0 * px + y0 * py;
    // (dot > 0: a zero-filled return passes 0 <= 0, but the reference's atan2(0, 0) = 0 starts a new azimuth there)
    suspect |= !(fabsf(cross) <= 4e-5f * dot) || !(dot > 0.f);
    const double d2 = (double)px * (double)px + (double)py * (double)py;
    if (d2 >= A.lo2 && d2 <= A.hi2) {
      if (pin > best_i) {  // a lane meets its bins in increasing order: strict '>' keeps the first
        best_i = pin;
        best_idx = b;
      }
    }
  };
  // lane t holds bins t, t + 256, ...: FILT_UNROLL loads per lane and round, all issued before the first is looked at (rows
  // beyond 3072 bins take further rounds)
  for (int c0 = 0; c0 < A.n_bins; c0 += FILT_UNROLL * FILT_BLOCK) {
    float4 pt[FILT_UNROLL];
#pragma unroll
    for (int u = 0; u < FILT_UNROLL; ++u) {  // bins past the end of the row re-read its last bin and are skipped when visited
      const int b0 = c0 + u * FILT_BLOCK + tid;
      const unsigned off = (unsigned)(b0 < A.n_bins ? b0 : A.n_bins - 1) * rec;
      if (PACKED) {
        pt[u] = *reinterpret_cast<const float4*>(rb + off);
      } else {
        const float* p = reinterpret_cast<const float*>(rb + off);
        pt[u].x = p[0];
        pt[u].y = p[1];
        pt[u].w = p[A.ioff];
      }
    }
#pragma unroll
    for (int u = 0; u < FILT_UNROLL; ++u) {
      const int b = c0 + u * FILT_BLOCK + tid;
      if (PACKED) asm volatile("" ::"v"(pt[u].z));  // keeps the record ONE dwordx4 load (z is not needed here: the compiler would split it in two)
      if (b < A.n_bins) visit(b, pt[u].x, pt[u].y, (!PACKED || I3) ? pt[u].w : (A.ioff == 2 ? pt[u].z : (A.ioff == 1 ? pt[u].y : pt[u].x)));
    }
  }
  // arg-max over the wavefront: larger intensity wins, ties -> smaller index (strict '>', first index wins)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float oi = __shfl_xor(best_i, off, 64);
    const int ox = __shfl_xor(best_idx, off, 64);
    if (oi > best_i || (oi == best_i && ox < best_idx)) {
      best_i = oi;
      best_idx = ox;
    }
  }
  const int flagged = __ballot(suspect) != 0ull ? 1 : 0;
  if (lane == 0) {
    s_best[wave] = best_i;
    s_idx[wave] = best_idx;
    s_flag[wave] = flagged;
  }
  __syncthreads();
  int any_flag = 0;
  best_i = s_best[0];
  best_idx = s_idx[0];
#pragma unroll
  for (int w = 0; w < FILT_WAVES; ++w) {
    any_flag |= s_flag[w];
    if (w > 0 && (s_best[w] > best_i || (s_best[w] == best_i && s_idx[w] < best_idx))) {
      best_i = s_best[w];
      best_idx = s_idx[w];
    }
  }
  int bad = 0;
  float a0 = 0.f;
  if (any_flag || near_cut) {  // uniform over the workgroup
    // the exact organisation test (rare: the row at the +-pi cut, and clouds that are not organised azimuth after azimuth):
    // the row once more, eight loads in flight per lane
    a0 = atan2f(y0, x0);
    const bool cut = !(fabsf(a0) < 3.14f);
    for (int c0 = 0; c0 < A.n_bins; c0 += 8 * FILT_BLOCK) {
      float qx[8], qy[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b0 = c0 + u * FILT_BLOCK + tid;
        const float* q = reinterpret_cast<const float*>(rb + (unsigned)(b0 < A.n_bins ? b0 : A.n_bins - 1) * rec);
        qx[u] = q[0];
        qy[u] = q[1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float cross = x0 * qy[u] - y0 * qx[u], dot = x0 * qx[u] + y0 * qy[u];
        if (c0 + u * FILT_BLOCK + tid < A.n_bins && (cut || !(fabsf(cross) <= 4e-5f * dot) || !(dot > 0.f))) {
          const float ang = atan2f(qy[u], qx[u]);
          if (fabsf(ang - a0) > 0.0001) bad = 1;  // an azimuth change inside the row
        }
      }
    }
    bad = __syncthreads_or(bad);
  }
  if (wave != 0) return;
  if (!(any_flag || near_cut)) a0 = atan2f(y0, x0);
  // the row's detection; quirk: the first boundary pushes current_max_idx = 0 even if azimuth 0 had no valid return;
  // the last azimuth is never flushed
  long long m = (best_i > 0.f && best_idx != 0x7fffffff) ? r0 + best_idx : -1;
  float pk_i = best_i;
  if (row == 0 && m < 0) {
    m = 0;
    pk_i = 0.f;
  }
  if (row == A.n_az - 1) m = -1;
  long long closer = 0, further = -1;
  int kept = 0;
  float range_m = 0.f;
  bool staged = false;
  if (m >= 0) staged = expand_both<PACKED>(A, base, n, m, lane, closer, further, kept, range_m, A.stage + (size_t)g * (2 * FILT_STAGE));  // uniform over the wavefront
  if (lane == 0) {
    RowRec rec;
    rec.m = (int32_t)m;
    rec.closer = (int32_t)closer;
    rec.further = (int32_t)further;
    rec.kept = kept;
    rec.angle = a0;
    rec.maxi = pk_i;
    rec.peak_range = range_m;
    rec.bad = (bad ? 1 : 0) | (staged ? 2 : 0);
    A.rows[g] = rec;
  }
}

}  // namespace

namespace {
// host twin of hypot_f (glibc: correctly rounded double sqrt, one rounding to float)
inline float hypot_from_d2(double d2) { return (float)sqrt(d2); }
// smallest non-negative double d2 with pred(d2) true, for a predicate that is monotone (false ... true) in d2
template <typename Pred>
double first_true(Pred pred) {
  unsigned long long lo = 0, hi = 0x7ff0000000000000ull;  // bit patterns of +0 .. +inf are ordered like the values
  if (pred(0.0)) return 0.0;
  if (!pred(INFINITY)) return NAN;  // never true: every comparison with the threshold is false
  while (hi - lo > 1) {
    const unsigned long long mid = lo + (hi - lo) / 2;
    double v;
    memcpy(&v, &mid, 8);
    if (pred(v)) hi = mid; else lo = mid;
  }
  double v;
  memcpy(&v, &hi, 8);
  return v;
}
}  // namespace

int launch_filter_scan(randt_ctx* ctx, const float* d_raw, int n_scans, int n_az, int n_bins, int stride, int ioff,
                       const randt_filter_params* fp, float* d_out_pts, int pitch_out, int32_t* d_out_counts, float* d_polar,
                       float* d_peaks, int32_t* d_peak_counts, int32_t* d_status, void* d_scratch) {
  randt_note_enqueue(ctx);  // (RANDT_SOLVE_AUTO of the process's other contexts: this one has work in flight)
  FilterArgs A;
  A.raw = d_raw;
  A.n_scans = n_scans;
  A.n_az = n_az;
  A.n_bins = n_bins;
  A.stride = stride;
  A.ioff = ioff;
  A.min_d = fp->min_range;
  A.max_d = fp->max_range;
  A.min_i = fp->min_intensity;
  A.thr = fp->beam_distance_increment_threshold;
  {
    const float mn = A.min_d, mx = A.max_d;
    A.lo2 = first_true([mn](double d2) { return (double)hypot_from_d2(d2) > (double)mn; });
    // largest d2 still below max_d = predecessor of the first d2 that is not
    const double first_not_below = first_true([mx](double d2) { return !((double)hypot_from_d2(d2) < (double)mx); });
    A.hi2 = first_not_below > 0.0 ? nextafter(first_not_below, -1.0) : (first_not_below == 0.0 ? -1.0 : (double)INFINITY);
  }
  for (int i = 0; i < 12; ++i) A.T[i] = fp->sensor_to_base[i];
  A.out_pts = d_out_pts;
  A.out_polar = d_polar;
  A.peaks = d_peaks;
  A.out_counts = d_out_counts;
  A.peak_counts = d_peak_counts;
  A.status = d_status;
  A.pitch_out = pitch_out;
  A.out_vec = ((uintptr_t)d_out_pts & 15) == 0;
  A.rows = (RowRec*)d_scratch;  // [n_scans * n_az] records, then the stage (api.hip sizes the workspace: FILT_WS_PER_ROW)
  A.stage = (float4*)((char*)d_scratch + (((size_t)n_scans * n_az * sizeof(RowRec) + 255) & ~(size_t)255));
  // one workgroup per azimuth row; the dispatcher back-fills the chip as workgroups retire
  if ((long long)n_scans * n_az > 0x7fffffffll) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "filter: more than 2^31 azimuth rows in one launch", hipSuccess);
  const int row_wgs = n_scans * n_az;
  if (stride == 4 && ioff == 3) {
    hipLaunchKernelGGL((k_filter_rows<true, true>), dim3(row_wgs), dim3(FILT_BLOCK), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_filter_emit<true>, dim3(n_scans), dim3(FILT_EBLOCK), 0, ctx->stream, A);
  } else if (stride == 4) {
    hipLaunchKernelGGL((k_filter_rows<true, false>), dim3(row_wgs), dim3(FILT_BLOCK), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_filter_emit<true>, dim3(n_scans), dim3(FILT_EBLOCK), 0, ctx->stream, A);
  } else {
    hipLaunchKernelGGL((k_filter_rows<false, false>), dim3(row_wgs), dim3(FILT_BLOCK), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_filter_emit<false>, dim3(n_scans), dim3(FILT_EBLOCK), 0, ctx->stream, A);
  }
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
