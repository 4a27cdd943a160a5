// Radar scan filter for gfx950 -- SURVEY row f-1, the HBM-streaming stage in front of the NDT build
// (compiled with -ffp-contract=off: the emitted points must equal the reference's bit for bit).
//
// Replaces RadarPreprocessor::filterScan (src/radar_preprocessing/radar_preprocessor.cpp:45-125):
// per azimuth the strongest return inside (min_range, max_range), then the run of monotonically
// decreasing intensity around it, thresholded by min_intensity, transformed sensor -> base.
//
//   k_filter_rows   256-thread workgroups that each walk a strided set of azimuth rows of one scan (one row each when a
//                   launch has few scans, several when it has many).  Per row:
//                   1. the row of the raw polar scan (16 B / point, 19.2 MB for 400 x 3000) is read exactly once with
//                      coalesced float4 loads, ALL of a lane's loads in flight at once (12 for 3000 bins); per-row
//                      arg-max (strict '>', first index wins) by shuffles + a 4-way LDS combine; the same pass verifies
//                      that the cloud really
//                      is organised azimuth after azimuth (the reference detects azimuth changes with
//                      |atan2 - current| > 1e-4 while walking the cloud sequentially);
//                   2. the row's detection is expanded towards / away from the sensor right there, while the row is
//                      cache-hot: wavefront 0 walks inwards and wavefront 1 outwards, 64 bins per step (one lane per
//                      bin evaluates the reference's stopping rule, the first lane that stops ends the walk) instead
//                      of one dependent load pair per bin; the same loads count the run's points that pass the output
//                      thresholds; a 32-byte row record goes to scratch.
//   k_filter_emit   one workgroup per scan: replays the reference's detection-list semantics (last azimuth never
//                   flushed, the very first boundary pushes index 0), block-scans the kept counts over the rows and
//                   emits the points and the per-azimuth peaks in the reference's order (run bins re-read from L2,
//                   eight independent loads at a time).
// Measured alternatives (16 scans per launch, 307 MB): emission fused into the row kernel behind a device-scope ticket
// 130 us (every row workgroup then ends with a store acknowledgement and a returning atomic under full read load);
// a second register set holding the workgroup's next row (191 registers, two workgroups per CU) 93 us; this version
// (128 registers, four per CU) 72 us; the same with non-temporal row loads (the run expansion then misses L2) 100 us; squeezed to 96 / 80 registers by the compiler (spills) 84 / 104 us.
#include "randt_internal.h"

#include <math.h>
#include <string.h>

#pragma clang fp contract(off)

#define FILT_BLOCK 256
#define FILT_WAVES (FILT_BLOCK / 64)
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
  int32_t bad;      // 1: an azimuth change inside the row (the scan is not organised azimuth after azimuth)
};

struct FilterArgs {
  const float* raw;   // [n_scans][n_az][n_bins][stride]
  int n_az, n_bins, stride, ioff;
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
  // scratch
  RowRec* rows;       // [n_scans][n_az]
};

// PACKED: 16-byte x y z I records (one dwordx4 load per point), known at compile time so that a row's loads are
// issued back to back with no branch between them.
template <bool PACKED>
__device__ __forceinline__ void fetch(const FilterArgs& A, const float* base, long long i, float& x, float& y, float& in) {
  if (PACKED) {
    const float4 p = reinterpret_cast<const float4*>(base)[i];
    x = p.x;
    y = p.y;
    in = A.ioff == 3 ? p.w : (A.ioff == 2 ? p.z : p.x);
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

// One direction of the run expansion (:80-108), DIR = -1 towards the sensor, +1 away from it: the first d >= 0 at which
//   (m + DIR (d + 1) leaves the cloud)  or  hypot(a) - hypot(b) > thr  or  I(a) <= I(b)  or  hypot(a) < min_range
// with a = m + DIR d, b = a + DIR; returns that a.  Sequential in the reference; here 64 values of d per step, one per
// lane, the lowest stopping lane wins (lanes beyond it evaluate bins the reference never touches: discarded).
// kept: how many bins of the walked part of the run (d = 0 .. stop for DIR = -1, d = 1 .. stop for DIR = +1, so that the
// detection itself is counted once) pass the output thresholds (:110-118) -- the same loads serve both questions.
template <int DIR, bool PACKED>
__device__ __forceinline__ long long expand_run(const FilterArgs& A, const float* base, long long n, long long m, int lane, int& kept, float* range_m = nullptr) {
  kept = 0;
  for (long long d0 = 0;; d0 += 64) {
    const long long a = m + DIR * (d0 + lane), b = a + DIR;
    const bool a_in = a >= 0 && a <= n - 1, b_in = b >= 0 && b <= n - 1;
    bool stop = true, keep = false;
    float ha = 0.f;
    if (a_in) {
      float ax, ay, ai, bx = 0.f, by = 0.f, bi = 0.f;
      fetch<PACKED>(A, base, a, ax, ay, ai);
      if (b_in) fetch<PACKED>(A, base, b, bx, by, bi);
      keep = keep_point(A, ax, ay, ai, ha) && (DIR < 0 || d0 + lane > 0);
      if (b_in) stop = ((double)(ha - hypot_f(bx, by)) > (double)A.thr) || (ai <= bi) || ((double)ha < (double)A.min_d);
    }
    if (range_m && d0 == 0) *range_m = __shfl(ha, 0, 64);
    const unsigned long long mask = __ballot(stop), kmask = __ballot(keep);
    if (mask) {
      const int first = __ffsll((long long)mask) - 1;
      kept += __popcll(kmask & (first == 63 ? ~0ull : ((1ull << (first + 1)) - 1ull)));
      return m + DIR * (d0 + (long long)first);
    }
    kept += __popcll(kmask);
  }
}

template <int NW>
__device__ __forceinline__ int block_excl_scan(int v, int* scratch, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  __syncthreads();
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int s = scratch[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

template <bool PACKED>
__global__ __launch_bounds__(FILT_EBLOCK) void k_filter_emit(FilterArgs A) {
  __shared__ int scratch[FILT_EBLOCK / 64];
  const int scan = blockIdx.x, tid = threadIdx.x;
  const float* base = A.raw + (size_t)scan * A.n_az * A.n_bins * A.stride;
  const RowRec* rows = A.rows + (size_t)scan * A.n_az;
  // consecutive azimuths must differ by more than the reference's 1e-4 rad threshold
  int bad = 0;
  for (int r = tid; r < A.n_az; r += FILT_EBLOCK) {
    const RowRec cur = rows[r];
    if (cur.bad) bad = 1;  // an azimuth change inside a row (k_filter_rows)
    if (r > 0 && !(fabsf(cur.angle - rows[r - 1].angle) > 0.0001)) bad = 1;
  }
  bad = __syncthreads_or(bad);

  int n_det = 0, n_out = 0;
  float* out = A.out_pts + (size_t)scan * A.pitch_out * 4;
  float* pol = A.out_polar ? A.out_polar + (size_t)scan * A.pitch_out * 2 : nullptr;
  float* pk = A.peaks ? A.peaks + (size_t)scan * A.n_az * 3 : nullptr;
  // the last azimuth is never flushed (the push happens when the NEXT azimuth starts): its record says "none"
  const int n_rows = A.n_az - 1;
  for (int r0 = 0; r0 < n_rows; r0 += FILT_EBLOCK) {
    const int r = r0 + tid;
    RowRec rec;
    rec.m = -1;
    rec.closer = 0;
    rec.further = -1;
    rec.kept = 0;
    rec.angle = rec.maxi = rec.peak_range = 0.f;
    rec.bad = 0;
    if (r < n_rows) rec = rows[r];
    const bool det = rec.m >= 0;
    int tot_det, tot_kept;
    const int det_at = n_det + block_excl_scan<FILT_EBLOCK / 64>(det ? 1 : 0, scratch, &tot_det);
    int out_at = n_out + block_excl_scan<FILT_EBLOCK / 64>(det ? rec.kept : 0, scratch, &tot_kept);
    if (det) {
      if (pk) {
        pk[3 * det_at + 0] = rec.angle;
        pk[3 * det_at + 1] = rec.peak_range;
        pk[3 * det_at + 2] = rec.maxi;
      }
      // the run, eight bins per round trip (the loads of a round are independent; the stores are not)
      for (long long j0 = rec.closer; j0 <= rec.further; j0 += 8) {
        float x[8], y[8], in[8], z[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const long long j = j0 + u;
          x[u] = y[u] = in[u] = z[u] = 0.f;
          if (j <= rec.further) {
            fetch<PACKED>(A, base, j, x[u], y[u], in[u]);
            z[u] = PACKED ? reinterpret_cast<const float4*>(base)[j].z : (A.stride > 2 ? base[(size_t)j * A.stride + 2] : 0.f);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float dist;
          if (j0 + u <= rec.further && keep_point(A, x[u], y[u], in[u], dist)) {
            if (out_at < A.pitch_out) {
              float* o = out + (size_t)out_at * 4;
              // pcl::transformPointCloud with initial_transform_radar_baselink_ (:124)
              o[0] = ((A.T[0] * x[u] + A.T[1] * y[u]) + A.T[2] * z[u]) + A.T[3];
              o[1] = ((A.T[4] * x[u] + A.T[5] * y[u]) + A.T[6] * z[u]) + A.T[7];
              o[2] = ((A.T[8] * x[u] + A.T[9] * y[u]) + A.T[10] * z[u]) + A.T[11];
              o[3] = in[u];
              if (pol) {
                pol[2 * (size_t)out_at + 0] = atan2f(y[u], x[u]);
                pol[2 * (size_t)out_at + 1] = dist;
              }
            }
            ++out_at;
          }
        }
      }
    }
    n_det += tot_det;
    n_out += tot_kept;
  }
  if (tid == 0) {
    A.out_counts[scan] = n_out <= A.pitch_out ? n_out : A.pitch_out;
    if (A.peak_counts) A.peak_counts[scan] = n_det;
    A.status[scan] = n_out > A.pitch_out ? 2 : (bad ? 1 : 0);  // written, not max-ed: no clearing launch in front of the two kernels
  }
}

// Four wavefronts per SIMD (<= 128 registers; the compiler would take 145 and drop to three): 79 -> 72 us per 16 scans.
#ifndef FILT_WPE
#define FILT_WPE 4
#endif
#define FILT_OCC __attribute__((amdgpu_waves_per_eu(FILT_WPE, FILT_WPE)))
template <bool PACKED>
__global__ __launch_bounds__(FILT_BLOCK) FILT_OCC void k_filter_rows(FilterArgs A) {
  __shared__ float s_i[2][FILT_WAVES];
  __shared__ int s_idx[2][FILT_WAVES];
  __shared__ int s_bad[2][FILT_WAVES];
  __shared__ long long s_run[2][2];  // closer, further (double-buffered: one barrier per use)
  __shared__ int s_kept[2][2];
  __shared__ float s_range[2];
  const int scan = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* base = A.raw + (size_t)scan * A.n_az * A.n_bins * A.stride;
  const long long n = (long long)A.n_az * A.n_bins;
  // packed rows of up to FILT_UNROLL * 256 bins: the whole row in flight at once, the next row behind it
  const bool one_shot = PACKED && A.n_bins <= FILT_UNROLL * FILT_BLOCK;
  int par = 0;

  // all loads of a row (one_shot): FILT_UNROLL per lane, no control flow between them (bins past the end of the row
  // re-read its last bin and are skipped when the row is visited)
  auto issue = [&](int row, float4* pt) {
    const float4* rp = reinterpret_cast<const float4*>(base) + (long long)row * A.n_bins;
#pragma unroll
    for (int u = 0; u < FILT_UNROLL; ++u) {
      const int b = tid + u * FILT_BLOCK;
      pt[u] = rp[b < A.n_bins ? b : A.n_bins - 1];
    }
  };
  // everything else of a row; pt: the row's points if one_shot (else the row is streamed here)
  auto finish = [&](int row, const float4* pt) {
    const long long r0 = (long long)row * A.n_bins;
    float best_i = 0.f;  // max_intensity starts at 0: only intensity > 0 can win
    int best_idx = 0x7fffffff;
    int bad = 0;
    // angle of the row's first point = the reference's current_angle for this azimuth
    float x0, y0, i0;
    fetch<PACKED>(A, base, r0, x0, y0, i0);
    const float a0 = atan2f(y0, x0);
    // The kernel must stay on the HBM roofline, so the two per-point tests are restated without transcendental
    // work: (a) range: hypot() in float after a double sqrt is monotone in d2 = x^2 + y^2, so the launcher
    // bisects the two double thresholds once; (b) organisation: a point whose direction is within 4e-5 rad of
    // the row's first point (|cross| <= 4e-5 dot) cannot differ from it by 1e-4 in atan2f; only other points
    // (none in an organised scan) and rows next to the +-pi cut take the exact atan2f comparison.
    const bool near_cut = !(fabsf(a0) < 3.14f);
    auto visit = [&](int b, float px, float py, float pin) {
      const float cross = x0 * py - y0 * px, dot = x0 * px + y0 * py;
      if (near_cut || !(fabsf(cross) <= 4e-5f * dot)) {
        const float ang = atan2f(py, px);
        if (fabsf(ang - a0) > 0.0001) bad = 1;  // an azimuth change inside the row
      }
      const double d2 = (double)px * (double)px + (double)py * (double)py;
      if (d2 >= A.lo2 && d2 <= A.hi2) {
        if (pin > best_i || (pin == best_i && pin > 0.f && b < best_idx)) {
          best_i = pin;
          best_idx = b;
        }
      }
    };
    if (one_shot) {
#pragma unroll
      for (int u = 0; u < FILT_UNROLL; ++u) {
        const int b = tid + u * FILT_BLOCK;
        if (b < A.n_bins) visit(b, pt[u].x, pt[u].y, A.ioff == 3 ? pt[u].w : (A.ioff == 2 ? pt[u].z : pt[u].x));
      }
    } else {
      for (int b0 = tid; b0 < A.n_bins; b0 += 4 * FILT_BLOCK) {
        float lx[4], ly[4], li[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = b0 + u * FILT_BLOCK;
          lx[u] = ly[u] = li[u] = 0.f;
          if (b < A.n_bins) fetch<PACKED>(A, base, r0 + b, lx[u], ly[u], li[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = b0 + u * FILT_BLOCK;
          if (b < A.n_bins) visit(b, lx[u], ly[u], li[u]);
        }
      }
    }
    // wave arg-max: larger intensity wins, ties -> smaller index
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float oi = __shfl_xor(best_i, off, 64);
      const int ox = __shfl_xor(best_idx, off, 64);
      if (oi > best_i || (oi == best_i && ox < best_idx)) {
        best_i = oi;
        best_idx = ox;
      }
      bad |= __shfl_xor(bad, off, 64);
    }
    if (lane == 0) {
      s_i[par][wave] = best_i;
      s_idx[par][wave] = best_idx;
      s_bad[par][wave] = bad;
    }
    __syncthreads();
    best_i = s_i[par][0];  // every thread: the same combine in the same order
    best_idx = s_idx[par][0];
    bad = s_bad[par][0];
#pragma unroll
    for (int w = 1; w < FILT_WAVES; ++w) {
      if (s_i[par][w] > best_i || (s_i[par][w] == best_i && s_idx[par][w] < best_idx)) {
        best_i = s_i[par][w];
        best_idx = s_idx[par][w];
      }
      bad |= s_bad[par][w];
    }
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
    if (m >= 0) {  // uniform over the workgroup
      if (wave == 0) {
        int kc = 0;
        long long c = 0;
        float rm = 0.f;
        c = expand_run<-1, PACKED>(A, base, n, m, lane, kc, &rm);  // (+ the detection's range for the peak record: lane 0's hypot)
        if (lane == 0) {
          s_run[par][0] = c;
          s_kept[par][0] = kc;
          s_range[par] = rm;
        }
      } else if (wave == 1) {
        int kf = 0;
        long long f = 0;
        f = expand_run<+1, PACKED>(A, base, n, m, lane, kf);
        if (lane == 0) {
          s_run[par][1] = f;
          s_kept[par][1] = kf;
        }
      }
      __syncthreads();
      closer = s_run[par][0];
      further = s_run[par][1];
      kept = s_kept[par][0] + s_kept[par][1];
      range_m = s_range[par];
    }
    if (tid == 0) {
      RowRec rec;
      rec.m = (int32_t)m;
      rec.closer = (int32_t)closer;
      rec.further = (int32_t)further;
      rec.kept = kept;
      rec.angle = a0;
      rec.maxi = pk_i;
      rec.peak_range = range_m;
      rec.bad = bad;
      A.rows[(size_t)scan * A.n_az + row] = rec;
    }
    par ^= 1;  // the next row's exchange goes through the other half of the LDS buffers
  };

  float4 pa[FILT_UNROLL];
  const int G = gridDim.x;
  for (int row = blockIdx.x; row < A.n_az; row += G) {
    if (one_shot) issue(row, pa);
    finish(row, pa);
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
  FilterArgs A;
  A.raw = d_raw;
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
  A.rows = (RowRec*)d_scratch;
  // Row workgroups per scan: one per row as long as all of them are resident at once; beyond that as many as the chip
  // holds (occupancy x CUs), each walking several rows with its next row's loads already in flight.
  int per_scan = n_az;
  {
    static int resident_of[64] = {0};  // resident workgroups per device (a process may hold contexts on several GPUs)
    int& resident = resident_of[ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0];
    if (resident == 0) {
      int per_cu = 0, cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_filter_rows<true>, FILT_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 256;
      resident = per_cu * cus;
    }
    const long long total = (long long)n_scans * n_az;
    if (total > resident) {
      const int rows_per_wg = (int)((total + resident - 1) / resident);
      per_scan = (n_az + rows_per_wg - 1) / rows_per_wg;
    }
#ifdef FILT_ROWS_PER_WG
    per_scan = (n_az + FILT_ROWS_PER_WG - 1) / FILT_ROWS_PER_WG;
#endif
  }
  if (stride == 4) {
    hipLaunchKernelGGL(k_filter_rows<true>, dim3(per_scan, n_scans), dim3(FILT_BLOCK), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_filter_emit<true>, dim3(n_scans), dim3(FILT_EBLOCK), 0, ctx->stream, A);
  } else {
    hipLaunchKernelGGL(k_filter_rows<false>, dim3(per_scan, n_scans), dim3(FILT_BLOCK), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_filter_emit<false>, dim3(n_scans), dim3(FILT_EBLOCK), 0, ctx->stream, A);
  }
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
