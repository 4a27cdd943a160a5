// Radar scan filter for gfx950 -- SURVEY row f-1, the HBM-streaming stage in front of the NDT build
// (compiled with -ffp-contract=off: the emitted points must equal the reference's bit for bit).
//
// Replaces RadarPreprocessor::filterScan (src/radar_preprocessing/radar_preprocessor.cpp:45-125):
// per azimuth the strongest return inside (min_range, max_range), then the run of monotonically
// decreasing intensity around it, thresholded by min_intensity, transformed sensor -> base.
//
//   k_filter_peaks   one 256-thread workgroup per azimuth row: the raw polar scan (16 B / point,
//                    19.2 MB for 400 x 3000) is read exactly once with coalesced float4 loads,
//                    four loads in flight per lane; per-row arg-max (strict '>', first index wins)
//                    by DPP-free shuffles + a 4-way LDS combine; also verifies that the cloud really
//                    is organised azimuth after azimuth (the reference detects azimuth changes with
//                    |atan2 - current| > 1e-4 while walking the cloud sequentially).
//   k_filter_expand  one workgroup per scan: replays the reference's detection list semantics (last
//                    azimuth never flushed, an empty azimuth re-uses the previous index, the very
//                    first boundary pushes index 0), expands each detection towards / away from the
//                    sensor (L2-resident re-reads), block-scans the kept counts and emits the
//                    points in the reference's order.
#include "randt_internal.h"

#include <math.h>
#include <string.h>

#pragma clang fp contract(off)

#define FILT_BLOCK 256
#define FILT_XBLOCK 512  // expansion kernel: all azimuth rows of a 400-row scan in one round

namespace {

// std::hypot(float, float) as glibc evaluates it: double sqrt, one rounding.
__device__ __forceinline__ float hypot_f(float x, float y) { return (float)sqrt((double)x * (double)x + (double)y * (double)y); }

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
  int32_t* row_max;   // [n_scans][n_az]
  float* row_angle;   // [n_scans][n_az]
  float* row_maxi;    // [n_scans][n_az]
};

__device__ __forceinline__ void fetch(const FilterArgs& A, const float* base, long long i, float& x, float& y, float& in) {
  if (A.stride == 4) {
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

__global__ __launch_bounds__(FILT_BLOCK) void k_filter_peaks(FilterArgs A) {
  __shared__ float s_i[4];
  __shared__ int s_idx[4];
  __shared__ int s_bad[4];
  const int row = blockIdx.x, scan = blockIdx.y, tid = threadIdx.x;
  const float* base = A.raw + (size_t)scan * A.n_az * A.n_bins * A.stride;
  const long long r0 = (long long)row * A.n_bins;
  // angle of the row's first point = the reference's current_angle for this azimuth
  float x0, y0, i0;
  fetch(A, base, r0, x0, y0, i0);
  const float a0 = atan2f(y0, x0);
  float best_i = 0.f;  // max_intensity starts at 0: only intensity > 0 can win
  int best_idx = 0x7fffffff;
  int bad = 0;
  // The kernel must stay on the HBM roofline, so the two per-point tests are restated without transcendental
  // work: (a) range: hypot() in float after a double sqrt is monotone in d2 = x^2 + y^2, so the launcher
  // bisects the two double thresholds once; (b) organisation: a point whose direction is within 4e-5 rad of
  // the row's first point (|cross| <= 4e-5 dot) cannot differ from it by 1e-4 in atan2f; only other points
  // (none in an organised scan) and rows next to the +-pi cut take the exact atan2f comparison.
  const bool near_cut = !(fabsf(a0) < 3.14f);
  for (int b0 = tid; b0 < A.n_bins; b0 += 4 * FILT_BLOCK) {
    float x[4], y[4], in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + u * FILT_BLOCK;
      x[u] = y[u] = in[u] = 0.f;
      if (b < A.n_bins) fetch(A, base, r0 + b, x[u], y[u], in[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + u * FILT_BLOCK;
      if (b < A.n_bins) {
        const float cross = x0 * y[u] - y0 * x[u], dot = x0 * x[u] + y0 * y[u];
        if (near_cut || !(fabsf(cross) <= 4e-5f * dot)) {
          const float ang = atan2f(y[u], x[u]);
          if (fabsf(ang - a0) > 0.0001) bad = 1;  // an azimuth change inside the row
        }
        const double d2 = (double)x[u] * (double)x[u] + (double)y[u] * (double)y[u];
        if (d2 >= A.lo2 && d2 <= A.hi2) {
          if (in[u] > best_i || (in[u] == best_i && in[u] > 0.f && b < best_idx)) {
            best_i = in[u];
            best_idx = b;
          }
        }
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
  if ((tid & 63) == 0) {
    s_i[tid >> 6] = best_i;
    s_idx[tid >> 6] = best_idx;
    s_bad[tid >> 6] = bad;
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      if (s_i[w] > best_i || (s_i[w] == best_i && s_idx[w] < best_idx)) {
        best_i = s_i[w];
        best_idx = s_idx[w];
      }
      bad |= s_bad[w];
    }
    const size_t o = (size_t)scan * A.n_az + row;
    A.row_max[o] = (best_i > 0.f && best_idx != 0x7fffffff) ? (int)(r0 + best_idx) : -1;
    A.row_angle[o] = a0;
    A.row_maxi[o] = best_i;
    if (bad) atomicMax(&A.status[scan], 1);
  }
}

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
  for (int w = 0; w < FILT_XBLOCK / 64; ++w) {
    const int s = scratch[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(FILT_XBLOCK) void k_filter_expand(FilterArgs A) {
  __shared__ int scratch[FILT_XBLOCK / 64];
  const int scan = blockIdx.x, tid = threadIdx.x;
  const float* base = A.raw + (size_t)scan * A.n_az * A.n_bins * A.stride;
  const long long n = (long long)A.n_az * A.n_bins;
  const int32_t* row_max = A.row_max + (size_t)scan * A.n_az;
  const float* row_angle = A.row_angle + (size_t)scan * A.n_az;
  const float* row_maxi = A.row_maxi + (size_t)scan * A.n_az;
  // consecutive azimuths must differ by more than the reference's 1e-4 rad threshold
  int bad = 0;
  for (int r = 1 + tid; r < A.n_az; r += FILT_XBLOCK)
    if (!(fabsf(row_angle[r] - row_angle[r - 1]) > 0.0001)) bad = 1;
  if (bad) atomicMax(&A.status[scan], 1);

  int n_det = 0, n_out = 0;
  float* out = A.out_pts + (size_t)scan * A.pitch_out * 4;
  float* pol = A.out_polar ? A.out_polar + (size_t)scan * A.pitch_out * 2 : nullptr;
  float* pk = A.peaks ? A.peaks + (size_t)scan * A.n_az * 3 : nullptr;
  // the last azimuth is never flushed (the push happens when the NEXT azimuth starts)
  const int n_rows = A.n_az - 1;
  for (int r0 = 0; r0 < n_rows; r0 += FILT_XBLOCK) {
    const int r = r0 + tid;
    long long m = -1;
    float pk_i = 0.f;
    if (r < n_rows) {
      m = row_max[r];
      pk_i = row_maxi[r];
      // quirk: the first boundary pushes current_max_idx = 0 even if azimuth 0 had no valid return
      if (r == 0 && m < 0) {
        m = 0;
        pk_i = 0.f;
      }
    }
    const bool det = m >= 0;
    long long closer = 0, further = -1;
    int kept = 0;
    if (det) {
      // :80-108 expansion towards the sensor, then away from it
      long long d = 0;
      for (;;) {
        const long long b = m - d - 1;
        if (b < 0 || b > n - 1) { closer = m - d; break; }
        const long long a = m - d;
        float ax, ay, ai, bx, by, bi;
        fetch(A, base, a, ax, ay, ai);
        fetch(A, base, b, bx, by, bi);
        const float ha = hypot_f(ax, ay);
        if (((double)(ha - hypot_f(bx, by)) > (double)A.thr) || (ai <= bi) || ((double)ha < (double)A.min_d)) { closer = a; break; }
        ++d;
      }
      d = 0;
      for (;;) {
        const long long b = m + d + 1;
        if (b < 0 || b > n - 1) { further = m + d; break; }
        const long long a = m + d;
        float ax, ay, ai, bx, by, bi;
        fetch(A, base, a, ax, ay, ai);
        fetch(A, base, b, bx, by, bi);
        const float ha = hypot_f(ax, ay);
        if (((double)(ha - hypot_f(bx, by)) > (double)A.thr) || (ai <= bi) || ((double)ha < (double)A.min_d)) { further = a; break; }
        ++d;
      }
      for (long long j = closer; j <= further; ++j) {
        float x, y, in;
        fetch(A, base, j, x, y, in);
        const float dist = hypot_f(x, y);
        kept += ((double)dist > (double)A.min_d && (double)dist < (double)A.max_d && (double)in > (double)A.min_i) ? 1 : 0;
      }
    }
    int tot_det, tot_kept;
    const int det_at = n_det + block_excl_scan(det ? 1 : 0, scratch, &tot_det);
    int out_at = n_out + block_excl_scan(kept, scratch, &tot_kept);
    if (det) {
      if (pk) {
        float mx, my, mi;
        fetch(A, base, m, mx, my, mi);
        pk[3 * det_at + 0] = row_angle[r];
        pk[3 * det_at + 1] = hypot_f(mx, my);
        pk[3 * det_at + 2] = pk_i;
      }
      for (long long j = closer; j <= further; ++j) {
        float x, y, in;
        fetch(A, base, j, x, y, in);
        const float dist = hypot_f(x, y);
        if ((double)dist > (double)A.min_d && (double)dist < (double)A.max_d && (double)in > (double)A.min_i) {
          if (out_at < A.pitch_out) {
            const float z = A.stride > 2 ? (A.stride == 4 ? reinterpret_cast<const float4*>(base)[j].z : base[(size_t)j * A.stride + 2]) : 0.f;
            float* o = out + (size_t)out_at * 4;
            // pcl::transformPointCloud with initial_transform_radar_baselink_ (:124)
            o[0] = ((A.T[0] * x + A.T[1] * y) + A.T[2] * z) + A.T[3];
            o[1] = ((A.T[4] * x + A.T[5] * y) + A.T[6] * z) + A.T[7];
            o[2] = ((A.T[8] * x + A.T[9] * y) + A.T[10] * z) + A.T[11];
            o[3] = in;
            if (pol) {
              pol[2 * (size_t)out_at + 0] = atan2f(y, x);
              pol[2 * (size_t)out_at + 1] = dist;
            }
          }
          ++out_at;
        }
      }
    }
    n_det += tot_det;
    n_out += tot_kept;
  }
  if (tid == 0) {
    A.out_counts[scan] = n_out <= A.pitch_out ? n_out : A.pitch_out;
    if (A.peak_counts) A.peak_counts[scan] = n_det;
    if (n_out > A.pitch_out) atomicMax(&A.status[scan], 2);
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
  A.row_max = (int32_t*)d_scratch;
  A.row_angle = (float*)(A.row_max + (size_t)n_scans * n_az);
  A.row_maxi = A.row_angle + (size_t)n_scans * n_az;
  RANDT_HIP_CHECK(ctx, hipMemsetAsync(d_status, 0, sizeof(int32_t) * n_scans, ctx->stream));
  hipLaunchKernelGGL(k_filter_peaks, dim3(n_az, n_scans), dim3(FILT_BLOCK), 0, ctx->stream, A);
  hipLaunchKernelGGL(k_filter_expand, dim3(n_scans), dim3(FILT_XBLOCK), 0, ctx->stream, A);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
