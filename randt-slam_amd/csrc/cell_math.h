// fp32 NDT cell arithmetic, device side.  Every function here must produce bit-identical results to
// the reference's Eigen fp32 code as restated by the CPU oracle, so translation units that include
// this header are compiled with -ffp-contract=off (no FMA contraction) and never with fast-math.
// HIP's float '/' and sqrtf are correctly rounded by default (-fhip-fp32-correctly-rounded-divide-sqrt).
//
// Citations are relative to /root/reference/ros/ndt_radar_slam/.
#pragma once

#include "randt_internal.h"

#pragma clang fp contract(off)

namespace randt_dev {

// static_cast<unsigned int>(double) as x86-64 evaluates it (Map::coordinateToIndex, ndt_map.h:87-90).
__device__ __forceinline__ uint32_t trunc_to_u32(double v) {
  if (!(v > -9.0e18 && v < 9.0e18)) return 0u;
  return (uint32_t)(long long)v;
}

// static_cast<int>(float) (Grid::cluster, grid.cpp:11).
__device__ __forceinline__ int32_t trunc_to_i32(float v) {
  if (!(v > -2.0e9f && v < 2.0e9f)) return 0;
  return (int32_t)v;
}

// Map::coordinateToIndex + getIndex (ndt_map.h:87-90,181-184): unsigned 32-bit arithmetic.
__device__ __forceinline__ uint32_t coord_to_index(const MapView& m, float x, float y) {
  uint32_t mx = trunc_to_u32(((double)x - m.offset_x) / m.res);
  uint32_t my = trunc_to_u32(((double)y - m.offset_y) / m.res);
  return my * (uint32_t)m.size_x + mx;
}

// Sophus::SE2d::cast<float>() (SO2 ctor normalises) -> Eigen::Affine2f (ndt_matcher.cpp:208,
// local_fuser.cpp:175).  aff = {c, s, tx, ty}.
__device__ __forceinline__ void pose_to_affine_f(const double* pose4, float aff[4]) {
  float c = (float)pose4[0], s = (float)pose4[1];
  float len = sqrtf(c * c + s * s);
  aff[0] = c / len;
  aff[1] = s / len;
  aff[2] = (float)pose4[2];
  aff[3] = (float)pose4[3];
}

// Closed-form fp32 symmetric 2x2 eigen-decomposition + regularisation (ndt_cell.cpp:102-112).
__device__ __forceinline__ void cell_regularize(randt_cell& c) {
  float a = c.cov[0], b = c.cov[1], d = c.cov[3];
  float t = 0.5f * (a + d);
  float h = 0.5f * (a - d);
  float r = sqrtf(h * h + b * b);
  float l0 = t - r, l1 = t + r;
  float vx, vy;
  if (b == 0.0f) {
    if (a <= d) { vx = 0.0f; vy = 1.0f; } else { vx = 1.0f; vy = 0.0f; }
  } else {
    if (h >= 0.0f) { vx = h + r; vy = b; } else { vx = b; vy = r - h; }
    float nrm = sqrtf(vx * vx + vy * vy);
    vx = vx / nrm;
    vy = vy / nrm;
  }
  float V0 = vy, V1 = vx, V2 = -vx, V3 = vy;
  l0 = fmaxf(l0, 0.001f * l1);
  float det = V0 * V3 - V1 * V2;
  float invdet = 1.0f / det;
  float I00 = V3 * invdet, I01 = -V1 * invdet;
  float I10 = -V2 * invdet, I11 = V0 * invdet;
  float T00 = V0 * l0, T01 = V1 * l1;
  float T10 = V2 * l0, T11 = V3 * l1;
  c.cov[0] = T00 * I00 + T01 * I10;
  c.cov[1] = T00 * I01 + T01 * I11;
  c.cov[3] = T10 * I01 + T11 * I11;
  c.cov[5] = (float)((double)c.cov[5] + 0.000001);
}

// Cell::transformCell (ndt_cell.cpp:117-123) with R = blockdiag(R2, 1).
__device__ __forceinline__ void cell_transform(randt_cell& cl, const float aff[4]) {
  float c = aff[0], s = aff[1];
  float x = cl.mean[0], y = cl.mean[1];
  cl.mean[0] = (c * x + (-s) * y) + aff[2];
  cl.mean[1] = (s * x + c * y) + aff[3];
  float S[3][3] = {{cl.cov[0], cl.cov[1], cl.cov[2]}, {cl.cov[1], cl.cov[3], cl.cov[4]}, {cl.cov[2], cl.cov[4], cl.cov[5]}};
  float R[3][3] = {{c, -s, 0.f}, {s, c, 0.f}, {0.f, 0.f, 1.f}};
  float T[3][3], O[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[i][j] = (R[i][0] * S[0][j] + R[i][1] * S[1][j]) + R[i][2] * S[2][j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) O[i][j] = (T[i][0] * R[j][0] + T[i][1] * R[j][1]) + T[i][2] * R[j][2];
  cl.cov[0] = O[0][0];
  cl.cov[1] = O[0][1];
  cl.cov[2] = O[0][2];
  cl.cov[3] = O[1][1];
  cl.cov[4] = O[1][2];
  cl.cov[5] = O[2][2];
}

// Cell::operator+= (ndt_cell.h:133-142), integer division (n*m)/(n+m) included.
__device__ __forceinline__ void cell_merge(randt_cell& dst, const randt_cell& src) {
  uint32_t n = dst.n;
  unsigned long long m = src.n;
  float wn = (float)(uint32_t)(n - 1u);
  float wm = (float)(unsigned long long)(m - 1u);
  float wk = (float)(unsigned long long)(((unsigned long long)n * m) / ((unsigned long long)n + m));
  float d[3] = {dst.mean[0] - src.mean[0], dst.mean[1] - src.mean[1], dst.mean[2] - src.mean[2]};
  const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
  float nc[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) nc[e] = (wn * dst.cov[e] + wm * src.cov[e]) + wk * (d[ii[e]] * d[jj[e]]);
  float fn = (float)n, fm = (float)m, fnm = (float)((unsigned long long)n + m);
#pragma unroll
  for (int e = 0; e < 3; ++e) dst.mean[e] = ((dst.mean[e] * fn) + (src.mean[e] * fm)) / fnm;
  dst.n = (uint32_t)(n + m);
  float den = (float)(uint32_t)(dst.n - 1u);
#pragma unroll
  for (int e = 0; e < 6; ++e) dst.cov[e] = nc[e] / den;
  if (src.max_intensity > dst.max_intensity) dst.max_intensity = src.max_intensity;
}

// Cell::mahalanobisSquaredIntensity (ndt_cell.cpp:172-176): Eigen 3.3 cofactor inverse, fp32.
__device__ __forceinline__ float mahalanobis3f(const randt_cell& q, const randt_cell& f) {
  float S[3][3];
  S[0][0] = f.cov[0] + q.cov[0];
  S[0][1] = S[1][0] = f.cov[1] + q.cov[1];
  S[0][2] = S[2][0] = f.cov[2] + q.cov[2];
  S[1][1] = f.cov[3] + q.cov[3];
  S[1][2] = S[2][1] = f.cov[4] + q.cov[4];
  S[2][2] = f.cov[5] + q.cov[5];
  float mu[3] = {f.mean[0] - q.mean[0], f.mean[1] - q.mean[1], f.mean[2] - q.mean[2]};
#define RANDT_COF(i, j) (S[((i) + 1) % 3][((j) + 1) % 3] * S[((i) + 2) % 3][((j) + 2) % 3] - S[((i) + 1) % 3][((j) + 2) % 3] * S[((i) + 2) % 3][((j) + 1) % 3])
  float c0 = RANDT_COF(0, 0), c1 = RANDT_COF(1, 0), c2 = RANDT_COF(2, 0);
  float det = (c0 * S[0][0] + c1 * S[1][0]) + c2 * S[2][0];
  float invdet = 1.0f / det;
  float inv[3][3];
  inv[0][0] = c0 * invdet;
  inv[0][1] = c1 * invdet;
  inv[0][2] = c2 * invdet;
  inv[1][0] = RANDT_COF(0, 1) * invdet;
  inv[1][1] = RANDT_COF(1, 1) * invdet;
  inv[1][2] = RANDT_COF(2, 1) * invdet;
  inv[2][0] = RANDT_COF(0, 2) * invdet;
  inv[2][1] = RANDT_COF(1, 2) * invdet;
  inv[2][2] = RANDT_COF(2, 2) * invdet;
#undef RANDT_COF
  float row[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) row[j] = (mu[0] * inv[0][j] + mu[1] * inv[1][j]) + mu[2] * inv[2][j];
  return (row[0] * mu[0] + row[1] * mu[1]) + row[2] * mu[2];
}

// 48-byte record <-> three 16-byte vector accesses.
__device__ __forceinline__ randt_cell load_cell(const randt_cell* p) {
  const float4* q = reinterpret_cast<const float4*>(p);
  float4 a = q[0], b = q[1], c = q[2];
  randt_cell r;
  r.mean[0] = a.x; r.mean[1] = a.y; r.mean[2] = a.z; r.cov[0] = a.w;
  r.cov[1] = b.x; r.cov[2] = b.y; r.cov[3] = b.z; r.cov[4] = b.w;
  r.cov[5] = c.x; r.n = __float_as_uint(c.y); r.max_intensity = c.z; r.reserved = __float_as_uint(c.w);
  return r;
}

__device__ __forceinline__ void store_cell(randt_cell* p, const randt_cell& r) {
  float4* q = reinterpret_cast<float4*>(p);
  q[0] = make_float4(r.mean[0], r.mean[1], r.mean[2], r.cov[0]);
  q[1] = make_float4(r.cov[1], r.cov[2], r.cov[3], r.cov[4]);
  q[2] = make_float4(r.cov[5], __uint_as_float(r.n), r.max_intensity, __uint_as_float(r.reserved));
}

// Inclusive prefix sum over the 64 lanes of a wavefront in six DPP adds: four shifts inside each 16-lane row, then the
// row totals travel with row_bcast15 (rows 1 and 3 take lane 15 of the row before) and row_bcast31 (rows 2 and 3 take
// lane 31).  No LDS crossbar (ds_bpermute) round trips.
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// Minimum / maximum over the 64 lanes (result in every lane): four in-row DPP steps, then the four row results.
template <bool MAX>
__device__ __forceinline__ int wave_minmax(int v) {
  auto pick = [](int a, int b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  v = pick(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v = pick(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v = pick(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));  // row_half_mirror
  v = pick(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));  // row_mirror
  return pick(pick(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
              pick(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// Exclusive prefix sum of one int per thread over a 256-thread block; returns the exclusive value,
// *total receives the block sum.  scratch: >= 4 ints of LDS.  Contains two barriers.
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* scratch, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = wave_inclusive_scan(v);
  __syncthreads();  // protect scratch reuse across calls
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    int s = scratch[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

}  // namespace randt_dev
