// Cauchy-Schwarz divergence between two NDT maps for gfx950 -- SURVEY row f-2, the loop-closure gate
// right after a pair registration (compiled with -ffp-contract=off: the fp32 3x3 inverse / determinant
// follow Eigen's operation order like the CPU path).
//
// Replaces Map::calculateCSDivergence (src/ndt_representation/ndt_map.cpp:42-99), called at
// local_fuser.cpp:338-339 after Map::transformMap of the moving map.  Three pairwise Gaussian-overlap
// sums: interaction (Nf x Nm), fixed self term (Nf^2/2), moving self term (Nm^2/2).  The reference
// leaves its accumulators uninitialised (:43-46); they start at 0 here.
//
//   k_cs_self   self term of each map of a batch, tiled: workgroup (tile, map) owns 256 outer cells,
//               inner cells stream through LDS in 256-cell tiles (48-byte records, 3 x 16-B loads);
//               the fixed map's term does not depend on the pair, so it is computed once per submap.
//   k_cs_pair   one workgroup per (submap, scan) pair: the scan's cells are transformed (fp32, the
//               reference's operation order) into LDS, every thread walks fixed cells against them;
//               fp64 partial sums, fixed-order block reduction (deterministic).
#include "cell_math.h"
#include "solve_math.h"

using namespace randt_dev;

#define CS_BLOCK 256
#define CS_PI 3.14159265358979323846

namespace {

__device__ __forceinline__ void full3(const randt_cell& c, float S[3][3]) {
  S[0][0] = c.cov[0]; S[0][1] = S[1][0] = c.cov[1]; S[0][2] = S[2][0] = c.cov[2];
  S[1][1] = c.cov[3]; S[1][2] = S[2][1] = c.cov[4]; S[2][2] = c.cov[5];
}
// Eigen bruteforce_det3_helper order
__device__ __forceinline__ float det3f(const float m[3][3]) {
#define RANDT_H3(a, b, c) (m[0][a] * (m[1][b] * m[2][c] - m[1][c] * m[2][b]))
  return RANDT_H3(0, 1, 2) - RANDT_H3(1, 0, 2) + RANDT_H3(2, 0, 1);
#undef RANDT_H3
}
__device__ __forceinline__ void inv3f(const float S[3][3], float inv[3][3]) {
#define RANDT_COF(i, j) (S[((i) + 1) % 3][((j) + 1) % 3] * S[((i) + 2) % 3][((j) + 2) % 3] - S[((i) + 1) % 3][((j) + 2) % 3] * S[((i) + 2) % 3][((j) + 1) % 3])
  const float c0 = RANDT_COF(0, 0), c1 = RANDT_COF(1, 0), c2 = RANDT_COF(2, 0);
  const float det = (c0 * S[0][0] + c1 * S[1][0]) + c2 * S[2][0];
  const float invdet = 1.0f / det;
  inv[0][0] = c0 * invdet; inv[0][1] = c1 * invdet; inv[0][2] = c2 * invdet;
  inv[1][0] = RANDT_COF(0, 1) * invdet; inv[1][1] = RANDT_COF(1, 1) * invdet; inv[1][2] = RANDT_COF(2, 1) * invdet;
  inv[2][0] = RANDT_COF(0, 2) * invdet; inv[2][1] = RANDT_COF(1, 2) * invdet; inv[2][2] = RANDT_COF(2, 2) * invdet;
#undef RANDT_COF
}
// (0.5 / sqrt(pi^2 det(Sf+Sq))) exp(-0.5 d^T (Sf+Sq)^-1 d)   (ndt_map.cpp:60-64)
__device__ __forceinline__ double cs_pair(const randt_cell& f, const randt_cell& q) {
  float Sf[3][3], Sq[3][3], M[3][3], inv[3][3];
  full3(f, Sf);
  full3(q, Sq);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[i][j] = Sf[i][j] + Sq[i][j];
  const float d[3] = {f.mean[0] - q.mean[0], f.mean[1] - q.mean[1], f.mean[2] - q.mean[2]};
  inv3f(M, inv);
  float row[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) row[j] = (d[0] * inv[0][j] + d[1] * inv[1][j]) + d[2] * inv[2][j];
  const double e = (double)((row[0] * d[0] + row[1] * d[1]) + row[2] * d[2]);
  // 0.5 / sqrt(pi^2 det): reciprocal square root to ~1 ulp (hardware seed + two Newton steps, solve_math.h) instead of a
  // correctly rounded fp64 sqrt and division (~50 instructions of the ~250 of an evaluation)
  return (0.5 * randt_solve::fast_rsqrt(CS_PI * CS_PI * (double)det3f(M))) * exp(-0.5 * e);
}
__device__ __forceinline__ bool cell_valid(const randt_cell& c, double* self) {
  float S[3][3], inv[3][3];
  full3(c, S);
  if ((double)det3f(S) < 0.00001) return false;   // :55,68,83
  inv3f(S, inv);
  *self = (double)sqrtf(det3f(inv)) / (2 * CS_PI);  // :71,86
  return true;
}
__device__ __forceinline__ void lds_store(float* l, const randt_cell& c) {
  l[0] = c.mean[0]; l[1] = c.mean[1]; l[2] = c.mean[2];
#pragma unroll
  for (int e = 0; e < 6; ++e) l[3 + e] = c.cov[e];
}
__device__ __forceinline__ randt_cell lds_load(const float* l) {
  randt_cell c;
  c.mean[0] = l[0]; c.mean[1] = l[1]; c.mean[2] = l[2];
#pragma unroll
  for (int e = 0; e < 6; ++e) c.cov[e] = l[3 + e];
  c.n = 0; c.max_intensity = 0.f; c.reserved = 0;
  return c;
}
// fixed-order block sum of one double per thread (NW wavefronts)
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += red[w];
  return s;
}

// partial[map][tile] = sum over the CS_SELF_OUTER outer cells f of the tile of [ self(f) + sum_{q < f} 2 pair(f, q) ].
// A thread's inner loop is a chain of dependent evaluations (~1000 issue cycles each: fp64 exp, sqrt, divide), so the tile is
// SMALL and the inner cells of an outer cell are dealt to CS_BLOCK / CS_SELF_OUTER threads: 16 outer cells x 16 inner lanes,
// a chain of N / 16 instead of the N of one-thread-per-outer-cell (round 3: 107 us for eight 300-cell submaps, 16 workgroups busy).
#define CS_SELF_OUTER RANDT_CS_SELF_OUTER
#define CS_SELF_SPLIT (CS_BLOCK / CS_SELF_OUTER)
__global__ __launch_bounds__(CS_BLOCK) void k_cs_self(MapView m, int first, int max_tiles, double* __restrict__ partial) {
  __shared__ float tile[CS_BLOCK * 9];
  __shared__ double red[CS_BLOCK / 64];
  const int map = first + blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  int N = m.counts[map];
  N = N > m.cap ? m.cap : N;
  if (t * CS_SELF_OUTER >= N) {  // uniform: nothing in this tile
    if (tid == 0) partial[(size_t)blockIdx.y * max_tiles + t] = 0.0;
    return;
  }
  const randt_cell* cells = m.cells + (size_t)map * m.cap;
  const int o = tid & (CS_SELF_OUTER - 1), sp = tid / CS_SELF_OUTER;
  const int f = t * CS_SELF_OUTER + o;
  double acc = 0.0;
  bool valid = false;
  randt_cell cf;
  if (f < N) {
    cf = load_cell(cells + f);
    double s;
    valid = cell_valid(cf, &s);
    if (valid && sp == 0) acc = s;
  }
  const int q_end = (t + 1) * CS_SELF_OUTER < N ? (t + 1) * CS_SELF_OUTER : N;  // inner cells needed by this tile
  for (int q0 = 0; q0 < q_end; q0 += CS_BLOCK) {
    __syncthreads();
    if (q0 + tid < N) lds_store(tile + tid * 9, load_cell(cells + q0 + tid));
    __syncthreads();
    if (valid) {
      const int lim = (f - q0) < CS_BLOCK ? (f - q0) : CS_BLOCK;   // q < f
      for (int j = sp; j < lim; j += CS_SELF_SPLIT) acc += 2 * cs_pair(cf, lds_load(tile + j * 9));
    }
  }
  const double s = block_sum<CS_BLOCK / 64>(acc, red);
  if (tid == 0) partial[(size_t)blockIdx.y * max_tiles + t] = s;
}

// One workgroup per (submap, scan) pair.  An evaluation costs ~1000 issue cycles, so what counts is that every lane has one to do:
// the (fixed cell, moving cell) pairs of a 256-cell tile of the fixed map are walked as ONE list by all threads (round 3 gave a
// thread a fixed cell and let it walk the moving cells: with 300 fixed cells the second round had 44 of 256 lanes busy, and the
// moving self term 75 -- 58 % lane efficiency, 150 us per 512 pairs), the moving map's triangle likewise.
#ifndef CS_PBLOCK
#define CS_PBLOCK 512
#endif
__global__ __launch_bounds__(CS_PBLOCK) void k_cs_pair(MapView fixed, const int32_t* __restrict__ fixed_idx, MapView moving,
                                                       int moving_first, const double* __restrict__ pose4,
                                                       const double* __restrict__ fixed_partial, int fixed_first, int max_tiles,
                                                       double* __restrict__ out, double* __restrict__ terms) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* mv = reinterpret_cast<float*>(smem);                    // [cap][9] transformed moving cells
  int* mvalid = reinterpret_cast<int*>(mv + (size_t)moving.cap * 9);  // [cap] det(S) >= 1e-5 (:83)
  __shared__ float fx[256 * 9];
  __shared__ int fvalid[256];
  __shared__ double red[CS_PBLOCK / 64];
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int fmap = fixed_idx ? fixed_idx[pair] : 0;
  const int mmap = moving_first + pair;
  int Nf = fixed.counts[fmap], M = moving.counts[mmap];
  Nf = Nf > fixed.cap ? fixed.cap : Nf;
  M = M > moving.cap ? moving.cap : M;
  const randt_cell* fcells = fixed.cells + (size_t)fmap * fixed.cap;
  const randt_cell* mcells = moving.cells + (size_t)mmap * moving.cap;
  float aff[4] = {1.f, 0.f, 0.f, 0.f};
  if (pose4) pose_to_affine_f(pose4 + 4 * (size_t)pair, aff);
  double mself = 0.0;
  for (int i = tid; i < M; i += CS_PBLOCK) {
    randt_cell c = load_cell(mcells + i);
    if (pose4) cell_transform(c, aff);  // m_loop_map.transformMap(trans), local_fuser.cpp:338
    lds_store(mv + i * 9, c);
    double s;
    const bool v = cell_valid(c, &s);
    mvalid[i] = v ? 1 : 0;
    if (v) mself += s;  // the diagonal of the moving self term (:86)
  }
  // interaction term: valid fixed cells x all moving cells (:50-65), a tile of the fixed map at a time
  double inter = 0.0;
  for (int f0 = 0; f0 < Nf; f0 += 256) {
    __syncthreads();  // (first round: the moving cells are staged; later: the previous tile is consumed)
    const int nt = (Nf - f0) < 256 ? (Nf - f0) : 256;
    if (tid < nt) {
      const randt_cell cf = load_cell(fcells + f0 + tid);
      double s;
      fvalid[tid] = cell_valid(cf, &s) ? 1 : 0;
      lds_store(fx + tid * 9, cf);
    }
    __syncthreads();
    const int items = nt * M;
    for (int i = tid; i < items; i += CS_PBLOCK) {
      const int fl = i / M, j = i - fl * M;
      if (fvalid[fl]) inter += cs_pair(lds_load(fx + fl * 9), lds_load(mv + j * 9));
    }
  }
  if (Nf <= 0) __syncthreads();
  // moving self term, off-diagonal (:81-95): the pairs (f, j < f) of the triangle as one list; item i = f (f - 1) / 2 + j
  const int T = M * (M - 1) / 2;
  for (int i = tid; i < T; i += CS_PBLOCK) {
    int f = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)i)) * 0.5f);
    while (f * (f - 1) / 2 > i) --f;
    while ((f + 1) * f / 2 <= i) ++f;
    const int j = i - f * (f - 1) / 2;
    if (mvalid[f]) mself += 2 * cs_pair(lds_load(mv + f * 9), lds_load(mv + j * 9));
  }
  inter = block_sum<CS_PBLOCK / 64>(inter, red);
  mself = block_sum<CS_PBLOCK / 64>(mself, red);
  if (tid == 0) {
    double fself = 0.0;
    const int tiles = (Nf + CS_SELF_OUTER - 1) / CS_SELF_OUTER;
    for (int t = 0; t < tiles; ++t) fself += fixed_partial[(size_t)(fmap - fixed_first) * max_tiles + t];
    out[pair] = -log(inter) + 0.5 * log(fself) + 0.5 * log(mself);  // :97
    if (terms) {
      terms[3 * (size_t)pair + 0] = inter;
      terms[3 * (size_t)pair + 1] = fself;
      terms[3 * (size_t)pair + 2] = mself;
    }
  }
}

}  // namespace

int launch_cs_divergence(randt_ctx* ctx, const MapView& fixed, int fixed_first, int fixed_count, const int32_t* d_fixed_idx,
                         const MapView& moving, int moving_first, int n_pairs, const double* d_pose4, double* d_partial,
                         double* d_out, double* d_terms) {
  randt_note_enqueue(ctx);  // (RANDT_SOLVE_AUTO of the process's other contexts: this one has work in flight)
  const int max_tiles = (fixed.cap + CS_SELF_OUTER - 1) / CS_SELF_OUTER;
  hipLaunchKernelGGL(k_cs_self, dim3(max_tiles, fixed_count), dim3(CS_BLOCK), 0, ctx->stream, fixed, fixed_first, max_tiles, d_partial);
  const size_t lds = (size_t)moving.cap * (9 * 4 + 4);  // transformed moving cells + their validity flags
  if (lds + 12 * 1024 > (size_t)ctx->lds_limit)  // (+ the kernel's static tile of fixed cells: 10.3 KB)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "moving-map capacity too large for the CS-divergence kernel", hipSuccess);
  RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_cs_pair), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_cs_pair, dim3(n_pairs), dim3(CS_PBLOCK), lds, ctx->stream, fixed, d_fixed_idx, moving, moving_first, d_pose4,
                     d_partial, fixed_first, max_tiles, d_out, d_terms);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
