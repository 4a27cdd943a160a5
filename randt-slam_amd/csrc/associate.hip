// Association kernel for gfx950 (compiled with -ffp-contract=off, see cell_math.h).
//
// Replaces the data-association half of Matcher::addNDTFactor (src/ndt_registration/ndt_matcher.cpp:200-215,
// 249-253) together with Map::getClosestCells / Map::getAdjacentIndizes
// (src/ndt_representation/ndt_map.cpp:101-175) and Cell::mahalanobisSquaredIntensity
// (src/ndt_representation/ndt_cell.cpp:172-176).
//
// One workgroup per (scan, submap) pair.  The submap's dense int32 index grid (40 KB for the
// 100x100 indoor map) is staged once into LDS with 16-byte coalesced loads; each wavefront then
// walks moving cells: the 64 lanes cover the (2R+1)^2 search window in the reference's x-major
// order, ballots count occupied slots per Chebyshev ring to find the first radius that holds >= k
// cells, candidates' 48-byte cell records are gathered from L2/HBM, the fp32 Mahalanobis (or
// Euclidean) distance is evaluated in the reference's operation order, and k rounds of a
// wave-level lexicographic arg-min on (distance, compact index) reproduce std::sort's order.
#include "cell_math.h"

using namespace randt_dev;

#define ASSOC_BLOCK 512
#define ASSOC_MAX_R 7            // window <= 15x15 = 225 slots = 4 lane passes
#define ASSOC_PASSES 4

namespace {

struct Cand {
  float dist;
  int32_t idx;
};

__device__ __forceinline__ bool cand_less(float da, int32_t ia, float db, int32_t ib) {
  // std::pair<double,size_t> ordering (ndt_map.cpp:122,147); idx < 0 = empty = +inf
  if (ia < 0) return false;
  if (ib < 0) return true;
  if (da < db) return true;
  if (db < da) return false;
  return ia < ib;
}

// duplicate test of Map::getAdjacentIndizes' std::find (ndt_map.cpp:169): slot (i, j) of the window
// of radius r repeats an EARLIER (x-offset-major) window entry iff some t >= 1 has
// i - t*size_x >= -r and j + t <= r.  Never true when size_x > 2r.
__device__ __forceinline__ bool window_dup(int i, int j, int r, int size_x) {
  for (int t = 1; i - t * size_x >= -r; ++t)
    if (j + t <= r) return true;
  return false;
}

template <bool STAGE_GRID>
__global__ __launch_bounds__(ASSOC_BLOCK) void k_associate(MapView fixed, const int32_t* __restrict__ fixed_idx,
                                                           MapView moving, int moving_first,
                                                           const double* __restrict__ guess4, int k, int metric_mahal,
                                                           int transform_full, int32_t* __restrict__ corr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* lgrid = reinterpret_cast<int32_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pair = blockIdx.x;
  const int fmap = fixed_idx ? fixed_idx[pair] : 0;
  const int mmap = moving_first + pair;
  const int32_t* ggrid = fixed.grid + (size_t)fmap * fixed.n_slots;
  const randt_cell* fcells = fixed.cells + (size_t)fmap * fixed.cap;
  const randt_cell* mcells = moving.cells + (size_t)mmap * moving.cap;
  int32_t* out = corr + (size_t)pair * moving.cap * k;
  const int M = moving.counts[mmap];
  const int n_slots = fixed.n_slots;

  if (STAGE_GRID) {
    if (((size_t)ggrid & 15) == 0) {
      const int n4 = n_slots >> 2;
      const int4* g4 = reinterpret_cast<const int4*>(ggrid);
      int4* l4 = reinterpret_cast<int4*>(lgrid);
      for (int i = tid; i < n4; i += ASSOC_BLOCK) l4[i] = g4[i];
      for (int i = (n4 << 2) + tid; i < n_slots; i += ASSOC_BLOCK) lgrid[i] = ggrid[i];
    } else {
      for (int i = tid; i < n_slots; i += ASSOC_BLOCK) lgrid[i] = ggrid[i];
    }
    __syncthreads();
  }
  const int32_t* grid = STAGE_GRID ? lgrid : ggrid;

  float aff[4];
  pose_to_affine_f(guess4 + 4 * (size_t)pair, aff);
  const int R = fixed.rmax - 1 > 0 ? fixed.rmax - 1 : 0;  // last radius the reference evaluates
  const int side = 2 * R + 1, nwin = side * side;

  for (int ci = wave; ci < M; ci += ASSOC_BLOCK / 64) {
    randt_cell q = load_cell(mcells + ci);
    if (transform_full) {
      cell_transform(q, aff);  // Cell::transformCell (ndt_matcher.cpp:207-209)
    } else {
      // initial_guess.cast<float>() * mean.xy (ndt_matcher.cpp:213,250)
      float x = q.mean[0], y = q.mean[1];
      q.mean[0] = (aff[0] * x - aff[1] * y) + aff[2];
      q.mean[1] = (aff[1] * x + aff[0] * y) + aff[3];
    }
    const uint32_t center = coord_to_index(fixed, q.mean[0], q.mean[1]);

    // window scan: per-ring counts of valid (A) and occupied (T) slots
    int32_t cidx[ASSOC_PASSES];
    int rho[ASSOC_PASSES], wi[ASSOC_PASSES], wj[ASSOC_PASSES];
    int T[ASSOC_MAX_R + 1], A[ASSOC_MAX_R + 1];
#pragma unroll
    for (int r = 0; r <= ASSOC_MAX_R; ++r) T[r] = A[r] = 0;
#pragma unroll
    for (int p = 0; p < ASSOC_PASSES; ++p) {
      const int w = p * 64 + lane;
      cidx[p] = -2;  // -2: not a valid window slot, -1: empty slot
      rho[p] = 1 << 20;
      wi[p] = wj[p] = 0;
      if (w < nwin) {
        const int i = w / side - R, j = w % side - R;  // i: x offset (outer), j: y offset (inner)
        const uint32_t ni = center + (uint32_t)i + (uint32_t)j * (uint32_t)fixed.size_x;
        if (ni < (uint32_t)n_slots) {
          cidx[p] = grid[ni];
          if (cidx[p] < -1) cidx[p] = -1;
          rho[p] = (i < 0 ? -i : i) > (j < 0 ? -j : j) ? (i < 0 ? -i : i) : (j < 0 ? -j : j);
          wi[p] = i;
          wj[p] = j;
        }
      }
#pragma unroll
      for (int r = 0; r <= ASSOC_MAX_R; ++r) {
        if (r <= R) {
          const bool uniq = cidx[p] >= -1 && rho[p] <= r && !window_dup(wi[p], wj[p], r, fixed.size_x);
          A[r] += __popcll(__ballot(uniq));
          T[r] += __popcll(__ballot(uniq && cidx[p] >= 0));
        }
      }
    }
    // while (targets.size() < n && adjacent.size() < n_cells_) {...; r++; if (r >= rmax) break;}
    int rstar = -1;
    {
      int nt = 0, nadj = 0, radius = 0;
      while (nt < k && nadj < n_slots) {
        nt = T[radius];
        nadj = A[radius];
        rstar = radius;
        ++radius;
        if (radius >= fixed.rmax) break;
      }
    }

    // candidates of the final window: distance in fp32, compared as (double)dist then index
    Cand cand[ASSOC_PASSES];
#pragma unroll
    for (int p = 0; p < ASSOC_PASSES; ++p) {
      cand[p].idx = -1;
      cand[p].dist = 0.f;
      const bool in = rstar >= 0 && cidx[p] >= 0 && rho[p] <= rstar && !window_dup(wi[p], wj[p], rstar, fixed.size_x);
      if (in) {
        const int32_t fi = cidx[p] < fixed.cap ? cidx[p] : fixed.cap - 1;
        randt_cell f = load_cell(fcells + fi);
        float d;
        if (metric_mahal) {
          d = mahalanobis3f(q, f);
        } else {
          const float dx = q.mean[0] - f.mean[0], dy = q.mean[1] - f.mean[1];
          d = sqrtf(dx * dx + dy * dy);
        }
        cand[p].dist = d;
        cand[p].idx = cidx[p];
      }
    }
    for (int kk = 0; kk < k; ++kk) {
      // lane-local best
      float bd = cand[0].dist;
      int32_t bi = cand[0].idx;
#pragma unroll
      for (int p = 1; p < ASSOC_PASSES; ++p)
        if (cand_less(cand[p].dist, cand[p].idx, bd, bi)) {
          bd = cand[p].dist;
          bi = cand[p].idx;
        }
      // wave arg-min
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off, 64);
        const int32_t oi = __shfl_xor(bi, off, 64);
        if (cand_less(od, oi, bd, bi)) {
          bd = od;
          bi = oi;
        }
      }
      if (lane == 0) out[(size_t)ci * k + kk] = bi;
#pragma unroll
      for (int p = 0; p < ASSOC_PASSES; ++p)
        if (bi >= 0 && cand[p].idx == bi) cand[p].idx = -1;
    }
  }
}

}  // namespace

int launch_associate(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving,
                     int moving_first, int n_pairs, const double* d_guess4, int k, int lookup_mahalanobis,
                     int use_intensity, int32_t* d_corr) {
  if (n_pairs <= 0) return RANDT_OK;
  if (!fixed.grid) return randt_set_error(ctx, RANDT_ERR_INVALID, "fixed maps need an index grid", hipSuccess);
  if (fixed.rmax - 1 > ASSOC_MAX_R)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "max_neighbour_dist/resolution > 8 not supported by the association kernel", hipSuccess);
  const int full = (use_intensity && lookup_mahalanobis) ? 1 : 0;
  size_t lds = (size_t)fixed.n_slots * 4;
  if (lds + 1024 <= (size_t)ctx->lds_limit / 2) {
    RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_associate<true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_associate<true>, dim3(n_pairs), dim3(ASSOC_BLOCK), lds, ctx->stream, fixed, d_fixed_idx,
                       moving, moving_first, d_guess4, k, full, full, d_corr);
  } else {
    hipLaunchKernelGGL(k_associate<false>, dim3(n_pairs), dim3(ASSOC_BLOCK), 0, ctx->stream, fixed, d_fixed_idx,
                       moving, moving_first, d_guess4, k, full, full, d_corr);
  }
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
