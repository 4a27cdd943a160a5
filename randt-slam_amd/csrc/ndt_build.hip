// NDT construction kernels for gfx950 (compiled with -ffp-contract=off, see cell_math.h):
//   k_ndt_build     one workgroup per radar scan: voxel key -> stable LDS counting sort -> per-cluster
//                   fp32 mean / covariance / regularisation -> compact cell table + index grid
// (Map::transformMap / mergeMapCell / insertCell live in mapops.hip.)
//
// Replaces (paths relative to /root/reference/ros/ndt_radar_slam/):
//   src/radar_preprocessing/grid.cpp:7-14                      Grid::cluster
//   src/radar_preprocessing/radar_preprocessor.cpp:151-169     ClusterGenerator::labelClouds
//   src/ndt_representation/ndt_map.cpp:238-245                 Map::insertCluster
//   src/ndt_representation/ndt_cell.cpp:25-114                 Cell::addPointCloud / updateCell
//
// Data layout: points are read once from HBM as 16-byte (stride 4) or strided records, kept in registers while
// they are sorted (scans <= 2048 points) and scattered as label-sorted SoA x/y/intensity into LDS; per-label
// bins and cluster bounds live in LDS as well, so the only HBM traffic is N*16 B in, M*48 B + grid out.  fp32
// sums run in the reference's sequential point order (one lane per accumulator chain, eight lanes per
// cluster) so the cell statistics are bit-identical to the CPU path.
#include <vector>

#include "cell_math.h"

using namespace randt_dev;

#define BUILD_BLOCK 256
#ifndef RANDT_CHAIN_FENCE
#define RANDT_CHAIN_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#ifdef RANDT_TIMING
__device__ long long g_randt_timing[32];
#define RANDT_TICK(slot)                                                              \
  do {                                                                                \
    __syncthreads();                                                                  \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_randt_timing[slot] = wall_clock64();   \
  } while (0)
extern "C" int randt_debug_timing(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_randt_timing), sizeof(long long) * 32);
}
#else
#define RANDT_TICK(slot) do {} while (0)
#endif

namespace {

__device__ __forceinline__ int32_t point_label(float x, float y, int row_size, float resolution) {
  // Grid::cluster (grid.cpp:7-14): label = int(x/res) + row * int(y/res), C truncation toward zero
  return trunc_to_i32(x / resolution) + row_size * trunc_to_i32(y / resolution);
}

// Per-cluster cell statistics with one lane per cluster, strictly in cluster order: the reference
// arithmetic spelled out once.  Used for the rare case where a cluster mean falls outside the map
// (the compact cell index of every later cluster then shifts), after the fast path below.
__device__ void cluster_stats_sequential(const float* sx, const float* sy, const float* si, const uint16_t* cstart, int nc,
                                         const MapView& out, randt_cell* cells, int32_t* grid, int* scratch, int* n_cells_out) {
  const int tid = threadIdx.x;
  int n_cells = 0;
  for (int c0 = 0; c0 < nc; c0 += BUILD_BLOCK) {
    const int c = c0 + tid;
    randt_cell cell;
    bool accept = false;
    uint32_t slot = 0;
    if (c < nc) {
      const int s = cstart[c], e = cstart[c + 1];
      const int k = e - s;
      if ((long long)k > (long long)out.min_points) {
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, maxi = 0.f;
        for (int j = s; j < e; ++j) {
          const float in = si[j];
          m0 += sx[j];
          m1 += sy[j];
          m2 += in;
          maxi = in > maxi ? in : maxi;
        }
        const float nf = (float)(uint32_t)k;
        m0 = m0 / nf;
        m1 = m1 / nf;
        m2 = m2 / nf;
        float c00 = 0.f, c11 = 0.f, c22 = 0.f, c01 = 0.f, c02 = 0.f, c12 = 0.f;
        for (int j = s; j < e; ++j) {
          const float d0 = sx[j] - m0, d1 = sy[j] - m1, d2 = si[j] - m2;
          c00 += (d0 * d0);
          c11 += (d1 * d1);
          c22 += (d2 * d2);
          c01 += (d0 * d1);
          c02 += (d0 * d2);
          c12 += (d1 * d2);
        }
        cell.mean[0] = m0;
        cell.mean[1] = m1;
        cell.mean[2] = m2;
        cell.cov[0] = c00 / nf;
        cell.cov[1] = c01 / nf;
        cell.cov[2] = c02 / nf;
        cell.cov[3] = c11 / nf;
        cell.cov[4] = c12 / nf;
        cell.cov[5] = c22 / nf;
        cell.n = (uint32_t)k;
        cell.max_intensity = maxi;
        cell.reserved = 0;
        cell_regularize(cell);
        slot = coord_to_index(out, cell.mean[0], cell.mean[1]);
        accept = slot < (uint32_t)out.n_slots;  // reference: vector::at throws otherwise
      }
    }
    int tot;
    int idx = n_cells + block_exclusive_scan_256(accept ? 1 : 0, scratch, &tot);
    if (accept && idx < out.cap) {
      store_cell(cells + idx, cell);
      // later cluster overwrites the slot, both cells stay in grid_ (quirk A.7-5)
      if (grid) atomicMax(&grid[slot], idx);
    }
    n_cells += tot;
  }
  *n_cells_out = n_cells;
}

// One workgroup per scan.  labelClouds' order (ascending label, input order inside a label) is
// produced by a STABLE counting sort: every wavefront owns a contiguous quarter of the points and
// keeps them in registers (PPT per lane); a 64-bit LDS word per label bin packs the four per-wave
// counts (4 x u16).  While counting, the lanes of a 64-point step that share a label find each
// other with one ballot per label bit, so every point learns its rank inside (label, wave) right
// there; a block scan turns the counts into per-(bin, wave) start positions and the placement is
// then a plain scatter into label-sorted SoA arrays.  Label ranges that do not fit the LDS bins
// (points far outside max_range) fall back to an O(N^2/256) rank-by-counting pass.
//
// Cell statistics: the fp32 sums must run in the reference's sequential point order to be
// bit-identical, so a cluster cannot be split over points -- but its ten accumulator chains
// (3 sums + max, then 6 covariance sums) are independent: a group of 8 lanes owns a cluster and
// each lane walks ONE chain.  Clusters are handed to the 32 groups in descending size so that a
// round's wavefronts finish together.
// Dynamic LDS: sx | sy | si [npad] | cstart[npad+2] | aux (bins u64[nb_cap] / labels / order + blist of the clusters that become cells) | scratch
// REG: scans of <= 2048 points keep their points and labels in registers (8 per lane, loops fully unrolled);
// larger scans keep the per-point word in LDS and re-read the points (L2-hot) with rolled loops.
// TP: see k_associate -- sharing the chip with other batches' solves, the kernel is held to 64 registers (eight wavefronts per
// SIMD; no spills, but the chain loops lose their second buffer of loads in flight: 30 -> 36 us for a lone batch, which
// therefore keeps its 88).
#ifndef RANDT_BUILD_TP_WPE
#define RANDT_BUILD_TP_WPE 8
#endif
template <bool REG, bool TP>
__global__ __launch_bounds__(BUILD_BLOCK) __attribute__((amdgpu_waves_per_eu(TP ? RANDT_BUILD_TP_WPE : 1, TP ? RANDT_BUILD_TP_WPE : 8))) void k_ndt_build(const float* __restrict__ pts, int pitch,
                                                           const int32_t* __restrict__ n_pts_arr, int stride,
                                                           int ioff, int row_size, float resolution, MapView out,
                                                           int first_map, int npad, int nb_cap, int aux_bytes, int bigcap,
                                                           int32_t* __restrict__ fallback_ws, int lane_ordered_atomics,
                                                           int32_t* misrank_word, int32_t* misrank_count) {
  constexpr int PPT = 8;
  constexpr bool KEEP = REG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // latency-bound chains (LDS atomics, the serial fp32 sums) that hold 30 KB of LDS: raised issue priority, so that solve
  // wavefronts sharing the SIMD do not stretch them (see k_associate)
  __builtin_amdgcn_s_setprio(RANDT_LATENCY_KERNEL_PRIO);
  // points in labelClouds order; the three arrays are shifted by 16 banks against each other because the
  // lanes of a cluster group read x[j], y[j] and i[j] in the same instruction.
  // REG  : [ sx | sy | si  ==  bins ] | cstart u16 | order u16 | blist u16 | scratch      (30.5 KB at N = 2000: 5 per CU)
  //        the label bins are dead once every point knows its final position (kept in registers), so the sorted
  //        points are scattered over them;
  // !REG : sx | sy | si | cstart u16 | bins | scratch | per-point words (order / blist alias them after the placement).
  float* sx = reinterpret_cast<float*>(smem);
  float* sy = sx + npad + 16;
  float* si = sy + npad + 16;
  char* after_pts = reinterpret_cast<char*>(si + npad + 16);
  const int cstart_bytes = ((npad + 2) * 2 + 15) & ~15;
  uint16_t* cstart;
  unsigned long long* bins;
  int* scratch;
  int32_t* plab = nullptr;
  // blist[i]: the cluster that becomes compact cell i (clusters above min_points, in label order; bit 15: its mean fell outside
  // the map); order[j]: those i in hand-out order.  At most n / (min_points + 1) clusters qualify (bigcap), so these lists take
  // 1.4 KB at N = 2000 instead of two per-cluster arrays of 8 KB: 30.5 KB per workgroup, FIVE workgroups per CU instead of four.
  uint16_t *order, *blist;
  if (REG) {
    bins = reinterpret_cast<unsigned long long*>(smem);
    char* p = smem + aux_bytes;  // aux_bytes = max(points, bins)
    cstart = reinterpret_cast<uint16_t*>(p);
    order = reinterpret_cast<uint16_t*>(p + cstart_bytes);
    blist = order + bigcap;
    scratch = reinterpret_cast<int*>(blist + bigcap);  // bigcap is a multiple of 8: 16-byte aligned
  } else {
    cstart = reinterpret_cast<uint16_t*>(after_pts);
    bins = reinterpret_cast<unsigned long long*>(after_pts + cstart_bytes);
    scratch = reinterpret_cast<int*>(reinterpret_cast<char*>(bins) + aux_bytes);  // [64]
    plab = scratch + 64;  // per-point word [npad] (label, later bin | rank)
    order = reinterpret_cast<uint16_t*>(plab);
    blist = order + bigcap;
  }
  // fallback only: unsorted / sorted labels in global memory (rare path, 2 x npad words per scan)
  int32_t* lab = fallback_ws + (size_t)blockIdx.x * 2 * npad;
  int32_t* slab = lab + npad;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scan = blockIdx.x;
  const int map = first_map + scan;
  int n = n_pts_arr ? n_pts_arr[scan] : pitch;
  n = n < 0 ? 0 : (n > pitch ? pitch : n);
  const float* sp = pts + (size_t)scan * pitch * stride;
  int32_t* grid = out.grid ? out.grid + (size_t)map * out.n_slots : nullptr;
  randt_cell* cells = out.cells + (size_t)map * out.cap;

  RANDT_TICK(0);
  // Map::initialize: index grid = -1 (ndt_map.cpp:13-16)
  if (grid) {
    int4* g4 = reinterpret_cast<int4*>(grid);
    const int n4 = out.n_slots >> 2;
    if (((size_t)grid & 15) == 0) {
      for (int i = tid; i < n4; i += BUILD_BLOCK) g4[i] = make_int4(-1, -1, -1, -1);
      for (int i = (n4 << 2) + tid; i < out.n_slots; i += BUILD_BLOCK) grid[i] = -1;
    } else {
      for (int i = tid; i < out.n_slots; i += BUILD_BLOCK) grid[i] = -1;
    }
  }

  RANDT_TICK(1);
  // ---- load points (16 B / lane coalesced for packed xyzI), labels, label range.  Wave w owns the
  // contiguous quarter [w*q, (w+1)*q) of the input; lane l holds points w*q + 64*j + l.
  const bool vec4 = (stride == 4) && (((size_t)sp & 15) == 0);
  int lmin = 0x7fffffff, lmax = (int)0x80000000;
#define RANDT_FETCH_POINT(i, x, y, in)                                             \
  do {                                                                             \
    if (vec4) {                                                                    \
      const float4 p_ = reinterpret_cast<const float4*>(sp)[i];                    \
      x = p_.x;                                                                    \
      y = p_.y;                                                                    \
      in = ioff == 3 ? p_.w : (ioff == 2 ? p_.z : (ioff == 1 ? p_.y : p_.x));      \
    } else {                                                                       \
      const float* p_ = sp + (size_t)(i) * stride;                                 \
      x = p_[0];                                                                   \
      y = p_[1];                                                                   \
      in = p_[ioff];                                                               \
    }                                                                              \
  } while (0)
  const int q = (n + 3) >> 2;
  const int w_beg = wave * q, w_end = (w_beg + q) < n ? (w_beg + q) : n;
  float px[KEEP ? PPT : 1], py[KEEP ? PPT : 1], pin[KEEP ? PPT : 1];
  int32_t pl[REG ? PPT : 1];  // label, later: rank inside (label, wave)
  const int nsteps = REG ? PPT : (q + 63) >> 6;
#define RANDT_PL(j, i) (*(REG ? &pl[REG ? (j) : 0] : &plab[i]))
#pragma unroll 8
  for (int j = 0; j < nsteps; ++j) {
    const int i = w_beg + 64 * j + lane;
    if (REG) pl[REG ? j : 0] = 0;
    if (i < w_end) {
      float x, y, in;
      RANDT_FETCH_POINT(i, x, y, in);
      if (KEEP) {
        px[j] = x;
        py[j] = y;
        pin[j] = in;
      }
      const int32_t l = point_label(x, y, row_size, resolution);
      RANDT_PL(j, i) = l;
      lmin = l < lmin ? l : lmin;
      lmax = l > lmax ? l : lmax;
    }
  }
  lmin = wave_minmax<false>(lmin);
  lmax = wave_minmax<true>(lmax);
  if (lane == 0) {
    scratch[8 + wave] = lmin;
    scratch[12 + wave] = lmax;
  }
  if (tid == 0) scratch[20] = 0;  // "the atomic ranking failed its order check" (set in the count phase below)
  __syncthreads();
  lmin = scratch[8];
  lmax = scratch[12];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    lmin = scratch[8 + w] < lmin ? scratch[8 + w] : lmin;
    lmax = scratch[12 + w] > lmax ? scratch[12 + w] : lmax;
  }
  RANDT_TICK(2);
  const long long range = n > 0 ? (long long)lmax - (long long)lmin + 1 : 0;
  const bool fast = range <= (long long)nb_cap;
  int nc = 0;  // number of clusters (distinct labels)

  if (fast) {
    const int nb = (int)range;
    for (int b = tid; b < nb; b += BUILD_BLOCK) bins[b] = 0ull;
    __syncthreads();
    // ---- count: per-(bin, wave) totals, and for every point its rank inside (bin, wave)
    const int nbits = nb > 1 ? 32 - __clz(nb - 1) : 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int sh = 16 * wave;
    // rank by one ballot per label bit: assumes nothing about the hardware.  from_packed: the per-point word already holds
    // bin | rank << 16 (redo behind a failed atomic ranking), otherwise the raw label
    auto rank_by_ballots = [&](const bool from_packed) {
#pragma unroll 8
      for (int j = 0; j < nsteps; ++j) {
        const int i = w_beg + 64 * j + lane;
        const bool valid = i < w_end;
        const int b = valid ? (from_packed ? (RANDT_PL(j, i) & 0xffff) : RANDT_PL(j, i) - lmin) : 0;
        unsigned long long mask = __ballot(valid);  // lanes of this step with my label
        if (mask == 0ull) break;                    // wave-uniform: this quarter is exhausted
        for (int bit = 0; bit < nbits; ++bit) {
          const bool one = (b >> bit) & 1;
          const unsigned long long m = __ballot(valid && one);
          mask &= one ? m : ~m;
        }
        int field = 0;
        const int leader = __ffsll((long long)mask) - 1;
        if (valid && lane == leader) {
          const unsigned long long old = atomicAdd(&bins[b], (unsigned long long)__popcll(mask) << sh);
          field = (int)((old >> sh) & 0xffff);
        }
        field = __shfl(field, valid ? leader : lane, 64);
        if (valid) RANDT_PL(j, i) = b | ((field + __popcll(mask & lt)) << 16);  // bin (<= 16 bits, nb_cap < 65536) | rank
      }
    };
    if (lane_ordered_atomics) {
      // The LDS serves the lanes of one atomic instruction that hit the same address in ascending lane order: the value a
      // lane gets back IS its rank among the lanes of the step with its label plus what earlier steps of this wavefront
      // added -- one returning atomic per point instead of a ballot per label bit; a wavefront's LDS operations execute in
      // program order, so the steps stay ordered.  That serving order is observed, not documented, so it is CHECKED: at
      // context creation under the real kernel's conditions (api.hip, k_lds_atomic_order_probe: four wavefronts on shared
      // 64-bit bins) and here in every step of every launch on a sample -- the lanes that share the first active lane's bin
      // must hold consecutive ranks in lane order (two ballots and a popcount).  A violation makes the WHOLE workgroup rank
      // again with the ballots below (the sums stay bit-exact) and is reported to the host, which stops using the atomic
      // ranking on this context (launch_ndt_build).  lane_ordered_atomics == 2: test hook, one rank of the sample is
      // misread on purpose so that the fallback path runs.
      int viol = 0;
#pragma unroll 8
      for (int j = 0; j < nsteps; ++j) {
        const int i = w_beg + 64 * j + lane;
        const bool valid = i < w_end;
        int b = -1, rk = 0;
        if (valid) {
          b = RANDT_PL(j, i) - lmin;
          const unsigned long long old = atomicAdd(&bins[b], 1ull << sh);
          rk = (int)((old >> sh) & 0xffff);
          RANDT_PL(j, i) = b | (rk << 16);  // bin (<= 16 bits, nb_cap < 65536) | rank
        }
        const unsigned long long act = __ballot(valid);
        if (act == 0ull) break;  // wave-uniform: this quarter is exhausted
        // the sample's reference lane moves from step to step (lane 13 j mod 64 if it is active, else the first active one), so
        // that over the steps of a launch many different bins are looked at -- still a SAMPLE: a wrong serving order in a bin
        // that is never the reference's goes unnoticed here (the creation-time probe covers 32 x 8 collision patterns once)
        const int want = (13 * j) & 63;
        const int first = ((act >> want) & 1ull) ? want : __ffsll((long long)act) - 1;
        const int b0 = __shfl(b, first, 64);
        const unsigned long long same = __ballot(valid && b == b0);
        const int lead = __ffsll((long long)same) - 1;  // lowest lane of the reference's bin: it holds the smallest rank
        const int rk0 = __shfl(rk, lead, 64);
        int seen = rk;
        if (lane_ordered_atomics == 2 && __popcll(same & lt) == 1) seen ^= 1;  // test hook: the sample's second lane misreads
        if (valid && b == b0 && seen != rk0 + __popcll(same & lt)) viol = 1;
      }
      if (__ballot(viol != 0) != 0ull && lane == 0) scratch[20] = 1;
    } else {
      rank_by_ballots(false);
    }
    __syncthreads();
    if (lane_ordered_atomics && scratch[20] != 0) {  // uniform over the workgroup (read behind the barrier)
      for (int b = tid; b < nb; b += BUILD_BLOCK) bins[b] = 0ull;
      __syncthreads();
      rank_by_ballots(true);
      if (tid == 0 && misrank_word) {
        // a plain store into the pinned host word ("some workgroup fell back": no PCIe atomics needed), the exact count in device memory
        *reinterpret_cast<volatile int32_t*>(misrank_word) = 1;
        if (misrank_count) atomicAdd(misrank_count, 1);
      }
      __syncthreads();
    }
    __syncthreads();
    RANDT_TICK(3);
    // block scan over bins: (points << 16 | non-empty) per thread chunk
    const int chunk = (nb + BUILD_BLOCK - 1) / BUILD_BLOCK;
    const int b0 = tid * chunk, b1 = (b0 + chunk) < nb ? (b0 + chunk) : nb;
    int local = 0;
    for (int b = b0; b < b1; ++b) {
      const unsigned long long c = bins[b];
      const int tot = (int)(c & 0xffff) + (int)((c >> 16) & 0xffff) + (int)((c >> 32) & 0xffff) + (int)((c >> 48) & 0xffff);
      local += (tot << 16) | (tot > 0 ? 1 : 0);
    }
    int total;
    int run = block_exclusive_scan_256(local, scratch, &total);
    nc = total & 0xffff;
    for (int b = b0; b < b1; ++b) {
      const unsigned long long c = bins[b];
      const int c0 = (int)(c & 0xffff), c1 = (int)((c >> 16) & 0xffff), c2 = (int)((c >> 32) & 0xffff), c3 = (int)((c >> 48) & 0xffff);
      const int tot = c0 + c1 + c2 + c3;
      const int start = run >> 16;
      if (tot > 0) cstart[run & 0xffff] = (uint16_t)start;
      // packed start positions of the four waves inside this bin
      bins[b] = (unsigned long long)start | ((unsigned long long)(start + c0) << 16) |
                ((unsigned long long)(start + c0 + c1) << 32) | ((unsigned long long)(start + c0 + c1 + c2) << 48);
      run += (tot << 16) | (tot > 0 ? 1 : 0);
    }
    if (tid == 0) cstart[nc] = (uint16_t)n;
    __syncthreads();
    RANDT_TICK(4);
    // ---- placement: position = start of (bin, wave) + rank inside it.  REG: all positions are taken first (the
    // bins are read for the last time), then the points are scattered over the bins' storage.
    int ppos[REG ? PPT : 1];
#pragma unroll 8
    for (int j = 0; j < nsteps; ++j) {
      const int i = w_beg + 64 * j + lane;
      if (i < w_end) {
        const int word = RANDT_PL(j, i);
        const int b = word & 0xffff, r = (int)((unsigned)word >> 16);
        const int pos = (int)((bins[b] >> sh) & 0xffff) + r;
        if (REG) {
          ppos[REG ? j : 0] = pos;
        } else {
          float x, y, in;
          RANDT_FETCH_POINT(i, x, y, in);
          sx[pos] = x;
          sy[pos] = y;
          si[pos] = in;
        }
      }
    }
    if (REG) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int i = w_beg + 64 * j + lane;
        if (i < w_end) {
          const int pos = ppos[REG ? j : 0];
          sx[pos] = px[KEEP ? j : 0];
          sy[pos] = py[KEEP ? j : 0];
          si[pos] = pin[KEEP ? j : 0];
        }
      }
    }
    __syncthreads();
  } else {
    // fallback: rank of (label, index) by counting -- unique keys => a permutation
#pragma unroll 8
    for (int j = 0; j < nsteps; ++j) {
      const int i = w_beg + 64 * j + lane;
      if (i < w_end) lab[i] = RANDT_PL(j, i);
    }
    __syncthreads();
    for (int i = tid; i < n; i += BUILD_BLOCK) {
      const int32_t li = lab[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const int32_t lj = lab[j];
        rank += (lj < li || (lj == li && j < i)) ? 1 : 0;
      }
      float x, y, in;
      RANDT_FETCH_POINT(i, x, y, in);
      sx[rank] = x;
      sy[rank] = y;
      si[rank] = in;
      slab[rank] = li;  // sorted labels
    }
    __syncthreads();  // (workgroup-scope fence: the global writes above are visible to the whole workgroup)
    const int chunk = (n + BUILD_BLOCK - 1) / BUILD_BLOCK;
    const int b0 = tid * chunk, b1 = (b0 + chunk) < n ? (b0 + chunk) : n;
    int heads = 0;
    for (int p = b0; p < b1; ++p) heads += (p == 0 || slab[p] != slab[p - 1]) ? 1 : 0;
    int cbase = block_exclusive_scan_256(heads, scratch, &nc);
    for (int p = b0; p < b1; ++p)
      if (p == 0 || slab[p] != slab[p - 1]) cstart[cbase++] = (uint16_t)p;
    if (tid == 0) cstart[nc] = (uint16_t)n;
    __syncthreads();
  }

  RANDT_TICK(5);
  // ---- Map::insertCluster per cluster in label order (ndt_map.cpp:238-245)
  // compact cell index of cluster c if every accepted-by-size cluster lands inside the map (the normal case)
  int n_cells = 0;
  for (int c0 = 0; c0 < nc; c0 += BUILD_BLOCK) {
    const int c = c0 + tid;
    // Cell::addPointCloud: n_points_(0) + size > min_points_per_cell_ (ndt_cell.cpp:26)
    const int big = (c < nc && (long long)(cstart[c + 1] - cstart[c]) > (long long)out.min_points) ? 1 : 0;
    int tot;
    const int idx = n_cells + block_exclusive_scan_256(big, scratch, &tot);
    if (big && idx < bigcap) blist[idx] = (uint16_t)c;  // (idx < bigcap always: at most n / (min_points + 1) clusters are this large)
    n_cells += tot;
  }
  RANDT_TICK(6);
  // hand-out order: roughly descending size (32 size classes, LDS counting sort).  The order only decides
  // which clusters share a round -- every cluster's arithmetic and output index are fixed -- so the
  // arbitrary order inside a class does not touch the result.
  int* bcount = scratch + 16;  // [32]
  if (tid < 32) bcount[tid] = 0;
  if (tid == 0) scratch[4] = 0;  // "a cluster mean fell outside the map"
  __syncthreads();
  const int n_big = n_cells < bigcap ? n_cells : bigcap;
  for (int i = tid; i < n_big; i += BUILD_BLOCK) {
    const int c = blist[i];
    const int kc = (cstart[c + 1] - cstart[c]) >> 3;
    atomicAdd(&bcount[31 - (kc < 31 ? kc : 31)], 1);
  }
  __syncthreads();
  if (tid < 64) {
    const int v = tid < 32 ? bcount[tid] : 0;
    const int incl = wave_inclusive_scan(v);
    if (tid < 32) bcount[tid] = incl - v;
  }
  __syncthreads();
  for (int i = tid; i < n_big; i += BUILD_BLOCK) {
    const int c = blist[i];
    const int kc = (cstart[c + 1] - cstart[c]) >> 3;
    order[atomicAdd(&bcount[31 - (kc < 31 ? kc : 31)], 1)] = (uint16_t)i;
  }
  __syncthreads();
  RANDT_TICK(7);

  const int g = tid & 7, gbase = lane & ~7, group = tid >> 3;
  // Eight lanes per cluster.  The sums are SEQUENTIAL fp32 chains in the reference's point order (one addition per point
  // and chain, nothing to parallelise), so what counts is how few instructions a point costs the wavefront:
  //   pass 1: lanes 0..2 carry sum x / y / i -- one v_add per point.  The maximum is order-free and exact: all eight lanes
  //           take every eighth point and combine afterwards (it used to ride in the serial loop, doubling its length);
  //   pass 2: every lane forms ONE deviation d = p[j] - mean per point and takes the second factor from a neighbour with a
  //           quad permutation (chains sit so that ONE pattern serves both quads): sub, mul, add instead of two subs, mul, add.
  //   lane : 0     1     2     3    4    5     6     7
  //   d of : x     y     x     y    i    i     x     y          second factor = d of quad lanes [0, 1, 1, 1]
  //   chain: c00   c11   c01   -    -    c22   c02   c12
  const float* p1 = g == 0 ? sx : (g == 1 ? sy : si);
  const float* pd = (g == 0 || g == 2 || g == 6) ? sx : ((g == 1 || g == 3 || g == 7) ? sy : si);
  const int id = (g == 0 || g == 2 || g == 6) ? 0 : ((g == 1 || g == 3 || g == 7) ? 1 : 2);
  for (int r0 = 0; r0 < n_big; r0 += BUILD_BLOCK / 8) {  // only clusters that become cells are handed out
    const int oc = r0 + group;
    int s = 0, e = 0, target = 0xffff, c = -1;
    if (oc < n_big) {
      target = order[oc];
      c = blist[target];
      s = cstart[c];
      e = cstart[c + 1];
    }
    const int k = e - s;
    // The serial loops take EIGHT points per trip, software-pipelined: the reads of the next eight are in flight while
    // the dependent chain consumes the current eight, and the per-lane bounds (exec masking) are paid once per eight points;
    // at most seven points are left for the plain loop behind.
    float acc = 0.f;
    {
      const float* q = p1 + s;
      int j = 0;
      if (k >= 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = q[u];
        for (; j + 24 <= k; j += 16) {  // two blocks per trip: the buffers swap roles, no register copies
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w[u] = q[j + 8 + u];
        if (TP) RANDT_CHAIN_FENCE();  // (64-register instantiation) the reads above are issued before the additions below start
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += v[u];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = q[j + 16 + u];
        if (TP) RANDT_CHAIN_FENCE();  // (64-register instantiation) the reads above are issued before the additions below start
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += w[u];
        }
        if (j + 16 <= k) {
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w[u] = q[j + 8 + u];
        if (TP) RANDT_CHAIN_FENCE();  // (64-register instantiation) the reads above are issued before the additions below start
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += v[u];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = w[u];
          j += 8;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
        j += 8;
      }
      for (; j < k; ++j) acc += q[j];
    }
    float accm = 0.f;  // std::max(max_intensity_, intensity) from +0 (NaN-free data): exact in any order
    {
      int j = s + g;
      for (; j + 24 < e; j += 32) {
        const float v0 = si[j], v1 = si[j + 8], v2 = si[j + 16], v3 = si[j + 24];
        const float a = v0 > v1 ? v0 : v1, b = v2 > v3 ? v2 : v3;
        const float c = a > b ? a : b;
        accm = c > accm ? c : accm;
      }
      for (; j < e; j += 8) {
        const float v = si[j];
        accm = v > accm ? v : accm;
      }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      const float o = __shfl_xor(accm, off, 64);
      accm = o > accm ? o : accm;
    }
    if (r0 == 0) RANDT_TICK(10);
    const float nf = (float)(uint32_t)k;
    const float mean = acc / nf;
    const float m0 = __shfl(mean, gbase + 0, 64), m1 = __shfl(mean, gbase + 1, 64), m2 = __shfl(mean, gbase + 2, 64);
    const float maxi = accm;
    const float md = id == 0 ? m0 : (id == 1 ? m1 : m2);
    float cacc = 0.f;
    {
      const float* q = pd + s;
      // second factor: the deviation a quad neighbour formed (a quad belongs to one cluster: its lanes run the same trips)
      auto term = [&](float v) {
        const float da = v - md;
        const float db = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(da), 0x54 /* quad_perm [0,1,1,1] */, 0xf, 0xf, true));
        cacc += (da * db);
      };
      int j = 0;
      if (k >= 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = q[u];
        for (; j + 24 <= k; j += 16) {
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w[u] = q[j + 8 + u];
        if (TP) RANDT_CHAIN_FENCE();  // (64-register instantiation) the reads above are issued before the additions below start
#pragma unroll
          for (int u = 0; u < 8; ++u) term(v[u]);
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = q[j + 16 + u];
        if (TP) RANDT_CHAIN_FENCE();  // (64-register instantiation) the reads above are issued before the additions below start
#pragma unroll
          for (int u = 0; u < 8; ++u) term(w[u]);
        }
        if (j + 16 <= k) {
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w[u] = q[j + 8 + u];
        if (TP) RANDT_CHAIN_FENCE();  // (64-register instantiation) the reads above are issued before the additions below start
#pragma unroll
          for (int u = 0; u < 8; ++u) term(v[u]);
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = w[u];
          j += 8;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) term(v[u]);
        j += 8;
      }
      for (; j < k; ++j) term(q[j]);
    }
    if (r0 == 0) RANDT_TICK(11);
    const float cv = cacc / nf;
    const float c00 = __shfl(cv, gbase + 0, 64), c11 = __shfl(cv, gbase + 1, 64), c22 = __shfl(cv, gbase + 5, 64);
    const float c01 = __shfl(cv, gbase + 2, 64), c02 = __shfl(cv, gbase + 6, 64), c12 = __shfl(cv, gbase + 7, 64);
    if (g == 0 && k > 0) {
      randt_cell cell;
      cell.mean[0] = m0;
      cell.mean[1] = m1;
      cell.mean[2] = m2;
      cell.cov[0] = c00;
      cell.cov[1] = c01;
      cell.cov[2] = c02;
      cell.cov[3] = c11;
      cell.cov[4] = c12;
      cell.cov[5] = c22;
      cell.n = (uint32_t)k;
      cell.max_intensity = maxi;
      cell.reserved = 0;
      cell_regularize(cell);
      const uint32_t slot = coord_to_index(out, cell.mean[0], cell.mean[1]);
      if (slot < (uint32_t)out.n_slots) {  // reference: vector::at throws otherwise
        if (target < out.cap) {
          store_cell(cells + target, cell);
          // later cluster overwrites the slot, both cells stay in grid_ (quirk A.7-5)
          if (grid) atomicMax(&grid[slot], target);
        }
      } else {
        // dropped: later cells move down
        atomicMax(&scratch[4], 1);
        blist[target] = (uint16_t)(c | 0x8000);
      }
    }
    if (r0 == 0) RANDT_TICK(12);
    if (r0 == 32) RANDT_TICK(13);
  }
  __syncthreads();
  RANDT_TICK(8);
  const int dropped = scratch[4];
  if (dropped == 1 && n_cells <= out.cap) {
    // Cluster means outside the map (dropped like the reference's vector::at would): the statistics are done,
    // only the compact indices behind a dropped cluster are too high.  Move the cells down in cluster order --
    // 256 clusters at a time: read, barrier, write (a cell's final index never exceeds its provisional one, and
    // a chunk's final range ends below the next chunk's provisional range) -- and redo the slot -> cell grid.
    if (grid) {
      for (int i = tid; i < out.n_slots; i += BUILD_BLOCK) grid[i] = -1;
    }
    __syncthreads();
    int n_final = 0;
    for (int c0 = 0; c0 < n_big; c0 += BUILD_BLOCK) {
      const int prov = c0 + tid;
      const bool keep = prov < n_big && (blist[prov] & 0x8000) == 0;
      randt_cell cell;
      if (keep) cell = load_cell(cells + prov);
      int tot;
      const int fin = n_final + block_exclusive_scan_256(keep ? 1 : 0, scratch, &tot);  // barriers: all reads done
      if (keep) {
        store_cell(cells + fin, cell);
        if (grid) atomicMax(&grid[coord_to_index(out, cell.mean[0], cell.mean[1])], fin);
      }
      n_final += tot;
      __syncthreads();
    }
    n_cells = n_final;
  } else if (dropped) {
    // (capacity overflow or > 65534 cells together with a dropped cluster) -- redo in strict cluster order
    if (grid) {
      for (int i = tid; i < out.n_slots; i += BUILD_BLOCK) grid[i] = -1;
    }
    __syncthreads();
    cluster_stats_sequential(sx, sy, si, cstart, nc, out, cells, grid, scratch, &n_cells);
  }
  RANDT_TICK(9);
  if (tid == 0) out.counts[map] = n_cells < out.cap ? n_cells : out.cap;
}

}  // namespace

int launch_ndt_build(randt_ctx* ctx, const float* d_points, int n_scans, int pitch, const int32_t* d_n_points,
                     int stride, int ioff, const randt_cluster_params* cp, const MapView& out, int first_map, const float* d_polar,
                     const float* beam_cov9) {
  if (n_scans <= 0) return RANDT_OK;
  const int npad = (pitch + 63) & ~63;
  if (pitch > 7168 || ctx->build_tiled || d_polar) {  // pNDT cells: always the tiled path (its statistics kernel carries the option)
    // scans beyond one workgroup's LDS: multi-workgroup stable counting sort in global memory (ndt_build_big.hip)
    if ((long long)pitch > (1ll << 26)) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan too large (max 2^26 points)", hipSuccess);
    const size_t want = ndt_build_big_ws_bytes(n_scans, pitch, d_polar ? 1 : 0);
    if (want > ctx->build_ws_bytes) {
      if (ctx->build_ws) {
        RANDT_HIP_CHECK(ctx, randt_sync(ctx));
        RANDT_HIP_CHECK(ctx, randt_hip_free(ctx, ctx->build_ws));
        ctx->build_ws = nullptr;
        ctx->build_ws_bytes = 0;
      }
      RANDT_HIP_CHECK(ctx, randt_hip_malloc(ctx, &ctx->build_ws, want));
      ctx->build_ws_bytes = want;
    }
    int rc = launch_ndt_build_big(ctx, d_points, n_scans, pitch, d_n_points, stride, ioff, cp, out, first_map, ctx->build_ws, d_polar, beam_cov9);
    if (rc) return rc;
    // The tiled path keeps <= 8192 label bins per tile.  A scan whose labels span more (points many times max_range away from
    // the sensor, a cluster grid of more than ~7900 clusters) is refused by it on the device; that is learnt here -- one
    // synchronisation on this cold path -- and such scans are built again through the sorting path (ndt_build_big.hip).
    std::vector<int32_t> st((size_t)4 * n_scans);
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(st.data(), ctx->build_ws, sizeof(int32_t) * 4 * n_scans, hipMemcpyDeviceToHost, ctx->stream));
    RANDT_HIP_CHECK(ctx, randt_sync(ctx));
    std::vector<int32_t> npts;
    for (int s = 0; s < n_scans; ++s) {
      if (st[4 * (size_t)s + 2] == 0) continue;
      if (d_n_points && npts.empty()) {
        npts.resize(n_scans);
        RANDT_HIP_CHECK(ctx, hipMemcpy(npts.data(), d_n_points, sizeof(int32_t) * n_scans, hipMemcpyDeviceToHost));
      }
      int n = d_n_points ? npts[s] : pitch;
      n = n < 0 ? 0 : (n > pitch ? pitch : n);
      rc = launch_ndt_build_big_wide(ctx, d_points + (size_t)s * pitch * stride, n, stride, ioff, cp, out, first_map + s,
                                     d_polar ? d_polar + (size_t)s * pitch * 2 : nullptr, beam_cov9);
      if (rc) return rc;
    }
    return RANDT_OK;
  }
  // Grid::cluster (grid.cpp:8-9)
  const int row_size = (int)sqrt((double)cp->n_clusters);
  const float resolution = cp->max_range * 2 / (float)row_size;
  const bool reg = pitch <= 2048;  // 8 points per lane in registers
  // label bins: coordinates inside +-max_range (what RadarPreprocessor hands over) in fast mode: int(x/res) and
  // int(y/res) in [-row/2, row/2]; anything wider takes the fallback (labels in the global scratch)
  int nb_want = row_size * row_size + 2 * row_size + 2;
  if (nb_want > 65535) nb_want = 65535;  // bin index is packed into 16 bits
  const size_t pts_bytes = (size_t)npad * 12 + 192;
  const size_t cstart_bytes = (((size_t)npad + 2) * 2 + 15) & ~(size_t)15;
  size_t lds, aux;
  int nb_cap;
  // clusters that can become cells: more than min_points points each (lists of two u16 per such cluster, see the kernel)
  const int mp_eff = out.min_points > 0 ? out.min_points : 0;
  int bigcap = npad / (mp_eff + 1) + 1;
  bigcap = (bigcap + 7) & ~7;
  if (bigcap > npad) bigcap = npad;
  if (reg) {
    // [points == bins] | cstart | order + blist | scratch
    const size_t tail = cstart_bytes + (size_t)bigcap * 4 + 256;
    // the smallest share of the CU's LDS (a fifth = five workgroups per CU, a quarter, ...) that holds every label bin the
    // cluster grid can produce AND the points with their 1 KB margin; the whole LDS (bins capped by the room left) otherwise
    const size_t want = tail + ((size_t)nb_want * 8 > pts_bytes ? (size_t)nb_want * 8 : pts_bytes);
    const size_t least = tail + pts_bytes + 1024;
    size_t budget = (size_t)ctx->lds_limit;
    for (int share = 5; share >= 2; --share) {
      const size_t b = (size_t)ctx->lds_limit / share;
      if (want <= b && least <= b) {
        budget = b;
        break;
      }
    }
    if (least > budget) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan too large for the LDS build kernel", hipSuccess);
    const size_t room = (budget - tail) / 8;
    nb_cap = (int)((size_t)nb_want < room ? (size_t)nb_want : room);
    aux = (size_t)nb_cap * 8 > pts_bytes ? (size_t)nb_cap * 8 : pts_bytes;
    aux = (aux + 15) & ~(size_t)15;
    lds = aux + tail;
  } else {
    // points | cstart | bins | scratch | per-point words
    const size_t fixed_bytes = pts_bytes + cstart_bytes + 256 + (size_t)npad * 4;
    size_t budget = (size_t)ctx->lds_limit / 2;
    if (fixed_bytes + (size_t)nb_want * 8 + 16 > budget) budget = (size_t)ctx->lds_limit;
    if (fixed_bytes + 1024 > budget) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan too large for the LDS build kernel", hipSuccess);
    const size_t room = (budget - fixed_bytes - 16) / 8;
    nb_cap = (int)((size_t)nb_want < room ? (size_t)nb_want : room);
    aux = ((size_t)nb_cap * 8 + 15) & ~(size_t)15;
    lds = fixed_bytes + aux;
  }
  // fallback scratch: 2 x npad label words per scan
  {
    const size_t want = sizeof(int32_t) * 2 * (size_t)npad * n_scans + 256;
    if (want > ctx->build_ws_bytes) {
      if (ctx->build_ws) {
        RANDT_HIP_CHECK(ctx, randt_sync(ctx));
        RANDT_HIP_CHECK(ctx, randt_hip_free(ctx, ctx->build_ws));
        ctx->build_ws = nullptr;
        ctx->build_ws_bytes = 0;
      }
      RANDT_HIP_CHECK(ctx, randt_hip_malloc(ctx, &ctx->build_ws, want + want / 4));
      ctx->build_ws_bytes = want + want / 4;
    }
  }
  int32_t* d_fallback = reinterpret_cast<int32_t*>(ctx->build_ws);
  // Atomic ranking (see k_ndt_build): every launch checks the serving order it relies on and falls back in-kernel; the
  // workgroups that had to are counted in a host-visible word, read here without a synchronisation -- once it is non-zero the
  // context ranks with ballots only.
  // (the downgrade is not an error -- results are unaffected -- and is not written to last_error, where it would outlive its
  // cause; randt_debug_build_rank_fallbacks reports it)
  if (ctx->lds_atomics_lane_ordered && ctx->misrank_word && *reinterpret_cast<volatile int32_t*>(ctx->misrank_word) != 0)
    ctx->lds_atomics_lane_ordered = 0;
  const int rank_mode = ctx->lds_atomics_lane_ordered ? (ctx->debug_force_misrank ? 2 : 1) : 0;
#define RANDT_BUILD_LAUNCH(REG, TP)                                                                                        \
  do {                                                                                                                     \
    RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ndt_build<REG, TP>),                          \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
    hipLaunchKernelGGL((k_ndt_build<REG, TP>), dim3(n_scans), dim3(BUILD_BLOCK), lds, ctx->stream, d_points, pitch,            \
                       d_n_points, stride, ioff, row_size, resolution, out, first_map, npad, nb_cap, (int)aux, bigcap, d_fallback,    \
                       rank_mode, ctx->misrank_word, ctx->d_misrank_count);                                                                       \
  } while (0)
  // placement (see the kernel): batches that share the chip with other batches' solves
  const bool tp = n_scans > 2 * ctx->n_cus || randt_throughput_placement(ctx);
  randt_note_enqueue(ctx);
  if (reg && tp) RANDT_BUILD_LAUNCH(true, true);
  else if (reg) RANDT_BUILD_LAUNCH(true, false);
  else RANDT_BUILD_LAUNCH(false, false);
#undef RANDT_BUILD_LAUNCH
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
