// NDT construction for scans that do not fit one workgroup's LDS (> 7168 points; tested to 10^6): the same result as
// k_ndt_build (ndt_build.hip) -- labelClouds order, sequential fp32 sums, compact indices in label order -- from a
// multi-workgroup stable counting sort in global memory.  Compiled with -ffp-contract=off (see cell_math.h).
//
// Replaces (paths relative to /root/reference/ros/ndt_radar_slam/):
//   src/radar_preprocessing/grid.cpp:7-14                      Grid::cluster
//   src/radar_preprocessing/radar_preprocessor.cpp:151-169     ClusterGenerator::labelClouds (no size limit there)
//   src/ndt_representation/ndt_map.cpp:238-245                 Map::insertCluster
//   src/ndt_representation/ndt_cell.cpp:25-114                 Cell::addPointCloud / updateCell
//
// Launch sequence per batch (all on the context's stream):
//   k_big_range    label range of every scan (tiles of 2048 points)
//   k_big_count    per tile: every point's rank inside (label bin, wavefront) by ballots -- the technique of the
//                  one-workgroup kernel -- and the tile's per-bin counts (4 x u16 per bin, one per wavefront)
//   k_big_scan     per scan: start of every (bin, tile) run in the sorted order (bin-major, tiles in input order =>
//                  stable), cluster bounds, acceptance by size, provisional compact indices; index grid := -1
//   k_big_scatter  per tile: points to their sorted position (SoA x / y / intensity in the workspace)
//   k_big_stats    eight lanes per cluster, one accumulator chain per lane, walking the sorted arrays sequentially
//   k_big_finish   per scan: cell count; moves cells down if a cluster mean fell outside the map (like the
//                  one-workgroup kernel's repair path)
// A scan whose labels span more than a tile's 8192 bins (points many times max_range away from the sensor, or a cluster grid of
// more than ~7900 clusters: grid.cpp:8-11 takes any n_clusters) takes launch_ndt_build_big_wide below instead of the counting sort: a
// stable device radix sort of (label, point index) -- rocPRIM, a plain library sort on a cold path -- IS labelClouds' order; cluster
// bounds are the label changes, and the statistics / finish kernels run unchanged on them.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "cell_math.h"

using namespace randt_dev;

#define BIG_BLOCK 256
#define BIG_TILE 2048      // points per tile: 8 per lane
#define BIG_PPT 8
#define BIG_NB_MAX 8192    // label bins a tile keeps in LDS (8 B each)

namespace {

struct BigArgs {
  const float* pts;
  const int32_t* n_pts_arr;
  int pitch, stride, ioff, row_size;
  float resolution;
  int n_tiles, first_map;
  // workspace (per scan strides in elements)
  int32_t* range;        // [n_scans][4]: lmin, lmax, status, dropped flag
  uint32_t* pword;       // [n_scans][npad]
  unsigned long long* hist;  // [n_scans][n_tiles][BIG_NB_MAX]  (only [0, nb) of a row is used)
  uint32_t* tstart;      // [n_scans][n_tiles][BIG_NB_MAX]
  uint32_t* bin_start;   // [n_scans][BIG_NB_MAX + 1]
  uint32_t* pre;         // [n_scans][BIG_NB_MAX]   provisional compact cell index, 0xffffffff = no cell
  float* sx;             // [n_scans][npad] x 3
  float* sy;
  float* si;
  int npad;
  int lane_ordered_atomics;  // randt_ctx::lds_atomics_lane_ordered
  // pNDT (Cell::updateCell with use_pndt, ndt_cell.cpp:67-82): polar coordinates of the points, sorted like sx / sy / si
  const float* polar;    // [n_scans][pitch][2] (angle, range) or nullptr
  float* sa;             // [n_scans][npad] x 2
  float* sr;
  float beam[9];         // NDTCellParameters::beam_cov, row-major
};

__device__ __forceinline__ int32_t big_label(float x, float y, int row_size, float resolution) {
  return trunc_to_i32(x / resolution) + row_size * trunc_to_i32(y / resolution);  // grid.cpp:7-14
}

__device__ __forceinline__ void fetch_point(const BigArgs& A, const float* sp, int i, float& x, float& y, float& in) {
  if (A.stride == 4 && (((size_t)sp & 15) == 0)) {
    const float4 p = reinterpret_cast<const float4*>(sp)[i];
    x = p.x;
    y = p.y;
    in = A.ioff == 3 ? p.w : (A.ioff == 2 ? p.z : (A.ioff == 1 ? p.y : p.x));
  } else {
    const float* p = sp + (size_t)i * A.stride;
    x = p[0];
    y = p[1];
    in = p[A.ioff];
  }
}

__device__ __forceinline__ int scan_points(const BigArgs& A, int scan) {
  int n = A.n_pts_arr ? A.n_pts_arr[scan] : A.pitch;
  return n < 0 ? 0 : (n > A.pitch ? A.pitch : n);
}

__global__ __launch_bounds__(BIG_BLOCK) void k_big_range(BigArgs A) {
  const int scan = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int n = scan_points(A, scan);
  const float* sp = A.pts + (size_t)scan * A.pitch * A.stride;
  int lmin = 0x7fffffff, lmax = (int)0x80000000;
  for (int j = 0; j < BIG_PPT; ++j) {
    const int i = tile * BIG_TILE + j * BIG_BLOCK + tid;
    if (i < n) {
      float x, y, in;
      fetch_point(A, sp, i, x, y, in);
      const int l = big_label(x, y, A.row_size, A.resolution);
      lmin = l < lmin ? l : lmin;
      lmax = l > lmax ? l : lmax;
    }
  }
  lmin = wave_minmax<false>(lmin);
  lmax = wave_minmax<true>(lmax);
  if ((tid & 63) == 0 && lmin <= lmax) {
    atomicMin(&A.range[4 * scan + 0], lmin);
    atomicMax(&A.range[4 * scan + 1], lmax);
  }
}

// Wave w of a tile owns the contiguous quarter [w * 512, (w + 1) * 512) of the tile; lane l holds points 64 j + l of it.
__global__ __launch_bounds__(BIG_BLOCK) void k_big_count(BigArgs A) {
  __shared__ unsigned long long bins[BIG_NB_MAX];
  const int scan = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = scan_points(A, scan);
  const int lmin = A.range[4 * scan + 0], lmax = A.range[4 * scan + 1];
  if (n == 0 || tile * BIG_TILE >= n) return;
  const long long range = (long long)lmax - (long long)lmin + 1;
  if (range > BIG_NB_MAX) {
    if (tid == 0) A.range[4 * scan + 2] = 1;  // label range too wide for the tiled path
    return;
  }
  const int nb = (int)range;
  for (int b = tid; b < nb; b += BIG_BLOCK) bins[b] = 0ull;
  __syncthreads();
  const float* sp = A.pts + (size_t)scan * A.pitch * A.stride;
  uint32_t* pw = A.pword + (size_t)scan * A.npad;
  const int w_beg = tile * BIG_TILE + wave * (BIG_TILE / 4);
  const int w_end = (w_beg + BIG_TILE / 4) < n ? (w_beg + BIG_TILE / 4) : n;
  const int nbits = nb > 1 ? 32 - __clz(nb - 1) : 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int sh = 16 * wave;
  if (A.lane_ordered_atomics) {
    // colliding LDS atomics of one instruction are served in lane order on this device (self-test at context creation):
    // the returned count is the rank (see k_ndt_build)
    for (int j = 0; j < BIG_PPT; ++j) {
      const int i = w_beg + 64 * j + lane;
      if (i < w_end) {
        float x, y, in;
        fetch_point(A, sp, i, x, y, in);
        const int b = big_label(x, y, A.row_size, A.resolution) - lmin;
        const unsigned long long old = atomicAdd(&bins[b], 1ull << sh);
        pw[i] = (uint32_t)b | ((uint32_t)((old >> sh) & 0xffff) << 16);
      }
    }
  } else {
    for (int j = 0; j < BIG_PPT; ++j) {
      const int i = w_beg + 64 * j + lane;
      const bool valid = i < w_end;
      int b = 0;
      if (valid) {
        float x, y, in;
        fetch_point(A, sp, i, x, y, in);
        b = big_label(x, y, A.row_size, A.resolution) - lmin;
      }
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
      if (valid) pw[i] = (uint32_t)b | ((uint32_t)(field + __popcll(mask & lt)) << 16);  // bin | rank inside (bin, wave)
    }
  }
  __syncthreads();
  unsigned long long* h = A.hist + ((size_t)scan * A.n_tiles + tile) * BIG_NB_MAX;
  for (int b = tid; b < nb; b += BIG_BLOCK) h[b] = bins[b];
}

__global__ __launch_bounds__(BIG_BLOCK) void k_big_scan(BigArgs A, MapView out) {
  __shared__ int scratch[8];
  const int scan = blockIdx.x, tid = threadIdx.x;
  const int map = A.first_map + scan;
  const int n = scan_points(A, scan);
  // Map::initialize: index grid = -1 (ndt_map.cpp:13-16)
  if (out.grid) {
    int32_t* grid = out.grid + (size_t)map * out.n_slots;
    for (int i = tid; i < out.n_slots; i += BIG_BLOCK) grid[i] = -1;
  }
  if (A.range[4 * scan + 2] != 0 || n == 0) return;
  const int lmin = A.range[4 * scan + 0], lmax = A.range[4 * scan + 1];
  const int nb = lmax - lmin + 1;
  const int n_tiles = (n + BIG_TILE - 1) / BIG_TILE;
  const unsigned long long* hist = A.hist + (size_t)scan * A.n_tiles * BIG_NB_MAX;
  uint32_t* tstart = A.tstart + (size_t)scan * A.n_tiles * BIG_NB_MAX;
  uint32_t* bstart = A.bin_start + (size_t)scan * (BIG_NB_MAX + 1);
  uint32_t* pre = A.pre + (size_t)scan * BIG_NB_MAX;
  uint32_t run_pts = 0;
  int run_cells = 0;
  for (int b0 = 0; b0 < nb; b0 += BIG_BLOCK) {
    const int b = b0 + tid;
    uint32_t cnt = 0;
    if (b < nb) {
      for (int t = 0; t < n_tiles; ++t) {  // tiles in input order: the sort is stable
        const unsigned long long c = hist[(size_t)t * BIG_NB_MAX + b];
        tstart[(size_t)t * BIG_NB_MAX + b] = cnt;  // relative to the bin's start, fixed up below
        cnt += (uint32_t)(c & 0xffff) + (uint32_t)((c >> 16) & 0xffff) + (uint32_t)((c >> 32) & 0xffff) + (uint32_t)((c >> 48) & 0xffff);
      }
    }
    // Cell::addPointCloud: n_points_(0) + size > min_points_per_cell_ (ndt_cell.cpp:26)
    const int big = (b < nb && (long long)cnt > (long long)out.min_points) ? 1 : 0;
    int tot_pts, tot_cells;
    const uint32_t start = run_pts + (uint32_t)block_exclusive_scan_256((int)cnt, scratch, &tot_pts);
    const int idx = run_cells + block_exclusive_scan_256(big, scratch + 4, &tot_cells);
    if (b < nb) {
      bstart[b] = start;
      pre[b] = big ? (uint32_t)idx : 0xffffffffu;
      for (int t = 0; t < n_tiles; ++t) tstart[(size_t)t * BIG_NB_MAX + b] += start;
    }
    run_pts += (uint32_t)tot_pts;
    run_cells += tot_cells;
  }
  if (tid == 0) bstart[nb] = run_pts;
}

__global__ __launch_bounds__(BIG_BLOCK) void k_big_scatter(BigArgs A) {
  const int scan = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = scan_points(A, scan);
  if (A.range[4 * scan + 2] != 0 || tile * BIG_TILE >= n) return;
  const float* sp = A.pts + (size_t)scan * A.pitch * A.stride;
  const uint32_t* pw = A.pword + (size_t)scan * A.npad;
  const unsigned long long* h = A.hist + ((size_t)scan * A.n_tiles + tile) * BIG_NB_MAX;
  const uint32_t* ts = A.tstart + ((size_t)scan * A.n_tiles + tile) * BIG_NB_MAX;
  float* sx = A.sx + (size_t)scan * A.npad;
  float* sy = A.sy + (size_t)scan * A.npad;
  float* si = A.si + (size_t)scan * A.npad;
  const int w_beg = tile * BIG_TILE + wave * (BIG_TILE / 4);
  const int w_end = (w_beg + BIG_TILE / 4) < n ? (w_beg + BIG_TILE / 4) : n;
  for (int j = 0; j < BIG_PPT; ++j) {
    const int i = w_beg + 64 * j + lane;
    if (i < w_end) {
      const uint32_t word = pw[i];
      const int b = (int)(word & 0xffff), r = (int)(word >> 16);
      const unsigned long long c = h[b];
      uint32_t off = 0;  // points of this bin in the lower wavefronts of the tile
      if (wave > 0) off += (uint32_t)(c & 0xffff);
      if (wave > 1) off += (uint32_t)((c >> 16) & 0xffff);
      if (wave > 2) off += (uint32_t)((c >> 32) & 0xffff);
      const uint32_t pos = ts[b] + off + (uint32_t)r;
      float x, y, in;
      fetch_point(A, sp, i, x, y, in);
      sx[pos] = x;
      sy[pos] = y;
      si[pos] = in;
      if (A.polar) {
        const float* pp = A.polar + ((size_t)scan * A.pitch + i) * 2;
        A.sa[(size_t)scan * A.npad + pos] = pp[0];
        A.sr[(size_t)scan * A.npad + pos] = pp[1];
      }
    }
  }
}

// 32 clusters per workgroup round, eight lanes per cluster (lanes 0-2 / 0-5 carry the sum / covariance chains).
__global__ __launch_bounds__(BIG_BLOCK) void k_big_stats(BigArgs A, MapView out) {
  const int scan = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int n = scan_points(A, scan);
  if (A.range[4 * scan + 2] != 0 || n == 0) return;
  const int map = A.first_map + scan;
  const int nb = A.range[4 * scan + 1] - A.range[4 * scan + 0] + 1;
  const uint32_t* bstart = A.bin_start + (size_t)scan * (BIG_NB_MAX + 1);
  uint32_t* pre = A.pre + (size_t)scan * BIG_NB_MAX;
  const float* sx = A.sx + (size_t)scan * A.npad;
  const float* sy = A.sy + (size_t)scan * A.npad;
  const float* si = A.si + (size_t)scan * A.npad;
  int32_t* grid = out.grid ? out.grid + (size_t)map * out.n_slots : nullptr;
  randt_cell* cells = out.cells + (size_t)map * out.cap;
  const int g = tid & 7, gbase = lane & ~7, group = tid >> 3;
  const float* p1 = g == 0 ? sx : (g == 1 ? sy : si);
  const float* pa = (g == 0 || g == 3 || g == 4) ? sx : ((g == 1 || g == 5) ? sy : si);
  const float* pb = (g == 0) ? sx : ((g == 1 || g == 3) ? sy : si);
  const int ia = (g == 0 || g == 3 || g == 4) ? 0 : ((g == 1 || g == 5) ? 1 : 2);
  const int ib = (g == 0) ? 0 : ((g == 1 || g == 3) ? 1 : 2);
  for (int b0 = blockIdx.x * (BIG_BLOCK / 8); b0 < nb; b0 += gridDim.x * (BIG_BLOCK / 8)) {
    const int b = b0 + group;
    uint32_t target = 0xffffffffu;
    int s = 0, e = 0;
    if (b < nb) {
      target = pre[b];
      if (target != 0xffffffffu) {
        s = (int)bstart[b];
        e = (int)bstart[b + 1];
      }
    }
    const int k = e - s;
    float acc = 0.f, accm = 0.f;
    for (int j = s; j < e; ++j) {
      const float v = p1[j];
      acc += v;
      accm = fmaxf(accm, v);
    }
    const float nf = (float)(uint32_t)k;
    const float mean = acc / nf;
    const float m0 = __shfl(mean, gbase + 0, 64), m1 = __shfl(mean, gbase + 1, 64), m2 = __shfl(mean, gbase + 2, 64);
    const float maxi = __shfl(accm, gbase + 3, 64);
    const float ma = ia == 0 ? m0 : (ia == 1 ? m1 : m2);
    const float mb = ib == 0 ? m0 : (ib == 1 ? m1 : m2);
    float cacc = 0.f;
    for (int j = s; j < e; ++j) {
      const float da = pa[j] - ma, db = pb[j] - mb;
      cacc += (da * db);
    }
    float cv = cacc / nf;
    if (A.polar) {
      // pNDT: lane g accumulates entry (ia, ib) of sum_j J_j S J_j^T in point order, J = d(x, y, i) / d(angle, range, i)
      // (ndt_cell.cpp:68-80); products as ((a0 b0 + a1 b1) + a2 b2) with (J S) first, sin / cos in double rounded once
      // (spec decision 10)
      const float* sa = A.sa + (size_t)scan * A.npad;
      const float* sr = A.sr + (size_t)scan * A.npad;
      float pacc = 0.f;
      for (int j = s; j < e; ++j) {
        const float a = sa[j], r = sr[j];
        const float sn = (float)sin((double)a), cs = (float)cos((double)a);
        const float Ja0 = ia == 0 ? -r * sn : (ia == 1 ? r * cs : 0.f), Ja1 = ia == 0 ? cs : (ia == 1 ? sn : 0.f), Ja2 = ia == 2 ? 1.f : 0.f;
        const float Jb0 = ib == 0 ? -r * sn : (ib == 1 ? r * cs : 0.f), Jb1 = ib == 0 ? cs : (ib == 1 ? sn : 0.f), Jb2 = ib == 2 ? 1.f : 0.f;
        const float JS0 = (Ja0 * A.beam[0] + Ja1 * A.beam[3]) + Ja2 * A.beam[6];
        const float JS1 = (Ja0 * A.beam[1] + Ja1 * A.beam[4]) + Ja2 * A.beam[7];
        const float JS2 = (Ja0 * A.beam[2] + Ja1 * A.beam[5]) + Ja2 * A.beam[8];
        pacc = pacc + ((JS0 * Jb0 + JS1 * Jb1) + JS2 * Jb2);
      }
      cv = cv + pacc / nf;
    }
    const float c00 = __shfl(cv, gbase + 0, 64), c11 = __shfl(cv, gbase + 1, 64), c22 = __shfl(cv, gbase + 2, 64);
    const float c01 = __shfl(cv, gbase + 3, 64), c02 = __shfl(cv, gbase + 4, 64), c12 = __shfl(cv, gbase + 5, 64);
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
      if (!A.polar) cell_regularize(cell);  // ndt_cell.cpp:102: not with use_pndt
      const uint32_t slot = coord_to_index(out, cell.mean[0], cell.mean[1]);
      if (slot < (uint32_t)out.n_slots) {  // reference: vector::at throws otherwise
        if (target < (uint32_t)out.cap) {
          store_cell(cells + target, cell);
          // later cluster overwrites the slot, both cells stay in grid_ (quirk A.7-5)
          if (grid) atomicMax(&grid[slot], (int32_t)target);
        }
      } else {
        // dropped: later cells move down (k_big_finish)
        A.range[4 * scan + 3] = 1;
        pre[b] = 0xfffffffeu;
      }
    }
  }
}

__global__ __launch_bounds__(BIG_BLOCK) void k_big_finish(BigArgs A, MapView out) {
  __shared__ int scratch[8];
  const int scan = blockIdx.x, tid = threadIdx.x;
  const int map = A.first_map + scan;
  const int n = scan_points(A, scan);
  if (A.range[4 * scan + 2] != 0 || n == 0) {
    if (tid == 0) out.counts[map] = 0;
    return;
  }
  const int nb = A.range[4 * scan + 1] - A.range[4 * scan + 0] + 1;
  const uint32_t* pre = A.pre + (size_t)scan * BIG_NB_MAX;
  randt_cell* cells = out.cells + (size_t)map * out.cap;
  int32_t* grid = out.grid ? out.grid + (size_t)map * out.n_slots : nullptr;
  const bool dropped = A.range[4 * scan + 3] != 0;
  if (dropped && grid) {
    for (int i = tid; i < out.n_slots; i += BIG_BLOCK) grid[i] = -1;
  }
  __syncthreads();
  // count / compaction in cluster order, 256 clusters at a time: read, barrier, write (a cell's final index never
  // exceeds its provisional one, and a chunk's final range ends below the next chunk's provisional range)
  int n_final = 0;
  for (int b0 = 0; b0 < nb; b0 += BIG_BLOCK) {
    const int b = b0 + tid;
    const uint32_t prov = b < nb ? pre[b] : 0xffffffffu;
    const bool keep = prov < 0xfffffffeu;
    randt_cell cell;
    const bool have = keep && prov < (uint32_t)out.cap;
    if (dropped && have) cell = load_cell(cells + prov);
    int tot;
    const int fin = n_final + block_exclusive_scan_256(keep ? 1 : 0, scratch, &tot);  // barriers: all reads done
    if (dropped && have && fin < out.cap) {
      store_cell(cells + fin, cell);
      if (grid) atomicMax(&grid[coord_to_index(out, cell.mean[0], cell.mean[1])], fin);
    }
    n_final += tot;
    __syncthreads();
  }
  if (tid == 0) out.counts[map] = n_final < out.cap ? n_final : out.cap;
}

__global__ void k_big_init(int32_t* range, int n_scans) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_scans) {
    range[4 * i + 0] = 0x7fffffff;
    range[4 * i + 1] = (int)0x80000000;
    range[4 * i + 2] = 0;
    range[4 * i + 3] = 0;
  }
}

// ---- wide label spans: sort instead of count (one scan at a time; see the header)
__global__ void k_wide_keys(BigArgs A, int n, uint32_t* keys, uint32_t* vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x, y, in;
  fetch_point(A, A.pts, i, x, y, in);
  keys[i] = (uint32_t)big_label(x, y, A.row_size, A.resolution) ^ 0x80000000u;  // signed order as unsigned order
  vals[i] = (uint32_t)i;
}
__global__ void k_wide_heads(const uint32_t* keys, int n, uint32_t* head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// cid: inclusive scan of head (cluster number + 1 of every sorted point)
__global__ void k_wide_starts(const uint32_t* head, const uint32_t* cid, int n, uint32_t* cstart, int32_t* range) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (head[i]) cstart[cid[i] - 1u] = (uint32_t)i;
  if (i == n - 1) {
    const uint32_t nc = cid[i];
    cstart[nc] = (uint32_t)n;
    range[0] = 0;                // the statistics / finish kernels see clusters 0 .. nc - 1 as their "bins"
    range[1] = (int32_t)nc - 1;
    range[2] = 0;
    range[3] = 0;
  }
}
__global__ void k_wide_accept(const uint32_t* cstart, const int32_t* range, int n, int min_points, uint32_t* accept) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int nc = range[1] + 1;
  // Cell::addPointCloud: n_points_(0) + size > min_points_per_cell_ (ndt_cell.cpp:26)
  accept[c] = (c < nc && (long long)(cstart[c + 1] - cstart[c]) > (long long)min_points) ? 1u : 0u;
}
__global__ void k_wide_pre(const uint32_t* accept, const uint32_t* excl, int n, uint32_t* pre) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) pre[c] = accept[c] ? excl[c] : 0xffffffffu;
}
__global__ void k_wide_scatter(BigArgs A, int n, const uint32_t* vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int src = (int)vals[i];
  float x, y, in;
  fetch_point(A, A.pts, src, x, y, in);
  A.sx[i] = x;
  A.sy[i] = y;
  A.si[i] = in;
  if (A.polar) {
    A.sa[i] = A.polar[2 * (size_t)src];
    A.sr[i] = A.polar[2 * (size_t)src + 1];
  }
}

}  // namespace

// One scan whose labels span more than BIG_NB_MAX values (host-side n: the caller has synchronised to learn that the tiled path
// refused it).  d_points / d_polar: THIS scan's points; map: its index in `out`.  Uses ctx->build_wide_ws (grown on demand).
int launch_ndt_build_big_wide(randt_ctx* ctx, const float* d_points, int n, int stride, int ioff, const randt_cluster_params* cp,
                              const MapView& out, int map, const float* d_polar, const float* beam_cov9) {
  if (n <= 0) return RANDT_OK;
  const size_t npad = ((size_t)n + 63) & ~(size_t)63;
  size_t t_sort = 0, t_scan = 0;
  {
    uint32_t* nul = nullptr;
    RANDT_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, t_sort, nul, nul, nul, nul, (size_t)n, 0, 32, ctx->stream));
    RANDT_HIP_CHECK(ctx, rocprim::inclusive_scan(nullptr, t_scan, nul, nul, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
    size_t t_ex = 0;
    RANDT_HIP_CHECK(ctx, rocprim::exclusive_scan(nullptr, t_ex, nul, nul, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
    t_scan = t_scan > t_ex ? t_scan : t_ex;
  }
  const size_t t_tmp = ((t_sort > t_scan ? t_sort : t_scan) + 255) & ~(size_t)255;
  const size_t need = 256 + 9 * (npad + 64) * 4 + (d_polar ? 2 * npad * 4 : 0) + t_tmp + 4096;
  if (need > ctx->build_wide_ws_bytes) {
    if (ctx->build_wide_ws) {
      RANDT_HIP_CHECK(ctx, randt_sync(ctx));
      (void)randt_hip_free(ctx, ctx->build_wide_ws);
      ctx->build_wide_ws = nullptr;
      ctx->build_wide_ws_bytes = 0;
    }
    RANDT_HIP_CHECK(ctx, randt_hip_malloc(ctx, &ctx->build_wide_ws, need));
    ctx->build_wide_ws_bytes = need;
  }
  char* w = static_cast<char*>(ctx->build_wide_ws);
  auto take = [&](size_t bytes) {
    char* p = w;
    w += (bytes + 255) & ~(size_t)255;
    return p;
  };
  int32_t* range = (int32_t*)take(16);
  uint32_t* k_in = (uint32_t*)take((npad + 64) * 4);
  uint32_t* k_out = (uint32_t*)take((npad + 64) * 4);
  uint32_t* v_in = (uint32_t*)take((npad + 64) * 4);
  uint32_t* v_out = (uint32_t*)take((npad + 64) * 4);
  uint32_t* cstart = (uint32_t*)take((npad + 64) * 4);
  uint32_t* pre = (uint32_t*)take((npad + 64) * 4);
  BigArgs A;
  memset(&A, 0, sizeof(A));
  A.polar = d_polar;
  for (int i = 0; i < 9; ++i) A.beam[i] = (d_polar && beam_cov9) ? beam_cov9[i] : 0.f;
  A.pts = d_points;
  A.n_pts_arr = nullptr;
  A.pitch = n;
  A.stride = stride;
  A.ioff = ioff;
  A.row_size = (int)sqrt((double)cp->n_clusters);           // Grid::cluster (grid.cpp:8-9)
  A.resolution = cp->max_range * 2 / (float)A.row_size;
  A.n_tiles = 1;
  A.first_map = map;
  A.npad = (int)npad;
  A.range = range;
  A.bin_start = cstart;
  A.pre = pre;
  A.sx = (float*)take(npad * 4);
  A.sy = (float*)take(npad * 4);
  A.si = (float*)take(npad * 4);
  if (d_polar) {
    A.sa = (float*)take(npad * 4);
    A.sr = (float*)take(npad * 4);
  }
  void* tmp = take(t_tmp);
  hipStream_t st = ctx->stream;
  const dim3 blk(256), grd((n + 255) / 256);
  hipLaunchKernelGGL(k_wide_keys, grd, blk, 0, st, A, n, k_in, v_in);
  size_t tb = t_tmp;
  RANDT_HIP_CHECK(ctx, rocprim::radix_sort_pairs(tmp, tb, k_in, k_out, v_in, v_out, (size_t)n, 0, 32, st));  // stable: labelClouds' order
  hipLaunchKernelGGL(k_wide_heads, grd, blk, 0, st, k_out, n, k_in /* head */);
  tb = t_tmp;
  RANDT_HIP_CHECK(ctx, rocprim::inclusive_scan(tmp, tb, k_in, v_in /* cluster number + 1 */, (size_t)n, rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_wide_starts, grd, blk, 0, st, k_in, v_in, n, cstart, range);
  hipLaunchKernelGGL(k_wide_accept, grd, blk, 0, st, cstart, range, n, out.min_points, k_in /* accept */);
  tb = t_tmp;
  RANDT_HIP_CHECK(ctx, rocprim::exclusive_scan(tmp, tb, k_in, v_in /* provisional compact index */, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_wide_pre, grd, blk, 0, st, k_in, v_in, n, pre);
  hipLaunchKernelGGL(k_wide_scatter, grd, blk, 0, st, A, n, v_out);
  hipLaunchKernelGGL(k_big_stats, dim3(64, 1), dim3(BIG_BLOCK), 0, st, A, out);
  hipLaunchKernelGGL(k_big_finish, dim3(1), dim3(BIG_BLOCK), 0, st, A, out);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

// Workspace bytes of the tiled path for a batch.
size_t ndt_build_big_ws_bytes(int n_scans, int pitch, int with_polar) {
  const size_t npad = ((size_t)pitch + 63) & ~(size_t)63;
  const size_t n_tiles = ((size_t)pitch + BIG_TILE - 1) / BIG_TILE;
  size_t per = 16 + npad * 4 + n_tiles * BIG_NB_MAX * 8 + n_tiles * BIG_NB_MAX * 4 + (BIG_NB_MAX + 1) * 4 + BIG_NB_MAX * 4 + npad * 12 +
               (with_polar ? npad * 8 + 512 : 0);
  per = (per + 255) & ~(size_t)255;
  return per * n_scans + 4096;
}

int launch_ndt_build_big(randt_ctx* ctx, const float* d_points, int n_scans, int pitch, const int32_t* d_n_points, int stride,
                         int ioff, const randt_cluster_params* cp, const MapView& out, int first_map, void* d_ws, const float* d_polar,
                         const float* beam_cov9) {
  BigArgs A;
  A.polar = d_polar;
  A.sa = A.sr = nullptr;
  for (int i = 0; i < 9; ++i) A.beam[i] = (d_polar && beam_cov9) ? beam_cov9[i] : 0.f;
  A.pts = d_points;
  A.n_pts_arr = d_n_points;
  A.pitch = pitch;
  A.stride = stride;
  A.ioff = ioff;
  A.row_size = (int)sqrt((double)cp->n_clusters);           // Grid::cluster (grid.cpp:8-9)
  A.resolution = cp->max_range * 2 / (float)A.row_size;
  A.n_tiles = (pitch + BIG_TILE - 1) / BIG_TILE;
  A.first_map = first_map;
  A.npad = (pitch + 63) & ~63;
  // the tiled path (scans above 7168 points, pNDT cells: not the headline path) ranks with ballots unless asked otherwise: the atomic
  // ranking's in-kernel order check and fallback live in k_ndt_build only
  A.lane_ordered_atomics = (ctx->lds_atomics_lane_ordered && getenv("RANDT_BUILD_BIG_ATOMIC_RANK")) ? 1 : 0;
  char* w = (char*)d_ws;
  auto take = [&](size_t bytes) {
    char* p = w;
    w += (bytes + 255) & ~(size_t)255;
    return p;
  };
  A.range = (int32_t*)take(sizeof(int32_t) * 4 * n_scans);
  A.pword = (uint32_t*)take(sizeof(uint32_t) * (size_t)A.npad * n_scans);
  A.hist = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)A.n_tiles * BIG_NB_MAX * n_scans);
  A.tstart = (uint32_t*)take(sizeof(uint32_t) * (size_t)A.n_tiles * BIG_NB_MAX * n_scans);
  A.bin_start = (uint32_t*)take(sizeof(uint32_t) * (size_t)(BIG_NB_MAX + 1) * n_scans);
  A.pre = (uint32_t*)take(sizeof(uint32_t) * (size_t)BIG_NB_MAX * n_scans);
  A.sx = (float*)take(sizeof(float) * (size_t)A.npad * n_scans);
  A.sy = (float*)take(sizeof(float) * (size_t)A.npad * n_scans);
  A.si = (float*)take(sizeof(float) * (size_t)A.npad * n_scans);
  if (d_polar) {
    A.sa = (float*)take(sizeof(float) * (size_t)A.npad * n_scans);
    A.sr = (float*)take(sizeof(float) * (size_t)A.npad * n_scans);
  }
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_big_init, dim3((n_scans + 255) / 256), dim3(256), 0, st, A.range, n_scans);
  const dim3 tiles(A.n_tiles, n_scans);
  hipLaunchKernelGGL(k_big_range, tiles, dim3(BIG_BLOCK), 0, st, A);
  hipLaunchKernelGGL(k_big_count, tiles, dim3(BIG_BLOCK), 0, st, A);
  hipLaunchKernelGGL(k_big_scan, dim3(n_scans), dim3(BIG_BLOCK), 0, st, A, out);
  hipLaunchKernelGGL(k_big_scatter, tiles, dim3(BIG_BLOCK), 0, st, A);
  int sb = (BIG_NB_MAX + 31) / 32;
  if (sb > 64) sb = 64;
  hipLaunchKernelGGL(k_big_stats, dim3(sb, n_scans), dim3(BIG_BLOCK), 0, st, A, out);
  hipLaunchKernelGGL(k_big_finish, dim3(n_scans), dim3(BIG_BLOCK), 0, st, A, out);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
