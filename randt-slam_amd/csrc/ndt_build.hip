// NDT construction kernels for gfx950 (compiled with -ffp-contract=off, see cell_math.h):
//   k_ndt_build     one workgroup per radar scan: voxel key -> stable LDS counting sort -> per-cluster
//                   fp32 mean / covariance / regularisation -> compact cell table + index grid
//   k_maps_transform  Map::transformMap
//   k_maps_merge      rolling-submap update (transform + Map::mergeMapCell), ordered
//
// Replaces (paths relative to /root/reference/ros/ndt_radar_slam/):
//   src/radar_preprocessing/grid.cpp:7-14                      Grid::cluster
//   src/radar_preprocessing/radar_preprocessor.cpp:151-169     ClusterGenerator::labelClouds
//   src/ndt_representation/ndt_map.cpp:238-245                 Map::insertCluster
//   src/ndt_representation/ndt_cell.cpp:25-114                 Cell::addPointCloud / updateCell
//   src/ndt_representation/ndt_map.cpp:177-207, ndt_cell.h:133-142   transformMap / mergeMapCell / operator+=
//
// Data layout: points are read once from HBM as 16-byte (stride 4) or strided records and staged
// as SoA x/y/intensity in LDS; labels, per-label bins and the sorted index list live in LDS as well, so the
// only HBM traffic is N*16 B in, M*48 B + grid out.  fp32 sums run in the reference's sequential
// point order (one lane per cluster) so the cell statistics are bit-identical to the CPU path.
#include "cell_math.h"

using namespace randt_dev;

#define BUILD_BLOCK 256

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

// One workgroup per scan.  labelClouds' order (ascending label, input order inside a label) is
// produced by a STABLE counting sort: every wavefront owns a contiguous quarter of the points;
// a 64-bit LDS word per label bin packs the four per-wave counts (4 x u16); a block scan turns the
// counts into per-(bin, wave) start positions; each wave then walks its quarter in order and ranks
// equal labels inside a 64-point step with ballots.  Label ranges that do not fit the LDS bins
// (points far outside max_range) fall back to an O(N^2/256) rank-by-counting pass.
// The placement pass re-reads each point (L2-hot) and writes it straight into label-sorted SoA
// arrays, so the per-cluster fp32 loops stream consecutive LDS words with no index indirection.
// Dynamic LDS: sx | sy | si | lab [npad] | cstart[npad+2] | bins[nb_cap] u64 | scratch
__global__ __launch_bounds__(BUILD_BLOCK) void k_ndt_build(const float* __restrict__ pts, int pitch,
                                                           const int32_t* __restrict__ n_pts_arr, int stride,
                                                           int ioff, int row_size, float resolution, MapView out,
                                                           int first_map, int npad, int nb_cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);  // points in labelClouds order
  float* sy = sx + npad;
  float* si = sy + npad;
  int32_t* lab = reinterpret_cast<int32_t*>(si + npad);
  int* cstart = lab + npad;                                                  // [npad + 1] (+1 pad)
  unsigned long long* bins = reinterpret_cast<unsigned long long*>(cstart + npad + 2);
  int* scratch = reinterpret_cast<int*>(bins + nb_cap);                      // [16]

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
  // ---- load points (16 B / lane coalesced for packed xyzI), labels, label range
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
  for (int i = tid; i < n; i += BUILD_BLOCK) {
    float x, y, in;
    RANDT_FETCH_POINT(i, x, y, in);
    (void)in;
    const int32_t l = point_label(x, y, row_size, resolution);
    lab[i] = l;
    lmin = l < lmin ? l : lmin;
    lmax = l > lmax ? l : lmax;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int a = __shfl_xor(lmin, off, 64), b = __shfl_xor(lmax, off, 64);
    lmin = a < lmin ? a : lmin;
    lmax = b > lmax ? b : lmax;
  }
  if (lane == 0) {
    scratch[8 + wave] = lmin;
    scratch[12 + wave] = lmax;
  }
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
    // per-(bin, wave) counts: wave w owns points [w*q, (w+1)*q)
    const int q = (n + 3) >> 2;
    const int w_beg = wave * q, w_end = (w_beg + q) < n ? (w_beg + q) : n;
    for (int i = w_beg + lane; i < w_end; i += 64)
      atomicAdd(&bins[lab[i] - lmin], 1ull << (16 * wave));
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
      if (tot > 0) cstart[run & 0xffff] = start;
      // packed start positions of the four waves inside this bin
      bins[b] = (unsigned long long)start | ((unsigned long long)(start + c0) << 16) |
                ((unsigned long long)(start + c0 + c1) << 32) | ((unsigned long long)(start + c0 + c1 + c2) << 48);
      run += (tot << 16) | (tot > 0 ? 1 : 0);
    }
    if (tid == 0) cstart[nc] = n;
    __syncthreads();
    RANDT_TICK(4);
    // stable placement: each wave walks its quarter in input order
    for (int base = w_beg; base < w_end; base += 64) {
      const int i = base + lane;
      const bool valid = i < w_end;
      const int b = valid ? lab[i] - lmin : -1;
      float x = 0.f, y = 0.f, in = 0.f;
      if (valid) RANDT_FETCH_POINT(i, x, y, in);
      unsigned long long todo = __ballot(valid);
      int pos = 0;
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lb = __shfl(b, leader, 64);
        const unsigned long long m = __ballot(valid && b == lb);
        const unsigned long long word = bins[lb];
        if (b == lb) pos = (int)((word >> (16 * wave)) & 0xffff) + __popcll(m & ((1ull << lane) - 1ull));
        if (lane == leader) atomicAdd(&bins[lb], (unsigned long long)__popcll(m) << (16 * wave));
        todo &= ~m;
      }
      if (valid) {
        sx[pos] = x;
        sy[pos] = y;
        si[pos] = in;
      }
    }
    __syncthreads();
  } else {
    // fallback: rank of (label, index) by counting -- unique keys => a permutation
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
      cstart[rank] = li;  // sorted labels, parked in cstart until the heads are known
    }
    __syncthreads();
    // cluster heads from the sorted labels (cstart[] is rewritten in place: a head's cluster index
    // never exceeds its position, and positions are consumed in ascending order per thread chunk,
    // so first copy the labels of this chunk's range to registers-free scratch in lab[])
    for (int p = tid; p < n; p += BUILD_BLOCK) lab[p] = cstart[p];
    __syncthreads();
    const int chunk = (n + BUILD_BLOCK - 1) / BUILD_BLOCK;
    const int b0 = tid * chunk, b1 = (b0 + chunk) < n ? (b0 + chunk) : n;
    int heads = 0;
    for (int p = b0; p < b1; ++p) heads += (p == 0 || lab[p] != lab[p - 1]) ? 1 : 0;
    int cbase = block_exclusive_scan_256(heads, scratch, &nc);
    for (int p = b0; p < b1; ++p)
      if (p == 0 || lab[p] != lab[p - 1]) cstart[cbase++] = p;
    if (tid == 0) cstart[nc] = n;
    __syncthreads();
  }

  RANDT_TICK(5);
  // Map::insertCluster per cluster in label order (ndt_map.cpp:238-245)
  int n_cells = 0;
  for (int c0 = 0; c0 < nc; c0 += BUILD_BLOCK) {
    const int c = c0 + tid;
    randt_cell cell;
    bool accept = false;
    uint32_t slot = 0;
    if (c < nc) {
      const int s = cstart[c], e = cstart[c + 1];
      const int k = e - s;
      // Cell::addPointCloud: n_points_(0) + size > min_points_per_cell_ (ndt_cell.cpp:26)
      if ((long long)k > (long long)out.min_points) {
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, maxi = 0.f;
#pragma unroll 8
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
#pragma unroll 8
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
  RANDT_TICK(6);
  if (tid == 0) out.counts[map] = n_cells < out.cap ? n_cells : out.cap;
}

__global__ __launch_bounds__(256) void k_maps_transform(MapView m, int first, int count, const double* __restrict__ pose4) {
  const int map = first + blockIdx.y;
  float aff[4];
  pose_to_affine_f(pose4 + 4 * blockIdx.y, aff);
  const int n = m.counts[map];
  randt_cell* cells = m.cells + (size_t)map * m.cap;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    randt_cell c = load_cell(cells + i);
    cell_transform(c, aff);
    store_cell(cells + i, c);
  }
}

// Single workgroup; moving maps applied strictly in order.  Within one moving map, cells that fall
// into the same fixed slot are applied in cell order by the thread owning the first of them; new
// cells receive compact indices in cell order (Map::insertCell).  LDS: slot[cap] u32 | scratch.
__global__ __launch_bounds__(256) void k_maps_merge(MapView fixed, int fixed_idx, MapView moving, int moving_first,
                                                    int n_moving, const double* __restrict__ pose4) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* slots = reinterpret_cast<uint32_t*>(smem);
  int* scratch = reinterpret_cast<int*>(slots + moving.cap);
  const int tid = threadIdx.x;
  randt_cell* fcells = fixed.cells + (size_t)fixed_idx * fixed.cap;
  int32_t* fgrid = fixed.grid + (size_t)fixed_idx * fixed.n_slots;
  int n_cells = fixed.counts[fixed_idx];
  for (int t = 0; t < n_moving; ++t) {
    const int mmap = moving_first + t;
    const randt_cell* mcells = moving.cells + (size_t)mmap * moving.cap;
    const int M = moving.counts[mmap];
    float aff[4];
    pose_to_affine_f(pose4 + 4 * t, aff);
    for (int i = tid; i < M; i += 256) {
      randt_cell c = load_cell(mcells + i);
      cell_transform(c, aff);
      slots[i] = coord_to_index(fixed, c.mean[0], c.mean[1]);
    }
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += 256) {
      const int i = i0 + tid;
      bool leader = false, insert = false;
      uint32_t s = 0;
      int32_t existing = -1;
      if (i < M) {
        s = slots[i];
        if (s < (uint32_t)fixed.n_slots) {  // ndt_map.cpp:196
          leader = true;
          for (int j = 0; j < i; ++j)
            if (slots[j] == s) { leader = false; break; }
          if (leader) {
            existing = fgrid[s];
            insert = existing < 0;
          }
        }
      }
      int tot;
      int newidx = n_cells + block_exclusive_scan_256(insert ? 1 : 0, scratch, &tot);
      if (leader) {
        int target = insert ? newidx : existing;
        bool have = true;
        randt_cell acc;
        if (insert) {
          if (newidx < fixed.cap) {
            acc = load_cell(mcells + i);
            cell_transform(acc, aff);
            fgrid[s] = newidx;
          } else {
            have = false;  // capacity exhausted: drop (oracle does the same)
          }
        } else {
          acc = load_cell(fcells + target);
          randt_cell c = load_cell(mcells + i);
          cell_transform(c, aff);
          cell_merge(acc, c);
        }
        if (have) {
          for (int j = i + 1; j < M; ++j) {
            if (slots[j] == s) {
              randt_cell c = load_cell(mcells + j);
              cell_transform(c, aff);
              cell_merge(acc, c);
            }
          }
          store_cell(fcells + target, acc);
        }
      }
      n_cells += tot;
      if (n_cells > fixed.cap) n_cells = fixed.cap;
    }
    __syncthreads();  // workgroup-scope fence + barrier: next moving map sees this one's cells/grid
  }
  if (tid == 0) fixed.counts[fixed_idx] = n_cells;
}

}  // namespace

int launch_ndt_build(randt_ctx* ctx, const float* d_points, int n_scans, int pitch, const int32_t* d_n_points,
                     int stride, int ioff, const randt_cluster_params* cp, const MapView& out, int first_map) {
  if (n_scans <= 0) return RANDT_OK;
  const int npad = (pitch + 63) & ~63;
  if (pitch > 7168) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan too large for the LDS build kernel (max 7168 points)", hipSuccess);
  // Grid::cluster (grid.cpp:8-9)
  const int row_size = (int)sqrt((double)cp->n_clusters);
  const float resolution = cp->max_range * 2 / (float)row_size;
  // label bins: coordinates up to ~1.5 x max_range in fast mode, anything else takes the fallback
  const size_t fixed_bytes = (size_t)npad * 16 + (size_t)(npad + 2) * 4 + 128;
  const int h = (3 * row_size) / 4 + 2;
  int nb_want = 2 * (h + row_size * h) + 1;
  size_t budget = (size_t)ctx->lds_limit / 2;
  if (fixed_bytes + (size_t)nb_want * 8 > budget) budget = (size_t)ctx->lds_limit;
  if (fixed_bytes + 1024 > budget) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan too large for the LDS build kernel", hipSuccess);
  const size_t room = (budget - fixed_bytes) / 8;
  const int nb_cap = (int)((size_t)nb_want < room ? (size_t)nb_want : room);
  const size_t lds = fixed_bytes + (size_t)nb_cap * 8;
  RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ndt_build),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_ndt_build, dim3(n_scans), dim3(BUILD_BLOCK), lds, ctx->stream, d_points, pitch, d_n_points,
                     stride, ioff, row_size, resolution, out, first_map, npad, nb_cap);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int launch_maps_transform(randt_ctx* ctx, const MapView& m, int first, int count, const double* d_pose4) {
  if (count <= 0) return RANDT_OK;
  int bx = (m.cap + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_maps_transform, dim3(bx, count), dim3(256), 0, ctx->stream, m, first, count, d_pose4);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int launch_maps_merge(randt_ctx* ctx, const MapView& fixed, int fixed_idx, const MapView& moving, int moving_first,
                      int n_moving, const double* d_pose4) {
  if (n_moving <= 0) return RANDT_OK;
  size_t lds = (size_t)moving.cap * 4 + 64;
  if (lds > (size_t)ctx->lds_limit) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "moving map capacity too large for merge kernel", hipSuccess);
  RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_maps_merge),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_maps_merge, dim3(1), dim3(256), lds, ctx->stream, fixed, fixed_idx, moving, moving_first,
                     n_moving, d_pose4);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
