// NDT construction kernels for gfx950 (compiled with -ffp-contract=off, see cell_math.h):
//   k_ndt_build     one workgroup per radar scan: voxel key -> LDS bitonic sort -> per-cluster
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
// as SoA x/y/intensity in LDS; the sort key (label << 32 | point index) lives in LDS as well, so the
// only HBM traffic is N*16 B in, M*48 B + grid out.  fp32 sums run in the reference's sequential
// point order (one lane per cluster) so the cell statistics are bit-identical to the CPU path.
#include "cell_math.h"

using namespace randt_dev;

#define BUILD_BLOCK 256

namespace {

__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* keys, int npad) {
  const int tid = threadIdx.x;
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (npad >> 1); t += BUILD_BLOCK) {
        int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int l = i | j;
        bool up = (i & k) == 0;
        unsigned long long a = keys[i], b = keys[l];
        if ((a > b) == up) {
          keys[i] = b;
          keys[l] = a;
        }
      }
      __syncthreads();
    }
  }
}

// One workgroup per scan.  Dynamic LDS: keys[npad] u64 | px[npad] | py[npad] | pi[npad] | cstart[npad+1] | scratch[8]
__global__ __launch_bounds__(BUILD_BLOCK) void k_ndt_build(const float* __restrict__ pts, int pitch,
                                                           const int32_t* __restrict__ n_pts_arr, int stride,
                                                           int ioff, int row_size, float resolution, MapView out,
                                                           int first_map, int npad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  float* px = reinterpret_cast<float*>(keys + npad);
  float* py = px + npad;
  float* pi = py + npad;
  int* cstart = reinterpret_cast<int*>(pi + npad);
  int* scratch = cstart + npad + 1;

  const int tid = threadIdx.x;
  const int scan = blockIdx.x;
  const int map = first_map + scan;
  int n = n_pts_arr ? n_pts_arr[scan] : pitch;
  n = n < 0 ? 0 : (n > pitch ? pitch : n);
  const float* sp = pts + (size_t)scan * pitch * stride;
  int32_t* grid = out.grid ? out.grid + (size_t)map * out.n_slots : nullptr;
  randt_cell* cells = out.cells + (size_t)map * out.cap;

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

  // Grid::cluster (grid.cpp:7-14): label = int(x/res) + row * int(y/res)
  const bool vec4 = (stride == 4) && (((size_t)sp & 15) == 0);
  for (int i = tid; i < npad; i += BUILD_BLOCK) {
    unsigned long long key = ~0ull;
    float x = 0.f, y = 0.f, in = 0.f;
    if (i < n) {
      if (vec4) {
        float4 p = reinterpret_cast<const float4*>(sp)[i];
        x = p.x;
        y = p.y;
        in = ioff == 3 ? p.w : (ioff == 2 ? p.z : (ioff == 1 ? p.y : p.x));
      } else {
        const float* p = sp + (size_t)i * stride;
        x = p[0];
        y = p[1];
        in = p[ioff];
      }
      int32_t label = trunc_to_i32(x / resolution) + row_size * trunc_to_i32(y / resolution);
      key = ((unsigned long long)((uint32_t)label ^ 0x80000000u) << 32) | (uint32_t)i;
    }
    keys[i] = key;
    px[i] = x;
    py[i] = y;
    pi[i] = in;
  }
  __syncthreads();

  // labelClouds (radar_preprocessor.cpp:151-169): ascending label, input order inside a label
  bitonic_sort_u64(keys, npad);

  // cluster heads -> cstart[] via block scan (each thread owns a contiguous chunk)
  const int chunk = npad / BUILD_BLOCK > 0 ? npad / BUILD_BLOCK : 1;
  const int beg = tid * chunk;
  int heads = 0;
  for (int i = beg; i < beg + chunk && i < n; ++i) {
    bool head = (i == 0) || ((keys[i] >> 32) != (keys[i - 1] >> 32));
    heads += head ? 1 : 0;
  }
  int nc;
  int cbase = block_exclusive_scan_256(heads, scratch, &nc);
  for (int i = beg; i < beg + chunk && i < n; ++i) {
    bool head = (i == 0) || ((keys[i] >> 32) != (keys[i - 1] >> 32));
    if (head) cstart[cbase++] = i;
  }
  if (tid == 0) cstart[nc] = n;
  __syncthreads();

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
        for (int j = s; j < e; ++j) {
          const int id = (int)(uint32_t)keys[j];
          const float in = pi[id];
          m0 += px[id];
          m1 += py[id];
          m2 += in;
          maxi = in > maxi ? in : maxi;
        }
        const float nf = (float)(uint32_t)k;
        m0 = m0 / nf;
        m1 = m1 / nf;
        m2 = m2 / nf;
        float c00 = 0.f, c11 = 0.f, c22 = 0.f, c01 = 0.f, c02 = 0.f, c12 = 0.f;
        for (int j = s; j < e; ++j) {
          const int id = (int)(uint32_t)keys[j];
          const float d0 = px[id] - m0, d1 = py[id] - m1, d2 = pi[id] - m2;
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

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

int launch_ndt_build(randt_ctx* ctx, const float* d_points, int n_scans, int pitch, const int32_t* d_n_points,
                     int stride, int ioff, const randt_cluster_params* cp, const MapView& out, int first_map) {
  if (n_scans <= 0) return RANDT_OK;
  int npad = next_pow2(pitch < BUILD_BLOCK ? BUILD_BLOCK : pitch);
  size_t lds = (size_t)npad * 8 + (size_t)npad * 12 + (size_t)(npad + 1) * 4 + 64;
  if (lds > (size_t)ctx->lds_limit) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan too large for the LDS build kernel (max 4096 points)", hipSuccess);
  // Grid::cluster (grid.cpp:8-9)
  int row_size = (int)sqrt((double)cp->n_clusters);
  float resolution = cp->max_range * 2 / (float)row_size;
  RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ndt_build),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_ndt_build, dim3(n_scans), dim3(BUILD_BLOCK), lds, ctx->stream, d_points, pitch, d_n_points,
                     stride, ioff, row_size, resolution, out, first_map, npad);
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
