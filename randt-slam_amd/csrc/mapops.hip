// Whole-map operations for gfx950 (compiled with -ffp-contract=off, see cell_math.h) -- everything that edits a map without
// building one (the build itself: ndt_build.hip):
//   k_maps_transform  Map::transformMap                                  src/ndt_representation/ndt_map.cpp:177-182, ndt_cell.cpp:117-123
//   k_maps_merge      rolling-submap update (transform + Map::mergeMapCell), ordered     ndt_map.cpp:191-207, ndt_cell.h:133-142
//   k_maps_append     Map::insertCell / the tail of Map::insertCluster    ndt_map.h:137-140, ndt_map.cpp:242-243
//   k_maps_reindex    NOT in the reference: rebuild a transformed map's index grid
// (paths relative to /root/reference/ros/ndt_radar_slam/)
#include "cell_math.h"

using namespace randt_dev;

namespace {

__global__ __launch_bounds__(256) void k_maps_transform(MapView m, int first, int count, const double* __restrict__ pose4) {
  const int map = first + blockIdx.y;
  float aff[4];
  pose_to_affine_f(pose4 + 4 * blockIdx.y, aff);
  const int n = min(m.counts[map], m.cap);  // (caller-provided storage may hold anything: never walk past the capacity)
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
  // a batch of independent merges (randt_maps_merge_batch): workgroup p merges moving maps [moving_first + p n_moving, + n_moving)
  // into fixed map fixed_idx + p (one pair: blockIdx.x = 0)
  fixed_idx += blockIdx.x;
  moving_first += blockIdx.x * n_moving;
  pose4 += 4 * (size_t)blockIdx.x * n_moving;
  randt_cell* fcells = fixed.cells + (size_t)fixed_idx * fixed.cap;
  int32_t* fgrid = fixed.grid + (size_t)fixed_idx * fixed.n_slots;
  int n_cells = min(max(fixed.counts[fixed_idx], 0), fixed.cap);
  for (int t = 0; t < n_moving; ++t) {
    const int mmap = moving_first + t;
    const randt_cell* mcells = moving.cells + (size_t)mmap * moving.cap;
    const int M = min(moving.counts[mmap], moving.cap);  // slots[] holds moving.cap entries
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

// Map::insertCell / the tail of Map::insertCluster (ndt_map.h:137-140, ndt_map.cpp:242-243): append the cells of one
// map to another in order; with set_grid the slot of each cell's mean points at its new compact index.  Sequential
// by construction (a later cell overwrites the slot of an earlier one), so one lane does it.  status[0] counts cells
// dropped for capacity, status[1] cells whose mean lies outside the index grid (std::vector::at throws there).
__global__ __launch_bounds__(64) void k_maps_append(MapView dst, int dst_idx, MapView src, int src_idx, int set_grid,
                                                    int32_t* __restrict__ status, int accumulate) {
  if (threadIdx.x != 0) return;
  const int n_src = min(src.counts[src_idx], src.cap);
  int n = dst.counts[dst_idx];
  randt_cell* dcells = dst.cells + (size_t)dst_idx * dst.cap;
  const randt_cell* scells = src.cells + (size_t)src_idx * src.cap;
  int32_t* grid = dst.grid ? dst.grid + (size_t)dst_idx * dst.n_slots : nullptr;
  int dropped = 0, outside = 0;
  for (int i = 0; i < n_src; ++i) {
    const randt_cell c = load_cell(scells + i);
    if (set_grid && grid) {
      const uint32_t slot = coord_to_index(dst, c.mean[0], c.mean[1]);
      if (slot >= (uint32_t)dst.n_slots) {
        ++outside;
        continue;
      }
      if (n >= dst.cap) {
        ++dropped;
        continue;
      }
      grid[slot] = n;
    } else if (n >= dst.cap) {
      ++dropped;
      continue;
    }
    store_cell(dcells + n, c);
    ++n;
  }
  dst.counts[dst_idx] = n;
  if (status) {
    if (accumulate) {  // asynchronous inserts: nobody reads the words between two calls (api.hip, deferred_status)
      if (dropped) status[0] += dropped;
      if (outside) status[1] += outside;
    } else {
      status[0] = dropped;
      status[1] = outside;
    }
  }
}

// Rebuild the index grid of a map from its cells' current means (later cells win a shared slot, as insertion order
// would have it).  The reference never does this after Map::transformMap (ndt_map.cpp:177-182 leaves grid_indizes_
// stale); this is the opt-in repair.  One workgroup per map.
__global__ __launch_bounds__(256) void k_maps_reindex(MapView m, int first) {
  const int map = first + blockIdx.x;
  if (!m.grid) return;
  int32_t* grid = m.grid + (size_t)map * m.n_slots;
  const randt_cell* cells = m.cells + (size_t)map * m.cap;
  const int n = min(m.counts[map], m.cap);
  for (int s = threadIdx.x; s < m.n_slots; s += 256) grid[s] = -1;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint32_t slot = coord_to_index(m, cells[i].mean[0], cells[i].mean[1]);
    if (slot < (uint32_t)m.n_slots) atomicMax(&grid[slot], i);
  }
}

int launch_maps_reindex(randt_ctx* ctx, const MapView& m, int first, int count) {
  if (count <= 0) return RANDT_OK;
  hipLaunchKernelGGL(k_maps_reindex, dim3(count), dim3(256), 0, ctx->stream, m, first);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int launch_maps_append(randt_ctx* ctx, const MapView& dst, int dst_idx, const MapView& src, int src_idx, int set_grid,
                       int32_t* d_status, int accumulate) {
  hipLaunchKernelGGL(k_maps_append, dim3(1), dim3(64), 0, ctx->stream, dst, dst_idx, src, src_idx, set_grid, d_status, accumulate);
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
                      int n_moving, const double* d_pose4, int n_pairs) {
  if (n_moving <= 0 || n_pairs <= 0) return RANDT_OK;
  size_t lds = (size_t)moving.cap * 4 + 64;
  if (lds > (size_t)ctx->lds_limit) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "moving map capacity too large for merge kernel", hipSuccess);
  // the attribute belongs to the kernel's code object on ONE device: remembered per context (a context is one device and one
  // caller), raised, never lowered -- not in a process-wide static that a second GPU or thread would trust
  if (lds > ctx->merge_lds_granted) {
    RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_maps_merge),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->merge_lds_granted = lds;
  }
  hipLaunchKernelGGL(k_maps_merge, dim3(n_pairs), dim3(256), lds, ctx->stream, fixed, fixed_idx, moving, moving_first,
                     n_moving, d_pose4);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
