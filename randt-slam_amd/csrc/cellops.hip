// Cell-level operations of the reference's Cell class, for callers that edit single cells through the facade
// (compiled with -ffp-contract=off, see cell_math.h).  Convenience paths: the batched map kernels are the hot path.
//
// Replaces (paths relative to /root/reference/ros/ndt_radar_slam/):
//   src/ndt_representation/ndt_cell.cpp:25-114        Cell::addPointCloud / updateCell incl. the recursive update (:84-89)
//   src/ndt_representation/ndt_cell.cpp:117-123       Cell::transformCell
//   src/ndt_representation/ndt_cell.cpp:158-169       Cell::mahalanobisSquared / mahalanobisSquaredIntensity
//   include/ndt_representation/ndt_cell.h:133-142     Cell::operator+=
#include "cell_math.h"

using namespace randt_dev;

namespace {

// Cell::mahalanobisSquared (ndt_cell.cpp:158-162): Eigen's 2x2 inverse is the adjugate times 1 / det; mu^T * inv first.
__device__ __forceinline__ float mahalanobis2f(const randt_cell& self, const randt_cell& sub) {
  const float s00 = sub.cov[0] + self.cov[0], s01 = sub.cov[1] + self.cov[1], s11 = sub.cov[3] + self.cov[3];
  const float mu0 = sub.mean[0] - self.mean[0], mu1 = sub.mean[1] - self.mean[1];
  const float det = s00 * s11 - s01 * s01;
  const float invdet = 1.0f / det;
  const float i00 = s11 * invdet, i01 = -s01 * invdet, i10 = -s01 * invdet, i11 = s00 * invdet;
  const float r0 = mu0 * i00 + mu1 * i10, r1 = mu0 * i01 + mu1 * i11;
  return r0 * mu0 + r1 * mu1;
}

// op 0: a[i] += b[i] (operator+=); 1: a[i] transformed by pose; 2 / 3: out[i] = a[i].mahalanobisSquared{Intensity,}(b[i])
__global__ __launch_bounds__(64) void k_cells_op(int op, randt_cell* a, const randt_cell* b, int n, const double* pose4, double* out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  randt_cell x = load_cell(a + i);
  if (op == 0) {
    cell_merge(x, load_cell(b + i));
    store_cell(a + i, x);
  } else if (op == 1) {
    float aff[4];
    pose_to_affine_f(pose4, aff);
    cell_transform(x, aff);
    store_cell(a + i, x);
  } else if (op == 2) {
    out[i] = (double)mahalanobis3f(x, load_cell(b + i));
  } else {
    out[i] = (double)mahalanobis2f(x, load_cell(b + i));
  }
}

// Cell::addPointCloud (ndt_cell.cpp:25-34) + updateCell (:36-114) for ONE cell that may already hold a distribution.
// Strictly sequential fp32 sums in point order (one lane): the reference's arithmetic spelled out.  Returns false (cell
// untouched) if the points are not taken (n + k <= min_points).
__device__ __forceinline__ bool cell_add_points(randt_cell& c, const float* pts, int k, int stride, int ioff, int min_points) {
  if (!((long long)c.n + (long long)k > (long long)min_points) || k <= 0) return false;
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, maxi = c.max_intensity;
  for (int j = 0; j < k; ++j) {
    const float* p = pts + (size_t)j * stride;
    m0 += p[0];
    m1 += p[1];
    m2 += p[ioff];
    maxi = p[ioff] > maxi ? p[ioff] : maxi;
  }
  const float nf = (float)(uint32_t)k;
  m0 = m0 / nf;
  m1 = m1 / nf;
  m2 = m2 / nf;
  float c00 = 0.f, c11 = 0.f, c22 = 0.f, c01 = 0.f, c02 = 0.f, c12 = 0.f;
  for (int j = 0; j < k; ++j) {
    const float* p = pts + (size_t)j * stride;
    const float d0 = p[0] - m0, d1 = p[1] - m1, d2 = p[ioff] - m2;
    c00 += (d0 * d0);
    c11 += (d1 * d1);
    c22 += (d2 * d2);
    c01 += (d0 * d1);
    c02 += (d0 * d2);
    c12 += (d1 * d2);
  }
  randt_cell add;
  add.mean[0] = m0;
  add.mean[1] = m1;
  add.mean[2] = m2;
  add.cov[0] = c00 / nf;
  add.cov[1] = c01 / nf;
  add.cov[2] = c02 / nf;
  add.cov[3] = c11 / nf;
  add.cov[4] = c12 / nf;
  add.cov[5] = c22 / nf;
  add.n = (uint32_t)k;
  add.max_intensity = 0.f;
  add.reserved = 0;
  if (c.n > 0) {
    cell_merge(c, add);  // the recursive update (:84-89) is operator+='s formula
  } else {
    c = add;
  }
  c.max_intensity = maxi;
  cell_regularize(c);
  return true;
}

__global__ __launch_bounds__(64) void k_cell_update(randt_cell* cell, const float* pts, int k, int stride, int ioff, int min_points,
                                                   int32_t* accepted) {
  if (threadIdx.x != 0) return;
  randt_cell c = load_cell(cell);
  if (!cell_add_points(c, pts, k, stride, ioff, min_points)) {
    *accepted = 0;
    return;
  }
  store_cell(cell, c);
  *accepted = 1;
}

// HierarchicalMap::addClusters (ndt_hierarchical_map.cpp:28-33): Map::insertCluster (ndt_map.cpp:238-245) for a whole list of
// already separated clusters in ONE launch -- cluster c = points [offsets[c], offsets[c + 1]) of one array.  A thread per cluster
// forms the cell (Cell::addPointCloud + updateCell on an empty cell: cell_add_points), the accepted ones are appended in CLUSTER
// ORDER (block scan of the accept flags = the reference's grid_.size() at each push_back) and every cell's slot points at it --
// a later cluster wins a shared slot, like the sequential loop (atomicMax on the compact index).  A cluster whose mean lies
// outside the index grid is dropped (std::vector::at throws there), one beyond the capacity too: counted in status[0 / 1].
#define INS_BLOCK 256
__global__ __launch_bounds__(INS_BLOCK) void k_maps_insert_clusters(MapView dst, int dst_idx, const float* __restrict__ pts, const int32_t* __restrict__ offsets,
                                                                   int n_clusters, int stride, int ioff, int32_t* __restrict__ status, int accumulate,
                                                                   int32_t* __restrict__ n_accepted) {
  __shared__ int s_wave[INS_BLOCK / 64];
  __shared__ int s_base, s_drop, s_out, s_acc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  randt_cell* dcells = dst.cells + (size_t)dst_idx * dst.cap;
  int32_t* grid = dst.grid ? dst.grid + (size_t)dst_idx * dst.n_slots : nullptr;
  if (tid == 0) {
    s_base = dst.counts[dst_idx];
    s_drop = s_out = s_acc = 0;
  }
  __syncthreads();
  for (int c0 = 0; c0 < n_clusters; c0 += INS_BLOCK) {
    const int c = c0 + tid;
    randt_cell cell;
    cell.n = 0;
    cell.max_intensity = 0.f;
    cell.reserved = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) cell.mean[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) cell.cov[i] = 0.f;
    bool take = false;
    uint32_t slot = 0;
    if (c < n_clusters) {
      const int first = offsets[c], k = offsets[c + 1] - first;
      const bool accepted = cell_add_points(cell, pts + (size_t)first * stride, k, stride, ioff, dst.min_points);
      if (accepted) {
        atomicAdd(&s_acc, 1);
        slot = grid ? coord_to_index(dst, cell.mean[0], cell.mean[1]) : 0u;
        if (grid && slot >= (uint32_t)dst.n_slots) atomicAdd(&s_out, 1);
        else take = true;
      }
    }
    // exclusive scan of `take` over the block, in cluster order
    const unsigned long long m = __ballot(take);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < INS_BLOCK / 64; ++w) {
      if (w < wave) before += s_wave[w];
      total += s_wave[w];
    }
    const int base = s_base;
    if (take) {
      const int at = base + before + in_wave;
      if (at < dst.cap) {
        store_cell(dcells + at, cell);
        if (grid) atomicMax(&grid[slot], at);
      } else {
        atomicAdd(&s_drop, 1);
      }
    }
    __syncthreads();
    if (tid == 0) s_base = base + total < dst.cap ? base + total : dst.cap;
    __syncthreads();
  }
  if (tid == 0) {
    dst.counts[dst_idx] = s_base;
    if (status) {
      if (accumulate) {
        if (s_drop) status[0] += s_drop;
        if (s_out) status[1] += s_out;
      } else {
        status[0] = s_drop;
        status[1] = s_out;
      }
    }
    if (n_accepted) *n_accepted = s_acc;
  }
}

// Cell::transformCellWithPointCloud's second half (ndt_cell.cpp:131-135): pcl::transformPointCloud of the cell's points with
// the Affine3f built from the 2-D pose (z row / column identity), fp32; intensity and z untouched.  PCL (un-vendored; 1.10 on
// ROS noetic) evaluates a transformed coordinate as m0 x + (m1 y + (m2 z + m3)) in its SSE path -- SPEC DECISION: that order.
__global__ __launch_bounds__(256) void k_points_transform(float* pts, int n, int stride, const double* pose4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float aff[4];
  pose_to_affine_f(pose4, aff);  // c, s, tx, ty as float (Sophus::SE2d::cast<float>().matrix())
  float* p = pts + (size_t)i * stride;
  const float x = p[0], y = p[1], z = p[2];
  p[0] = aff[0] * x + (-aff[1] * y + (0.0f * z + aff[2]));
  p[1] = aff[1] * x + (aff[0] * y + (0.0f * z + aff[3]));
}

}  // namespace

int launch_points_transform(randt_ctx* ctx, float* d_pts, int n, int stride, const double* d_pose4) {
  if (n <= 0) return RANDT_OK;
  hipLaunchKernelGGL(k_points_transform, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_pts, n, stride, d_pose4);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int launch_cells_op(randt_ctx* ctx, int op, randt_cell* d_a, const randt_cell* d_b, int n, const double* d_pose4, double* d_out) {
  if (n <= 0) return RANDT_OK;
  hipLaunchKernelGGL(k_cells_op, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, op, d_a, d_b, n, d_pose4, d_out);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int launch_maps_insert_clusters(randt_ctx* ctx, const MapView& dst, int dst_idx, const float* d_pts, const int32_t* d_offsets, int n_clusters,
                                int stride, int ioff, int32_t* d_status, int accumulate, int32_t* d_n_accepted) {
  if (n_clusters <= 0) return RANDT_OK;
  hipLaunchKernelGGL(k_maps_insert_clusters, dim3(1), dim3(INS_BLOCK), 0, ctx->stream, dst, dst_idx, d_pts, d_offsets, n_clusters, stride, ioff,
                     d_status, accumulate, d_n_accepted);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int launch_cell_update(randt_ctx* ctx, randt_cell* d_cell, const float* d_pts, int k, int stride, int ioff, int min_points,
                       int32_t* d_accepted) {
  hipLaunchKernelGGL(k_cell_update, dim3(1), dim3(64), 0, ctx->stream, d_cell, d_pts, k, stride, ioff, min_points, d_accepted);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
