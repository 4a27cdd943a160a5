// What can a read-only streaming kernel reach on this MI355X, with the access pattern of k_filter_rows (filter.hip)?
// The polar filter's roofline line prices the stage against 8 TB/s; the guide's measured figure is 6.29 TB/s for a float4
// COPY.  This probe puts a number on a pure READ of the filter's own input shape -- 16 scans x 400 rows x 3000 bins x 16 B =
// 307 MB -- so that "how far is the row kernel from what the memory system gives a kernel of its shape" has an answer:
//   rows     one 48 KB row per workgroup visit, 256 threads, U float4 loads per lane in flight (U = 12: a whole row),
//            workgroups walk rows with stride G like the row kernel; the loaded values are max-reduced per lane and one
//            value per workgroup is written (no barrier, no LDS): the kernel with everything but its loads removed
//   flat     grid-stride float4 loop, U loads in flight per lane
// for several grid sizes / occupancies.  HIP events over REPS launches on one stream; a cold pass first.
//   hipcc --offload-arch=gfx950 -O2 tools/hbm_stream_probe.hip -o /tmp/hbm_stream_probe && /tmp/hbm_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int N_SCANS = 16, N_AZ = 400, N_BINS = 3000;

template <int U, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_rows(const float4* __restrict__ raw, float* out, int n_rows) {
  float best = 0.f;
  for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const float4* rp = raw + (long long)row * N_BINS;
    float4 pt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = threadIdx.x + u * 256;
      pt[u] = rp[b < N_BINS ? b : N_BINS - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) best = fmaxf(best, fmaxf(fmaxf(pt[u].x, pt[u].y), fmaxf(pt[u].z, pt[u].w)));
  }
  if (best == 12345.678f) out[blockIdx.x] = best;
}

template <int U, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_flat(const float4* __restrict__ raw, float* out, long long n) {
  float best = 0.f;
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long i0 = (long long)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += stride) {
    float4 pt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + (long long)u * 256;
      pt[u] = raw[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) best = fmaxf(best, fmaxf(fmaxf(pt[u].x, pt[u].y), fmaxf(pt[u].z, pt[u].w)));
  }
  if (best == 12345.678f) out[blockIdx.x] = best;
}

// one wavefront per row, the row in chunks of U x 64 bins (12 KB contiguous per wavefront for U = 12): k_filter_rows' loads;
// VISIT = 1 adds its per-point arithmetic (cross / dot organisation test, fp64 squared range, arg-max)
template <int U, int WPE, int VISIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wave_rows(const float4* __restrict__ raw, float* out, int n_rows, double lo2, double hi2) {
  const int lane = threadIdx.x & 63;
  const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_rows) return;
  const float4* rp = raw + g * N_BINS;
  float best = 0.f, x0 = 0.f, y0 = 0.f;
  int best_idx = 0x7fffffff, bad = 0;
  for (int c0 = 0; c0 < N_BINS; c0 += U * 64) {
    float4 pt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = c0 + u * 64 + lane;
      pt[u] = rp[b < N_BINS ? b : N_BINS - 1];
    }
    if (c0 == 0) {
      x0 = __shfl(pt[0].x, 0, 64);
      y0 = __shfl(pt[0].y, 0, 64);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (VISIT) {
        const int b = c0 + u * 64 + lane;
        const float px = pt[u].x, py = pt[u].y, pin = pt[u].w;
        const float cross = x0 * py - y0 * px, dot = x0 * px + y0 * py;
        if (!(fabsf(cross) <= 4e-5f * dot)) bad = 1;
        const double d2 = (double)px * (double)px + (double)py * (double)py;
        if (b < N_BINS && d2 >= lo2 && d2 <= hi2 && pin > best) {
          best = pin;
          best_idx = b;
        }
      } else {
        best = fmaxf(best, fmaxf(fmaxf(pt[u].x, pt[u].y), fmaxf(pt[u].z, pt[u].w)));
      }
    }
  }
  if (best == 12345.678f || bad == 77) out[g] = best + best_idx;
}

template <typename F>
double time_us(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return 1e3 * ms / reps;
}

int main() {
  const long long n_rows = (long long)N_SCANS * N_AZ, n = n_rows * N_BINS;
  const double bytes = (double)n * 16.0;
  float4* d_raw;
  float* d_out;
  CHECK(hipMalloc(&d_raw, (size_t)n * 16));
  CHECK(hipMalloc(&d_out, 1 << 20));
  CHECK(hipMemset(d_raw, 0x3c, (size_t)n * 16));
  const int reps = 20;
  printf("kernel,loads_in_flight,waves_per_simd,workgroups,us_per_launch,TB_per_s,frac_of_8\n");
  auto report = [&](const char* name, int u, int wpe, int g, double us) {
    printf("%s,%d,%d,%d,%.2f,%.3f,%.3f\n", name, u, wpe, g, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0);
  };
  for (int g : {928, 1024, 1600, 2048, 3200, 6400}) {
    report("rows", 12, 4, g, time_us([&] { hipLaunchKernelGGL((k_rows<12, 4>), dim3(g), dim3(256), 0, 0, d_raw, d_out, (int)n_rows); }, reps));
  }
  for (int g : {1536, 2048, 3200, 6400}) {
    report("rows", 12, 8, g, time_us([&] { hipLaunchKernelGGL((k_rows<12, 8>), dim3(g), dim3(256), 0, 0, d_raw, d_out, (int)n_rows); }, reps));
  }
  for (int g : {1024, 2048, 4096, 8192}) {
    report("flat", 4, 8, g, time_us([&] { hipLaunchKernelGGL((k_flat<4, 8>), dim3(g), dim3(256), 0, 0, d_raw, d_out, n); }, reps));
    report("flat", 8, 8, g, time_us([&] { hipLaunchKernelGGL((k_flat<8, 8>), dim3(g), dim3(256), 0, 0, d_raw, d_out, n); }, reps));
    report("flat", 12, 4, g, time_us([&] { hipLaunchKernelGGL((k_flat<12, 4>), dim3(g), dim3(256), 0, 0, d_raw, d_out, n); }, reps));
  }
  {
    const int g = (int)((n_rows + 3) / 4);
    report("wave_rows", 12, 4, g, time_us([&] { hipLaunchKernelGGL((k_wave_rows<12, 4, 0>), dim3(g), dim3(256), 0, 0, d_raw, d_out, (int)n_rows, 1.0, 1e9); }, reps));
    report("wave_rows_visit", 12, 4, g, time_us([&] { hipLaunchKernelGGL((k_wave_rows<12, 4, 1>), dim3(g), dim3(256), 0, 0, d_raw, d_out, (int)n_rows, 1.0, 1e9); }, reps));
    report("wave_rows_visit", 6, 8, g, time_us([&] { hipLaunchKernelGGL((k_wave_rows<6, 8, 1>), dim3(g), dim3(256), 0, 0, d_raw, d_out, (int)n_rows, 1.0, 1e9); }, reps));
  }
  {
    // the same kernels over 64 scans (1.23 GB: five times the 256 MB Infinity Cache), per 16 scans: what the memory gives
    // when consecutive launches cannot meet their own lines in the cache
    float4* d_big;
    const long long rows4 = 4 * n_rows;
    CHECK(hipMalloc(&d_big, (size_t)rows4 * N_BINS * 16));
    CHECK(hipMemset(d_big, 0x3c, (size_t)rows4 * N_BINS * 16));
    const int g = (int)((rows4 + 3) / 4);
    report("wave_rows_64scans_per16", 12, 4, g, time_us([&] { hipLaunchKernelGGL((k_wave_rows<12, 4, 0>), dim3(g), dim3(256), 0, 0, d_big, d_out, (int)rows4, 1.0, 1e9); }, reps) / 4);
    report("wave_rows_visit_64scans_per16", 12, 4, g, time_us([&] { hipLaunchKernelGGL((k_wave_rows<12, 4, 1>), dim3(g), dim3(256), 0, 0, d_big, d_out, (int)rows4, 1.0, 1e9); }, reps) / 4);
    report("rows_64scans_per16", 12, 4, 2048, time_us([&] { hipLaunchKernelGGL((k_rows<12, 4>), dim3(2048), dim3(256), 0, 0, d_big, d_out, (int)rows4); }, reps) / 4);
    // ... and as FOUR launches of 16 scans each over the four quarters of that buffer (what a stream of filter launches over
    // distinct inputs does): every launch pays its own fill and drain
    {
      const int gq = (int)((n_rows + 3) / 4);
      int turn = 0;
      report("wave_rows_16scans_rotating_quarters", 12, 4, gq, time_us([&] {
        hipLaunchKernelGGL((k_wave_rows<12, 4, 0>), dim3(gq), dim3(256), 0, 0, d_big + (size_t)(turn++ & 3) * n, d_out, (int)n_rows, 1.0, 1e9);
      }, reps));
      turn = 0;
      report("wave_rows_visit_16scans_rotating_quarters", 12, 4, gq, time_us([&] {
        hipLaunchKernelGGL((k_wave_rows<12, 4, 1>), dim3(gq), dim3(256), 0, 0, d_big + (size_t)(turn++ & 3) * n, d_out, (int)n_rows, 1.0, 1e9);
      }, reps));
    }
    CHECK(hipFree(d_big));
  }
  {
    // the same 307 MB filled with pseudo-random finite floats instead of a constant byte
    unsigned int* h = (unsigned int*)malloc((size_t)n * 16);
    unsigned int x = 12345u;
    for (long long i = 0; i < n * 4; ++i) {
      x = x * 1664525u + 1013904223u;
      h[i] = 0x3f000000u | (x >> 9);  // [0.5, 1)
    }
    CHECK(hipMemcpy(d_raw, h, (size_t)n * 16, hipMemcpyHostToDevice));
    free(h);
    const int g = (int)((n_rows + 3) / 4);
    report("wave_rows_random_data", 12, 4, g, time_us([&] { hipLaunchKernelGGL((k_wave_rows<12, 4, 0>), dim3(g), dim3(256), 0, 0, d_raw, d_out, (int)n_rows, 1.0, 1e9); }, reps));
    report("wave_rows_visit_random_data", 12, 4, g, time_us([&] { hipLaunchKernelGGL((k_wave_rows<12, 4, 1>), dim3(g), dim3(256), 0, 0, d_raw, d_out, (int)n_rows, 1.0, 1e9); }, reps));
    report("rows_random_data", 12, 4, 2048, time_us([&] { hipLaunchKernelGGL((k_rows<12, 4>), dim3(2048), dim3(256), 0, 0, d_raw, d_out, (int)n_rows); }, reps));
  }
  // one workgroup per row, all rows in one grid (the round-2 shape)
  report("rows_one_per_wg", 12, 4, (int)n_rows, time_us([&] { hipLaunchKernelGGL((k_rows<12, 4>), dim3((int)n_rows), dim3(256), 0, 0, d_raw, d_out, (int)n_rows); }, reps));
  return 0;
}
