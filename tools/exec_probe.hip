// Does the gfx950 vector ALU skip the 16-lane passes of a wave64 instruction whose EXEC bits are all zero?
// (round-4 question behind the "transposed LM algebra" of k_solve: the solver algebra between two residual passes is
// wave-uniform fp64 work executed on all 64 lanes.  If a v_fma_f64 under EXEC = lanes 0..15 issued in one pass instead of
// four, masking the algebra would be worth as much as transposing it.)
//
// The same dependent-chain loop as tools/clock_probe.hip (8 chains of v_fma_f64 / v_rcp_f64 / v_fma_f32), run with 64, 32, 16 and 1
// active lanes per wavefront at 1 / 2 / 4 wavefronts per SIMD, timed with HIP events over the launch.
//   hipcc --offload-arch=gfx950 -O2 tools/exec_probe.hip -o /tmp/exec_probe && /tmp/exec_probe
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

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

#define KERNEL(NAME, T, ASM)                                                                                 \
  __global__ __launch_bounds__(256) void NAME(double* out, double seed, int active) {                        \
    T a[CHAINS];                                                                                             \
    const T b = (T)(seed * 1.0000001), c = (T)(seed * 0.5);                                                  \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) a[i] = (T)(seed + i + threadIdx.x);                   \
    if ((int)(threadIdx.x & 63) < active) {                                                                  \
      for (int it = 0; it < ITERS; ++it) {                                                                   \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
      }                                                                                                      \
    }                                                                                                        \
    T s = 0;                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += a[i];                                            \
    if (s == (T)12345.678) out[0] = (double)s;                                                               \
  }

KERNEL(k_fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fma_f64, double, "v_fma_f64 %0, %0, %1, %2")
KERNEL(k_rcp_f64, double, "v_rcp_f64 %0, %0")

typedef void (*kern_t)(double*, double, int);

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* d_out;
  CHECK(hipMalloc(&d_out, 64));
  struct { const char* name; kern_t k; double spec_cycles; } probes[] = {
      {"v_fma_f32", k_fma_f32, 2.0}, {"v_fma_f64", k_fma_f64, 4.0}, {"v_rcp_f64", k_rcp_f64, 16.0}};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("instruction,waves_per_simd,active_lanes,us_per_launch,cycles_per_instr_at_2.4GHz,spec_cycles_full_wave\n");
  for (auto& p : probes) {
    for (int wps : {1, 2, 4}) {
      for (int active : {64, 32, 16, 1}) {
        const dim3 grid(cus * wps), block(256);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(p.k, grid, block, 0, 0, d_out, 1.5, active);
        CHECK(hipDeviceSynchronize());
        const int reps = 10;
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(p.k, grid, block, 0, 0, d_out, 1.5, active);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const double instr_per_simd = (double)wps * ITERS * CHAINS;
        printf("%s,%d,%d,%.2f,%.3f,%.1f\n", p.name, wps, active, us, us * 1e-6 * 2.4e9 / instr_per_simd, p.spec_cycles);
      }
    }
  }
  return 0;
}
