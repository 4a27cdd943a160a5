// Which clock do the "ticks" of tools/valu_rate_probe.hip count, and what does a wave64 fp64 FMA cost in it?
// (round-2 verdict, item 5: the probe prices v_fma_f64 at 3.33 ticks while the 78.6 TFLOP/s vector-fp64 figure means
// 16 lanes per clock per SIMD = 4.0 cycles.)
//
// Every wavefront brackets its loop with BOTH counters: s_memtime (clock64(): the shader clock, moves with DVFS) and
// s_memrealtime (wall_clock64(): the constant 100 MHz reference), and the host brackets the launches with HIP events.
// Three independent time bases for the same instruction stream:
//   shader ticks / instruction, realtime ticks -> seconds -> instructions per second per SIMD, event time -> the same.
// shader clock during the loop = d(s_memtime) / d(s_memrealtime) x 100 MHz.
// Run at 1, 2, 4 and 8 wavefronts per SIMD (256-thread workgroups = one wavefront per SIMD; 1 / 2 / 4 / 8 per CU).
//   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e = (x);                                                                 \
    if (e != hipSuccess) {                                                              \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);      \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

#define KERNEL(NAME, T, ASM)                                                                               \
  __global__ __launch_bounds__(256) void NAME(double* out, double seed) {                                  \
    T a[CHAINS];                                                                                           \
    const T b = (T)(seed * 1.0000001), c = (T)(seed * 0.5);                                                \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) a[i] = (T)(seed + i + threadIdx.x);                 \
    const long long r0 = wall_clock64();                                                                   \
    const long long t0 = clock64();                                                                        \
    for (int it = 0; it < ITERS; ++it) {                                                                   \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                                      \
    const long long t1 = clock64();                                                                        \
    const long long r1 = wall_clock64();                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {                                                 \
      out[1] = (double)(t1 - t0);                                                                          \
      out[2] = (double)(r1 - r0);                                                                          \
    }                                                                                                      \
    T s = 0;                                                                                               \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += a[i];                                          \
    if (s == (T)12345.678) out[0] = (double)s;                                                             \
  }

KERNEL(k_fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fma_f64, double, "v_fma_f64 %0, %0, %1, %2")
KERNEL(k_mul_f64, double, "v_mul_f64 %0, %0, %1")
KERNEL(k_rcp_f64, double, "v_rcp_f64 %0, %0")
KERNEL(k_rcp_f32, float, "v_rcp_f32 %0, %0")

typedef void (*kern_t)(double*, double);

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int wall_khz = 0;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  double* d_out;
  CHECK(hipMalloc(&d_out, 64));
  struct { const char* name; kern_t k; double spec_cycles; } probes[] = {
      {"v_fma_f32", k_fma_f32, 2.0}, {"v_fma_f64", k_fma_f64, 4.0}, {"v_mul_f64", k_mul_f64, 4.0},
      {"v_rcp_f32", k_rcp_f32, 8.0}, {"v_rcp_f64", k_rcp_f64, 16.0},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double real_hz = wall_khz > 0 ? wall_khz * 1e3 : 100e6;
  printf("device: %d CUs, clockRate %d kHz, wall clock (s_memrealtime) %d kHz\n", cus, prop.clockRate, wall_khz);
  printf("instruction,waves_per_simd,us_per_launch_events,shader_ticks_per_instr,realtime_us_in_kernel,shader_clock_GHz,"
         "Minstr_per_s_per_simd_events,cycles_per_instr_at_2.4GHz_events,spec_cycles,achieved_over_spec_rate_at_2.4GHz\n");
  for (auto& p : probes) {
    for (int wps : {1, 2, 4, 8}) {
      const dim3 grid(cus * wps), block(256);
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(p.k, grid, block, 0, 0, d_out, 1.5);
      CHECK(hipDeviceSynchronize());
      const int reps = 10;
      CHECK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(p.k, grid, block, 0, 0, d_out, 1.5);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      double h[3] = {0, 0, 0};
      CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
      const double instr_per_simd = (double)wps * ITERS * CHAINS;
      const double real_us = h[2] / real_hz * 1e6;
      const double shader_ghz = h[2] > 0 ? h[1] / (h[2] / real_hz) * 1e-9 : 0.0;
      const double rate = instr_per_simd / us;                         // M wave-instructions per second per SIMD
      const double cyc24 = us * 1e-6 * 2.4e9 / instr_per_simd;
      printf("%s,%d,%.2f,%.3f,%.2f,%.3f,%.1f,%.3f,%.1f,%.3f\n", p.name, wps, us, h[1] / instr_per_simd, real_us, shader_ghz, rate, cyc24,
             p.spec_cycles, p.spec_cycles / cyc24);
    }
  }
  return 0;
}
