// Issue cost of the VALU instruction classes the registration kernels are made of, measured at saturation on gfx950:
// every SIMD holds 4 wavefronts, each running 8 independent chains of ONE instruction kind.  The result is printed as
// SIMD-cycles per wave-instruction, calibrated on v_fma_f32 = 2 cycles (MI355X_MICROARCH.md: a wave64 VALU instruction
// issues over 2 cycles on a SIMD-32) and, independently, on the wall clock at the nominal 2.4 GHz.
// These constants turn the SQ_INSTS_VALU_* counters of profiles/r02_*_sq_summary.csv into the VALU-issue cycles behind
// bench.py's `roofline` (bound "valu_issue").
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                          \
  do {                                                                    \
    hipError_t e = (x);                                                   \
    if (e != hipSuccess) {                                                \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                            \
    }                                                                     \
  } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

// 64-bit destination / sources
#define KERNEL_D(NAME, ASM)                                                              \
  __global__ __launch_bounds__(256) void NAME(double* out, double seed) {                \
    double a[CHAINS];                                                                    \
    const double b = seed * 1.0000001, c = seed * 0.5;                                   \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) a[i] = seed + i + threadIdx.x;    \
    const long long t0 = clock64();                                                      \
    for (int it = 0; it < ITERS; ++it) {                                                 \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                    \
    const long long t1 = clock64();                                                      \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) out[1] = (double)(t1 - t0);     \
    double s = 0;                                                                        \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += a[i];                        \
    if (s == 12345.678) out[0] = s;                                                      \
  }
// 32-bit
#define KERNEL_F(NAME, ASM)                                                              \
  __global__ __launch_bounds__(256) void NAME(double* out, double seed) {                \
    float a[CHAINS];                                                                     \
    const float b = (float)seed * 1.0000001f, c = (float)seed * 0.5f;                    \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) a[i] = (float)seed + i + threadIdx.x; \
    const long long t0 = clock64();                                                      \
    for (int it = 0; it < ITERS; ++it) {                                                 \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                    \
    const long long t1 = clock64();                                                      \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) out[1] = (double)(t1 - t0);     \
    float s = 0;                                                                         \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += a[i];                        \
    if (s == 12345.678f) out[0] = s;                                                     \
  }
// f32 -> f64 conversion: 64-bit destination, 32-bit source
#define KERNEL_CVT(NAME, ASM)                                                            \
  __global__ __launch_bounds__(256) void NAME(double* out, double seed) {                \
    double a[CHAINS];                                                                    \
    float f[CHAINS];                                                                     \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { a[i] = 0; f[i] = (float)seed + i + threadIdx.x; } \
    const long long t0 = clock64();                                                      \
    for (int it = 0; it < ITERS; ++it) {                                                 \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(f[i])); \
    }                                                                                    \
    const long long t1 = clock64();                                                      \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) out[1] = (double)(t1 - t0);     \
    double s = 0;                                                                        \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += a[i];                        \
    if (s == 12345.678) out[0] = s;                                                      \
  }

KERNEL_F(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL_F(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL_F(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL_F(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_F(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_F(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_F(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F(k_permlane32_swap, "v_permlane32_swap_b32 %0, %1")
KERNEL_D(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_D(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_D(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_D(k_max_f64, "v_max_f64 %0, %0, %1")
KERNEL_D(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_D(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_D(k_lshl_b64, "v_lshlrev_b64 %0, 1, %0")
KERNEL_CVT(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
// 64-bit ALU with the DPP row_newbcast modifier (gfx90a+: the only DPP control the DP ALU takes) -- the window solve's
// pivot-row broadcast (csrc/window.hip, pivot_group)
KERNEL_D(k_fmac_f64, "v_fmac_f64 %0, %1, %2")
KERNEL_D(k_fmac_f64_dpp, "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf")
KERNEL_D(k_mov_b64_dpp, "v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf")

__global__ __launch_bounds__(256) void k_readlane(double* out, double seed) {
  float a = (float)seed + threadIdx.x;
  int acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      int s;
      asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a));
      acc += s;
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) out[1] = (double)(t1 - t0);
  if (acc == 123457) out[0] = acc;
}

typedef void (*kern_t)(double*, double);
struct Probe {
  const char* name;
  kern_t k;
  const char* cls;
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int simds = cus * 4, waves_per_simd = 4;
  double* d_out;
  CHECK(hipMalloc(&d_out, 64));
  Probe probes[] = {
      {"v_fma_f32", k_fma_f32, "fp32"},       {"v_mul_f32", k_mul_f32, "fp32"},        {"v_mov_b32", k_mov_b32, "move"},
      {"v_mov_b32_dpp", k_mov_dpp, "move"},   {"v_add_u32", k_add_u32, "int32"},       {"v_mul_lo_u32", k_mul_lo_u32, "int32"},
      {"v_cndmask_b32", k_cndmask, "move"},   {"v_rcp_f32", k_rcp_f32, "trans_f32"},   {"v_permlane32_swap", k_permlane32_swap, "move"},
      {"v_readlane_b32", k_readlane, "move"}, {"v_fma_f64", k_fma_f64, "fp64"},        {"v_mul_f64", k_mul_f64, "fp64"},
      {"v_add_f64", k_add_f64, "fp64"},       {"v_max_f64", k_max_f64, "fp64"},        {"v_rcp_f64", k_rcp_f64, "trans_f64"},
      {"v_rsq_f64", k_rsq_f64, "trans_f64"},  {"v_lshlrev_b64", k_lshl_b64, "int64"},  {"v_cvt_f64_f32", k_cvt_f64_f32, "cvt"},
      {"v_fmac_f64", k_fmac_f64, "fp64"},     {"v_fmac_f64_dpp(row_newbcast)", k_fmac_f64_dpp, "fp64_dpp"},
      {"v_mov_b64_dpp(row_newbcast)", k_mov_b64_dpp, "move64_dpp"},
  };
  const int n = sizeof(probes) / sizeof(probes[0]);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const dim3 grid(cus * waves_per_simd), block(256);  // 4 waves per workgroup = one per SIMD; waves_per_simd workgroups per CU
  double t_ref = 0.0;
  printf("device: %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  printf("instruction,class,us_per_launch,cycles_per_wave_instr_vs_fma_f32,cycles_per_wave_instr_at_2.4GHz,s_memtime_ticks_per_wave_instr,ticks_per_us\n");
  for (int p = 0; p < n; ++p) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probes[p].k, grid, block, 0, 0, d_out, 1.5);
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probes[p].k, grid, block, 0, 0, d_out, 1.5);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    if (p == 0) t_ref = us;
    const double instr_per_simd = (double)waves_per_simd * ITERS * CHAINS;
    double h[2] = {0, 0};
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    printf("%s,%s,%.2f,%.3f,%.3f,%.3f,%.1f\n", probes[p].name, probes[p].cls, us, 2.0 * us / t_ref, us * 1e-6 * 2.4e9 / instr_per_simd,
           h[1] / instr_per_simd, h[1] / us);
  }
  // shader clock seen by a nearly idle chip: ONE workgroup (the fixed-lag window solve is such a launch)
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_fma_f64, dim3(1), block, 0, 0, d_out, 1.5);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_fma_f64, dim3(1), block, 0, 0, d_out, 1.5);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double h[2] = {0, 0};
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    printf("single_workgroup_v_fma_f64,us_per_launch %.2f,ticks_per_wave_instr %.3f (1 wave per SIMD),ticks_per_us %.1f\n", ms * 1e3 / 20,
           h[1] / ((double)ITERS * CHAINS), h[1] / (ms * 1e3 / 20));
  }
  // the same instruction kinds issued by ONE wavefront on an otherwise idle chip (8 independent chains): what the serial
  // phases of the window solve pay per instruction
  printf("lone_wavefront: instruction,ticks_per_wave_instr,ticks_per_us\n");
  for (int p = 0; p < n; ++p) {
    hipLaunchKernelGGL(probes[p].k, dim3(1), dim3(64), 0, 0, d_out, 1.5);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probes[p].k, dim3(1), dim3(64), 0, 0, d_out, 1.5);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double h[2] = {0, 0};
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    printf("lone_wavefront: %s,%.3f,%.1f\n", probes[p].name, h[1] / ((double)ITERS * CHAINS), h[1] / (ms * 1e3));
  }
  (void)simds;
  return 0;
}
