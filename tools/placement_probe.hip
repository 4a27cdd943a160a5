// Diagnostic: where do single-wavefront workgroups land?  Records HW_ID / XCC_ID per workgroup while
// all of them are co-resident (each spins ~200 us).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o /tmp/pp && /tmp/pp 512 64 4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ void k_probe(unsigned* out, long long spin_cycles, int waves_per_wg) {
  const long long t0 = wall_clock64();
  unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
  unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11));
  double x = threadIdx.x;
  while (wall_clock64() - t0 < spin_cycles) x = x * 1.0000001 + 1e-9;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * waves_per_wg + (threadIdx.x >> 6);
    out[2 * w] = hw;
    out[2 * w + 1] = xcc | (x == 12345.0 ? 0x80000000u : 0u);
  }
}

int main(int argc, char** argv) {
  const int n_wg = argc > 1 ? atoi(argv[1]) : 512, block = argc > 2 ? atoi(argv[2]) : 64, n_streams = argc > 3 ? atoi(argv[3]) : 1;
  const int wpw = block / 64;
  std::vector<hipStream_t> st(n_streams);
  std::vector<unsigned*> bufs(n_streams);
  for (int i = 0; i < n_streams; ++i) {
    hipStreamCreate(&st[i]);
    hipMalloc(&bufs[i], sizeof(unsigned) * 2 * n_wg * wpw);
  }
  for (int rep = 0; rep < 2; ++rep)
    for (int i = 0; i < n_streams; ++i) hipLaunchKernelGGL(k_probe, dim3(n_wg), dim3(block), 0, st[i], bufs[i], 20000LL /* 100 MHz clock: 200 us */, wpw);
  hipDeviceSynchronize();
  std::map<unsigned, int> per_simd, per_cu;
  for (int i = 0; i < n_streams; ++i) {
    std::vector<unsigned> h(2 * n_wg * wpw);
    hipMemcpy(h.data(), bufs[i], h.size() * 4, hipMemcpyDeviceToHost);
    for (int w = 0; w < n_wg * wpw; ++w) {
      const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned cu_key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
      per_cu[cu_key]++;
      per_simd[(cu_key << 2) | simd]++;
    }
  }
  std::map<int, int> hist_simd, hist_cu;
  for (auto& kv : per_simd) hist_simd[kv.second]++;
  for (auto& kv : per_cu) hist_cu[kv.second]++;
  printf("wgs=%d block=%d streams=%d: distinct CUs %zu, distinct SIMDs %zu\n", n_wg, block, n_streams, per_cu.size(), per_simd.size());
  printf("  waves per SIMD histogram:");
  for (auto& kv : hist_simd) printf(" %d:%d", kv.first, kv.second);
  printf("\n  waves per CU histogram:");
  for (auto& kv : hist_cu) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
