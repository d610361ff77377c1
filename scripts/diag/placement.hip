// Diagnostic: where does the dispatcher put 512 workgroups of 256 threads / 80 KiB LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    out[3 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[3 * blockIdx.x + 2] = (unsigned)__builtin_amdgcn_s_memtime();
  }
  smem[threadIdx.x] = 1;
  for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);  // stay resident ~ 0.7 ms
}
int main() {
  const int n = 512;
  unsigned* d; hipMalloc(&d, 3 * n * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 81920, 0, d);
  hipDeviceSynchronize();
  std::vector<unsigned> h(3 * n); hipMemcpy(h.data(), d, 3 * n * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < n; ++b) cu[((h[3 * b + 1] & 15) << 8) | ((h[3 * b] >> 8) & 0xFF)].push_back(b);
  printf("distinct CU keys: %zu\n", cu.size());
  int hist[8] = {0}; for (auto& kv : cu) hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
  for (int i = 0; i < 8; ++i) if (hist[i]) printf("  CUs holding %d workgroups: %d\n", i, hist[i]);
  int shown = 0;
  for (auto& kv : cu) { if (shown++ >= 12) break; printf("key %03x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
  printf("hw_id sample: %08x xcc %08x\n", h[0], h[1]);
  return 0;
}
