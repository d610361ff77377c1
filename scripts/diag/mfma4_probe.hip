// Probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (hipcc --offload-arch=gfx950 -o mfma4_probe mfma4_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A /*16 blocks x 4*/, const float* B /*16 blocks x 4*/, float* D /*16 blocks x 4 x 4*/) {
  const int l = threadIdx.x;
  // assumed: A lane l -> block l / 4, row i = l % 4; B lane l -> block l / 4, column j = l % 4;
  //          D lane l, register r -> block l / 4, row i = r, column j = l % 4
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(l / 4) * 16 + r * 4 + l % 4] = c[r];
}
int main() {
  float hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 11) - 5.f; hB[i] = (float)((i * 5) % 13) - 6.f; }
  for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) ref[b * 16 + i * 4 + j] = hA[4 * b + i] * hB[4 * b + j];
  float *dA, *dB, *dD;
  (void)hipMalloc(&dA, 256); (void)hipMalloc(&dB, 256); (void)hipMalloc(&dD, 1024);
  (void)hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("v_mfma_f32_4x4x1_16b_f32 layout as assumed: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  return bad != 0;
}
