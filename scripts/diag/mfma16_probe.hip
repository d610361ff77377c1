// Probe: operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950 (hipcc --offload-arch=gfx950 -o mfma16_probe mfma16_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A /*16x4 row-major*/, const float* B /*4x16 row-major*/, float* D /*16x16*/, int* ok) {
  const int l = threadIdx.x;
  // assumed: A lane l -> A[l % 16][l / 16]; B lane l -> B[l / 16][l % 16]; D lane l, reg r -> D[4 * (l / 16) + r][l % 16]
  const float a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = c[r];
}
int main() {
  float hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 11) - 5.f; hB[i] = (float)((i * 5) % 13) - 6.f; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int q = 0; q < 4; ++q) s += hA[i * 4 + q] * hB[q * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dD; int* dk;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024); hipMalloc(&dk, 4);
  hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dk);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("v_mfma_f32_16x16x4_f32 layout as assumed: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  return bad != 0;
}
