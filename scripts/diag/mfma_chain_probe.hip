// Diagnostics (r4): cycles per v_mfma_f32_32x32x16_f16 when ONE wave per SIMD issues a chain of MFMAs that each accumulate
// into the previous one's result, against 2 and 4 independent chains (and 2 waves per SIMD).  Answers: is a single
// accumulation chain per wave enough to keep the matrix pipe busy (32 cycles per MFMA)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain_probe scripts/diag/mfma_chain_probe.hip && /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* cyc, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f16v acc[CH];
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
typedef float f4v __attribute__((ext_vector_type(4)));
// the same for v_mfma_f32_16x16x32_f16 (fused_train16.h): half the work of the 32 x 32 x 16 shape per instruction
template <int CH>
__global__ void __launch_bounds__(256) probe16(float* out, unsigned long long* cyc, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f4v acc[CH];
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 4; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH> void run16(float* out, unsigned long long* cyc) {
  const int iters = 200;
  probe16<CH><<<256, 256>>>(out, cyc, iters);
  probe16<CH><<<256, 256>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long c = 0;
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("16 x 16 x 32: %d chain(s) per wave, one wave per SIMD: %.1f cycles per MFMA\n", CH, c / ((double)iters * 16 * CH));
}
template <int CH> void run(int waves_per_simd, float* out, unsigned long long* cyc) {
  const int iters = 200;
  // one workgroup of 256 threads per CU = 1 wave per SIMD; 512 workgroups -> 2 per CU (registers allow it) = 2 waves per SIMD
  probe<CH><<<256 * waves_per_simd, 256>>>(out, cyc, iters);
  probe<CH><<<256 * waves_per_simd, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * CH;
  printf("%d chain(s) per wave, %d wave(s) per SIMD: %.1f cycles per MFMA per wave -> pipe busy %.2f\n", CH, waves_per_simd, c / n,
         32.0 * waves_per_simd / (c / n));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 256 * 512 * 2); hipMalloc(&cyc, 8);
  run<1>(1, out, cyc); run<2>(1, out, cyc); run<4>(1, out, cyc);
  run<1>(2, out, cyc); run<2>(2, out, cyc);
  run16<1>(out, cyc); run16<2>(out, cyc); run16<4>(out, cyc);
  return 0;
}
