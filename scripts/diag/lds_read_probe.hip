// Diagnostics (r4): what does ds_read_b128 deliver per CU?  Every wave reads lane-linear 1-KiB fragments (lane * 16 + k * 1024:
// the weight fragments of fused_fwd.h / fused_train.h) out of a 64-KiB region, DEPTH reads in flight behind a counted
// lgkmcnt wait, 4 waves (one per SIMD) or 8 per CU; with and without an MFMA per fragment.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_read_probe scripts/diag/lds_read_probe.hip && /tmp/lds_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ int g_random_operands = 0;  // 1: operands with random mantissas (the matrix pipe's power depends on the bits that toggle)
template <int DEPTH, bool MFMA>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (g_random_operands) {
    unsigned x = 12345u + threadIdx.x * 2654435761u;
    for (int i = threadIdx.x; i < 65536 / 2; i += 256) {
      x = x * 1664525u + 1013904223u;
      ((_Float16*)smem)[i] = (_Float16)(((int)(x >> 9) % 2001 - 1000) * 0.001f);   // uniform in [-1, 1]
    }
  } else {
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 255);
  }
  __syncthreads();
  const unsigned base = (unsigned)(size_t)smem + (threadIdx.x & 63) * 16;
  h8 q[DEPTH];
  h8 b;
  for (int i = 0; i < 8; ++i) b[i] = g_random_operands ? (_Float16)((((threadIdx.x * 131 + i * 71) % 199) - 99) * 0.01f) : (_Float16)(0.002f * (i + 1));
  f16v acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  unsigned long long t0, t1;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[d]) : "v"(base), "n"(d * 1024));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      constexpr int dummy = 0; (void)dummy;
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(q[u % DEPTH]) : "n"(DEPTH - 1));
      if constexpr (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[u % DEPTH], b, acc, 0, 0, 0);
      else { acc[0] += (float)q[u % DEPTH][0]; }
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[u % DEPTH]) : "v"(base), "n"(((u + DEPTH) % 64) * 1024));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int d = 0; d < DEPTH; ++d) s += (float)q[d][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int DEPTH, bool MFMA> void run(int wg_per_cu, float* out, unsigned long long* cyc) {
  const int iters = 2000;   // (~2 ms per launch: long enough for the power management to act)
  hipFuncSetAttribute((const void*)probe<DEPTH, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int r = 0; r < 2; ++r) probe<DEPTH, MFMA><<<256 * wg_per_cu, 256, 65536>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 32;
  printf("%s depth %d, %d waves per CU: %.1f cycles per fragment per wave -> %.1f B/clk per CU\n", MFMA ? "read + MFMA" : "read only  ", DEPTH,
         4 * wg_per_cu, c / n, 1024.0 * 4 * wg_per_cu / (c / n));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 256 * 512 * 2); hipMalloc(&cyc, 8);
  run<1, false>(1, out, cyc); run<2, false>(1, out, cyc); run<4, false>(1, out, cyc); run<8, false>(1, out, cyc);
  run<2, false>(2, out, cyc); run<4, false>(2, out, cyc);
  for (int rnd = 0; rnd < 2; ++rnd) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_random_operands), &rnd, 4);
    printf("--- %s operands\n", rnd ? "random" : "smooth");
    run<1, true>(1, out, cyc); run<2, true>(1, out, cyc); run<4, true>(1, out, cyc); run<8, true>(1, out, cyc);
    run<2, true>(2, out, cyc); run<4, true>(2, out, cyc);
  }
  return 0;
}
