// mfma4_rate_probe.hip -- issue cadence of v_mfma_f32_4x4x1_16b_f32 and v_mfma_f32_16x16x4_f32 on gfx950: cycles per
// MFMA for one wave with A accumulators in rotation, and for W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma4_rate_probe.hip -o mfma4_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int A>
__global__ void __launch_bounds__(1024) rate_kernel(int n, unsigned long long* ticks, float* sink) {
  f32x4 acc[A];
#pragma unroll
  for (int i = 0; i < A; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        if (KIND == 4) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < A; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int KIND, int A>
void run(int waves, unsigned long long* dt, float* sink) {
  const int n = 2000;
  unsigned long long t = 0;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((rate_kernel<KIND, A>), dim3(1), dim3(64 * waves), 0, 0, n, dt, sink);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
  }
  const double per_wave = (double)n * 8 * A;
  printf("%s  accumulators %d  waves %2d (%d per SIMD): %.2f ticks per MFMA per wave, %.2f ticks per MFMA per SIMD\n", KIND == 4 ? "4x4x1  " : "16x16x4", A, waves,
         (waves + 3) / 4, (double)t / per_wave, (double)t / (per_wave * ((waves + 3) / 4)));
}

int main() {
  unsigned long long* dt; float* sink;
  (void)hipMalloc(&dt, 64); (void)hipMalloc(&sink, 4096);
  for (int waves : {1, 4, 8, 16}) {
    run<4, 1>(waves, dt, sink); run<4, 2>(waves, dt, sink); run<4, 4>(waves, dt, sink);
    run<16, 1>(waves, dt, sink); run<16, 2>(waves, dt, sink);
  }
  return 0;
}
