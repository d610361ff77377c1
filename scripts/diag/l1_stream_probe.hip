// l1_stream_probe.hip -- how fast can ONE workgroup stream an L2-resident buffer into registers (16-byte loads, 1 KiB per
// wave-instruction), as a function of waves per CU and loads in flight per wave?  (The fp32 chain kernels are bound by
// this path at small batches: train_chain32s.h.)   hipcc --offload-arch=gfx950 -O3 l1_stream_probe.hip -o l1_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ void __launch_bounds__(1024) stream_kernel(const f32x4* __restrict__ src, long long frags_per_wave, int reps, unsigned long long* ticks, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* p = src + (long long)wave * frags_per_wave * 64 + lane;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned long long t0 = 0, t1 = 0;
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    if (r == 1) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");  // pass 0 warms L2
    for (long long f = 0; f < frags_per_wave; f += D) {
      f32x4 v[D];
#pragma unroll
      for (int j = 0; j < D; ++j) v[j] = p[(f + j) * 64];
#pragma unroll
      for (int j = 0; j < D; ++j) acc += v[j];
    }
  }
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[threadIdx.x] = acc[0];
}

int main() {
  const long long bytes = 1408 * 1024;  // ~ one direction of the autoencoder's fp32 stream
  f32x4* d; unsigned long long* dt; float* sink;
  hipMalloc(&d, bytes); hipMemset(d, 0, bytes); hipMalloc(&dt, 64); hipMalloc(&sink, 4096);
  const int reps = 5;
  for (int waves : {4, 8, 12, 16}) {
    const long long fpw = bytes / 1024 / waves / 16 * 16;
    for (int D : {2, 4, 8, 16}) {
      unsigned long long t = 0;
      for (int it = 0; it < 2; ++it) {
        if (D == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(1), dim3(64 * waves), 0, 0, d, fpw, reps, dt, sink);
        if (D == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(1), dim3(64 * waves), 0, 0, d, fpw, reps, dt, sink);
        if (D == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(1), dim3(64 * waves), 0, 0, d, fpw, reps, dt, sink);
        if (D == 16) hipLaunchKernelGGL(stream_kernel<16>, dim3(1), dim3(64 * waves), 0, 0, d, fpw, reps, dt, sink);
        hipDeviceSynchronize();
        hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
      }
      const double moved = (double)fpw * waves * 1024 * (reps - 1);
      printf("waves %2d  loads in flight/wave %2d : %8llu cycles (s_memtime) for %.0f KiB x %d passes -> %.1f B per cycle\n", waves, D, t, moved / (reps - 1) / 1024, reps - 1, moved / (double)t);
    }
  }
  return 0;
}
