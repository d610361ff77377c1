// cold_stream_probe.hip -- how fast does ONE workgroup (16 waves, 16-byte loads, two 1-KiB loads in flight per wave) stream a
// buffer that is NOT in its XCD's L2 -- the state in which every chain-kernel launch finds its weight streams, since a kernel
// boundary invalidates L2 -- and does it help when other workgroups of the same launch touch the buffer's lines first
// (the chain kernels' "prefetcher" workgroups)?   hipcc --offload-arch=gfx950 -O3 cold_stream_probe.hip -o cold_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(1024) probe(const f32x4* __restrict__ src, long long frags_per_wave, long long lines, int npref,
                                              unsigned long long* ticks, float* sink) {
  if (blockIdx.x >= 8) {  // prefetchers: blocks 8 + 8 p + x touch every 128-byte line once (as chain_prefetch does)
    const int p = (blockIdx.x - 8) >> 3;
    for (long long i = (long long)p * blockDim.x + threadIdx.x; i < lines; i += (long long)npref * blockDim.x) {
      unsigned v;
      asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(reinterpret_cast<const char*>(src) + (i << 7)) : "memory");
    }
    return;
  }
  // consumers: one workgroup per XCD (blocks 0..7), all streaming the same buffer
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* p = src + (long long)wave * frags_per_wave * 64 + lane;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (long long f = 0; f < frags_per_wave; f += 2) {
    const f32x4 a = p[f * 64], b = p[(f + 1) * 64];
    acc += a; acc += b;
  }
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[threadIdx.x] = acc[0];
}
__global__ void scribble(float* p, long long n, float v) {  // rewrites the buffer from every XCD: the next launch finds it beyond L2
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

int main() {
  const long long bytes = 1280 * 1024;
  f32x4* d; unsigned long long* dt; float* sink;
  (void)hipMalloc(&d, bytes); (void)hipMalloc(&dt, 64); (void)hipMalloc(&sink, 4096);
  const long long fpw = bytes / 1024 / 16;
  for (int npref : {0, 1, 4, 8, 16}) {
    unsigned long long t[8] = {0};
    double sum = 0;
    for (int it = 0; it < 6; ++it) {
      hipLaunchKernelGGL(scribble, dim3(256), dim3(256), 0, 0, (float*)d, bytes / 4, (float)it);
      hipLaunchKernelGGL(probe, dim3(8 + 8 * npref), dim3(1024), 0, 0, d, fpw, bytes / 128, npref, dt, sink);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(t, dt, 64, hipMemcpyDeviceToHost);
      if (it) { double m = 0; for (int x = 0; x < 8; ++x) m += (double)t[x]; sum += m / 8; }
    }
    printf("cold buffer of %lld KiB, one streaming workgroup per XCD, %2d prefetcher workgroups per XCD: %.0f cycles = %.1f B per cycle per workgroup\n",
           bytes / 1024, npref, sum / 5, (double)bytes / (sum / 5));
  }
  return 0;
}
