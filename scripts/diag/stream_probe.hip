// Diagnostic: how fast can ONE CU pull a weight stream that every workgroup reads (L2-resident, 1 MiB)?
// The training chain kernel is bound by exactly this (DESIGN.md K3''): every workgroup streams the whole
// packed weight set.  Variants:
//   reg  : W waves, each keeps U 1-KiB fragment loads (16 B per lane) in flight into registers
//   dma  : W loader waves, LDS-DMA (global_load_lds_dwordx4) into an LDS ring, D pieces in flight per wave
// Prints GB/s per workgroup for grids of 8 / 128 / 256 workgroups (one per CU).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/diag/stream_probe scripts/diag/stream_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void __launch_bounds__(1024) reg_kernel(const u32x4* __restrict__ src, int nfrag, int reps, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (int f0 = wave * U; f0 < nfrag; f0 += nw * U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[(size_t)((f0 + u) % nfrag) * 64 + lane];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if (acc[0] == 0x12345678u && acc[1] == 77u) sink[0] = acc[2] ^ acc[3];
}

// double-buffered register variant: loads of group g+1 are issued before group g is consumed
template <int U>
__global__ void __launch_bounds__(1024) reg2_kernel(const u32x4* __restrict__ src, int nfrag, int reps, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  const int groups = nfrag / (nw * U) * reps;
  u32x4 a[U], b[U];
  int g = 0;
  auto fr = [&](int gi, int u) { return (size_t)(((gi % (nfrag / (nw * U))) * nw + wave) * U + u) * 64 + lane; };
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = src[fr(0, u)];
  for (; g + 2 <= groups; g += 2) {
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = src[fr(g + 1, u)];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= a[u];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = src[fr(g + 2, u)];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= b[u];
  }
  if (acc[0] == 0x12345678u && acc[1] == 77u) sink[0] = acc[2] ^ acc[3];
}

// every wave streams ITS OWN contiguous 1/nw of the buffer (the chain kernel's pattern), two register sets of U
template <int U>
__global__ void __launch_bounds__(1024) own_kernel(const u32x4* __restrict__ src, int nfrag, int reps, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  const int per = nfrag / nw, groups = per / U;
  const u32x4* base = src + (size_t)wave * per * 64 + lane;
  u32x4 a[U], b[U];
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = base[(size_t)u * 64];
    for (int g = 0; g + 2 <= groups; g += 2) {
#pragma unroll
      for (int u = 0; u < U; ++u) b[u] = base[(size_t)((g + 1) * U + u) * 64];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= a[u];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = base[(size_t)(((g + 2) % groups) * U + u) * 64];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= b[u];
    }
  }
  if (acc[0] == 0x12345678u && acc[1] == 77u) sink[0] = acc[2] ^ acc[3];
}
// one launch: (optional) touch pass over the whole buffer at `stride` bytes, barrier, then the own-range stream once;
// s_memtime of both phases for workgroup 0 -> out[0..2]
template <int U>
__global__ void __launch_bounds__(1024) touch_own_kernel(const u32x4* __restrict__ src, int nfrag, int stride, int wide, unsigned long long* out, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const unsigned nper = (gridDim.x + 7) / 8, slot = blockIdx.x / 8;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned t = 0;
  if (stride > 0) {
    const long long all = (long long)nfrag * 1024 / stride, lo = all * slot / nper, hi = all * (slot + 1) / nper;
    if (wide) {
      u32x4 tt = {0, 0, 0, 0};
      for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) tt ^= *(const u32x4*)((const char*)src + i * stride);
      t = tt[0] ^ tt[1] ^ tt[2] ^ tt[3];
    } else {
      for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) t ^= *(const unsigned*)((const char*)src + i * stride);
    }
  }
  asm volatile("" ::"v"(t));
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  u32x4 acc = {0, 0, 0, 0};
  const int per = nfrag / nw, groups = per / U;
  const u32x4* base = src + (size_t)wave * per * 64 + lane;
  u32x4 a[U], b[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = base[(size_t)u * 64];
  for (int g = 0; g + 2 <= groups; g += 2) {
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = base[(size_t)((g + 1) * U + u) * 64];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= a[u];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = base[(size_t)(((g + 2) % groups) * U + u) * 64];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= b[u];
  }
  if (acc[0] == 0x12345678u && acc[1] == 77u) sink[0] = acc[2] ^ acc[3] ^ t;
  __syncthreads();
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
}
__global__ void fill_kernel(u32x4* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = u32x4{1, 2, 3, 4};
}

__device__ __forceinline__ void glds16(const unsigned char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// D pieces in flight per wave; ring of RING KiB per wave (so waves never collide); no consumer (a floor for the DMA)
template <int D>
__global__ void __launch_bounds__(1024) dma_kernel(const unsigned char* __restrict__ src, int nfrag, int reps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem + wave * (2 * D) * 1024;
  int slot = 0;
  for (int r = 0; r < reps; ++r) {
    for (int f = wave; f < nfrag; f += nw) {
      glds16(src + (size_t)f * 1024, lane * 16, base + slot * 1024);
      slot = (slot + 1) % (2 * D);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (smem[threadIdx.x] == 0x5a && smem[threadIdx.x + 1] == 0x17) sink[0] = 1;
}

template <class F>
static float time_it(F launch, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main(int argc, char** argv) {
  const int nfrag = argc > 1 ? atoi(argv[1]) : 1024;  // 1 MiB
  const int reps = 4;
  unsigned char* d; unsigned* sink;
  hipMalloc(&d, (size_t)nfrag * 1024 + 65536); hipMalloc(&sink, 64);
  std::vector<unsigned> h((size_t)nfrag * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const double bytes = (double)nfrag * 1024 * reps;
  printf("stream of %d KiB read %d times by every workgroup; GB/s PER WORKGROUP (= per CU at <= 256 workgroups)\n", nfrag, reps);
  {  // own-range pattern, warm (4 passes) and cold (1 pass after a 512 MiB fill that evicts L2 and the Infinity Cache)
    u32x4* big; hipMalloc(&big, (size_t)512 << 20);
    for (int g : {16, 128, 256}) {
      for (int waves : {8}) {
#define OWN(U) { float ms = time_it([&] { hipLaunchKernelGGL(own_kernel<U>, dim3(g), dim3(64 * waves), 0, 0, (const u32x4*)d, nfrag, reps, sink); }, 20); \
                 printf("grid %3d  own   waves %2d  U %2d warm x%d  %7.1f GB/s  (%.1f us)\n", g, waves, U, reps, bytes / ms / 1e6, ms * 1e3); }
        OWN(4) OWN(8)
        {
          hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
          float tot = 0;
          for (int it = 0; it < 5; ++it) {
            hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, big, ((size_t)512 << 20) / 16);
            hipEventRecord(a);
            hipLaunchKernelGGL(own_kernel<8>, dim3(g), dim3(64 * waves), 0, 0, (const u32x4*)d, nfrag, 1, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); tot += ms;
          }
          printf("grid %3d  own   waves %2d  U  8 COLD x1   %7.1f GB/s  (%.1f us incl. launch)\n", g, waves, (double)nfrag * 1024 / (tot / 5) / 1e6, tot / 5 * 1e3);
        }
      }
    }
  }
  {  // does a touch pass warm the XCD's L2 for the stream that follows?  (cold start: 512 MiB fill before every launch)
    u32x4* big; hipMalloc(&big, (size_t)512 << 20);
    unsigned long long* out; hipMalloc(&out, 64);
    for (int g : {8, 128}) {
      for (int mode = 0; mode < 6; ++mode) {
        const int stride = mode == 0 ? 0 : mode == 1 ? 128 : mode == 2 ? 64 : mode == 3 ? 32 : 16, wide = mode >= 4;
        const int st2 = mode == 5 ? 64 : stride;
        unsigned long long acc0 = 0, acc1 = 0;
        for (int it = 0; it < 5; ++it) {
          hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, big, ((size_t)512 << 20) / 16);
          hipLaunchKernelGGL(touch_own_kernel<8>, dim3(g), dim3(512), 0, 0, (const u32x4*)d, nfrag, st2, wide, out, sink);
          unsigned long long h2[2]; hipMemcpy(h2, out, 16, hipMemcpyDeviceToHost);
          acc0 += h2[0]; acc1 += h2[1];
        }
        printf("grid %3d  touch stride %3d %s: touch %6llu cycles, stream of %d KiB %6llu cycles\n", g, st2, wide ? "(16 B per lane)" : "(4 B per lane) ", acc0 / 5, nfrag, acc1 / 5);
      }
    }
  }
  const int grids[] = {128};
  for (int g : grids) {
    for (int waves : {4, 8, 16}) {
#define REG(U) { float ms = time_it([&] { hipLaunchKernelGGL(reg_kernel<U>, dim3(g), dim3(64 * waves), 0, 0, (const u32x4*)d, nfrag, reps, sink); }, 20); \
                 printf("grid %3d  reg   waves %2d  U %2d (%3d KiB in flight)  %7.1f GB/s  (%.1f us)\n", g, waves, U, waves * U, bytes / ms / 1e6, ms * 1e3); }
      REG(4) REG(8) REG(16) REG(32)
#define REG2(U) { float ms = time_it([&] { hipLaunchKernelGGL(reg2_kernel<U>, dim3(g), dim3(64 * waves), 0, 0, (const u32x4*)d, nfrag, reps, sink); }, 20); \
                 printf("grid %3d  reg2  waves %2d  U %2d (%3d KiB in flight)  %7.1f GB/s  (%.1f us)\n", g, waves, U, 2 * waves * U, bytes / ms / 1e6, ms * 1e3); }
      REG2(4) REG2(8) REG2(16)
    }
    for (int waves : {1, 2, 4, 8}) {
#define DMA(D) { const int lds = waves * 2 * D * 1024; \
                 hipFuncSetAttribute((const void*)dma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
                 float ms = time_it([&] { hipLaunchKernelGGL(dma_kernel<D>, dim3(g), dim3(64 * waves), lds, 0, d, nfrag, reps, sink); }, 20); \
                 printf("grid %3d  dma   waves %2d  D %2d (%3d KiB in flight)  %7.1f GB/s  (%.1f us)\n", g, waves, D, waves * D, bytes / ms / 1e6, ms * 1e3); }
      DMA(2) DMA(4) DMA(8)
      if (waves <= 4) DMA(16)
    }
  }
  return 0;
}
