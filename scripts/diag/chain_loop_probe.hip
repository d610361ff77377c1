// chain_loop_probe.hip -- the inner loop of the 16-bit chain kernel in isolation: ONE workgroup of 16 waves; every wave
// streams "its tile" (NCH chunks of four 1-KiB weight fragments, two-buffer rolling prefetch exactly as train_chain.h) from
// a warm L2, with or without the four dependent MFMAs per chunk and the four activation-operand reads from LDS per chunk.
// Which ingredient takes the stream from the 63 B/clk of a pure stream (l1_stream_probe) to the ~38 B/clk of the kernel?
//   hipcc --offload-arch=gfx950 -O3 chain_loop_probe.hip -o chain_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool MFMA, bool LDS, int ACCS>
__global__ void __launch_bounds__(1024) loop_kernel(const half8* __restrict__ w, int nch, int tiles_per_wave, int reps, unsigned long long* ticks, float* sink) {
  __shared__ __attribute__((aligned(16))) _Float16 act[32 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32 * 520; i += 1024) act[i] = (_Float16)(0.001f * (i & 63));
  __syncthreads();
  const _Float16* ap = act + (lane & 31) * 520 + 8 * (lane >> 5);
  f32x16 acc[ACCS];
  for (int a = 0; a < ACCS; ++a)
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  half8 wa[4], wb[4], bc[4], bn[4];
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    if (r == 1) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");  // rep 0 warms L2
    for (int t = 0; t < tiles_per_wave; ++t) {
      const half8* src = w + ((long long)(wave * tiles_per_wave + t) * nch * 4) * 64 + lane;
#pragma unroll
      for (int j = 0; j < 4; ++j) wa[j] = src[j * 64];
      if (LDS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const half8*>(ap + j * 16);
      }
      for (int c = 0; c + 2 <= nch; c += 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = src[(4 * (c + 1) + j) * 64];
        __builtin_amdgcn_sched_barrier(0);
        if (LDS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const half8*>(ap + ((4 * (c + 1) + j) & 31) * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MFMA) acc[j % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[j], LDS ? bc[j] : wa[j], acc[j % ACCS], 0, 0, 0);
          else asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[0][0]) : "v"(__builtin_bit_cast(float4, wa[j]).x));
        }
        const int cn = c + 2 < nch ? c + 2 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) wa[j] = src[(4 * cn + j) * 64];
        __builtin_amdgcn_sched_barrier(0);
        if (LDS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const half8*>(ap + ((4 * (c + 2) + j) & 31) * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MFMA) acc[j % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[j], LDS ? bn[j] : wb[j], acc[j % ACCS], 0, 0, 0);
          else asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[0][0]) : "v"(__builtin_bit_cast(float4, wb[j]).x));
        }
      }
    }
  }
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
  float s = 0.f;
  for (int a = 0; a < ACCS; ++a)
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

template <bool MFMA, bool LDS, int ACCS>
void run(const char* what, const half8* d, int nch, int tpw, unsigned long long* dt, float* sink) {
  const int reps = 9;
  unsigned long long t = 0;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((loop_kernel<MFMA, LDS, ACCS>), dim3(1), dim3(1024), 0, 0, d, nch, tpw, reps, dt, sink);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
  }
  const double bytes = 16.0 * tpw * nch * 4096.0;
  printf("%-58s %2d tiles/wave x %d chunks: %7.0f cycles per pass of %4.0f KiB = %.1f B per cycle\n", what, tpw, nch, (double)t / (reps - 1), bytes / 1024,
         bytes * (reps - 1) / (double)t);
}

// The same loop with what a LAYER adds: a barrier between tiles, an epilogue's worth of idle time (EPI cycles of s_sleep-free
// VALU work), and only the chunks prefetched BEFORE the barrier in flight across it: AHEAD = 1 (train_chain.h today: the next
// tile's first chunk) or 2 (its first two).
template <int AHEAD>
__global__ void __launch_bounds__(1024) layer_kernel(const half8* __restrict__ w, int nch, int tiles, int reps, unsigned long long* ticks, float* sink) {
  __shared__ __attribute__((aligned(16))) _Float16 act[32 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32 * 520; i += 1024) act[i] = (_Float16)(0.001f * (i & 63));
  __syncthreads();
  const _Float16* ap = act + (lane & 31) * 520 + 8 * (lane >> 5);
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  half8 wa[4], wb[4], bc[4], bn[4];
  auto tile_src = [&](int t) { return w + ((long long)(wave * tiles + (t % tiles)) * nch * 4) * 64 + lane; };
  for (int r = 0; r < reps; ++r) {
    if (r == 1) { __syncthreads(); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); }
    {
      const half8* s0 = tile_src(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) wa[j] = s0[j * 64];
      if (AHEAD == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = s0[(4 + j) * 64];
      }
    }
    for (int t = 0; t < tiles; ++t) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the layer's barrier
      const half8* src = tile_src(t);
      const half8* nxt = tile_src(t + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const half8*>(ap + j * 16);
      for (int c = 0; c + 2 <= nch; c += 2) {
        if (AHEAD == 1 || c > 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wb[j] = src[(4 * (c + 1) + j) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const half8*>(ap + ((4 * (c + 1) + j) & 31) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[j], bc[j], acc, 0, 0, 0);
        {
          const half8* p = c + 2 < nch ? src + (4 * (c + 2)) * 64 : nxt;
#pragma unroll
          for (int j = 0; j < 4; ++j) wa[j] = p[j * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const half8*>(ap + ((4 * (c + 2) + j) & 31) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[j], bn[j], acc, 0, 0, 0);
        if (AHEAD == 2 && c + 2 >= nch) {  // the next tile's SECOND chunk leaves before the barrier as well
#pragma unroll
          for (int j = 0; j < 4; ++j) wb[j] = nxt[(4 + j) * 64];
        }
      }
      // an epilogue's worth of work that needs the accumulator (bias, ReLU, packing): ~150 VALU instructions
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaxf(acc[i] * 1.0001f, -1.0f);
    }
  }
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}
template <int AHEAD>
void run_layers(const char* what, const half8* d, int nch, int tiles, unsigned long long* dt, float* sink) {
  const int reps = 9;
  unsigned long long t = 0;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((layer_kernel<AHEAD>), dim3(1), dim3(1024), 0, 0, d, nch, tiles, reps, dt, sink);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
  }
  const double bytes = 16.0 * tiles * nch * 4096.0;
  printf("%-58s %2d tiles/wave x %d chunks: %7.0f cycles per tile round = %.1f B per cycle\n", what, tiles, nch, (double)t / (reps - 1) / tiles,
         bytes * (reps - 1) / (double)t);
}

int main() {
  const long long bytes = 16ll * 4 * 8 * 4096;  // 16 waves x up to 4 tiles x 8 chunks
  half8* d; unsigned long long* dt; float* sink;
  (void)hipMalloc(&d, bytes); (void)hipMemset(d, 0, bytes); (void)hipMalloc(&dt, 64); (void)hipMalloc(&sink, 4096);
  for (int tpw : {1, 4}) {
    run<false, false, 1>("stream only (one VALU per fragment)", d, 8, tpw, dt, sink);
    run<true, false, 1>("+ 4 MFMAs per chunk into ONE accumulator", d, 8, tpw, dt, sink);
    run<true, false, 2>("+ 4 MFMAs per chunk into TWO accumulators", d, 8, tpw, dt, sink);
    run<false, true, 1>("+ 4 operand reads from LDS per chunk (no MFMA)", d, 8, tpw, dt, sink);
    run<true, true, 1>("+ MFMAs (one accumulator) + LDS operand reads", d, 8, tpw, dt, sink);
    run<true, true, 2>("+ MFMAs (two accumulators) + LDS operand reads", d, 8, tpw, dt, sink);
  }
  run_layers<1>("barrier + epilogue per tile, ONE chunk ahead across it", d, 8, 4, dt, sink);
  run_layers<2>("barrier + epilogue per tile, TWO chunks ahead across it", d, 8, 4, dt, sink);
  run_layers<1>("barrier + epilogue per tile, ONE chunk ahead across it", d, 6, 4, dt, sink);
  run_layers<2>("barrier + epilogue per tile, TWO chunks ahead across it", d, 6, 4, dt, sink);
  return 0;
}
