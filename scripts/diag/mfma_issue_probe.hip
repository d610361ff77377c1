// Diagnostics (r4): one wave per SIMD runs  [LDS fragment read, counted wait, MFMA]  per step (lds_read_probe.hip: 32.4 cycles per
// step with two or more reads in flight) plus EXTRA independent vector instructions per step -- how many can ride under an MFMA
// before the step gets longer?  (The epilogue of fused_train.h puts ~2-7 of them between two MFMAs.)
//   hipcc -std=c++20 --offload-arch=gfx950 -O3 -o /tmp/mfma_issue_probe scripts/diag/mfma_issue_probe.hip && /tmp/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int EXTRA, int KIND, int UNROLL = 32>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* cyc, int iters, const unsigned char* gsrc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  constexpr int DEPTH = 4;
  const unsigned base = (unsigned)(size_t)smem + (threadIdx.x & 63) * 16;
  h8 q[DEPTH], b;
  for (int i = 0; i < 8; ++i) b[i] = (_Float16)(0.002f * (i + 1));
  f16v acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.5f + threadIdx.x * 0.001f * i;
  unsigned long long t0, t1;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[d]) : "v"(base), "n"(d * 1024));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  const unsigned dma_voff = (threadIdx.x & 63) * 16, dma_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + 65536 + (threadIdx.x >> 6) * 1024);
  auto step = [&q, &acc, &b, &x, base, dma_voff, dma_dst, gsrc](auto u_) __attribute__((always_inline)) {
    constexpr int u = decltype(u_)::value;
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(q[u % DEPTH]) : "n"(DEPTH - 1));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[u % DEPTH], b, acc, 0, 0, 0);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[u % DEPTH]) : "v"(base), "n"(((u + DEPTH) % 64) * 1024));
#pragma unroll
    for (int e = 0; e < EXTRA; ++e) {
      if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[e % 8]) : "v"(x[(e + 1) % 8]));          // plain VALU
      else if constexpr (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "+v"(x[e % 8]) : "v"(x[(e + 1) % 8]));    // conversion
      else if constexpr (KIND == 2) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");                                           // scalar
      else if constexpr (KIND == 4) {  // one LDS-DMA piece (1 KiB per wave, global -> LDS) every 4th step, as the weight ring's refill
        if (e == 0 && (u & 3) == 0)
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(dma_voff), "s"(gsrc + (u & 63) * 4096), "s"(dma_dst) : "memory");
      }
      else asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[e % 8]) : "v"(x[(e + 1) % 8]) : "vcc"); // compare + select
    }
  };
  for (int it = 0; it < iters; ++it) {
    static_for<UNROLL / 32>([&](auto o_) __attribute__((always_inline)) {
      static_for<32>([&](auto i_) __attribute__((always_inline)) {
        step(std::integral_constant<int, decltype(o_)::value * 32 + decltype(i_)::value>{});
      });
    });
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int d = 0; d < DEPTH; ++d) s += (float)q[d][1];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
static unsigned char* g_src = nullptr;
template <int EXTRA, int KIND, int UNROLL = 32> void run(float* out, unsigned long long* cyc) {
  const int iters = 9600 / UNROLL;
  (void)hipFuncSetAttribute((const void*)probe<EXTRA, KIND, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
  for (int r = 0; r < 2; ++r) probe<EXTRA, KIND, UNROLL><<<256, 256, 65536 + 4096>>>(out, cyc, iters, g_src);
  (void)hipDeviceSynchronize();
  unsigned long long c = 0;
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  static const char* kinds[] = {"v_fma_f32", "v_cvt_pk_f16_f32", "s_add_u32", "v_cmp + s_nop 1 + v_cndmask", "(LDS-DMA piece every 4th step)"};
  printf("%d x %-28s per MFMA step, %4d steps unrolled (~%3d KB of code): %.1f cycles per step\n", EXTRA, kinds[KIND], UNROLL,
         UNROLL * (24 + 8 * EXTRA) / 1024, c / ((double)iters * UNROLL));
  fflush(stdout);
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4 * 256 * 256); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&g_src, 1 << 20); (void)hipMemset(g_src, 0, 1 << 20);
  run<0, 0>(out, cyc); run<2, 0>(out, cyc); run<4, 0>(out, cyc); run<6, 0>(out, cyc); run<8, 0>(out, cyc); run<12, 0>(out, cyc);
  run<4, 1>(out, cyc); run<8, 1>(out, cyc);
  run<2, 3>(out, cyc); run<4, 3>(out, cyc);
  run<1, 4>(out, cyc);
  // straight-line code larger than the instruction cache (64 KB per two CUs): every instruction is fetched from L2
  run<2, 0, 480>(out, cyc); run<2, 0, 1600>(out, cyc); run<2, 0, 3200>(out, cyc); run<2, 0, 4800>(out, cyc);
  return 0;
}
