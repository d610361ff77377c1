// gemm.h -- generic LDS-tiled MFMA GEMM for arbitrary layer shapes (gfx950).
//
// Serves (a) the per-layer fallback of Model.predict for stacks that have no fused
// kernel and (b) every contraction of a training step (Keras fit(), reference call
// sites emulator.py:369, :739, :756):
//     forward   Z = H W + b, H' = relu(Z)          (A k-contiguous, B n-contiguous)
//     backward  dH = dZ W^T  (.) [H > 0]           (A k-contiguous, B k-contiguous)
//     weights   [dW; db] = [H^T; 1^T] dZ           (A m-contiguous, B n-contiguous)
// The bias gradient rides along as one extra row of ones appended to H^T, so that the
// (K+1) x N result is exactly the [kernel | bias] slice of the flat gradient arena.
//
// 64x64 output tile per 256-thread workgroup (4 waves as 2x2, one 32x32 MFMA tile
// each), BK = 32.  Operands are fp32 in memory and are converted to the compute type
// while being staged into LDS ([row][k] images, k contiguous, rows padded by 16 B).
// f32 mode uses v_mfma_f32_32x32x2_f32 (exact); f16/bf16 use v_mfma_f32_32x32x16.
#pragma once
#include <hip/hip_runtime.h>

#include "fused_fwd.h"

namespace v21 {

struct GemmArgs {
  const float* A; long long sa_m, sa_k;   // A(m,k) = A[m*sa_m + k*sa_k]
  const float* B; long long sb_k, sb_n;   // B(k,n) = B[k*sb_k + n*sb_n]
  float* C; long long ldc;                // C(m,n) = C[m*ldc + n]
  int M, N, K;
  const float* bias;                      // EP_BIAS*: N floats
  const float* mask; long long ldmask;    // EP_MASK: multiply by [mask(m,n) > 0]
  int ones_row;                           // A(m == ones_row, k) = 1  (-1: none)
  float alpha;                            // C = alpha * acc (EP_PLAIN)
};

enum { EP_PLAIN = 0, EP_BIAS = 1, EP_BIAS_RELU = 2, EP_MASK = 3 };

constexpr int kBM = 64, kBN = 64, kBK = 32;

template <class P> struct GemmTraits;
template <> struct GemmTraits<PrecF32> { using T = float; static constexpr int PITCH = kBK + 4; };
template <> struct GemmTraits<PrecF16> { using T = _Float16; static constexpr int PITCH = kBK + 8; };
template <> struct GemmTraits<PrecBF16> { using T = __bf16; static constexpr int PITCH = kBK + 8; };

template <class P, int EP>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs g) {
  using T = typename GemmTraits<P>::T;
  constexpr int PITCH = GemmTraits<P>::PITCH;
  __shared__ __attribute__((aligned(16))) T As[kBM * PITCH];
  __shared__ __attribute__((aligned(16))) T Bs[kBN * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // staging maps: "k-contiguous" operand -> thread owns (row = tid/4, 8 k's);
  // "row-contiguous" operand -> thread owns (row = tid%64, 8 k's) so that consecutive
  // lanes touch consecutive addresses either way
  const bool a_kc = (g.sa_k == 1), b_kc = (g.sb_k == 1);
  const int a_row = a_kc ? (tid >> 2) : (tid & 63), a_k8 = a_kc ? (tid & 3) * 8 : (tid >> 6) * 8;
  const int b_row = b_kc ? (tid >> 2) : (tid & 63), b_k8 = b_kc ? (tid & 3) * 8 : (tid >> 6) * 8;

  for (int k0 = 0; k0 < g.K; k0 += kBK) {
    {
      const int m = m0 + a_row;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + a_k8 + i;
        float v = 0.f;
        if (m < g.M && k < g.K) v = (m == g.ones_row) ? 1.0f : g.A[m * g.sa_m + k * g.sa_k];
        As[a_row * PITCH + a_k8 + i] = (T)v;
      }
      const int n = n0 + b_row;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + b_k8 + i;
        float v = 0.f;
        if (n < g.N && k < g.K) v = g.B[k * g.sb_k + n * g.sb_n];
        Bs[b_row * PITCH + b_k8 + i] = (T)v;
      }
    }
    __syncthreads();
    const T* ap = As + (wm * 32 + li) * PITCH;
    const T* bp = Bs + (wn * 32 + li) * PITCH;
    if constexpr (std::is_same<P, PrecF32>::value) {
#pragma unroll
      for (int q = 0; q < kBK / 8; ++q) {
        const f32x4 av = *(const f32x4*)(ap + 8 * q + 4 * lh);
        const f32x4 bv = *(const f32x4*)(bp + 8 * q + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
      }
    } else {
      using frag = typename P::frag;
#pragma unroll
      for (int kk = 0; kk < kBK / 16; ++kk) {
        const frag av = *(const frag*)(ap + 16 * kk + 8 * lh);
        const frag bv = *(const frag*)(bp + 16 * kk + 8 * lh);
        acc = P::template mfma<false>(av, bv, acc);
      }
    }
    __syncthreads();
  }

  // C/D map of the 32x32 tile: col = lane&31, row = (i&3) + 8(i>>2) + 4(lane>>5)
  const int n = n0 + wn * 32 + li;
  if (n < g.N) {
    float bias = 0.f;
    if constexpr (EP == EP_BIAS || EP == EP_BIAS_RELU) bias = g.bias[n];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
      if (m < g.M) {
        float v = acc[i];
        if constexpr (EP == EP_BIAS) v = v + bias;
        if constexpr (EP == EP_BIAS_RELU) v = fmaxf(v + bias, 0.f);
        if constexpr (EP == EP_MASK) v = (g.mask[m * g.ldmask + n] > 0.f) ? v : 0.f;
        if constexpr (EP == EP_PLAIN) v = v * g.alpha;
        g.C[m * g.ldc + n] = v;
      }
    }
  }
}

}  // namespace v21
