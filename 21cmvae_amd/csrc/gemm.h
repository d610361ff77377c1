// gemm.h -- generic LDS-tiled MFMA GEMM for arbitrary layer shapes (gfx950).
//
// The per-layer fallback of Model.predict (emulator.py:402, :753-754) for stacks that
// have no fused kernel: Z = H W + b, H' = relu(Z), operands as they sit in the Keras-order
// arena (A k-contiguous, B n-contiguous).  Training does NOT use this kernel any more: a
// K-loop GEMM pays one global round trip per k-tile, which made a batch-256 step 267 us;
// see gemm_nt.h.
//
// 64x64 output tile per 256-thread workgroup (4 waves as 2x2, one 32x32 MFMA tile
// each), BK = 64, double-buffered LDS (one barrier per k-tile).  Operands are fp32 in memory; each thread fetches its 8 elements of
// the NEXT k-tile into registers before the MFMAs of the current one (so the global
// latency hides under compute), converts them to the compute type and writes them to
// the [row][k] LDS images after the barrier.  k-contiguous operands are fetched as
// 128-byte row segments (float4 per lane when 16-byte aligned), row-contiguous operands
// as 256-byte column segments.  f32 mode uses v_mfma_f32_32x32x2_f32 (exact);
// f16/bf16 use v_mfma_f32_32x32x16.
#pragma once
#include <hip/hip_runtime.h>

#include "fused_fwd.h"

namespace v21 {

struct GemmArgs {
  const float* A; long long sa_m, sa_k;   // A(m,k) = A[m*sa_m + k*sa_k]
  const float* B; long long sb_k, sb_n;   // B(k,n) = B[k*sb_k + n*sb_n]
  float* C; long long ldc;                // C(m,n) = C[m*ldc + n]
  int M, N, K;
  const float* bias;                      // N floats
};

enum { EP_BIAS = 1, EP_BIAS_RELU = 2 };

constexpr int kBM = 64, kBN = 64, kBK = 64;

template <class P> struct GemmTraits;
template <> struct GemmTraits<PrecF32> { using T = float; static constexpr int PITCH = kBK + 4; };
template <> struct GemmTraits<PrecF16> { using T = _Float16; static constexpr int PITCH = kBK + 8; };
template <> struct GemmTraits<PrecBF16> { using T = __bf16; static constexpr int PITCH = kBK + 8; };

// One operand tile (64 rows x 64 k) : fetch this thread's 16 values.
//   KC (k-contiguous):  thread -> rows t/16 + 16*{0..3}, k = 4*(t%16) .. +3   (256-B row segments)
//   RC (row-contiguous): thread -> row t%64, k = 16*(t/64) .. +15            (256-B column segments)
template <bool KC>
__device__ __forceinline__ void fetch_tile(float (&v)[16], const float* __restrict__ base, long long s_row,
                                           long long s_k, int row0, int nrows, int k0, int kend, int tid) {
  if constexpr (KC) {
    const int kk = k0 + 4 * (tid & 15);
#pragma unroll
    for (int half = 0; half < 4; ++half) {
      const int row = row0 + (tid >> 4) + 16 * half;
      float* o = v + 4 * half;
      o[0] = o[1] = o[2] = o[3] = 0.f;
      if (row < nrows) {
        const float* p = base + (long long)row * s_row + kk;  // s_k == 1
        if (kk + 3 < kend && ((reinterpret_cast<unsigned long long>(p) & 15ull) == 0)) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) if (kk + i < kend) o[i] = p[i];
        }
      }
    }
  } else {
    const int row = row0 + (tid & 63);
    const int kk = k0 + 16 * (tid >> 6);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float x = 0.f;
      if (row < nrows && kk + i < kend) x = base[(long long)row * s_row + (long long)(kk + i) * s_k];
      v[i] = x;
    }
  }
}
template <bool KC, class T, int PITCH>
__device__ __forceinline__ void store_tile(T* __restrict__ lds, const float (&v)[16], int tid) {
  typedef T T4 __attribute__((ext_vector_type(4)));
  if constexpr (KC) {
#pragma unroll
    for (int half = 0; half < 4; ++half) {
      T4 q = {(T)v[4 * half], (T)v[4 * half + 1], (T)v[4 * half + 2], (T)v[4 * half + 3]};
      *(T4*)(lds + ((tid >> 4) + 16 * half) * PITCH + 4 * (tid & 15)) = q;  // one 8/16-byte LDS write
    }
  } else {
    T* o = lds + (tid & 63) * PITCH + 16 * (tid >> 6);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      T4 q = {(T)v[4 * i], (T)v[4 * i + 1], (T)v[4 * i + 2], (T)v[4 * i + 3]};
      *(T4*)(o + 4 * i) = q;
    }
  }
}

template <class P, int EP, bool AKC, bool BKC>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs g) {
  using T = typename GemmTraits<P>::T;
  constexpr int PITCH = GemmTraits<P>::PITCH;
  __shared__ __attribute__((aligned(16))) T As[2][kBM * PITCH];  // double-buffered: one barrier per k-tile
  __shared__ __attribute__((aligned(16))) T Bs[2][kBN * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int li = lane & 31, lh = lane >> 5;
  const int kbeg = 0, kend = g.K;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // A(m,k): rows = m.  KC: s_row = sa_m (sa_k == 1); RC: rows contiguous (sa_m == 1), s_k = sa_k
  const long long a_srow = AKC ? g.sa_m : 1, a_sk = AKC ? 1 : g.sa_k;
  // B(k,n): rows = n.  KC: s_row = sb_n (sb_k == 1); RC: s_row = 1 (sb_n == 1), s_k = sb_k
  const long long b_srow = BKC ? g.sb_n : 1, b_sk = BKC ? 1 : g.sb_k;

  float va[16], vb[16];
  fetch_tile<AKC>(va, g.A, a_srow, a_sk, m0, g.M, kbeg, kend, tid);
  fetch_tile<BKC>(vb, g.B, b_srow, b_sk, n0, g.N, kbeg, kend, tid);

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kBK, buf ^= 1) {
    // buffer `buf` was last read two iterations ago, before the previous barrier
    store_tile<AKC, T, PITCH>(As[buf], va, tid);
    store_tile<BKC, T, PITCH>(Bs[buf], vb, tid);
    __syncthreads();
    if (k0 + kBK < kend) {  // next tile's global loads fly while this tile is multiplied
      fetch_tile<AKC>(va, g.A, a_srow, a_sk, m0, g.M, k0 + kBK, kend, tid);
      fetch_tile<BKC>(vb, g.B, b_srow, b_sk, n0, g.N, k0 + kBK, kend, tid);
    }
    const T* ap = As[buf] + (wm * 32 + li) * PITCH;
    const T* bp = Bs[buf] + (wn * 32 + li) * PITCH;
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
  }

  // C/D map of the 32x32 tile: col = lane&31, row = (i&3) + 8(i>>2) + 4(lane>>5)
  float* C = g.C;
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
        C[(long long)m * g.ldc + n] = v;
      }
    }
  }
}

}  // namespace v21
