// gemm_nt.h -- latency-oriented MFMA GEMM for the training step (gfx950).
//
// A Keras fit() step at the reference's batch of 256 (emulator.py:372) is ~0.5 GFLOP cut
// into ~15 dependent contractions: every one of them is bound by memory LATENCY, not by
// FLOPs.  A K-loop GEMM pays one global round trip per k-tile; this kernel pays ONE per
// contraction:
//   * every operand is stored with the contraction index contiguous ("NT" form:
//     C[m][n] = sum_k A[m][k] * Bm[n][k]); producers write the transposed copies the
//     consumers need (H and H^T, dZ and dZ^T, W^T and row-padded W), so an MFMA operand
//     fragment is two 16-byte loads straight from global/L2 into registers -- no LDS
//     staging, no barrier in the loop;
//   * a 256-thread workgroup owns ONE 32x32 output tile and its four waves split the
//     contraction range four ways; each wave issues ALL loads of its quarter (<= 8 k-steps
//     of 16 for f16/bf16, <= 16 of 8 for f32) before the first MFMA, so the whole range is
//     in flight at once; the four partial tiles meet in LDS (one barrier) and each wave
//     finishes a quarter of the rows;
//   * contraction ranges longer than kMaxKPerWg (the batch dimension of the weight
//     gradient) are split over blockIdx.z into slabs that are summed in a fixed order.
// Epilogues: forward (bias, ReLU; writes H and H^T), backward (ReLU mask; writes dZ and
// dZ^T), weight gradient (plain store into the [kernel|bias] slab).
#pragma once
#include <hip/hip_runtime.h>

#include "fused_fwd.h"

namespace v21 {

struct NtArgs {
  const float* A; long long lda;   // A(m,k)  = A[m*lda + k]
  const float* B; long long ldb;   // Bm(n,k) = B[n*ldb + k]
  float* C; long long ldc;         // C(m,n); slab z at C + z*slab_stride
  float* CT; long long ldct;       // optional transposed copy CT(n,m)
  int M, N, K;
  const float* bias;               // NT_FWD*: N floats
  const float* mask; long long ldmask;  // NT_DX_MASK: keep where mask(m,n) > 0
  int k_chunk;                     // contraction range of slice z: [z*k_chunk, min(K, (z+1)*k_chunk))
  long long slab_stride;
  int ep;                          // NT_* epilogue
  int nx, ny, nz;                  // tiles along n, m and contraction slices (set by the launcher)
};
// up to two independent contractions per launch (the backward of one layer: dW and dX
// both consume dZ of that layer); blocks [0, nx0*ny0*nz0) belong to the first
struct NtGroup {
  NtArgs p[2];
  int count;
};
enum { NT_FWD = 0, NT_FWD_RELU = 1, NT_DX = 2, NT_DX_MASK = 3, NT_DW = 4 };

constexpr int kNtMaxKPerWg = 512;

template <class P> struct NtTraits;
template <> struct NtTraits<PrecF32> { static constexpr int KSTEP = 8, MAXSTEPS = 16, REGS = 4; };
template <> struct NtTraits<PrecF16> { static constexpr int KSTEP = 16, MAXSTEPS = 8, REGS = 8; };
template <> struct NtTraits<PrecBF16> { static constexpr int KSTEP = 16, MAXSTEPS = 8, REGS = 8; };

template <class P>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const NtGroup grp) {
  using TR = NtTraits<P>;
  int bid = blockIdx.x;
  const int n0blocks = grp.p[0].nx * grp.p[0].ny * grp.p[0].nz;
  const bool second = grp.count > 1 && bid >= n0blocks;
  if (second) bid -= n0blocks;
  const NtArgs& g = second ? grp.p[1] : grp.p[0];
  const int bx = bid % g.nx, by = (bid / g.nx) % g.ny, bz = bid / (g.nx * g.ny);
  const int EP = g.ep;
  constexpr int KSTEP = TR::KSTEP, MAXSTEPS = TR::MAXSTEPS, REGS = TR::REGS;
  __shared__ __attribute__((aligned(16))) float part[4][16][64];  // partial tiles of the 4 waves

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = by * 32, n0 = bx * 32;
  const int kbeg = bz * g.k_chunk;
  const int kend = min(g.K, kbeg + g.k_chunk);
  // this wave's quarter, in whole k-steps
  const int nsteps = (kend - kbeg + KSTEP - 1) / KSTEP;
  const int per = (nsteps + 3) / 4;
  const int s0 = wave * per, s1 = min(nsteps, s0 + per);

  const int am = min(m0 + li, g.M - 1), bn = min(n0 + li, g.N - 1);  // clamp: edge rows are discarded
  const float* ap = g.A + (long long)am * g.lda + kbeg + (REGS == 8 ? 8 : 4) * lh;
  const float* bp = g.B + (long long)bn * g.ldb + kbeg + (REGS == 8 ? 8 : 4) * lh;

  float va[MAXSTEPS][REGS], vb[MAXSTEPS][REGS];
  // ---- issue every load of this wave's range (rows are padded: reads stay in bounds)
#pragma unroll
  for (int s = 0; s < MAXSTEPS; ++s) {
    if (s0 + s < s1) {
      const int ko = (s0 + s) * KSTEP;
#pragma unroll
      for (int q = 0; q < REGS / 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(ap + ko + 4 * q);
        const float4 y = *reinterpret_cast<const float4*>(bp + ko + 4 * q);
        va[s][4 * q] = x.x; va[s][4 * q + 1] = x.y; va[s][4 * q + 2] = x.z; va[s][4 * q + 3] = x.w;
        vb[s][4 * q] = y.x; vb[s][4 * q + 1] = y.y; vb[s][4 * q + 2] = y.z; vb[s][4 * q + 3] = y.w;
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int s = 0; s < MAXSTEPS; ++s) {
    if (s0 + s < s1) {
      // zero the elements past the end of the contraction range (padding may hold anything)
      const int kk = kbeg + (s0 + s) * KSTEP + (REGS == 8 ? 8 : 4) * lh;
#pragma unroll
      for (int e = 0; e < REGS; ++e)
        if (kk + e >= kend) { va[s][e] = 0.f; vb[s][e] = 0.f; }
      if constexpr (REGS == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s][e], vb[s][e], acc, 0, 0, 0);
      } else {
        typename P::frag fa, fb;
#pragma unroll
        for (int e = 0; e < 8; ++e) { fa[e] = (typename P::elem)va[s][e]; fb[e] = (typename P::elem)vb[s][e]; }
        acc = P::template mfma<false>(fa, fb, acc);
      }
    }
  }
  // ---- meet in LDS; wave w finishes accumulator registers 4w..4w+3 = rows 8w + 4h + {0..3}
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][i][lane] = acc[i];
  __syncthreads();
  float r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    r[e] = (part[0][4 * wave + e][lane] + part[1][4 * wave + e][lane]) + (part[2][4 * wave + e][lane] + part[3][4 * wave + e][lane]);

  const int n = n0 + li;
  const int mrow = m0 + 8 * wave + 4 * lh;  // rows mrow .. mrow+3
  if (n >= g.N) return;
  float bias = 0.f;
  if (EP == NT_FWD || EP == NT_FWD_RELU) bias = g.bias[n];
  float* C = g.C + (long long)bz * g.slab_stride;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = mrow + e;
    float v = r[e];
    if (EP == NT_FWD) v = v + bias;
    if (EP == NT_FWD_RELU) v = fmaxf(v + bias, 0.f);
    if (EP == NT_DX_MASK) v = (m < g.M && g.mask[(long long)m * g.ldmask + n] > 0.f) ? v : 0.f;
    r[e] = v;
    if (m < g.M) C[(long long)m * g.ldc + n] = v;
  }
  if (EP != NT_DW) {
    if (g.CT) {  // rows of the transposed copy are padded to a multiple of 32: no bound check on m
      float4 t = make_float4(mrow + 0 < g.M ? r[0] : 0.f, mrow + 1 < g.M ? r[1] : 0.f, mrow + 2 < g.M ? r[2] : 0.f,
                             mrow + 3 < g.M ? r[3] : 0.f);
      *reinterpret_cast<float4*>(g.CT + (long long)n * g.ldct + mrow) = t;
    }
  }
}

}  // namespace v21
