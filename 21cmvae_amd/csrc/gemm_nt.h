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
#include "train_kernels.h"  // StepCtx

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
  int tile;                        // 32 or 64: workgroup tile edge (set by the launcher)
  // f16/bf16 operand conversion: A*a_scale, B*b_scale, result*out_scale.  Gradients of a large
  // batch (2 w_i (p-y)/B ~ 1e-7) sit below the f16 normal range: the backward contractions
  // scale dZ by a power of two on the way into the MFMA and undo it on the way out (exact).
  float a_scale, b_scale, out_scale;
  const float* aff_mean; float aff_std;  // NT_FWD_UNPRE
};
// several independent contractions per launch: the backward of one layer (dW and dX both
// consume dZ of that layer), and the same layer of every model of a sweep
// (BASELINE configs[4]).  Blocks are dealt to the problems in order; first[i] = first block
// of problem i (set by the launcher).
constexpr int kNtMaxGroup = 16;
template <int CAP>
struct NtGroupT {  // CAP = 2 for a single model (small kernel-argument block), 16 for sweeps
  NtArgs p[CAP];
  int first[CAP + 1];
  int count;
};
using NtGroup = NtGroupT<2>;
using NtGroupBig = NtGroupT<kNtMaxGroup>;
// NT_FWD_UNPRE: last layer of predict with preprocess.unpreproc fused in: (z + bias) * aff_std + aff_mean[n]
enum { NT_FWD = 0, NT_FWD_RELU = 1, NT_DX = 2, NT_DX_MASK = 3, NT_DW = 4, NT_FWD_UNPRE = 5 };

constexpr int kNtMaxKPerWg = 512;

// NT_DW with Adam in the epilogue (f32 chain steps on one rank whose batch fits one contraction slice, train_chain32.h):
// the workgroup that forms a 32 x 32 tile of [dW; db] over the whole batch applies the update to the tile's arena
// elements and rewrites their places in the packed fp32 weight streams -- no separate Adam launch (9 us of a 77-us step:
// 3 us of launch, the gradient read back, scattered 4-byte stores either way).  Common to every problem of the group:
struct NtAdamLayer { long long arena_off, fw_off, bw_off; int K, KS, NS; };  // per problem: [W; b] block, stream offsets (floats)
struct NtAdamInfo {
  float *w, *m, *v;            // arena, first and second moments (same layout as the gradient arena C points into)
  float* fw; float* bw;        // packed fp32 streams (train_chain32.h)
  float alpha, omb1, omb2, eps;
  StepCtx sc;                  // replayed step: alpha and the loss slot come from the descriptor
  unsigned long long* loss_acc; float* loss_out; float* loss_out2; int loss_slot;  // the step's batch loss (thread 0 of block 0)
  int fmt;                     // 3: streams of train_chain32.h (16-row kernel), 4: of train_chain32s.h (8-row kernel)
#ifdef V21_CHAIN_FINE
  unsigned long long* dbg;     // diagnostic build: phase stamps of a few workgroups (scripts/diag/dwadam_stamps.py)
#endif
  NtAdamLayer lt[kNtMaxGroup];
};

template <class P> struct NtTraits;
template <> struct NtTraits<PrecF32> { static constexpr int KSTEP = 8, MAXSTEPS = 16, REGS = 4; };
template <> struct NtTraits<PrecF16> { static constexpr int KSTEP = 16, MAXSTEPS = 8, REGS = 8; };
template <> struct NtTraits<PrecBF16> { static constexpr int KSTEP = 16, MAXSTEPS = 8, REGS = 8; };

// T = MFMA tiles per side of the workgroup tile: T = 1 -> 32x32 (latency: small batches),
// T = 2 -> 64x64 (twice the arithmetic intensity per loaded byte: large batches; the
// contraction range of a wave is then walked in rounds of <= MAXSTEPS/2 k-steps).
#ifdef V21_CHAIN_FINE
#define NTFINE(i) do { if constexpr (ADAM) { if ((threadIdx.x & 63) == 0 && ad->dbg && (blockIdx.x % 47) == 0 && blockIdx.x / 47 < 8) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); ad->dbg[((blockIdx.x / 47) * 8 + (i)) * 4 + (threadIdx.x >> 6)] = t_; } } } while (0)
#else
#define NTFINE(i)
#endif
template <class P, int T, class GROUP, bool ADAM = false>
__device__ __forceinline__ void gemm_nt_body(const GROUP& grp, const NtAdamInfo* ad) {
  using TR = NtTraits<P>;
  NTFINE(0);
  int pi = 0;
  if constexpr (ADAM) {  // (no dependent chain of scalar loads: the whole table, then compares)
#pragma unroll
    for (int i = 1; i < kNtMaxGroup; ++i) pi += (i < grp.count && (int)blockIdx.x >= grp.first[i]) ? 1 : 0;
  } else {
    while (pi + 1 < grp.count && (int)blockIdx.x >= grp.first[pi + 1]) ++pi;  // uniform: scalar loop
  }
  const int bid = blockIdx.x - grp.first[pi];
  const NtArgs& g = grp.p[pi];
  const int bx = bid % g.nx, by = (bid / g.nx) % g.ny, bz = bid / (g.nx * g.ny);
  const int EP = ADAM ? (int)NT_DW : g.ep;  // (the fused gradient + Adam launch has one epilogue: no branches on g.ep)
  constexpr int KSTEP = TR::KSTEP, REGS = TR::REGS;
  constexpr int ROUND = TR::MAXSTEPS / (T * T > 1 ? 2 : 1);  // k-steps in flight per round
  constexpr int HOFF = (REGS == 8 ? 8 : 4);
  __shared__ __attribute__((aligned(16))) float part[4][T * T][16][64];  // partial tiles of the 4 waves

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = by * (32 * T), n0 = bx * (32 * T);
  const int kbeg = bz * g.k_chunk;
  const int kend = min(g.K, kbeg + g.k_chunk);
  // this wave's quarter, in whole k-steps
  const int nsteps = (kend - kbeg + KSTEP - 1) / KSTEP;
  const int per = (nsteps + 3) / 4;
  const int s0 = wave * per, s1 = min(nsteps, s0 + per);

  // ADAM, 32 x 32 tiles: this thread's four arena elements (rows m0 + 8 wave + 4 lh + e, column n0 + li) are requested
  // NOW, before the operands: they arrive in the same ~2-us window, where the epilogue used to start its own round trip
  // after the tiles had met (phase stamps, scripts/diag/dwadam_stamps.py: operands +7.3 k cycles, MFMAs +4.8 k, meeting
  // +0.7 k, epilogue +8.4 k of a 22-k-cycle workgroup).  Elements past the edge are clamped to a valid one, never stored.
  // Everything the epilogue needs of the problem and of the Adam block, read ONCE here: both live in the kernel-argument
  // segment and are indexed by `pi`, so every use further down was a scalar load of its own, waited for on the spot -- some
  // thirty ~200-cycle round trips in a row were the epilogue's 8 k cycles.
  struct { float* C; long long ldc; int M, N; long long arena_off, fw_off, bw_off; int K, KS, NS, fmt; float alpha, omb1, omb2, eps;
           float *w, *m, *v, *fw, *bw; } h{};
  if constexpr (ADAM) {
    const NtAdamLayer& al = ad->lt[pi];
    h.C = g.C; h.ldc = g.ldc; h.M = g.M; h.N = g.N;
    h.arena_off = al.arena_off; h.fw_off = al.fw_off; h.bw_off = al.bw_off; h.K = al.K; h.KS = al.KS; h.NS = al.NS; h.fmt = ad->fmt;
    h.alpha = ad->sc.desc ? ad->sc.desc[*ad->sc.cur].alpha : ad->alpha;
    h.omb1 = ad->omb1; h.omb2 = ad->omb2; h.eps = ad->eps;
    h.w = ad->w; h.m = ad->m; h.v = ad->v; h.fw = ad->fw; h.bw = ad->bw;
  }
  float am[4], av[4], aw[4];
  if constexpr (ADAM && T == 1) {
    const NtAdamLayer& al = ad->lt[pi];
    const int n = min(n0 + li, g.N - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = min(m0 + 8 * wave + 4 * lh + e, g.M - 1);
      const long long i = al.arena_off + (long long)m * g.ldc + n;
      am[e] = ad->m[i]; av[e] = ad->v[i]; aw[e] = ad->w[i];
    }
    __builtin_amdgcn_sched_barrier(0);  // (the requests stay ahead of the operand loads)
  }
  const float* ap[T];
  const float* bp[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {  // clamp: edge rows are loaded twice and discarded at the store
    ap[t] = g.A + (long long)min(m0 + 32 * t + li, g.M - 1) * g.lda + kbeg + HOFF * lh;
    bp[t] = g.B + (long long)min(n0 + 32 * t + li, g.N - 1) * g.ldb + kbeg + HOFF * lh;
  }
  f32x16 acc[T][T];
#pragma unroll
  for (int ti = 0; ti < T; ++ti)
#pragma unroll
    for (int tj = 0; tj < T; ++tj)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ti][tj][i] = 0.f;

  for (int r0 = s0; r0 < s1; r0 += ROUND) {
    float va[ROUND][T][REGS], vb[ROUND][T][REGS];
    // ---- issue every load of this round (rows are padded: reads stay in bounds)
#pragma unroll
    for (int s = 0; s < ROUND; ++s) {
      if (r0 + s < s1) {
        const int ko = (r0 + s) * KSTEP;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int q = 0; q < REGS / 4; ++q) {
            const float4 x = *reinterpret_cast<const float4*>(ap[t] + ko + 4 * q);
            const float4 y = *reinterpret_cast<const float4*>(bp[t] + ko + 4 * q);
            va[s][t][4 * q] = x.x; va[s][t][4 * q + 1] = x.y; va[s][t][4 * q + 2] = x.z; va[s][t][4 * q + 3] = x.w;
            vb[s][t][4 * q] = y.x; vb[s][t][4 * q + 1] = y.y; vb[s][t][4 * q + 2] = y.z; vb[s][t][4 * q + 3] = y.w;
          }
      }
    }
    NTFINE(1);
#pragma unroll
    for (int s = 0; s < ROUND; ++s) {
      if (r0 + s < s1) {
        // zero the elements past the end of the contraction range (padding may hold anything)
        const int kk = kbeg + (r0 + s) * KSTEP + HOFF * lh;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int e = 0; e < REGS; ++e)
            if (kk + e >= kend) { va[s][t][e] = 0.f; vb[s][t][e] = 0.f; }
        if constexpr (REGS == 4) {
#pragma unroll
          for (int ti = 0; ti < T; ++ti)
#pragma unroll
            for (int tj = 0; tj < T; ++tj)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s][ti][e], vb[s][tj][e], acc[ti][tj], 0, 0, 0);
        } else {
          typename P::frag fa[T], fb[T];
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              fa[t][e] = (typename P::elem)(va[s][t][e] * g.a_scale);
              fb[t][e] = (typename P::elem)(vb[s][t][e] * g.b_scale);
            }
#pragma unroll
          for (int ti = 0; ti < T; ++ti)
#pragma unroll
            for (int tj = 0; tj < T; ++tj) acc[ti][tj] = P::template mfma<false>(fa[ti], fb[tj], acc[ti][tj]);
        }
      }
    }
  }
  NTFINE(2);
  // ---- meet in LDS; wave w finishes accumulator registers 4w..4w+3 = rows 8w + 4h + {0..3} of each tile
#pragma unroll
  for (int ti = 0; ti < T; ++ti)
#pragma unroll
    for (int tj = 0; tj < T; ++tj)
#pragma unroll
      for (int i = 0; i < 16; ++i) part[wave][ti * T + tj][i][lane] = acc[ti][tj][i];
  __syncthreads();
  NTFINE(3);
  if constexpr (ADAM) {
    // ---- gradient element -> Keras Adam (train_kernels.h: adam_update_element) -> arena, moments, packed fp32 streams.
    // A thread holds rows mrow .. mrow + 3 of one column n: in the 8-row kernel's format (fmt 4) that is ONE 16-byte word
    // of the forward stream, and after a 4 x 4 transpose inside the quad of lanes that holds columns n & ~3 .. + 3, one
    // 16-byte word of the backward stream per lane -- 2 stores instead of 8 scattered 4-byte ones.
    typedef float f32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ti = 0; ti < T; ++ti) {
#pragma unroll
      for (int tj = 0; tj < T; ++tj) {
        const int n = n0 + 32 * tj + li;
        const int mrow = m0 + 32 * ti + 8 * wave + 4 * lh;  // rows mrow .. mrow+3
        const bool nvalid = n < h.N;
        float wnew[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int reg = 4 * wave + e, tt = ti * T + tj;
          const float v = (part[0][tt][reg][lane] + part[1][tt][reg][lane]) + (part[2][tt][reg][lane] + part[3][tt][reg][lane]);
          const int m = mrow + e;
          if (nvalid && m < h.M) {
            const long long i = h.arena_off + (long long)m * h.ldc + n;
            float m0v, v0v, w0v;
            if constexpr (T == 1) { m0v = am[e]; v0v = av[e]; w0v = aw[e]; }
            else { m0v = h.m[i]; v0v = h.v[i]; w0v = h.w[i]; }
            const float mi = m0v + (v - m0v) * h.omb1;
            const float vi = v0v + (v * v - v0v) * h.omb2;
            const float wi = w0v - (mi * h.alpha) / (sqrtf(vi) + h.eps);
            h.C[(long long)m * h.ldc + n] = v;
            h.m[i] = mi; h.v[i] = vi; h.w[i] = wi;
            if (m < h.K) wnew[e] = wi;  // (the bias row has no packed copy)
          }
        }
        if (h.fmt == 4 && mrow + 3 < h.K) {  // (the same for the four lanes of a quad: they share wave, lh, ti)
          if (nvalid)
            *reinterpret_cast<f32x4v*>(h.fw + h.fw_off + ((((long long)(n >> 6) * h.KS + (mrow >> 2)) * 64 + (n & 63)) << 2)) =
                f32x4v{wnew[0], wnew[1], wnew[2], wnew[3]};
          // 4 x 4 transpose in the quad (n & 3 == li & 3: the tile starts at a multiple of 32): two butterfly stages
          const int j = li & 3;
          auto xchg1 = [&](float give) __attribute__((always_inline)) -> float {
            return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
          };
          auto xchg2 = [&](float give) __attribute__((always_inline)) -> float {
            return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0x4E, 0xf, 0xf, true));
          };
          {  // partner j ^ 1 (quad_perm [1,0,3,2]): pairs (0,1) and (2,3)
            const float g0 = xchg1((j & 1) ? wnew[0] : wnew[1]), g1 = xchg1((j & 1) ? wnew[2] : wnew[3]);
            if (j & 1) { wnew[0] = g0; wnew[2] = g1; } else { wnew[1] = g0; wnew[3] = g1; }
          }
          {  // partner j ^ 2 (quad_perm [2,3,0,1]): pairs (0,2) and (1,3)
            const float g0 = xchg2((j & 2) ? wnew[0] : wnew[2]), g1 = xchg2((j & 2) ? wnew[1] : wnew[3]);
            if (j & 2) { wnew[0] = g0; wnew[1] = g1; } else { wnew[2] = g0; wnew[3] = g1; }
          }
          // now wnew[i] = W[mrow + j][nb + i], nb = n & ~3
          const int mj = mrow + j, nb = n & ~3;
          if (nb < h.N)
            *reinterpret_cast<f32x4v*>(h.bw + h.bw_off + ((((long long)(mj >> 6) * h.NS + (nb >> 2)) * 64 + (mj & 63)) << 2)) =
                f32x4v{wnew[0], wnew[1], wnew[2], wnew[3]};
        } else if (nvalid) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = mrow + e;
            if (m >= h.K) continue;
            long long qf, qb;  // (train_kernels.h: adam_repack_element spells the two formats out)
            if (h.fmt == 4) {
              qf = h.fw_off + ((((long long)(n >> 6) * h.KS + (m >> 2)) * 64 + (n & 63)) << 2) + (m & 3);
              qb = h.bw_off + ((((long long)(m >> 6) * h.NS + (n >> 2)) * 64 + (m & 63)) << 2) + (n & 3);
            } else {
              qf = h.fw_off + ((((long long)(n >> 5) * h.KS + 2 * (m >> 4) + ((n >> 4) & 1)) * 64 + (n & 15) + 16 * ((m >> 2) & 3)) << 2) + (m & 3);
              qb = h.bw_off + ((((long long)(m >> 5) * h.NS + 2 * (n >> 4) + ((m >> 4) & 1)) * 64 + (m & 15) + 16 * ((n >> 2) & 3)) << 2) + (n & 3);
            }
            h.fw[qf] = wnew[e];
            h.bw[qb] = wnew[e];
          }
        }
      }
    }
  } else {
  float* C = g.C + (long long)bz * g.slab_stride;
  const int gM = g.M, gN = g.N;
  const long long gldc = g.ldc;
#pragma unroll
  for (int ti = 0; ti < T; ++ti) {
#pragma unroll
    for (int tj = 0; tj < T; ++tj) {
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int reg = 4 * wave + e, tt = ti * T + tj;
        r[e] = (part[0][tt][reg][lane] + part[1][tt][reg][lane]) + (part[2][tt][reg][lane] + part[3][tt][reg][lane]);
        if constexpr (REGS == 8) r[e] *= g.out_scale;
      }
      const int n = n0 + 32 * tj + li;
      const int mrow = m0 + 32 * ti + 8 * wave + 4 * lh;  // rows mrow .. mrow+3
      if (n >= gN) continue;
      float bias = 0.f;
      if (EP == NT_FWD || EP == NT_FWD_RELU || EP == NT_FWD_UNPRE) bias = g.bias[n];
      const float amean = EP == NT_FWD_UNPRE ? g.aff_mean[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = mrow + e;
        float v = r[e];
        if (EP == NT_FWD) v = v + bias;
        if (EP == NT_FWD_RELU) v = fmaxf(v + bias, 0.f);
        if (EP == NT_FWD_UNPRE) v = (v + bias) * g.aff_std + amean;  // rounded as numpy does: (p * std) + mean
        if (EP == NT_DX_MASK) v = (m < gM && g.mask[(long long)m * g.ldmask + n] > 0.f) ? v : 0.f;
        r[e] = v;
        if (m < gM) C[(long long)m * gldc + n] = v;
      }
      if (EP != NT_DW && g.CT) {  // rows of the transposed copy are padded past the batch: no bound check on m
        float4 t4 = make_float4(mrow + 0 < g.M ? r[0] : 0.f, mrow + 1 < g.M ? r[1] : 0.f, mrow + 2 < g.M ? r[2] : 0.f,
                                mrow + 3 < g.M ? r[3] : 0.f);
        *reinterpret_cast<float4*>(g.CT + (long long)n * g.ldct + mrow) = t4;
      }
    }
  }
  }
  NTFINE(4);
#ifdef V21_CHAIN_FINE
  if constexpr (ADAM) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
  NTFINE(5);
}

template <class P, int T, class GROUP>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const GROUP grp) {
  gemm_nt_body<P, T, GROUP, false>(grp, nullptr);
}
// every weight gradient of an f32 chain step + Adam + the packed copies in one launch (NtAdamInfo above)
template <int T>
__global__ void __launch_bounds__(256) gemm_nt_dwadam_kernel(const NtGroupBig grp, const NtAdamInfo ad) {
  if (ad.loss_acc && blockIdx.x == 0 && threadIdx.x == 0) {
    const float f = (float)((double)(long long)*ad.loss_acc * (1.0 / 4294967296.0));
    *ad.loss_out = f;
    if (ad.loss_out2 && (ad.sc.desc || ad.loss_slot >= 0)) ad.loss_out2[ad.sc.desc ? ad.sc.desc[*ad.sc.cur].slot : ad.loss_slot] = f;
    *ad.loss_acc = 0ull;
  }
  gemm_nt_body<PrecF32, T, NtGroupBig, true>(grp, &ad);
}

}  // namespace v21
