// train_chain.h -- forward + backward of a whole training step for one block of 32 batch
// rows in ONE kernel (gfx950, f16 / bf16 operands, fp32 accumulation).
//
// The per-layer launches of gemm_nt.h cost ~5 us each and a Keras fit() step (emulator.py:
// 369-378, batch 256) is ~10 of them.  Rows of a batch are independent through the forward
// pass and through the activation-gradient chain, so a workgroup can carry its 32 rows through
// every layer with the activations in LDS and only workgroup barriers between layers:
//
//   gather x[idx] -> LDS (f16) and H0^T (global, for the weight gradient)
//   for each layer:  Z^T(n, m) = sum_k W^T(n, k) H^T(k, m)      "transposed orientation":
//       MFMA A operand = weights (rows = output features), B operand = activations (columns
//       = batch rows), so a lane owns ONE batch row and 4 x 4 consecutive features: the
//       result goes back to LDS as four 8-byte writes in the [row][feature] layout the
//       next layer reads with one ds_read_b128 per MFMA.
//       Weights come straight from L2 into registers as 1-KiB fragments, pre-packed by the
//       Adam kernel (train_kernels.h) in exactly the order a wave reads them; a wave issues
//       every load of a tile before its first MFMA, two waves per SIMD cover each other.
//   last layer: loss_i and dL/dz = 2 w_i (p - y) / B in the epilogue (relative_mse_loss,
//       emulator.py:68-81, as a row weight); row losses are reduced in a fixed order.
//   for each layer, top down:  dX^T(k, m) = sum_n W(k, n) dZ^T(n, m), masked by the ReLU
//       bits the forward epilogue left in LDS.
//   H_l^T and dZ_l^T go to global memory (fp32, batch-contiguous): the weight gradients
//   contract over the WHOLE batch and stay in the NT kernel (one grouped launch).
//
// Gradients are carried through the f16 operands multiplied by a power of two `gs`
// (see gemm_nt.h: a_scale) and stored to global memory unscaled.
#pragma once
#include <hip/hip_runtime.h>

#include "fused_fwd.h"

namespace v21 {

constexpr int kChainMaxDim = 512;             // widest layer the LDS layout holds
constexpr int kChainPitch = kChainMaxDim + 8; // halfs; 1040 B = 16 B mod 128 B: conflict-free ds_read_b128
constexpr int kChainWaves = 8;
constexpr int kChainMaskTiles = 112;
constexpr int kChainBufBytes = 2 * 32 * kChainPitch * 2;
constexpr int kChainMaskBytes = kChainMaskTiles * 64 * 2;
constexpr int kChainLdsBytes = kChainBufBytes + kChainMaskBytes + kChainWaves * 32 * 4 + 32 * 4 + 32 * 8 + 16;

struct ChainLayer {
  int K, N;            // Dense input / output width
  int KS4, NT;         // forward: k-steps of 16 (padded to a multiple of 4), 32-wide output tiles
  int NS4, KT;         // backward: n-steps of 16 (padded to 4), 32-wide input tiles
  int relu;            // ReLU on this layer's output
  int mask_tile;       // first tile of this layer's output mask in LDS (-1: none)
  long long fw_off, bw_off;  // fragment offsets (units of 8 elements) into the packed streams
  long long b_off;     // bias offset in the arena
  float* ht;           // input of this layer, transposed (K x Bp), written here
  float* dzt;          // gradient w.r.t. this layer's output, transposed (N x Bp), written here
};
// what stays the same from step to step: one per model (a sweep keeps a table of them in HBM)
struct ChainModel {
  int L;
  ChainLayer lt[16];
  const void* fw; const void* bw;  // packed weight streams
  const float* w;                  // arena (biases)
  long long Bp;                    // pitch of the transposed buffers
  float* partial;                  // per-workgroup loss
  float* loss_out;                 // batch loss (sum over rows): the all-reduce slot
  float* steploss;                 // nullable: per-step losses of the epoch
  unsigned* ticket;
  unsigned long long* stamps;      // diagnostics: s_memtime of workgroup 0 at every phase boundary
};
// the batch of this step (shared by every model of a sweep)
struct ChainStep {
  const float* x; long long ldx;   // source rows
  const float* y; long long ldy;   // targets (nullptr: y == x, the autoencoder)
  const float* rw;                 // row weights w_i
  const int* idx; long long first; // row m of the batch = source row idx[first + m] (or first + m)
  int rows;                        // rows of this rank's batch
  float scale;                     // 2 / B_global
  float gs;                        // gradient operand scale (power of two)
  long long step_index;            // >= 0: also store the loss at steploss[step_index]
};
struct ChainArgs : ChainModel, ChainStep {};
__device__ __forceinline__ void chain_stamp(const ChainModel& a, int i) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    a.stamps[i] = t;
  }
}

template <class P>
__device__ __forceinline__ void train_chain_body(const ChainModel& a, const ChainStep& st);

// one model: everything in the kernel-argument block
template <class P>
__global__ void __launch_bounds__(64 * kChainWaves) train_chain_kernel(const ChainArgs a) {
  train_chain_body<P>(a, a);
}
// a sweep: blockIdx.y = model, the per-model blocks in device memory
template <class P>
__global__ void __launch_bounds__(64 * kChainWaves) train_chain_group_kernel(const ChainModel* __restrict__ tab,
                                                                             const ChainStep st) {
  train_chain_body<P>(tab[blockIdx.y], st);
}

template <class P>
__device__ __forceinline__ void train_chain_body(const ChainModel& a, const ChainStep& st) {
  using frag = typename P::frag;
  using elem = typename P::elem;
  constexpr int NW = kChainWaves, PITCH = kChainPitch;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  elem(*buf)[32 * PITCH] = reinterpret_cast<elem(*)[32 * PITCH]>(chain_smem);
  unsigned short(*masks)[64] = reinterpret_cast<unsigned short(*)[64]>(chain_smem + kChainBufBytes);
  float(*red)[32] = reinterpret_cast<float(*)[32]>(chain_smem + kChainBufBytes + kChainMaskBytes);
  float* rwl = reinterpret_cast<float*>(chain_smem + kChainBufBytes + kChainMaskBytes + NW * 32 * 4);
  long long* srow = reinterpret_cast<long long*>(rwl + 32);
  int& is_last = *reinterpret_cast<int*>(srow + 32);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 32;

  chain_stamp(a, 0);
  // ---- gather: x[idx] -> buf[0] (compute type) and H0^T (fp32)
  if (tid < 32) {
    const int m = m0 + tid;
    const bool ok = m < st.rows;
    const long long s = ok ? (st.idx ? (long long)st.idx[st.first + m] : st.first + m) : 0;
    srow[tid] = s;
    rwl[tid] = ok ? st.rw[s] : 0.f;
  }
  __syncthreads();
  {
    const int K0 = a.lt[0].K, K0p = a.lt[0].KS4 * 16;
    const int m = tid & 31;
    const bool ok = m0 + m < st.rows;
    const float* xs = st.x + srow[m] * st.ldx;
    float* ht0 = a.lt[0].ht;
    constexpr int KG = (64 * NW) >> 5, NV = kChainMaxDim / KG;
    float v[NV];  // every load first (one memory round trip), then the stores
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k = (tid >> 5) + KG * i;
      v[i] = (ok && k < K0) ? xs[k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k = (tid >> 5) + KG * i;
      if (k < K0p) buf[0][m * PITCH + k] = (elem)v[i];
      if (k < K0) ht0[(long long)k * a.Bp + m0 + m] = v[i];
    }
  }
  __syncthreads();

  chain_stamp(a, 1);
  const frag* fw = reinterpret_cast<const frag*>(a.fw);
  const frag* bw = reinterpret_cast<const frag*>(a.bw);
  float lsum = 0.f;  // this lane's share of the row losses
  int cur = 0;

  // one 32-wide tile: acc(rows = features of the tile, col = batch row li) over `nch` chunks of 4 k-steps
  auto contract = [&](const frag* wsrc, const elem* act, int nch, f32x16& acc) {
    frag wv[32];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[4 * c + j] = wsrc[(4 * c + j) * 64];
      }
    const elem* ap = act + li * PITCH + 8 * lh;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const frag bv = *reinterpret_cast<const frag*>(ap + (4 * c + j) * 16);
          acc = P::template mfma<false>(wv[4 * c + j], bv, acc);
        }
      }
  };

  // ---- forward
  for (int l = 0; l < a.L; ++l) {
    const ChainLayer& ly = a.lt[l];
    const bool last = l == a.L - 1;
    const elem* act = buf[cur];
    elem* out = buf[cur ^ 1];
    const int nch = ly.KS4 >> 2;
    const float* bias = a.w + ly.b_off;
    float* htn = last ? nullptr : a.lt[l + 1].ht;
    const float* yrow = (st.y ? st.y : st.x) + srow[li] * (st.y ? st.ldy : st.ldx);
    const float wi = rwl[li];
    for (int t = wave; t < ly.NT; t += NW) {
      const int n0 = 32 * t;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        acc[r] = n < ly.N ? bias[n] : 0.f;
      }
      float yv[16];
      if (last) {  // targets of this tile: in flight together with the weights
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + 8 * (r >> 2) + 4 * lh + (r & 3);
          yv[r] = n < ly.N ? yrow[n] : 0.f;
        }
      }
      contract(fw + ly.fw_off + ((long long)t * ly.KS4) * 64 + lane, act, nch, acc);
      if (!last) {
        unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (ly.relu) acc[r] = fmaxf(acc[r], 0.f);
          bits |= (acc[r] > 0.f ? 1u : 0u) << r;
        }
        if (ly.mask_tile >= 0) masks[ly.mask_tile + t][lane] = (unsigned short)bits;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 8 * g + 4 * lh;
          uint2 pk = {P::pack2(acc[4 * g], acc[4 * g + 1]), P::pack2(acc[4 * g + 2], acc[4 * g + 3])};
          *reinterpret_cast<uint2*>(out + li * PITCH + n) = pk;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < ly.N) htn[(long long)(n + e) * a.Bp + m0 + li] = acc[4 * g + e];
        }
      } else {  // loss_i = w_i sum_j (p - y)^2,  dL/dp = scale w_i (p - y)
        const float gsc = st.scale * wi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 8 * g + 4 * lh;
          float d[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float df = n + e < ly.N ? acc[4 * g + e] - yv[4 * g + e] : 0.f;
            lsum += df * df;
            d[e] = gsc * df;
            if (n + e < ly.N) ly.dzt[(long long)(n + e) * a.Bp + m0 + li] = d[e];
          }
          uint2 pk = {P::pack2(d[0] * st.gs, d[1] * st.gs), P::pack2(d[2] * st.gs, d[3] * st.gs)};
          *reinterpret_cast<uint2*>(out + li * PITCH + n) = pk;
        }
      }
    }
    // columns the tiles did not cover, up to the next contraction's padded range: zero
    {
      const int c0 = 32 * ly.NT, c1 = last ? ly.NS4 * 16 : a.lt[l + 1].KS4 * 16;
      for (int i = tid; i < 32 * (c1 - c0); i += 64 * NW) {
        const int m = i / (c1 - c0), c = c0 + i % (c1 - c0);
        out[m * PITCH + c] = (elem)0.f;
      }
    }
    __syncthreads();
    cur ^= 1;
    chain_stamp(a, 2 + l);
  }

  // ---- loss: lanes -> rows -> workgroup, fixed order
  lsum += __shfl_xor(lsum, 32, 64);
  if (lh == 0) red[wave][li] = lsum * rwl[li];
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w][tid];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (tid == 0) {
      a.partial[blockIdx.x] = s;
      __threadfence();
      const unsigned tk = atomicAdd(a.ticket, 1u);
      is_last = tk == gridDim.x - 1;
    }
  }
  __syncthreads();
  if (is_last && tid < 64) {  // the last workgroup to get here adds the partials in block order
    __threadfence();
    double s = 0.0;
    for (int i = tid; i < (int)gridDim.x; i += 64) s += (double)a.partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (tid == 0) {
      a.loss_out[0] = (float)s;
      if (a.steploss && st.step_index >= 0) a.steploss[st.step_index] = (float)s;
      *a.ticket = 0u;
    }
  }

  chain_stamp(a, 2 + a.L);
  // ---- backward: layer l consumes dZ_l (gs-scaled, in buf[cur]) and produces dZ_{l-1}
  for (int l = a.L - 1; l >= 1; --l) {
    const ChainLayer& ly = a.lt[l];
    const ChainLayer& below = a.lt[l - 1];
    const elem* act = buf[cur];
    elem* out = buf[cur ^ 1];
    const int nch = ly.NS4 >> 2;
    const float inv = 1.0f / st.gs;
    for (int t = wave; t < ly.KT; t += NW) {
      const int k0 = 32 * t;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      contract(bw + ly.bw_off + ((long long)t * ly.NS4) * 64 + lane, act, nch, acc);
      if (below.relu) {
        const unsigned bits = masks[below.mask_tile + t][lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (bits >> r) & 1u ? acc[r] : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = k0 + 8 * g + 4 * lh;
        uint2 pk = {P::pack2(acc[4 * g], acc[4 * g + 1]), P::pack2(acc[4 * g + 2], acc[4 * g + 3])};
        *reinterpret_cast<uint2*>(out + li * PITCH + k) = pk;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < ly.K) below.dzt[(long long)(k + e) * a.Bp + m0 + li] = acc[4 * g + e] * inv;
      }
    }
    {
      const int c0 = 32 * ly.KT, c1 = below.NS4 * 16;
      for (int i = tid; i < 32 * (c1 - c0); i += 64 * NW) {
        const int m = i / (c1 - c0), c = c0 + i % (c1 - c0);
        out[m * PITCH + c] = (elem)0.f;
      }
    }
    __syncthreads();
    cur ^= 1;
    chain_stamp(a, 3 + a.L + (a.L - 1 - l));
  }
}

}  // namespace v21
