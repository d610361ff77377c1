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
constexpr int kChainYPitch = kChainMaxDim + 4;  // floats; 2064 B = 16 B mod 128 B
constexpr int kChainSmallBytes = kChainWaves * 32 * 4 + 32 * 4 + 32 * 8 + 16;
constexpr int kChainMaxLatent = 32;               // variational head: [z_mean | z_log_var] up to 64 wide
constexpr int kChainZPitch = 2 * kChainMaxLatent + 4;
constexpr int kChainZBytes = 32 * kChainZPitch * 4 + 32 * 4;  // fp32 (mu | lv) rows + kl_weight * KL per row
constexpr int kChainLdsBytes = kChainBufBytes + kChainMaskBytes + kChainSmallBytes + 32 * kChainYPitch * 4 + kChainZBytes;

struct ChainLayer {
  int K, N;            // Dense input / output width
  int KS4, NT;         // forward: k-steps of 16 (padded to a multiple of 4), 32-wide output tiles
  int NS4, KT;         // backward: n-steps of 16 (padded to 4), 32-wide input tiles
  int relu;            // ReLU on this layer's output
  int gauss;           // variational head (V21_ACT_GAUSS): N = 2*latent Dense outputs [z_mean | z_log_var]; the
                       // next layer sees z = z_mean + exp(z_log_var/2) eps (latent wide)
  int mask_tile;       // first tile of this layer's output mask in LDS (-1: none)
  long long fw_off, bw_off;  // fragment offsets (units of 8 elements) into the packed streams
  long long b_off;     // bias offset in the arena
  // operands of the weight gradient (gemm_dw16_kernel below), written here in MFMA-fragment order:
  // element (feature f, batch row b) at ((f/32 * BS + b/16) * 64 + 32*((b%16)/8) + f%32) * 8 + b%8
  void* ht16;          // input of this layer (K features; feature K is a constant row of ones)
  void* dzt16;         // gs * gradient w.r.t. this layer's output (N features)
};
// what stays the same from step to step: one per model (a sweep keeps a table of them in HBM)
struct ChainModel {
  int L;
  ChainLayer lt[16];
  const void* fw; const void* bw;  // packed weight streams
  const float* w;                  // arena (biases)
  long long BS;                    // batch steps of 16 per feature tile of the transposed buffers
  // batch loss: every workgroup adds its rows' losses as 2^-32 fixed point (an integer sum does not
  // depend on the order of arrival); gemm_dw16_kernel turns it into the float slot and clears it
  unsigned long long* loss_acc;
  unsigned long long* stamps;      // diagnostics: s_memtime of workgroup 0 at every phase boundary
  // variational head (per model, so the members of a sweep may differ): loss_i += kl_weight * KL_i;
  // eps keyed on (seed, step, row0 + row, d) as in train_kernels.h
  float kl_weight;
  int sample;
  unsigned long long seed, step;
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
  float inv_b;                     // 1 / B_global (the KL term's own gradient)
  unsigned long long row0;         // position of this rank's first row in the global batch (noise key)
  unsigned long long step_off;     // steps since the per-model `step` was stored (a sweep stores it once per epoch)
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
  float* ystg = reinterpret_cast<float*>(chain_smem + kChainBufBytes + kChainMaskBytes + kChainSmallBytes);
  constexpr int YP = kChainYPitch, ZP = kChainZPitch;
  float* zs = ystg + 32 * YP;      // (mu | lv) of the variational head, fp32
  float* klb = zs + 32 * ZP;       // kl_weight * KL_i

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 32;
  if (tid < 32) klb[tid] = 0.f;

  chain_stamp(a, 0);
  // ---- gather: x[idx] -> buf[0] (compute type) and H0^T (fp32)
  {  // wave w moves rows 4w..4w+3, lanes run along the row (256-byte segments); targets stay in LDS as fp32.
     // Two dependent round trips: the wave's four row indices, then rows + row weights together.
    const int K0 = a.lt[0].K, K0p = a.lt[0].KS4 * 16, DO = a.lt[a.L - 1].N;
    long long sr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * wave + r;  // wave-uniform: scalar loads
      sr[r] = m < st.rows ? (st.idx ? (long long)st.idx[st.first + m] : st.first + m) : 0;
    }
    if (lane < 4) {
      const int m = 4 * wave + lane;
      const long long s = lane == 0 ? sr[0] : lane == 1 ? sr[1] : lane == 2 ? sr[2] : sr[3];
      rwl[m] = m0 + m < st.rows ? st.rw[s] : 0.f;
    }
    float v[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = m0 + 4 * wave + r < st.rows;
      const float* xs = st.x + sr[r] * st.ldx;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = lane + 64 * i;
        v[r][i] = (ok && k < K0) ? xs[k] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * wave + r;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = lane + 64 * i;
        if (k < K0p) buf[0][m * PITCH + k] = (elem)v[r][i];
        if (!st.y) ystg[m * YP + k] = v[r][i];
      }
    }
    if (st.y) {  // separate targets: a second pass through the same registers
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = m0 + 4 * wave + r < st.rows;
        const float* ys = st.y + sr[r] * st.ldy;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = lane + 64 * i;
          v[r][i] = (ok && k < DO) ? ys[k] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) ystg[(4 * wave + r) * YP + lane + 64 * i] = v[r][i];
    }
  }
  __syncthreads();

  chain_stamp(a, 1);
  const frag* fw = reinterpret_cast<const frag*>(a.fw);
  const frag* bw = reinterpret_cast<const frag*>(a.bw);
  float lsum = 0.f;  // this lane's share of the row losses
  int cur = 0;

  // the 32 rows x F features in `act` -> fragment-ordered transposed copy (16 bytes per store: 8 batch
  // rows of one feature); rows past the end of the batch are written as zeros
  // `tiles`: tile count of the contraction that follows -- the waves that get one tile fewer do the flush
  auto flush_t = [&](const elem* act, int F, void* dst, int tiles) {
    const int F32 = (F + 31) & ~31;
    frag* d = reinterpret_cast<frag*>(dst);
    const int w0 = tiles % NW;  // waves [0, w0) carry the extra tile (w0 == 0: all alike)
    if (wave < w0) return;
    for (int i = tid - 64 * w0; i < 4 * F32; i += 64 * (NW - w0)) {
      const int q = i / F32, f = i % F32;
      if (f >= F) continue;
      frag v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (m0 + 8 * q + j < st.rows) ? act[(8 * q + j) * PITCH + f] : (elem)0.f;
      d[((long long)(f >> 5) * a.BS + (m0 >> 4) + (q >> 1)) * 64 + (q & 1) * 32 + (f & 31)] = v;
    }
  };

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
    frag bc[4], bn[4];  // activation fragments of chunk c+1 are read from LDS under the MFMAs of chunk c
#pragma unroll
    for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const frag*>(ap + j * 16);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nch) {
        if (c + 1 < nch) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const frag*>(ap + (4 * (c + 1) + j) * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = P::template mfma<false>(wv[4 * c + j], bc[j], acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) bc[j] = bn[j];
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
    flush_t(act, ly.K, ly.ht16, ly.NT);  // this layer's input -> operand of its weight gradient
    const float wi = rwl[li];
    for (int t = wave; t < ly.NT; t += NW) {
      const int n0 = 32 * t;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        acc[r] = n < ly.N ? bias[n] : 0.f;
      }
      contract(fw + ly.fw_off + ((long long)t * ly.KS4) * 64 + lane, act, nch, acc);
      if (ly.gauss) {  // keep (mu | lv) in fp32: the sampling pass below turns them into z
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 8 * g + 4 * lh;
          if (n < ZP) *reinterpret_cast<f32x4*>(zs + li * ZP + n) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        }
      } else if (!last) {
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
        }
      } else {  // loss_i = w_i sum_j (p - y)^2,  dL/dp = scale w_i (p - y)
        const float gsc = st.scale * wi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 8 * g + 4 * lh;
          const f32x4 yq = *reinterpret_cast<const f32x4*>(ystg + li * YP + n);
          float d[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float df = n + e < ly.N ? acc[4 * g + e] - yq[e] : 0.f;
            lsum += df * df;
            d[e] = gsc * df;
          }
          uint2 pk = {P::pack2(d[0] * st.gs, d[1] * st.gs), P::pack2(d[2] * st.gs, d[3] * st.gs)};
          *reinterpret_cast<uint2*>(out + li * PITCH + n) = pk;
        }
      }
    }
    if (ly.gauss) {  // z = mu + exp(lv/2) eps -> the next layer's operand image; KL_i -> the row's loss
      __syncthreads();
      const int LAT = ly.N >> 1, c1 = a.lt[l + 1].KS4 * 16;
      if (tid < 32) {
        const bool ok = m0 + tid < st.rows;
        float kl = 0.f;
        for (int d = 0; d < c1; ++d) {
          float z = 0.f;
          if (d < LAT) {
            const float mu = zs[tid * ZP + d], lv = zs[tid * ZP + LAT + d];
            const float sd = expf(0.5f * lv);
            const float e = a.sample ? gauss_eps(a.seed, a.step + st.step_off, st.row0 + m0 + tid, d) : 0.f;
            z = mu + sd * e;
            kl += -0.5f * (1.0f + lv - mu * mu - sd * sd);
          }
          out[tid * PITCH + d] = (elem)z;
        }
        klb[tid] = ok ? a.kl_weight * kl : 0.f;
      }
    } else
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

  // ---- loss: lanes -> rows -> workgroup (fixed order) -> one fixed-point atomic per workgroup
  lsum += __shfl_xor(lsum, 32, 64);
  if (lh == 0) red[wave][li] = lsum * rwl[li];
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w][tid];
    s += klb[tid];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (tid == 0) atomicAdd(a.loss_acc, (unsigned long long)(long long)llrint((double)s * 4294967296.0));
  }

  chain_stamp(a, 2 + a.L);
  // gs * dL/dz (latent wide, in `b`) -> gs * dL/d[mu | lv] in place:  d mu = dz + beta mu,
  // d lv = dz eps exp(lv/2)/2 + beta (exp lv - 1)/2, beta = kl_weight / B (the KL term's own gradient)
  auto gauss_backward = [&](elem* b, int LAT) {
    if (tid < 32) {
      const int pad = ((2 * LAT + 63) & ~63);
      for (int d = 0; d < LAT; ++d) {
        const float mu = zs[tid * ZP + d], lv = zs[tid * ZP + LAT + d];
        const float sd = expf(0.5f * lv);
        const float e = a.sample ? gauss_eps(a.seed, a.step + st.step_off, st.row0 + m0 + tid, d) : 0.f;
        const float g = (float)b[tid * PITCH + d];
        const bool ok = m0 + tid < st.rows;
        const float kb = ok ? st.gs * a.kl_weight * st.inv_b : 0.f;
        b[tid * PITCH + d] = (elem)(g + kb * mu);                                          // d mu (in place)
        b[tid * PITCH + LAT + d] = (elem)(g * e * 0.5f * sd + kb * 0.5f * (sd * sd - 1.0f));  // d lv (columns past dz)
      }
      for (int d = 2 * LAT; d < pad; ++d) b[tid * PITCH + d] = (elem)0.f;
    }
  };
  // ---- backward: layer l consumes dZ_l (gs-scaled, in buf[cur]) and produces dZ_{l-1}
  for (int l = a.L - 1; l >= 1; --l) {
    const ChainLayer& ly = a.lt[l];
    const ChainLayer& below = a.lt[l - 1];
    const elem* act = buf[cur];
    elem* out = buf[cur ^ 1];
    const int nch = ly.NS4 >> 2;
    if (ly.gauss) { gauss_backward(buf[cur], ly.N >> 1); __syncthreads(); }
    flush_t(act, ly.N, ly.dzt16, ly.KT);  // gs * dZ of this layer's output -> operand of its weight gradient
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
  if (a.lt[0].gauss) { gauss_backward(buf[cur], a.lt[0].N >> 1); __syncthreads(); }
  flush_t(buf[cur], a.lt[0].N, a.lt[0].dzt16, 0);
}

// ---- weight gradients from the fragment-ordered operands: [dW; db](k, n) = sum_b HT(k, b) dZT(n, b).
// 64x64 output tile per 256-thread workgroup; the four waves split the batch range of the slice,
// every fragment is one 1-KiB wave load straight into MFMA operand registers (no conversion, no
// masking: rows past the batch are zeros); partial tiles meet in LDS; slices -> slabs (fixed order).
struct Dw16Args {
  const void* A; const void* B;  // ht16 (M = K+1 features), dzt16 (N features)
  float* C; long long ldc;       // slab z at C + z*slab_stride
  int M, N;
  int nx, ny, nz;                // 64-tiles along N, along M; batch slices
  int steps, steps_per_slice;    // batch steps of 16 in all / per slice
  long long BS;
  long long slab_stride;
  float out_scale;
  // on the first problem of a model: fixed-point batch loss -> float slot(s), accumulator cleared
  unsigned long long* loss_acc; float* loss_out; float* loss_out2;
};
struct Dw16Group {
  Dw16Args p[kNtMaxGroup];
  int first[kNtMaxGroup + 1];
  int count;
};
template <class P>
__global__ void __launch_bounds__(256) gemm_dw16_kernel(const Dw16Group grp) {
  using frag = typename P::frag;
  int pi = 0;
  while (pi + 1 < grp.count && (int)blockIdx.x >= grp.first[pi + 1]) ++pi;
  const Dw16Args& g = grp.p[pi];
  const int bid = blockIdx.x - grp.first[pi];
  const int bx = bid % g.nx, by = (bid / g.nx) % g.ny, bz = bid / (g.nx * g.ny);
  if (g.loss_acc && bid == 0 && threadIdx.x == 0) {
    const float f = (float)((double)(long long)*g.loss_acc * (1.0 / 4294967296.0));
    *g.loss_out = f;
    if (g.loss_out2) *g.loss_out2 = f;
    *g.loss_acc = 0ull;
  }
  __shared__ __attribute__((aligned(16))) float part[4][4][16][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int sbeg = bz * g.steps_per_slice, send = min(g.steps, sbeg + g.steps_per_slice);
  const int per = (send - sbeg + 3) / 4;
  const int s0 = sbeg + wave * per, s1 = min(send, s0 + per);
  const frag* A = reinterpret_cast<const frag*>(g.A);
  const frag* B = reinterpret_cast<const frag*>(g.B);
  const frag* ap[2] = {A + ((long long)(2 * by) * g.BS) * 64 + lane, A + ((long long)(2 * by + 1) * g.BS) * 64 + lane};
  const frag* bp[2] = {B + ((long long)(2 * bx) * g.BS) * 64 + lane, B + ((long long)(2 * bx + 1) * g.BS) * 64 + lane};
  // feature tiles past the end of the operand (odd tile counts) are clamped: their rows are never stored
  const int mt = (g.M + 31) / 32, nt = (g.N + 31) / 32;
  if (2 * by + 1 >= mt) ap[1] = ap[0];
  if (2 * bx + 1 >= nt) bp[1] = bp[0];
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr int MAXS = kNtMaxKPerWg / 16 / 4;  // 8 batch steps per wave
  frag fa[MAXS][2], fb[MAXS][2];
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s0 + s < s1) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[s][t] = ap[t][(long long)(s0 + s) * 64];
        fb[s][t] = bp[t][(long long)(s0 + s) * 64];
      }
    }
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s0 + s < s1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = P::template mfma<false>(fa[s][i], fb[s][j], acc[i][j]);
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wave][2 * i + j][r][lane] = acc[i][j][r];
  __syncthreads();
  float* C = g.C + (long long)bz * g.slab_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = 64 * bx + 32 * j + li;
      const int mrow = 64 * by + 32 * i + 8 * wave + 4 * lh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int reg = 4 * wave + e, tt = 2 * i + j;
        const float v = ((part[0][tt][reg][lane] + part[1][tt][reg][lane]) + (part[2][tt][reg][lane] + part[3][tt][reg][lane])) * g.out_scale;
        if (n < g.N && mrow + e < g.M) C[(long long)(mrow + e) * g.ldc + n] = v;
      }
    }
}

// ---- the same contraction for large batches: 128x128 output tile per workgroup, operands staged through LDS.
// gemm_dw16_kernel pulls 1 KiB per MFMA into the CU (every wave loads its own A and B fragments); here a batch
// step's 4 A + 4 B fragments enter LDS once (LDS-DMA, already lane-linear) and serve all four waves, each of
// which owns a 64x64 quadrant over the whole batch slice: 0.5 KiB per MFMA and no cross-wave reduction.
// Two stages of kDwStageSteps batch steps double-buffer the stream: one barrier per stage.
#ifndef V21_DW_STAGE
#define V21_DW_STAGE 8
#endif
constexpr int kDwStageSteps = V21_DW_STAGE;
constexpr int kDwLdsBytes = 2 * kDwStageSteps * 8 * kFragBytes;
template <class P>
__global__ void __launch_bounds__(256) gemm_dw16_lds_kernel(const Dw16Group grp) {
  using frag = typename P::frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
  int pi = 0;
  while (pi + 1 < grp.count && (int)blockIdx.x >= grp.first[pi + 1]) ++pi;
  const Dw16Args& g = grp.p[pi];
  const int bid = blockIdx.x - grp.first[pi];
  const int bx = bid % g.nx, by = (bid / g.nx) % g.ny, bz = bid / (g.nx * g.ny);
  if (g.loss_acc && bid == 0 && threadIdx.x == 0) {
    const float f = (float)((double)(long long)*g.loss_acc * (1.0 / 4294967296.0));
    *g.loss_out = f;
    if (g.loss_out2) *g.loss_out2 = f;
    *g.loss_acc = 0ull;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wi = wave >> 1, wj = wave & 1;
  const int sbeg = bz * g.steps_per_slice, send = min(g.steps, sbeg + g.steps_per_slice);
  const int nst = (send - sbeg + kDwStageSteps - 1) / kDwStageSteps;
  const int mt = (g.M + 31) / 32, nt = (g.N + 31) / 32;
  // fragment q of a step: q < 4 -> A tile 4*by + q, else B tile 4*bx + q - 4 (tiles past the operand: clamped,
  // their rows/columns are never stored).  Wave w moves fragments w and w + 4 of every step of a stage.
  const unsigned char* src[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = wave + 4 * u;
    const int tile = q < 4 ? min(4 * by + q, mt - 1) : min(4 * bx + q - 4, nt - 1);
    src[u] = reinterpret_cast<const unsigned char*>(q < 4 ? g.A : g.B) + ((long long)tile * g.BS) * kFragBytes;
  }
  auto issue = [&](int st) {  // stage st -> buffer st & 1; steps past the slice re-read its last step (unused)
    const unsigned base = lds_addr(dw_smem) + (st & 1) * (kDwStageSteps * 8 * kFragBytes);
#pragma unroll
    for (int s = 0; s < kDwStageSteps; ++s) {
      const int step = min(sbeg + st * kDwStageSteps + s, send - 1);
#pragma unroll
      for (int u = 0; u < 2; ++u)
        glds16(src[u] + (long long)step * kFragBytes, lane * 16, base + (s * 8 + wave + 4 * u) * kFragBytes);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  issue(0);
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // stage st is in LDS; everyone is done with st-1
    if (st + 1 < nst) issue(st + 1);
    const unsigned char* buf = dw_smem + (st & 1) * (kDwStageSteps * 8 * kFragBytes) + lane * 16;
    const int ns = min(kDwStageSteps, send - sbeg - st * kDwStageSteps);
#pragma unroll
    for (int s = 0; s < kDwStageSteps; ++s)
      if (s < ns) {
        frag fa[2], fb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[t] = *reinterpret_cast<const frag*>(buf + (s * 8 + 2 * wi + t) * kFragBytes);
          fb[t] = *reinterpret_cast<const frag*>(buf + (s * 8 + 4 + 2 * wj + t) * kFragBytes);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = P::template mfma<false>(fa[i], fb[j], acc[i][j]);
      }
  }
  float* C = g.C + (long long)bz * g.slab_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = 128 * bx + 64 * wj + 32 * j + li;
      const int m0r = 128 * by + 64 * wi + 32 * i + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0r + (r & 3) + 8 * (r >> 2);
        if (n < g.N && m < g.M) C[(long long)m * g.ldc + n] = acc[i][j][r] * g.out_scale;
      }
    }
}

}  // namespace v21
