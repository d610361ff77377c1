// train_chain.h -- forward + backward of a whole training step for one block of 32 batch
// rows in ONE kernel (gfx950, f16 / bf16 operands, fp32 accumulation).
//
// The per-layer launches of gemm_nt.h cost ~5 us each and a Keras fit() step (emulator.py:
// 369-378, batch 256) is ~10 of them.  Rows of a batch are independent through the forward
// pass and through the activation-gradient chain, so a workgroup can carry its 32 rows through
// every layer with the activations in LDS and only workgroup barriers between layers:
//
//   gather x[idx] -> LDS (f16; the fp32 rows stay in LDS as the loss targets when y == x)
//   for each layer:  Z^T(n, m) = sum_k W^T(n, k) H^T(k, m)      "transposed orientation":
//       MFMA A operand = weights (rows = output features), B operand = activations (columns
//       = batch rows), so a lane owns ONE batch row and 4 x 4 consecutive features: the
//       result goes back to LDS as four 8-byte writes in the [row][feature] layout the
//       next layer reads with one ds_read_b128 per MFMA.
//       Weights come straight from L2 into registers as 1-KiB fragments, pre-packed by the
//       Adam kernel (train_kernels.h) in exactly the order a wave reads them.  A wave owns the
//       output tiles t = wave, wave + 16, ... and walks their fragments with a ROLLING prefetch:
//       while the four fragments of one chunk feed the MFMAs, the next chunk -- of this tile, of
//       the wave's next tile, or of its first tile in a LATER layer -- is in flight (weights do
//       not depend on activations, so the loads cross tile ends, layer barriers and the
//       forward/backward turn).  What bounds the kernel is how fast ONE CU pulls the packed
//       weight set (1.2 MB for the autoencoder; every workgroup needs all of it): the load path
//       moves 64 B/clk per CU (~130 GB/s, scripts/diag/stream_probe.hip), i.e. >= 9 us.
//   last layer: loss_i and dL/dz = 2 w_i (p - y) / B in the epilogue (relative_mse_loss,
//       emulator.py:68-81, as a row weight); row losses are reduced in a fixed order.
//   for each layer, top down:  dX^T(k, m) = sum_n W(k, n) dZ^T(n, m), masked by the ReLU
//       bits the forward epilogue left in LDS.
//   H_l^T and dZ_l^T leave the kernel as f16 MFMA fragments (hardware-transposed LDS reads,
//   16-byte coalesced stores): the weight gradients contract over the WHOLE batch and are a
//   separate grouped launch (gemm_dw16*).
//
// Gradients are carried through the f16 operands multiplied by a power of two `gs`
// (see gemm_nt.h: a_scale) and stored to global memory unscaled.
#pragma once
#include <hip/hip_runtime.h>

#include "fused_fwd.h"
#include "train_kernels.h"

namespace v21 {

constexpr int kChainMaxDim = 512;             // widest layer the LDS layout holds
constexpr int kChainPitch = kChainMaxDim + 8; // halfs; 1040 B = 16 B mod 128 B: conflict-free ds_read_b128
constexpr int kChainWaves = 16;  // 4 per SIMD: an 11- or 15-tile layer is ONE round of tiles, and 16 streams keep the CU's load path busy
constexpr int kChainMaskTiles = 112;
constexpr int kChainBufBytes = 2 * 32 * kChainPitch * 2;
constexpr int kChainMaskBytes = kChainMaskTiles * 64 * 2;
constexpr int kChainYPitch = kChainMaxDim + 4;  // floats; 2064 B = 16 B mod 128 B
constexpr int kChainSmallBytes = kChainWaves * 32 * 4 + 32 * 4 + 32 * 8 + 16;
constexpr int kChainMaxLatent = 32;               // variational head: [z_mean | z_log_var] up to 64 wide
constexpr int kChainZPitch = 2 * kChainMaxLatent + 4;
constexpr int kChainZBytes = 32 * kChainZPitch * 4 + 32 * 4;  // fp32 (mu | lv) rows + kl_weight * KL per row
constexpr int kChainLdsBytes = kChainBufBytes + kChainMaskBytes + kChainSmallBytes + 32 * kChainYPitch * 4 + kChainZBytes;

// (ChainLayer / ChainModel / ChainStep / ChainArgs: chain_types.h)
__device__ __forceinline__ void chain_stamp(const ChainModel& a, int i) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.stamps) {  // (physical block 0 is row block 0)
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    a.stamps[i] = t;
  }
}

// diagnostic builds (wrong results): the kernel without its weight loads / without its MFMAs.  r3, autoencoder stack, 4,096
// rows, cycles of workgroup 0 (scripts/train_probe.py): whole kernel 51.5 k; without the weight loads 38.9 k; without
// loads AND MFMAs 33.6 k -- the gather (6.2 k), the activation operands read from LDS by every wave for its tile (32 KB
// per wave and layer), flush, epilogues and barriers are two thirds of the kernel; with the loads but without MFMAs 60.9 k:
// a chunk then has no matrix work to hide behind and a wave pays ~1.5 k cycles per chunk, eight chunks per big layer.
// Without the operand reads as well (-DV21_C16_NOLDS): 32.4 k -- a layer with no contraction at all still takes 2 k cycles
// (bias permutes, epilogue, search for the next tile, barrier), 4.4-4.9 k where few waves carry the flush.
#ifdef V21_C16_NOLOAD
#define C16LOAD(dst, src) asm volatile("" : "+v"(dst))
#else
#define C16LOAD(dst, src) dst = src
#endif
#ifdef V21_C16_NOLDS
#define C16LDS(dst, src) asm volatile("" : "+v"(dst))
#else
#define C16LDS(dst, src) dst = src
#endif
#ifdef V21_C16_NOMFMA
#define C16MFMA(w, b) do { const f32x4 w4_ = __builtin_bit_cast(f32x4, w), b4_ = __builtin_bit_cast(f32x4, b); asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[0]) : "v"(w4_[0]), "v"(b4_[0])); } while (0)
#else
#define C16MFMA(w, b) acc = P::template mfma<false>(w, b, acc)
#endif
#ifdef V21_CHAIN_FINE  // (diagnostic build: per-wave stamps of workgroup 0, scripts/diag/chain_wave_stamps.py)
#define CFINE(i) do { if (blockIdx.x == 0 && lane == 0 && a.stamps) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); a.stamps[64 + (i) * 16 + wave] = t_; } } while (0)
#else
#define CFINE(i)
#endif
// FEAT: which features of the general body an instantiation carries (see train_chain_body)
constexpr int kChainGauss = 1, kChainJoint = 2, kChainOut = 4, kChainAll = 7, kChainFwd = 8;  // (kChainFwd REMOVES code: no backward pass)
template <class P, int FEAT = kChainAll>
__device__ __forceinline__ void train_chain_body(const ChainModel& a, const ChainStep& st, int bid);

// Every row-block workgroup streams the model's whole packed weight set, and the Adam kernel has just rewritten it:
// the lines are in no L2.  All workgroups of an XCD ask for the same line at about the same time, so each of them
// waits for the miss (~700 ns against ~300 for a hit), once per chunk of its rolling prefetch -- and a CU pulls only
// ~55 GB/s of lines that miss its L2.  A batch leaves CUs idle (4,096 rows: 128 of 256; 256 rows: 248): workgroups
// placed there touch every line of the streams once, npref of them per XCD in parallel, while the row-block
// workgroups are still gathering their rows.  (The same touches issued by the row-block workgroups themselves were
// measured useless in r2: they queue in the very miss path they are meant to relieve.)  Measured: chain 66.5 k ->
// 62.4 k cycles at 4,096 rows, 68.3 k -> 61.3 k at 256 rows; 4 prefetchers per XCD do as well as 16.
__device__ __forceinline__ void chain_prefetch(const ChainModel& a, const ChainStep& st) {
  const int p = ((int)blockIdx.x - st.ncons) >> 3;
  const long long nf = (a.fw_bytes + 127) >> 7, nb = (a.bw_bytes + 127) >> 7;
  for (long long i = (long long)p * blockDim.x + threadIdx.x; i < nf + nb; i += (long long)st.npref * blockDim.x) {
    const char* ptr = i < nf ? reinterpret_cast<const char*>(a.fw) + (i << 7) : reinterpret_cast<const char*>(a.bw) + ((i - nf) << 7);
    unsigned v;
    // (waited for inside the statement: the destination register must not be reused while the load is in flight)
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
  }
}

// one model: everything in the kernel-argument block
template <class P, int FEAT = kChainAll>
__global__ void __launch_bounds__(64 * kChainWaves) train_chain_kernel(const ChainArgs a) {
  if ((int)blockIdx.x >= a.ncons) { chain_prefetch(a, a); return; }
  train_chain_body<P, FEAT>(a, a, (int)blockIdx.x);
}
// a sweep: G models, the per-model blocks in device memory.  r5: a one-dimensional grid of 8 * ncons * ceil(G / 8)
// workgroups, dealt so that ALL row blocks of a model carry the same XCD label (workgroups b, b + 8, ... share an XCD:
// observed round-robin dispatch; a wrong guess costs speed, never results): label x = b % 8 serves the models x, x + 8, ...,
// slot j = b / 8 is row block j % ncons of that label's (j / ncons)-th model.  Until r4 the grid was (ncons, G) with
// blockIdx.y = model: row block x of EVERY model sat on XCD x, every XCD pulled every member's packed weights through its
// own L2 (32 members: 318 MB of HBM-side traffic per launch against ~40 MB of weights; profiles/r5/pmc_sweep_f16_m32.json).
template <class P, bool GAUSS = true>  // GAUSS = false: no member of the sweep has a variational layer
__global__ void __launch_bounds__(64 * kChainWaves) train_chain_group_kernel(const ChainModel* __restrict__ tab,
                                                                             const ChainStep st, const int G) {
  const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
  const int model = x + 8 * (j / st.ncons);
  if (model >= G) return;  // (G is not a multiple of 8: the last round of labels is partly empty)
  train_chain_body<P, GAUSS ? kChainGauss : 0>(tab[model], st, j % st.ncons);
}

// joint step (BASELINE configs[2]): the autoencoder (signals -> signals) and the latent emulator (parameters -> latent)
// take one optimizer step each on the same rows, and the emulator's targets are the latents the encoder produces for
// those rows in this very step: the reference's frozen-encoder targets (emulator.py:753-754) without leaving the chip.
// TWO families of row-block workgroups run side by side (r2 ran both models one after the other in ONE workgroup: the
// step lasted as long as the two chains together):
//   blocks [0, ncons)          autoencoder: the plain chain body;
//   blocks [ncons, 2 ncons)    emulator: first the ENCODER alone on these rows (forward layers [0, zcap_layer], ~2 of the
//                              autoencoder's layers; the latents stay in LDS as fp32), then the emulator's chain body
//                              with those latents as targets.  Recomputing the encoder costs ~1/6 of an autoencoder
//                              pass and removes every dependence between the two families;
//   blocks [2 ncons, ...)      prefetchers of both models' weight streams.
template <class P, bool GAUSS = true>  // GAUSS = false: the autoencoder is not variational (the emulator never is)
__global__ void __launch_bounds__(64 * kChainWaves) train_chain_joint_kernel(const ChainModel* __restrict__ tab /* [2], device */,
                                                                             const ChainStep sa, const ChainStep sb) {
  constexpr int FA = kChainJoint | (GAUSS ? kChainGauss : 0);
  if ((int)blockIdx.x >= 2 * sa.ncons) {
    ChainStep sp = sa;
    sp.ncons = 2 * sa.ncons;
    chain_prefetch(tab[0], sp); chain_prefetch(tab[1], sp);
    return;
  }
  if ((int)blockIdx.x < sa.ncons) { train_chain_body<P, FA>(tab[0], sa, (int)blockIdx.x); return; }
  ChainStep se = sa;  // the encoder alone, for the emulator's rows
  se.fwd_only = 1; se.nfwd = tab[0].zcap_layer + 1; se.blk0 = sa.ncons;
  train_chain_body<P, FA>(tab[0], se, (int)blockIdx.x);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the encoder's last LDS accesses precede the emulator's gather
  train_chain_body<P, kChainJoint>(tab[1], sb, (int)blockIdx.x);
}

// LDS-only rendezvous: own LDS writes retired, then the barrier.  Deliberately NOT __syncthreads():
// its fence would also drain the weight loads that are in flight across the barrier (vmcnt(0)).
__device__ __forceinline__ void chain_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (chain_s4 / chain_s8 / chain_tr_read: chain_types.h)

// FEAT: the features this instantiation carries -- kChainGauss: a variational layer (sampling, KL, its backward);
// kChainJoint: the joint step's captured latents, targets from LDS, encoder-only passes, shifted block numbers;
// kChainOut: FORWARD mode (outputs to global memory, fused parameter transform; no loss, no backward pass); kChainFwd: the
// launch is forward-only (validation, FORWARD mode: st.fwd_only is set) -- the backward pass is not compiled.  The flag set is
// a promise of the host (which model, which launch), and it folds every branch on an absent feature away at compile time:
// the general body (all three) sits at its 128 vector registers with 43 scalar registers spilled into lanes of one of
// them and 5 vector registers in scratch; FEAT = 0 -- what a trainer of the reference's stacks launches -- has 15 and 0,
// and its step is 1.4 us shorter at every batch size (r3: 46.0 -> 44.6 us at 4,096 rows, 33.3 -> 31.9 at 256).
template <class P, int FEAT>
__device__ __forceinline__ void train_chain_body(const ChainModel& a, const ChainStep& st, const int bid /* the workgroup's block number: blockIdx.x, or the slot a grouped launch assigns */) {
#define GG(cond) ((FEAT & kChainGauss) ? (bool)(cond) : false)
#define GJ(cond) ((FEAT & kChainJoint) ? (bool)(cond) : false)
#define GO(cond) ((FEAT & kChainOut) ? (bool)(cond) : false)
  using frag = typename P::frag;
  using elem = typename P::elem;
  constexpr int NW = kChainWaves, PITCH = kChainPitch, RPW = 32 / NW;  // RPW: batch rows a wave gathers
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
  const long long first = st.sc.desc ? st.sc.desc[*st.sc.cur].first : st.first;
  // XCD-major row blocks: workgroups b, b + 8, ... share an XCD (observed round-robin dispatch; speed only), and
  // XCD x carries the CONSECUTIVE row blocks x * ceil(nb / 8) ...: the batch slice whose weight-gradient tiles
  // gemm_dw16* then runs on the same XCD, so that it finds the operands this kernel wrote in its own L2.
  const int nrb = (st.rows + 31) >> 5;
  const int bidx = bid - ((FEAT & kChainJoint) ? st.blk0 : 0);
  const int rb = (bidx & 7) * ((nrb + 7) >> 3) + (bidx >> 3);
  if (rb >= nrb) return;  // (the grid is rounded up to a multiple of 8)
  const int m0 = rb * 32;
  const int vrows = st.rows - m0;  // valid rows of this block (>= 1; < 32 only in the last block)
  if (tid < 32) klb[tid] = 0.f;

  const frag* fw = reinterpret_cast<const frag*>(a.fw);
  const frag* bw = reinterpret_cast<const frag*>(a.bw);

  // ---- the wave's weight stream (see the header): a prefetch fetches a chunk of four fragments and, for a
  // forward tile, its bias (lane i: bias[32 t + i], 0 past the layer's width; the tile start hands the 32
  // values out by readlane).  Vector loads on purpose: the compiler cannot prove the arena read-only (no
  // scalar loads), and sixteen dependent loads per lane at the start of a tile would stall the stream.
  // (r3, measured and dropped: splitting the CONTRACTION of single-tile layers across the waves, as train_chain32.h does --
  // 352 -> 9 forward and 32 <- 352 backward keep one wave busy for six chunks end to end, 3.5 k + 3.1 k cycles of 55 k.
  // This kernel sits at its 128 registers with 43 scalar registers already spilled; a second site of the contraction or of
  // the epilogue, or both folded into a two-pass loop with one site each, spills 68-156 vector registers to scratch: the
  // step went from 46 to 114 us.  The partial tiles also cannot simply meet in the free activation image here: as 16-bit
  // operands fp32 sums are garbage -- one half-word in 32 carries the exponent of a NaN / Inf -- and the image's padding
  // columns are read again against zero weights, so the area must be cleared after the meeting.)
  struct Job { const frag* w; const float* b; bool bok; };
  auto fwd_job = [&](int l, int t) __attribute__((always_inline)) -> Job {
    const ChainLayer& ly = a.lt[l];
    const int n = 32 * t + li;
    return Job{fw + ly.fw_off + ((long long)t * ly.KS) * 64 + lane, a.w + ly.b_off + (n < ly.N ? n : ly.N - 1), n < ly.N};
  };
  auto bwd_job = [&](int l, int t) __attribute__((always_inline)) -> Job {
    const ChainLayer& ly = a.lt[l];
    return Job{bw + ly.bw_off + ((long long)t * ly.NS) * 64 + lane, a.w, false};
  };
  auto bwd_from = [&](int l) __attribute__((always_inline)) -> Job {  // first tile of this wave in the activation-gradient layers <= l
    for (; l >= 1; --l)
      if (wave < a.lt[l].KT) return bwd_job(l, wave);
    return Job{fw + lane, a.w, false};  // nothing left: any valid address (the data is never used)
  };
  auto fwd_from = [&](int l) __attribute__((always_inline)) -> Job {  // first tile of this wave in forward layers >= l, else backward
    for (; l < a.L; ++l)
      if (wave < a.lt[l].NT) return fwd_job(l, wave);
    return bwd_from(a.L - 1);
  };
  // the wave's row indices first: their load is the oldest in flight, so the gather below waits for it alone and not
  // for the (cold) weight fragments requested next
  const int mq = m0 + RPW * wave + (lane & (RPW - 1));
  long long srow = mq < st.rows ? first + mq : first + m0;  // clamped: always a valid position
  if (st.idx) srow = st.idx[srow];
  // (r3, measured and dropped: THREE chunk buffers in rotation -- two chunks of a tile in flight while the third feeds the
  // MFMAs, the registers taken from the activation operands (a ring of four fragments read three k-steps ahead instead of
  // two sets of four) -- with the contraction instantiated for the three rotation phases, so that no registers move.
  // Correct, and 2-3x slower: which buffer holds the next tile's first chunk across an epilogue depends on the tile's
  // chunk count modulo 3, so all 48 registers stay live there, and 300-700 vector registers go to scratch in the hot
  // paths: 103-146 us per step.  With EIGHT waves of up to 256 registers (kChainWaves = 8: each wave takes two tiles of
  // a big layer one after the other) the same ring compiles to 255 registers, nothing spilled -- and is no faster: the big
  // layers take 10.3 / 11.4 / 8.5 k cycles against 9.7 / 9.8 / 8.0 k, the small ones more (every tile padded to two
  // chunks): 48.3 against 44.3 us per step.  Eight waves with the two-buffer scheme: 45.2 us -- half the waves, twice the
  // tiles each, the same time per layer.  The chunk round trips are not what a big layer waits for.)
  frag wa[4], wb[4];  // chunk in use / chunk in flight (the roles alternate)
  float bnext;        // bias values of the wave's next forward tile (in flight with its first chunk)
  {
    const Job j0 = fwd_from(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[j] = j0.w[j * 64];
    bnext = j0.bok ? *j0.b : 0.f;
  }

  chain_stamp(a, 0);
  // buf[1] <- 0, once.  A contraction reads its operand image up to the padded k range; the columns past a layer's last
  // 32-wide tile are never written again, and whatever finite value they hold meets a zero weight.  What must not be
  // there is the NaN / Inf bit patterns another kernel may have left in LDS: hence this clear (buf[0] is written in
  // full by the gather).  It replaces a zero-fill per layer (~250 cycles each, ten layers).
  for (int i = tid; i < 32 * PITCH * (int)sizeof(elem) / 16; i += 64 * NW)
    reinterpret_cast<uint4*>(buf[1])[i] = uint4{0u, 0u, 0u, 0u};
  // ---- gather: x[idx] -> buf[0] (compute type) and the fp32 target rows
  {  // wave w moves rows RPW w .. RPW w + RPW - 1, lanes run along the row (256-byte segments); targets stay in LDS as fp32.
     // Two dependent round trips: the wave's row indices (one vector load, broadcast by readlane), then
     // rows + row weights together.  Branch-free: loads past the end of a row / of the batch are clamped to
     // a valid address and replaced by 0, and every lane writes its whole strip of the LDS images (columns
     // >= K are the zero padding the first contraction reads).
    const int K0 = a.lt[0].K, DO = a.lt[a.L - 1].N;
    const float rwv = st.rw ? st.rw[srow] : 0.f;
    long long sr[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      sr[r] = ((long long)__builtin_amdgcn_readlane((int)(srow >> 32), r) << 32) |
              (unsigned)__builtin_amdgcn_readlane((int)srow, r);
    }
    if (lane < RPW) rwl[RPW * wave + lane] = mq < st.rows ? rwv : 0.f;
    float v[RPW][8];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      // (a select on a wave-uniform condition would make hipcc branch around every load and drain
      // vmcnt(0) after each: the row's validity is folded into the per-lane column bound instead)
      const int kmax = m0 + RPW * wave + r < st.rows ? K0 : 0;
      const float* xs = st.x + sr[r] * st.ldx;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = lane + 64 * i;
        float t = xs[k < K0 ? k : K0 - 1];
        if (i == 0 && GO(st.tin)) {  // par_transform on the (<= 8) input columns, as affine_in_kernel does it
          const int jc = lane < K0 ? lane : 0;
          t = par_transform_f32(t, st.tin->log_mask[jc], st.tin->zero_floor[jc], st.tin->lo[jc], st.tin->span[jc]);
        }
        v[r][i] = k < kmax ? t : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int m = RPW * wave + r;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = lane + 64 * i;  // <= 511 < PITCH, YP
        buf[0][m * PITCH + k] = (elem)v[r][i];
        if (!st.y && !GJ(st.y_from_lds) && !GO(st.out)) ystg[m * YP + k] = v[r][i];
      }
    }
    if (GJ(st.y_from_lds)) {  // rows of the previous model's captured layer (fp32, in the variational head's buffer)
      const int DOl = a.lt[a.L - 1].N;
      for (int i = tid; i < 32 * DOl; i += 64 * NW) ystg[(i / DOl) * YP + i % DOl] = zs[(i / DOl) * ZP + i % DOl];
    } else if (st.y && !GO(st.out)) {  // separate targets: a second pass through the same registers
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int kmax = m0 + RPW * wave + r < st.rows ? DO : 0;
        const float* ys = st.y + sr[r] * st.ldy;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = lane + 64 * i;
          const float t = ys[k < DO ? k : DO - 1];
          v[r][i] = k < kmax ? t : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) ystg[(RPW * wave + r) * YP + lane + 64 * i] = v[r][i];
    }
  }
  chain_barrier();

  chain_stamp(a, 1);
  float lsum = 0.f;  // this lane's share of the row losses
  int cur = 0;

  // The 32 rows x F features in `act` ([row][feature], 16-bit) -> the weight-gradient operand in MFMA
  // fragment order (one 1-KiB fragment = 32 features x 16 batch rows, 8 consecutive rows of a feature
  // per lane): two transposed LDS reads and ONE fully coalesced 16-byte-per-lane store per fragment.
  // Rows past the end of the batch are written as zeros; features >= F are left alone (feature K of
  // an input operand is the constant row of ones).
  // `tiles`: tile count of the contraction that follows.  The waves WITHOUT a tile in it do the flush while the
  // others already stream their weights (measured: done by everyone at the head of the layer it cost 6.3 k
  // cycles of the 67 k; `act` does not change during the layer, so there is no ordering to keep).
  // Two fragments per turn (r3: one at a time a fragment took ~500 cycles -- two LDS reads, then a store that queues
  // behind the weight loads saturating the CU's load/store path -- and in a 15-tile layer the ONE wave without a tile
  // has 22 of them: it left the layer at 11.4 k cycles when the last tile was done at 9.2 k, per-wave stamps of
  // scripts/diag/chain_wave_stamps.py.  Handing the fragments out through an LDS counter, the tile waves joining in when
  // done, was slower still: every wave pays the counter's round trips in every layer, 58.9 k cycles against 55.5 k.
  // Nor does it help to make the tile waves finish together -- s_setprio by progress, so that a wave that is ahead
  // yields: they then all end at ~10 k where the oldest ended at 5 k and the youngest at 9.4 k; the layer's weights pass
  // the CU's load path at ~38 B/clk either way.)
  auto flush_t = [&](const elem* act, int F, void* dst, int tiles) __attribute__((always_inline)) {
    frag* d = reinterpret_cast<frag*>(dst);
    const int g = lane >> 4, i = lane & 15;
    const int w0 = tiles < NW ? tiles : 0;  // waves [0, w0) have a tile (every wave has one: all of them flush)
    // (A wave that has ENDED leaves the barrier count on gfx950 -- the loader waves of gemm_dw16_lds_kernel below rely
    //  on that when they return before the compute waves' last s_barrier.  Here nothing ends: this `return` leaves the
    //  lambda, and every wave of the workgroup goes on to the layer's barrier.)
    if (wave < w0 || (FEAT & kChainFwd) || st.fwd_only) return;
    // A flushing wave takes whole feature tiles (both 16-row halves of a tile: the two fragments of a turn), ft = its
    // number among the flushing waves, + their count, ...: the LDS and global addresses then advance by constants (the
    // first cut worked every fragment's address out from its number: a 64-bit product and ~40 scalar instructions per
    // fragment, 880 for the lone flushing wave of a 15-tile layer).
    const int nfl = NW - w0, ntile = (F + 31) >> 5;
    const int r0 = 8 * (g >> 1);  // (+ 16 for the second half)
    const elem* p = act + (r0 + (i >> 2)) * PITCH + 16 * (g & 1) + 4 * (i & 3) + 32 * (wave - w0);
    frag* q = d + ((long long)(wave - w0) * a.BS + (m0 >> 4)) * 64 + lane;
    const long long qstep = (long long)nfl * a.BS * 64;
    __builtin_amdgcn_s_setprio(3);  // (the waves without a tile are the YOUNGEST of the workgroup: see above)
    for (int ft = wave - w0; ft < ntile; ft += nfl, p += 32 * nfl, q += qstep) {
      const chain_s4 lo0 = chain_tr_read(p), hi0 = chain_tr_read(p + 4 * PITCH);
      const chain_s4 lo1 = chain_tr_read(p + 16 * PITCH), hi1 = chain_tr_read(p + 20 * PITCH);
      chain_s8 v0 = {lo0[0], lo0[1], lo0[2], lo0[3], hi0[0], hi0[1], hi0[2], hi0[3]};
      chain_s8 v1 = {lo1[0], lo1[1], lo1[2], lo1[3], hi1[0], hi1[1], hi1[2], hi1[3]};
      if (vrows < 32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (r0 + j >= vrows) v0[j] = 0;
          if (16 + r0 + j >= vrows) v1[j] = 0;
        }
      }
      if (32 * ft + (lane & 31) < F) {
        q[0] = __builtin_bit_cast(frag, v0);
        q[64] = __builtin_bit_cast(frag, v1);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // One 32-wide tile: acc(rows = features of the tile, col = batch row li) over `nch` chunks of four
  // k-steps.  On entry chunk 0 of `wsrc` is in `wa` (in flight or landed); on exit chunk 0 of `nxt` --
  // the wave's next tile -- is in flight in `wa` (returns false) or in `wb` (returns true: nch was odd;
  // the caller moves it to `wa` AFTER its epilogue, when the loads have had time to land).
  auto contract = [&](const frag* wsrc, const elem* act, int nch, f32x16& acc, const Job nxt) __attribute__((always_inline)) -> bool {
    const elem* ap = act + li * PITCH + 8 * lh;
    bnext = nxt.bok ? *nxt.b : 0.f;  // (the caller has consumed this tile's values; issued first: it returns first)
    // (r3, measured and dropped: TWO accumulators, even and odd k-steps.  In isolation -- scripts/diag/chain_loop_probe.hip:
    // this loop alone, one workgroup of 16 waves on a warm L2 -- four dependent MFMAs per chunk hold the stream at 47-49
    // B/clk where two accumulators reach the bare stream's 55-56; in the kernel the step went from 44.0 to 45.0 us, with
    // the registers for it taken from the activation operands (a ring of four fragments, three k-steps ahead) 45.5.)
    // (r3, also measured and dropped: the next tile's SECOND chunk requested before the layer's barrier too -- after a
    // barrier every wave starts with an empty pipeline, and the probe with a barrier and an epilogue per tile gains 8-9 %
    // from it (45.7 -> 49.9 B/clk).  In the kernel the second chunk's buffer is then sometimes filled already when a tile
    // starts, the first load of the loop becomes conditional, and behind that branch hipcc drains vmcnt(0) before the
    // tile's first MFMAs: 55.6 us per step.  Unconditional, it needs every tile to have an even number >= 2 of chunks.)
    frag bc[4], bn[4];  // activation fragments of the next chunk are read from LDS under the MFMAs of this one
#pragma unroll
    for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const frag*>(ap + j * 16);
    int c = 0;
    for (; c + 2 <= nch; c += 2) {
      {
        const frag* p = wsrc + (long long)(4 * (c + 1)) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) C16LOAD(wb[j], p[j * 64]);
      }
      // (without the fence hipcc hoists the first MFMA -- and with it the wait for the CURRENT chunk -- above these
      // loads: the next chunk would leave only after this one has landed, and the prefetch distance collapses)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) C16LDS(bn[j], *reinterpret_cast<const frag*>(ap + (4 * (c + 1) + j) * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) C16MFMA(wa[j], bc[j]);
      {
        const frag* p = c + 2 < nch ? wsrc + (long long)(4 * (c + 2)) * 64 : nxt.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) C16LOAD(wa[j], p[j * 64]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) C16LDS(bc[j], *reinterpret_cast<const frag*>(ap + (4 * (c + 2) + j) * 16));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) C16MFMA(wb[j], bn[j]);
    }
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 4; ++j) C16LOAD(wb[j], nxt.w[j * 64]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) C16MFMA(wa[j], bc[j]);
      return true;
    }
    return false;
  };
  auto settle = [&](bool odd) __attribute__((always_inline)) {
    if (odd) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wa[j] = wb[j];
    }
  };

  // ---- forward
  const int LF = GJ(st.nfwd > 0) ? st.nfwd : a.L;
  for (int l = 0; l < LF; ++l) {
    const ChainLayer& ly = a.lt[l];
    const bool last = l == a.L - 1;
    const elem* act = buf[cur];
    elem* out = buf[cur ^ 1];
    const int nch = ly.KS >> 2;
    CFINE(4 * l);
    flush_t(act, ly.K, ly.ht16, ly.NT);  // this layer's input -> operand of its weight gradient
    CFINE(4 * l + 1);
    const float wi = rwl[li];
    for (int t = wave; t < ly.NT; t += NW) {
      const int n0 = 32 * t;
      f32x16 acc;
      {  // accumulator = bias: register r of lane half lh is feature n0 + (r & 3) + 8 (r >> 2) + 4 lh, which lane
         // (r & 3) + 8 (r >> 2) + 4 lh of `bnext` holds (lanes past the layer's width hold 0).  Sixteen lane permutes
         // through the LDS crossbar (one address register, immediate offsets): the readlane / select form was 80
         // VALU instructions per tile, a third of the kernel's.
        const int bbits = __builtin_bit_cast(int, bnext);
        const int pbase = 16 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pbase + 4 * ((r & 3) + 8 * (r >> 2)), bbits));
      }
      const Job nxt = t + NW < ly.NT ? fwd_job(l, t + NW) : fwd_from(l + 1);
      const bool odd = contract(fw + ly.fw_off + ((long long)t * ly.KS) * 64 + lane, act, nch, acc, nxt);
      if (GG(ly.gauss)) {  // keep (mu | lv) in fp32: the sampling pass below turns them into z
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 8 * g + 4 * lh;
          if (n < ZP) *reinterpret_cast<f32x4*>(zs + li * ZP + n) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        }
      } else if (!last) {
        if (GJ(l == a.zcap_layer)) {  // joint step: this layer's outputs are the next model's targets (fp32)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = n0 + 8 * g + 4 * lh;
            if (n < ZP) *reinterpret_cast<f32x4*>(zs + li * ZP + n) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
          }
        }
        if (ly.relu) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
          if (ly.mask_tile >= 0) {
            unsigned bits = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) bits |= (acc[r] > 0.f ? 1u : 0u) << r;
            masks[ly.mask_tile + t][lane] = (unsigned short)bits;
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 8 * g + 4 * lh;
          uint2 pk = {P::pack2(acc[4 * g], acc[4 * g + 1]), P::pack2(acc[4 * g + 2], acc[4 * g + 3])};
          *reinterpret_cast<uint2*>(out + li * PITCH + n) = pk;
        }
      } else if (GO(st.out)) {
        // FORWARD mode: the tile (32 rows x 32 features, this lane: one row, 4 x 4 features) goes through this wave's
        // 4.5 KB of the (unused) target area as [row][feature] and leaves as 16-byte stores -- eight lanes per row,
        // 128 contiguous bytes -- instead of 32 rows x 16 bytes per instruction
        constexpr int SP = 36;  // floats per staged row: 144 B, rows shifted by 4 banks
        float* stg = ystg + wave * (32 * SP);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(stg + li * SP + 8 * g + 4 * lh) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (own writes only: the area is private to the wave)
        typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
        const int c4 = 4 * (lane & 7), n = n0 + c4;
        f32x4 mean4 = {0.f, 0.f, 0.f, 0.f};
        if (st.out_mean) {
#pragma unroll
          for (int e = 0; e < 4; ++e) mean4[e] = st.out_mean[n + e < ly.N ? n + e : ly.N - 1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + (lane >> 3);
          f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + row * SP + c4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = (ly.relu ? fmaxf(v4[e], 0.f) : v4[e]) * st.out_std + mean4[e];  // (a ReLU OUTPUT layer: v21_mlp_create allows one)
          if (row < vrows) {
            float* dst = st.out + (long long)(m0 + row) * st.ldo + n;
            if (n + 3 < ly.N) *reinterpret_cast<f32x4_u*>(dst) = v4;
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < ly.N) dst[e] = v4[e];
            }
          }
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
      settle(odd);
    }
    CFINE(4 * l + 2);
    if (GG(ly.gauss)) {  // z = mu + exp(lv/2) eps -> the next layer's operand image; KL_i -> the row's loss
      chain_barrier();
      // one (batch row, latent dimension) per thread (latent <= 32 = 1,024 / 32): the draw of eps -- a hash, a logarithm,
      // a cosine -- was a serial loop over the latent dimensions in 32 threads (5 us of a 58-us step, forward + backward)
      const int LAT = ly.N >> 1, c1 = a.lt[l + 1].KS * 16;
      const int row = tid & 31;
      int d = tid >> 5;
      asm volatile("" : "+v"(d));  // (keeps the draw below inside this branch: it is loop-invariant, and hoisted out of the
                                   //  layer loop every model paid for it, variational or not: +35 % VALU instructions)
      float klt = 0.f;
      if (d < LAT) {
        const float mu = zs[row * ZP + d], lv = zs[row * ZP + LAT + d];
        const float sd = expf(0.5f * lv);
        const float e = a.sample ? gauss_eps(a.seed, a.step + st.step_off, st.row0 + m0 + row, d) : 0.f;
        out[row * PITCH + d] = (elem)(mu + sd * e);
        klt = -0.5f * (1.0f + lv - mu * mu - sd * sd);
      }
      for (int dd = LAT + d; dd < c1; dd += 32) out[row * PITCH + dd] = (elem)0.f;
      // KL_i: dimensions 2 w and 2 w + 1 meet in wave w, the waves in a fixed order after the layer's barrier
      klt += __shfl_xor(klt, 32, 64);
      if (lh == 0) red[wave][li] = klt;
    }
    // (Columns the tiles did not cover, up to the next contraction's padded range, keep what they held: the weights of
    // those k-steps are zero, and every 16-bit value an image ever holds is finite -- see the one-time clear above.)
    if (last && !GO(st.out)) {  // this lane's share of the row losses joins the other waves' behind the same barrier
      lsum += __shfl_xor(lsum, 32, 64);
      if (lh == 0) red[wave][li] = lsum * rwl[li];
    }
    chain_barrier();
    CFINE(4 * l + 3);
    if (GG(ly.gauss) && tid < 32) {
      float kl = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) kl += red[w][tid];
      klb[tid] = m0 + tid < st.rows ? a.kl_weight * kl : 0.f;
    }
    cur ^= 1;
    chain_stamp(a, 2 + l);
  }

  if (GJ(LF < a.L) || GO(st.out)) return;  // the encoder alone (joint step: its latents are in `zs`), or FORWARD mode
  // ---- loss: lanes -> rows (written before the last layer's barrier, above) -> workgroup (fixed order) -> one
  // fixed-point atomic per workgroup
  if (tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w][tid];
    if (FEAT & kChainGauss) s += klb[tid];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (tid == 0) atomicAdd(a.loss_acc, (unsigned long long)(long long)llrint((double)s * 4294967296.0));
  }

  chain_stamp(a, 2 + a.L);
  if ((FEAT & kChainFwd) || st.fwd_only) return;
  // gs * dL/dz (latent wide, in `b`) -> gs * dL/d[mu | lv] in place:  d mu = dz + beta mu,
  // d lv = dz eps exp(lv/2)/2 + beta (exp lv - 1)/2, beta = kl_weight / B (the KL term's own gradient)
  auto gauss_backward = [&](elem* b, int LAT, int pad) __attribute__((always_inline)) {
    const int row = tid & 31;  // one (batch row, latent dimension) per thread, as in the forward pass
    int d = tid >> 5;
    asm volatile("" : "+v"(d));  // (see the forward pass: keeps the draw out of the models that have no such layer)
    if (d < LAT) {
      const float mu = zs[row * ZP + d], lv = zs[row * ZP + LAT + d];
      const float sd = expf(0.5f * lv);
      const float e = a.sample ? gauss_eps(a.seed, a.step + st.step_off, st.row0 + m0 + row, d) : 0.f;
      const float g = (float)b[row * PITCH + d];
      const float kb = m0 + row < st.rows ? st.gs * a.kl_weight * st.inv_b : 0.f;
      b[row * PITCH + d] = (elem)(g + kb * mu);                                          // d mu (in place)
      b[row * PITCH + LAT + d] = (elem)(g * e * 0.5f * sd + kb * 0.5f * (sd * sd - 1.0f));  // d lv (columns past dz)
    }
    for (int dd = 2 * LAT + d; dd < pad; dd += 32) b[row * PITCH + dd] = (elem)0.f;
  };
  // ---- backward: layer l consumes dZ_l (gs-scaled, in buf[cur]) and produces dZ_{l-1}
  for (int l = a.L - 1; l >= 1; --l) {
    const ChainLayer& ly = a.lt[l];
    const ChainLayer& below = a.lt[l - 1];
    const elem* act = buf[cur];
    elem* out = buf[cur ^ 1];
    const int nch = ly.NS >> 2;
    if (GG(ly.gauss)) { gauss_backward(buf[cur], ly.N >> 1, ly.NS * 16); chain_barrier(); }
    flush_t(act, ly.N, ly.dzt16, ly.KT);  // gs * dZ of this layer's output -> operand of its weight gradient
    for (int t = wave; t < ly.KT; t += NW) {
      const int k0 = 32 * t;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const Job nxt = t + NW < ly.KT ? bwd_job(l, t + NW) : bwd_from(l - 1);
      const bool odd = contract(bw + ly.bw_off + ((long long)t * ly.NS) * 64 + lane, act, nch, acc, nxt);
      if (below.relu) {
        const unsigned bits = masks[below.mask_tile + t][lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // (a sign-extended one-bit field is the AND mask; the element goes through a
          const float v = acc[r];        //  scalar: __builtin_bit_cast applied to `acc[r]` itself read element 0)
          acc[r] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & (((int)(bits << (31 - r))) >> 31));
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = k0 + 8 * g + 4 * lh;
        uint2 pk = {P::pack2(acc[4 * g], acc[4 * g + 1]), P::pack2(acc[4 * g + 2], acc[4 * g + 3])};
        *reinterpret_cast<uint2*>(out + li * PITCH + k) = pk;
      }
      settle(odd);
    }
    chain_barrier();
    cur ^= 1;
    chain_stamp(a, 3 + a.L + (a.L - 1 - l));
  }
  if (GG(a.lt[0].gauss)) { gauss_backward(buf[cur], a.lt[0].N >> 1, a.lt[0].NS * 16); chain_barrier(); }
  flush_t(buf[cur], a.lt[0].N, a.lt[0].dzt16, 0);
#undef GG
#undef GJ
#undef GO
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
  StepCtx sc;  // replayed step: the loss also goes to loss_out2[slot of the step]
};
// r5 (opt-in, V21_DW_XROWS=1): the layer-0 operand of a step on rows of the RESIDENT training set is not a flushed copy but
// the set's 16-bit rows themselves (v21_trainer::d_x16: [row][feature], pitch `ld` halves, a multiple of 32), gathered
// through the step's row table by gemm_dw16_lds_kernel's loader waves.  The problem of the group whose A is nullptr reads
// it.  Measured (autoencoder stack, f16, 16,384 rows, profiles/r5/layer0_operand_gathered_vs_flushed/): 15 MB fewer bytes
// per step and the fused kernel 1.5 us shorter, but this launch 23.1 -> 29.1 us -- with only the loaders gathering 27.9,
// with only the compute waves on transposed 8-byte reads 26.0: the launch moves 96 KB through LDS per 32-KB stage and is
// sensitive to both -- hence opt-in.
struct DwXRows {
  const unsigned short* x16; long long ld;
  const int* idx; long long first;  // batch row i = set row idx[first + i] (idx == nullptr: first + i)
  int rows;                         // batch rows of the step (rows past it repeat the last one: their dZ is zero)
};
struct Dw16Group {
  Dw16Args p[kNtMaxGroup];     // all with the same nz
  int first[kNtMaxGroup + 1];  // output tiles (nx * ny) of the problems before this one
  int count;
  DwXRows xr;
};
template <class P>
__global__ void __launch_bounds__(256) gemm_dw16_kernel(const Dw16Group grp) {
  using frag = typename P::frag;
  // XCD-aware order.  Workgroups b, b + 8, ... land on one XCD (observed round-robin dispatch; a wrong guess
  // costs speed, never results); physical block p works on logical block (p % 8) * ceil(n / 8) + p / 8, and the
  // logical order is SLICE-major over all problems of the group (grp.first counts output tiles): the tiles of
  // batch slice z -- which read the same operand fragments -- sit on one XCD, and with 8 slices on XCD z, where
  // train_chain_kernel wrote those fragments (its row blocks are XCD-major too).
  const int ntile = grp.first[grp.count], nblk = ntile * grp.p[0].nz, xper = (nblk + 7) >> 3;
  const int lb = (int)(blockIdx.x & 7) * xper + (int)(blockIdx.x >> 3);
  if (lb >= nblk) return;  // (the grid is rounded up to a multiple of 8)
  const int bz = lb / ntile, tl = lb - bz * ntile;
  int pi = 0;
  while (pi + 1 < grp.count && tl >= grp.first[pi + 1]) ++pi;
  const Dw16Args& g = grp.p[pi];
  const int bid = tl - grp.first[pi];
  const int bx = bid % g.nx, by = bid / g.nx;
  if (g.loss_acc && bid == 0 && bz == 0 && threadIdx.x == 0) {
    const float f = (float)((double)(long long)*g.loss_acc * (1.0 / 4294967296.0));
    *g.loss_out = f;
    if (g.loss_out2) g.loss_out2[g.sc.desc ? g.sc.desc[*g.sc.cur].slot : 0] = f;
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
// step's 4 A + 4 B fragments enter LDS once (LDS-DMA, already lane-linear) and serve four COMPUTE waves, each of
// which owns a 64x64 quadrant over the whole batch slice: 0.5 KiB per MFMA and no cross-wave reduction.
// Eight more waves do nothing but issue the LDS-DMA: an in-order wave that issues 16 DMA pieces (60-180 cycles
// each) cannot issue MFMAs meanwhile, and with one workgroup per CU nobody else covers for it (r1: 16 us for
// ~2 us of MFMA work; with four loaders a stage still took 910 cycles against 512 of MFMAs).  Loader q moves
// fragment q of every batch step; the loaders run three stages ahead through a ring of four 32-KiB stages (4
// batch steps each) behind a counted vmcnt -- they issue nothing else, so the count is exact -- and one barrier
// per stage hands a stage over.
constexpr int kDwStageSteps = 4;
constexpr int kDwRing = 4;
constexpr int kDwStageBytes = kDwStageSteps * 8 * kFragBytes;
constexpr int kDwLdsBytes = kDwRing * kDwStageBytes;
constexpr int kDwIdxSlots = 8;                                // row tables of a gathering loader wave (DwXRows): 64 ints per stage
constexpr int kDwLdsTotal = kDwLdsBytes + 4 * kDwIdxSlots * 256;  // the operand ring + four loaders' row-table rings
constexpr int kDwThreads = 768;  // 4 compute + 8 loader waves
template <class P>
__global__ void __launch_bounds__(kDwThreads) gemm_dw16_lds_kernel(const Dw16Group grp) {
  using frag = typename P::frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
  // XCD-aware order.  Workgroups b, b + 8, ... land on one XCD (observed round-robin dispatch; a wrong guess
  // costs speed, never results); physical block p works on logical block (p % 8) * ceil(n / 8) + p / 8, and the
  // logical order is SLICE-major over all problems of the group (grp.first counts output tiles): the tiles of
  // batch slice z -- which read the same operand fragments -- sit on one XCD, and with 8 slices on XCD z, where
  // train_chain_kernel wrote those fragments (its row blocks are XCD-major too).
  const int ntile = grp.first[grp.count], nblk = ntile * grp.p[0].nz, xper = (nblk + 7) >> 3;
  const int lb = (int)(blockIdx.x & 7) * xper + (int)(blockIdx.x >> 3);
  if (lb >= nblk) return;  // (the grid is rounded up to a multiple of 8)
  const int bz = lb / ntile, tl = lb - bz * ntile;
  int pi = 0;
  while (pi + 1 < grp.count && tl >= grp.first[pi + 1]) ++pi;
  const Dw16Args& g = grp.p[pi];
  const int bid = tl - grp.first[pi];
  const int bx = bid % g.nx, by = bid / g.nx;
  if (g.loss_acc && bid == 0 && bz == 0 && threadIdx.x == 0) {
    const float f = (float)((double)(long long)*g.loss_acc * (1.0 / 4294967296.0));
    *g.loss_out = f;
    if (g.loss_out2) g.loss_out2[g.sc.desc ? g.sc.desc[*g.sc.cur].slot : 0] = f;
    *g.loss_acc = 0ull;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sbeg = bz * g.steps_per_slice, send = min(g.steps, sbeg + g.steps_per_slice);
  const int nst = (send - sbeg + kDwStageSteps - 1) / kDwStageSteps;
  if (wave >= 4) {
    // ---- loader wave q: fragment q of every step (q < 4: A tile 4 by + q, else B tile 4 bx + q - 4; tiles past
    // the operand are clamped -- their rows / columns are never stored; steps past the slice re-read its last
    // step, so that every stage is exactly kDwStageSteps pieces per loader)
    const int q = wave - 4;
    const int mt = (g.M + 31) / 32, nt = (g.N + 31) / 32;
    const int tile = q < 4 ? min(4 * by + q, mt - 1) : min(4 * bx + q - 4, nt - 1);
    if (q < 4 && g.A == nullptr) {
      // ---- the A tiles from the resident rows (DwXRows).  The workgroup's four feature tiles are 256 contiguous bytes of a
      // row, and a DMA instruction that takes them whole moves as many full cache lines as a flushed fragment does (first
      // form: a piece = one tile of 16 rows = sixteen 64-byte runs -- twice the requests per byte, the launch 23 -> 30 us).
      // Loader q moves rows q, q + 4, q + 8, q + 12 of every batch step: lane l = (row q + 4 (l / 16), 16-byte slot l % 16),
      // and the slot holds feature chunk (l % 16 + 4 q) % 16 of the 256 bytes -- the rotation puts the four rows of a
      // transposed read (one from each loader's piece, 1 KiB apart) on four different bank groups.  A stage is 64
      // consecutive batch rows: lane i fetches the set row of batch row 64 st + i (ONE operation per stage, counted with the
      // pieces).  Columns past the row pitch (the bias feature's tile of a stack whose input width is a multiple of 32) are
      // clamped: the compute waves overwrite that feature, the others are never stored.
      const DwXRows& xr = grp.xr;
      int col = 128 * by + 8 * (((lane & 15) + 4 * q) & 15);
      col = min(col, (int)xr.ld - 8);
      const unsigned short* const colbase = xr.x16 + col;
      const int R0 = 16 * sbeg;
      const int* const ip = xr.idx ? xr.idx + xr.first : nullptr;
      // The row table of a stage travels by LDS-DMA too (64 x 4 bytes into this wave's own eight-slot ring behind the
      // operand ring): a load into a REGISTER that is still on its way must not be copied, and the compiler does copy
      // registers around inline asm and across loop edges (first form of this loop: a v_mov of the destination ahead of
      // the counted wait).  Nothing is in flight in a register here; the wave reads its rows with plain LDS loads once the
      // counted wait has covered the DMA.
      unsigned char* const itab = dw_smem + kDwLdsBytes + q * (kDwIdxSlots * 256);
      auto fetch = [&](int st) __attribute__((always_inline)) {
        if (!ip) return;
        const int rr = min(R0 + 64 * st + lane, xr.rows - 1);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(ip + rr), "s"(lds_addr(itab) + (st % kDwIdxSlots) * 256) : "memory");
      };
      auto issue_x = [&](int st) __attribute__((always_inline)) {
        const unsigned base = lds_addr(dw_smem) + (st % kDwRing) * kDwStageBytes;
        const int* tab = reinterpret_cast<const int*>(itab + (st % kDwIdxSlots) * 256);
#pragma unroll
        for (int s = 0; s < kDwStageSteps; ++s) {
          const int j = 16 * s + q + 4 * (lane >> 4);
          const int srow = ip ? tab[j] : (int)xr.first + min(R0 + 64 * st + j, xr.rows - 1);
          const unsigned short* p = colbase + (long long)srow * xr.ld;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(p), "s"(base + (s * 8 + q) * kFragBytes) : "memory");
        }
      };
      // DMA three stages ahead (slot (st + 3) % 4 is the slot of stage st - 1, which every compute wave has finished reading
      // when barrier st completes); row tables five stages ahead -- with the table only one iteration ahead of its use every
      // iteration of this loop waited for a dependent load (measured: the launch 23.9 -> 28.8 us at 16,384 rows).
      // In flight at the top of iteration st, oldest first:
      //   [pieces st] [table st+3] [pieces st+1] [table st+4] [pieces st+2]          (iteration st issues table st+5, pieces st+3)
      // so "at most pieces st+1, table st+4, pieces st+2 left" = stage st has landed and so has the row table of stage st+3.
      fetch(0);
      if (nst > 1) fetch(1);
      if (nst > 2) fetch(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (nst > 3) fetch(3);
      if (nst > 4) fetch(4);
      issue_x(0);
      if (nst > 1) issue_x(1);
      if (nst > 2) issue_x(2);
      for (int st = 0; st < nst; ++st) {
        const int left = (st + 1 < nst ? 4 : 0) + (st + 2 < nst ? 4 : 0) + ((ip && st > 0 && st + 4 < nst) ? 1 : 0);
        switch (left) {
          case 9: asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory"); break;
          case 8: asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
        }
        if (st + 5 < nst) fetch(st + 5);
        if (st + 3 < nst) issue_x(st + 3);
      }
      return;
    }
    const unsigned char* src = reinterpret_cast<const unsigned char*>(q < 4 ? g.A : g.B) + ((long long)tile * g.BS) * kFragBytes;
    auto issue = [&](int st) __attribute__((always_inline)) {
      const unsigned base = lds_addr(dw_smem) + (st % kDwRing) * kDwStageBytes;
#pragma unroll
      for (int s = 0; s < kDwStageSteps; ++s) {
        const int step = min(sbeg + st * kDwStageSteps + s, send - 1);
        glds16(src + (long long)step * kFragBytes, lane * 16, base + (s * 8 + q) * kFragBytes);
      }
    };
    issue(0);
    if (nst > 1) issue(1);
    if (nst > 2) issue(2);
    for (int st = 0; st < nst; ++st) {
      // stage st has landed (this wave's share) once at most the pieces of stages st+1 and st+2 are outstanding; after the
      // barrier the compute waves are done with stage st-1, i.e. with the ring slot of stage st+3 (r5: three stages ahead,
      // two until r4)
      static_assert(kDwStageSteps == 4 && kDwRing == 4, "the counted waits below are whole stages of four pieces; slot (st + 3) % 4 = slot of st - 1");
      if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      else if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (st + 3 < nst) issue(st + 3);
    }
    // The loader waves END here, before the compute waves' last s_barrier (the epilogue's): on gfx950 an ended wave no
    // longer counts towards its workgroup's barrier, so the compute waves meet among themselves.  (Do not add a barrier
    // after this point that the loaders are meant to attend.)
    return;
  }
  // ---- compute wave (wi, wj): quadrant rows 64 wi .., columns 64 wj ..
  const int li = lane & 31, lh = lane >> 5;
  const int wi = wave >> 1, wj = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // (xg: this problem's A tiles come as row images of the resident set -- DwXRows -- and are read through the hardware
  // transpose: lane = (8-row group kh, feature f) as in a flushed fragment, rows 8 kh + 0..3 and 8 kh + 4..7 in two reads,
  // i.e. the very 16 bytes a flushed fragment holds for that lane; the bias feature M - 1 is a constant one)
  auto contract = [&](auto xg_) __attribute__((always_inline)) {
    constexpr bool XG = decltype(xg_)::value;
    // (the image of a batch step, DwXRows: piece qq = rows qq + 4 r at r * 256 bytes, feature chunk c of the workgroup's 256
    //  bytes in slot (c - 4 qq) % 16; a transposed read takes rows 4 b + 0..3 -- lane 4 qq + p of a 16-lane group supplies row
    //  qq, columns 4 p .. 4 p + 3 of the group's 16 features -- with b = 2 kh for the first read, 2 kh + 1 for the second)
    const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    int tr_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
      tr_off[t] = qq * 1024 + 2 * (gi >> 1) * 256 + (((4 * (2 * wi + t) + 2 * (gi & 1) + (pp >> 1) - 4 * qq) & 15) << 4) + 8 * (pp & 1);
    const int mt = (g.M + 31) / 32;
    const unsigned one2 = P::pack2(1.f, 1.f);
    bool is_one[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) is_one[t] = XG && (4 * by + 2 * wi + t == mt - 1) && ((lane & 31) == ((g.M - 1) & 31));
    auto read_a = [&](const unsigned char* stage, int s, int t) __attribute__((always_inline)) -> frag {
      if constexpr (!XG) return *reinterpret_cast<const frag*>(stage + lane * 16 + (s * 8 + 2 * wi + t) * kFragBytes);
      else {
        const unsigned char* img = stage + s * 8 * kFragBytes + tr_off[t];
        const chain_s4 f0 = chain_tr_read(img), f1 = chain_tr_read(img + 256);
        const chain_s8 v = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
        u32x4 w = __builtin_bit_cast(u32x4, v);
        if (is_one[t]) w = u32x4{one2, one2, one2, one2};
        return __builtin_bit_cast(frag, w);
      }
    };
    for (int st = 0; st < nst; ++st) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // stage st is in LDS; this wave's reads of st-1 are done
      const unsigned char* stage = dw_smem + (st % kDwRing) * kDwStageBytes;
      const unsigned char* buf = stage + lane * 16;
      const int ns = min(kDwStageSteps, send - sbeg - st * kDwStageSteps);
      // the fragments of step s+1 are read under the MFMAs of step s (steps past the slice hold a re-read of its last
      // step: reading them is harmless, they are not multiplied)
      frag fa[2][2], fb[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[0][t] = read_a(stage, 0, t);
        fb[0][t] = *reinterpret_cast<const frag*>(buf + (4 + 2 * wj + t) * kFragBytes);
      }
#pragma unroll
      for (int s = 0; s < kDwStageSteps; ++s) {
        if (s + 1 < kDwStageSteps) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            fa[(s + 1) & 1][t] = read_a(stage, s + 1, t);
            fb[(s + 1) & 1][t] = *reinterpret_cast<const frag*>(buf + ((s + 1) * 8 + 4 + 2 * wj + t) * kFragBytes);
          }
        }
        if (s < ns) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = P::template mfma<false>(fa[s & 1][i], fb[s & 1][j], acc[i][j]);
        }
      }
    }
  };
  if (g.A == nullptr) contract(std::true_type{});
  else contract(std::false_type{});
  // ---- epilogue.  Straight from the accumulators a lane would store one float per instruction (lane = column,
  // registers = rows): 64 store instructions per wave, and a row-per-lane store tail is ISSUE-bound (~170 cycles
  // each: measured 10.9k cycles here, more than the whole contraction).  So the 64x64 quadrant goes through
  // this wave's 16 KiB of the (now idle) ring as [row][column] and leaves as 16 stores of 16 bytes per lane:
  // one instruction = 4 rows x 256 contiguous bytes.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every compute wave is done reading the ring (loaders have left)
  float* tile = reinterpret_cast<float*>(dw_smem) + wave * (64 * 68);  // pitch 68 floats: 4 compute waves x 17 KiB
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * 68 + 32 * j + li] = acc[i][j][r] * g.out_scale;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (own writes only: the tile is private to the wave)
  float* C = g.C + (long long)bz * g.slab_stride;
  const int n0 = 128 * bx + 64 * wj + 4 * (lane & 15);
  // (16-byte stores at 4-byte alignment: the arena offsets and odd widths such as 451 are not multiples of 16 bytes;
  // gfx950 under ROCm runs with unaligned vector access enabled)
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
  const bool vec = n0 + 3 < g.N;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int row = 4 * k + (lane >> 4), m = 128 * by + 64 * wi + row;
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * 68 + 4 * (lane & 15));
    if (m < g.M) {
      float* dst = C + (long long)m * g.ldc + n0;
      if (vec) *reinterpret_cast<f32x4_u*>(dst) = v;
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + e < g.N) dst[e] = v[e];
      }
    }
  }
}

}  // namespace v21
