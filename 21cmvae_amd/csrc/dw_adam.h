// dw_adam.h -- the end of a chain step on ONE rank in one launch: weight gradients, Adam, packed weight copies (gfx950).
//
// train_chain_kernel leaves, per layer, the operands of the weight gradient in MFMA-fragment order (ht16: the
// layer's input plus the constant row of ones; dzt16: gs * dL/dz).  gemm_dw16*_kernel + adam_repack_kernel turn them
// into [dW; db] and the update in two launches with a split-K round trip in between (8 slabs of the whole arena
// written and read back).  Measured (scripts/diag/adam_probe.hip, profiles/r2): a dependent launch costs ~3 us
// before it does anything, the slab round trip ~21 MB of HBM traffic, together 19.4 us of a 52-us step.
//
// Here one 512-thread workgroup owns ONE 32 x 32 tile of [dW; db] over the WHOLE batch: its eight waves split the
// batch steps (every fragment is one 1-KiB wave load straight into MFMA operand registers, eight steps = 16 KiB
// in flight per wave), the eight partial tiles meet in LDS in a fixed order, and the same threads apply Adam to
// the tile's 1,024 arena elements (m, v, w were requested before the contraction started) and rebuild the tile's
// four packed fragments (2 forward + 2 backward) as whole 16-byte-per-lane stores.  No slabs, no second launch.
// Not bit-identical to the two-launch path (another summation order), deterministic all the same.
// Data-parallel steps keep the two-launch path: the gradients of all ranks must be summed before Adam.
#pragma once
#include "train_chain.h"

namespace v21 {

struct DwAdamLayer {
  const void* A; const void* B;  // ht16 (K + 1 features), dzt16 (N features)
  long long BS;                  // batch steps of 16 per feature tile of the operands
  float *w, *m, *v, *g;          // this layer's [W; b] block ((K + 1) x N, row-major) of the arena / moments / gradient
  void *fw, *bw;                 // the model's packed streams; this layer's fragments start at fw_off / bw_off (elements)
  long long fw_off, bw_off;
  int KS, NS, K, N;
  int nt;                        // 32-tiles along N
  int first;                     // first logical block of this layer
  // first layer of a model: fixed-point batch loss -> float slot(s), accumulator cleared (as gemm_dw16_kernel)
  unsigned long long* loss_acc; float* loss_out; float* loss_out2;
};
struct DwAdamModel {
  int L, nblk;
  float omb1, omb2, eps;
  int cprec;  // 1: f16, 2: bf16
  // Which tiles an XCD works on (single-model launches; nullptr: contiguous runs of the row-major tile list).
  // order[x * xper + i] = i-th tile of XCD x or -1: per layer the R x C grid of tiles is cut into 8 two-dimensional
  // blocks, one per XCD, so that an XCD pulls R/rb rows of A-tiles and C/cb columns of B-tiles from HBM instead of
  // R/8 rows and ALL C columns (r3: the launch's HBM-side reads were 57 MB for 20 MB of unique operands; every XCD
  // starts with a cold L2 after the kernel boundary, so what it reads, it reads from HBM or the Infinity Cache)
  const int* order;
  int xper;
#ifdef V21_CHAIN_FINE
  unsigned long long* dbg;  // diagnostic build: phase stamps of a few workgroups (scripts/diag/dwadam_stamps.py)
#endif
  DwAdamLayer lt[16];
};
// what changes from step to step (per model of a group)
struct DwAdamStep {
  int steps;  // batch steps of 16 of this step
  int slot;   // eager epochs: the step's loss also goes to loss_out2[slot] (-1: nowhere)
  float alpha[kSweepMax], out_scale[kSweepMax];
  StepCtx sc;  // replayed step (single model): alpha and the loss slot come from the descriptor
};
constexpr int kDwAdamWaves = 8, kDwAdamInFlight = 8, kDwAdamPitch = 40;
static_assert(sizeof(DwAdamModel) + sizeof(DwAdamStep) <= 4096, "dw16_adam_kernel takes both by value: kernel arguments are limited to 4 KiB");

#ifdef V21_CHAIN_FINE
#define DWFINE(i) do { if ((threadIdx.x & 63) == 0 && md.dbg && (blockIdx.x % 47) == 0 && blockIdx.x / 47 < 8) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); md.dbg[((blockIdx.x / 47) * 8 + (i)) * 8 + (threadIdx.x >> 6)] = t_; } } while (0)
#else
#define DWFINE(i)
#endif
template <class P, int U = kDwAdamInFlight>
__device__ __forceinline__ void dw16_adam_body(const DwAdamModel& md, const int lb, const float alpha, const float out_scale,
                                               const int steps, const int slot, const StepCtx& sc) {
  using frag = typename P::frag;
  constexpr int NW = kDwAdamWaves;
  __shared__ __attribute__((aligned(16))) float part[NW][16][64];
  __shared__ __attribute__((aligned(16))) unsigned short pk[32 * kDwAdamPitch];
  DWFINE(0);
  int pi = 0;
  while (pi + 1 < md.L && lb >= md.lt[pi + 1].first) ++pi;  // scalar
  const DwAdamLayer& g = md.lt[pi];
  const int q = lb - g.first;
  const int ti = q / g.nt, tj = q - ti * g.nt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (g.loss_acc && q == 0 && tid == 0) {
    const float f = (float)((double)(long long)*g.loss_acc * (1.0 / 4294967296.0));
    *g.loss_out = f;
    if (g.loss_out2 && (sc.desc || slot >= 0)) g.loss_out2[sc.desc ? sc.desc[*sc.cur].slot : slot] = f;
    *g.loss_acc = 0ull;
  }
  // this thread's two arena elements (tile rows tid/32 and tid/32 + 16, column tid%32): requested now, used
  // after the contraction.  Elements past the edge of [W; b] are clamped to a valid one and never stored.
  const int mrow0 = tid >> 5, ncol = tid & 31;
  const int n = 32 * tj + ncol;
  long long idx[2];
  bool ok[2];
  float w0[2], m0[2], v0[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int k = 32 * ti + mrow0 + 16 * r;
    ok[r] = k <= g.K && n < g.N;
    idx[r] = (long long)(k <= g.K ? k : g.K) * g.N + (n < g.N ? n : g.N - 1);
    w0[r] = g.w[idx[r]]; m0[r] = g.m[idx[r]]; v0[r] = g.v[idx[r]];
  }
  // ---- contraction over the batch: wave w takes steps w, w + 8, ...
  const frag* A = reinterpret_cast<const frag*>(g.A) + ((long long)ti * g.BS) * 64 + lane;
  const frag* B = reinterpret_cast<const frag*>(g.B) + ((long long)tj * g.BS) * 64 + lane;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  DWFINE(1);
  // Rounds of U steps (2 U fragments, 16 KiB in flight per wave).  (U = 10: 128 VGPRs, the step unchanged -- 44.4-44.5
  // against 44.0-44.5 us at 4,096 rows; U = 12: 132 VGPRs, one workgroup per CU, 48.3 us.)  A rolling refill of each consumed slot was
  // measured SLOWER (19.3 vs 16.9 us at batch 4,096): the kernel is bound by the bytes one CU can pull (~55 GB/s
  // when most lines come from beyond L2; 666 KB per CU at batch 4,096), not by the round trips.
  // (r3: 64 x 32 tiles -- two A-tiles against one B-tile per workgroup, the B fragments loaded once for two MFMAs:
  // 147 MB through the load paths instead of 194 MB, 196 workgroups, one per CU -- were SLOWER as well: 52.0 against
  // 46.0 us per step at 4,096 rows, 62.3 against 53.3 at 6,144 -- with six steps (18 KiB) in flight per wave, and just as
  // slow with ten (30 KiB per wave, 213 VGPRs: 52.4 / 61.7 us): 196 workgroups, one per CU, leave 60 CUs idle and put eight
  // waves on a load path that sixteen keep at 52 B/clk here (phase stamps, scripts/diag/dwadam_stamps.py, 4,096 rows: 20 k
  // of a workgroup's 27.5 k cycles are this loop, 1 MB through each of the 124 CUs that hold two workgroups; ~3 k pass
  // before the first request -- the kernel-argument block itself arrives cold from device memory at every launch, and
  // reading this layer's block once instead of field by field did not shorten that: 33.8 against 33.3 us at 256 rows).)
  int s = wave;
  for (; s + NW * (U - 1) < steps; s += NW * U) {
    frag fa[U], fb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      fa[u] = A[(long long)(s + NW * u) * 64];
      fb[u] = B[(long long)(s + NW * u) * 64];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = P::template mfma<false>(fa[u], fb[u], acc);
  }
  if (s < steps) {  // the last, partial round
    frag fa[U], fb[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s + NW * u < steps) {
        fa[u] = A[(long long)(s + NW * u) * 64];
        fb[u] = B[(long long)(s + NW * u) * 64];
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s + NW * u < steps) acc = P::template mfma<false>(fa[u], fb[u], acc);
  }
  DWFINE(2);
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
  __syncthreads();
  DWFINE(3);
  // ---- the eight partial tiles in a fixed order, Adam, the tile as 16-bit values for the packed copies.
  // element (row m, column c) of a 32 x 32 accumulator tile: lane c + 32 ((m % 8) / 4), register 4 (m / 8) + m % 4
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int mrow = mrow0 + 16 * r;
    const int reg = 4 * (mrow >> 3) + (mrow & 3), pl = ncol + 32 * ((mrow & 7) >> 2);
    float gi = part[0][reg][pl];
#pragma unroll
    for (int w = 1; w < NW; ++w) gi += part[w][reg][pl];
    gi *= out_scale;
    const float mi = m0[r] + (gi - m0[r]) * md.omb1;
    const float vi = v0[r] + (gi * gi - v0[r]) * md.omb2;
    const float wi = w0[r] - (mi * alpha) / (sqrtf(vi) + md.eps);
    if (ok[r]) { g.g[idx[r]] = gi; g.m[idx[r]] = mi; g.v[idx[r]] = vi; g.w[idx[r]] = wi; }
    const float pv = (ok[r] && 32 * ti + mrow < g.K) ? wi : 0.f;  // the bias row has no packed copy
    unsigned short bits;
    if (md.cprec == 1) bits = __builtin_bit_cast(unsigned short, (_Float16)pv);
    else bits = __builtin_bit_cast(unsigned short, (__bf16)pv);
    pk[mrow * kDwAdamPitch + ncol] = bits;
  }
  __syncthreads();
  DWFINE(4);
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  if (wave < 2) {
    // forward fragment (n-tile tj, k-step 2 ti + wave): lane = 32 ((k % 16) / 8) + n % 32, element k % 8
    const int ks = 2 * ti + wave;
    if (16 * ks < g.K) {  // (a k-step without a kernel row -- the bias row's tile -- may lie past the stream)
      u16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = pk[(16 * wave + 8 * (lane >> 5) + e) * kDwAdamPitch + (lane & 31)];
      *reinterpret_cast<u16x8*>(reinterpret_cast<unsigned short*>(g.fw) + g.fw_off +
                                ((((long long)tj * g.KS + ks) * 64 + lane) << 3)) = v;
    }
  } else if (wave < 4) {
    // backward fragment (k-tile ti, n-step 2 tj + wave - 2): lane = 32 ((n % 16) / 8) + k % 32, element n % 8
    const int h = wave - 2, ns = 2 * tj + h;
    if (32 * ti < g.K && 16 * ns < g.N) {
      const u16x8 v = *reinterpret_cast<const u16x8*>(&pk[(lane & 31) * kDwAdamPitch + 16 * h + 8 * (lane >> 5)]);
      *reinterpret_cast<u16x8*>(reinterpret_cast<unsigned short*>(g.bw) + g.bw_off +
                                ((((long long)ti * g.NS + ns) * 64 + lane) << 3)) = v;
    }
  }
  DWFINE(5);
}

// XCD-major logical blocks (workgroups b, b + 8, ... share an XCD: observed round-robin dispatch; speed only): an XCD
// owns a contiguous run of tiles in row-major order, i.e. a few rows of A-tiles against all B-tiles of a layer.
__device__ __forceinline__ int dw_adam_logical_block(int nblk) {
  const int xper = (nblk + 7) >> 3;
  if ((int)(blockIdx.x >> 3) >= xper) return nblk;  // (a group's grid is sized for its largest model: no aliasing into the next XCD's run)
  return (int)(blockIdx.x & 7) * xper + (int)(blockIdx.x >> 3);
}
// one model, everything in the kernel-argument block
template <class P>
__global__ void __launch_bounds__(64 * kDwAdamWaves) dw16_adam_kernel(const DwAdamModel md, const DwAdamStep st) {
  int lb;
  if (md.order) {  // (workgroups b, b + 8, ... share an XCD: observed round-robin dispatch; speed only)
    if ((int)(blockIdx.x >> 3) >= md.xper) return;
    lb = md.order[(int)(blockIdx.x & 7) * md.xper + (int)(blockIdx.x >> 3)];
    if (lb < 0) return;
  } else {
    lb = dw_adam_logical_block(md.nblk);
  }
  if (lb >= md.nblk) return;
  dw16_adam_body<P>(md, lb, st.sc.desc ? st.sc.desc[*st.sc.cur].alpha : st.alpha[0], st.out_scale[0], st.steps, st.slot, st.sc);
}
// a group (sweep, joint step): blockIdx.y = model, the per-model blocks in device memory.
// U: fragment pairs in flight per wave.  A group step of <= 512 rows gives a wave at most four batch steps: with U = 2 the
// kernel needs half the registers (63 against 116), four workgroups fit a CU instead of two, and twice as many of the arena
// reads that bound a large group's step (12 MB per member) are in flight -- r5, f16 sweeps of 8 / 16 /
// 32 / 64 members: 110 -> 119, 160 -> 185, 198 -> 207-214, 183 -> 197-199 k model-steps/s; same sums in the same order.
// (A single model's launch -- 315 workgroups -- does not change: 31.5 against 31.7 us per 256-row step; it keeps U = 8.)
template <class P, int U>
__global__ void __launch_bounds__(64 * kDwAdamWaves, U == 2 ? 8 : 1) dw16_adam_group_kernel(const DwAdamModel* __restrict__ tab, const DwAdamStep st) {
  const DwAdamModel& md = tab[blockIdx.y];
  const int lb = dw_adam_logical_block(md.nblk);
  if (lb >= md.nblk) return;
  dw16_adam_body<P, U>(md, lb, st.alpha[blockIdx.y], st.out_scale[blockIdx.y], st.steps, st.slot, st.sc);
}

}  // namespace v21
