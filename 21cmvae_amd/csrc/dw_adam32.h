// dw_adam32.h -- every weight gradient of an f32 chain step + Adam + the packed fp32 streams + the batch loss, for steps of
// up to 2,048 rows: the operands pass through LDS in whole rows, 256 batch rows at a time (gfx950).
//
// gemm_nt_dwadam_kernel (gemm_nt.h) loads its MFMA operands straight into registers: lane (li, lh) of a wave takes 16
// bytes of feature row li, so ONE wave instruction touches 32 rows x 32 bytes -- 32 cache lines for 1 KiB, and the CU's
// load path works through lines, not bytes (phase stamps of scripts/diag/dwadam_stamps.py: 8.4 k cycles of a 19-k-cycle
// workgroup pass before its 32 + 32 loads per wave have even been issued; CDNA guide, "fragment-shaped loads").  Here a
// wave instruction is one feature row of 256 batch floats = 1 KiB contiguous, written to LDS by the load itself
// (`global_load_lds_dwordx4`: no registers, no ds_write); 64 rows (32 of H^T, 32 of dZ^T) = 16 instructions per wave
// instead of 64.  The LDS image is lane-linear per row (the instruction demands it), so the bank spread comes from the
// SOURCE side: lane c of row r fetches chunk c ^ (r & 15), and the MFMA side reads chunk q of row r at position
// q ^ (r & 15) -- sixteen rows, sixteen different 16-byte bank groups.  The four waves split the batch, the partial tiles
// meet in the same 64 KB (the operands are dead by then), and the epilogue is gemm_nt.h's: Adam on the tile's arena
// elements (their m, v, w requested before the operands), the 8-row format's packed words as 16-byte stores.
// One workgroup per 32 x 32 tile of [dW; db]; K = rows of the step; steps of more than 256 rows pass through the same 64 KB
// in slabs of 256 rows (end of r3: 512 rows 51.3 -> 48.4 us per step, 1,024 rows 69.2 -> 60.9, 2,048 rows 90.0 -> 81.8).
// (Measured and dropped: the same whole rows through REGISTERS -- sixteen coalesced 16-byte loads per lane, then sixteen
// ds_write_b128 at the swizzled positions; bit-identical, 43.3-43.4 against 42.9-43.0 us per step at batch 256.  The
// ~280 cycles per row request are not the LDS-DMA mechanism: the rows were written by the chain kernel a launch ago and
// come from beyond this XCD's L2 either way.)
#pragma once
#include "gemm_nt.h"

namespace v21 {

constexpr int kDw32MaxRows = 2048;  // (slabs of 256 rows; = kC32sMaxBatch, the largest step of the trainers whose stream format it writes)

#ifdef V21_CHAIN_FINE  // (diagnostic build: phase stamps of a few workgroups, scripts/diag/dwadam_stamps.py)
#define D32FINE(i) do { if ((threadIdx.x & 63) == 0 && ad.dbg && (blk % 47) == 0 && blk / 47 < 8) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); ad.dbg[((blk / 47) * 8 + (i)) * 4 + (threadIdx.x >> 6)] = t_; } } while (0)
#else
#define D32FINE(i)
#endif
// what changes from step to step in a sweep (the per-model blocks stay in device memory)
struct Dw32Model { NtGroupBig grp; NtAdamInfo ad; };
struct Dw32Step { int rows; int slot; float alpha[kSweepMax]; };
// `kov` >= 0: a sweep's launch -- the contraction length, the step size and the loss slot come from the step block
// SLAB: batch rows that pass through LDS at a time.  256 (64 KB: two workgroups per CU) for a single model, whose ~380
// workgroups are resident at once and want the shortest chain of dependent waits; 128 (32 KB: four per CU) for the grouped
// launch of a LARGE sweep, which is thousands of workgroups deep and bound by how many operand rows are in flight (r5, end).
template <int SLAB>
__device__ __forceinline__ void dwadam32_body(const NtGroupBig& grp, const NtAdamInfo& ad, const int blk, const int kov, const float alpha_ov, const int slot_ov) {
  static_assert(SLAB == 256 || SLAB == 128, "a wave instruction is one row of 256 floats or two of 128");
  __shared__ __attribute__((aligned(16))) float smem[64 * SLAB];  // operand rows, then the four partial tiles
  D32FINE(0);
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  if (ad.loss_acc && blk == 0 && threadIdx.x == 0) {
    const float f = (float)((double)(long long)*ad.loss_acc * (1.0 / 4294967296.0));
    *ad.loss_out = f;
    const int slot = kov >= 0 ? slot_ov : ad.loss_slot;
    if (ad.loss_out2 && (ad.sc.desc || slot >= 0)) ad.loss_out2[ad.sc.desc ? ad.sc.desc[*ad.sc.cur].slot : slot] = f;
    *ad.loss_acc = 0ull;
  }
  int pi = 0;  // (no dependent chain of scalar loads: the whole table, then compares)
#pragma unroll
  for (int i = 1; i < kNtMaxGroup; ++i) pi += (i < grp.count && blk >= grp.first[i]) ? 1 : 0;
  const int bid = blk - grp.first[pi];
  // everything the kernel needs of its problem and of the Adam block, read once (both are indexed by `pi` in the
  // kernel-argument segment: every later use would be a scalar load of its own)
  const NtArgs& g = grp.p[pi];
  const NtAdamLayer& al = ad.lt[pi];
  const float* gA = g.A; const float* gB = g.B;
  const long long lda = g.lda, ldb = g.ldb, ldc = g.ldc;
  const int M = g.M, N = g.N, K = kov >= 0 ? kov : g.K, nx = g.nx;
  float* gC = g.C;
  const long long arena_off = al.arena_off, fw_off = al.fw_off, bw_off = al.bw_off;
  const int aK = al.K, KS = al.KS, NS = al.NS, fmt = ad.fmt;
  const float alpha = kov >= 0 ? alpha_ov : (ad.sc.desc ? ad.sc.desc[*ad.sc.cur].alpha : ad.alpha), omb1 = ad.omb1, omb2 = ad.omb2, eps = ad.eps;
  float* aw_ = ad.w; float* am_ = ad.m; float* av_ = ad.v; float* afw = ad.fw; float* abw = ad.bw;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int bx = bid % nx, by = bid / nx;
  const int m0 = 32 * by, n0 = 32 * bx;

  D32FINE(6);
  // this thread's four arena elements (rows m0 + 8 wave + 4 lh + e, column n0 + li): requested before the operands
  float pm[4], pv[4], pw[4];
  {
    const int n = min(n0 + li, N - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = min(m0 + 8 * wave + 4 * lh + e, M - 1);
      const long long i = arena_off + (long long)m * ldc + n;
      pm[e] = am_[i]; pv[e] = av_[i]; pw[e] = aw_[i];
    }
  }
  D32FINE(7);
  // ---- stage: wave w brings rows w, w + 4, ... (0 .. 31: H^T rows m0 + r; 32 .. 63: dZ^T rows n0 + r - 32).
  // r & 15 = wave + 4 (i & 3): four swizzled lane offsets per operand, worked out once; the row pointer itself is scalar.
  // (Phase stamps: the kernel-argument batch is read at ~2.1 k cycles, the Adam state requested at ~2.7 k, and the
  // sixteen row loads of a wave take until ~7.1 k to ISSUE -- ~280 cycles each, whether a row costs 30 instructions or 15,
  // and whether or not the table of first blocks is preloaded into SGPRs: the LDS-DMA requests themselves are paced.)
  // Steps of more than 256 rows: the batch passes through the same 64 KB in SLABS of 256 rows (stage, contract, next slab;
  // the accumulator stays in registers) -- one launch for any step a small-batch f32 trainer takes (<= 2,048 rows).
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float* arow = smem + li * SLAB;
  const float* brow = smem + (32 + li) * SLAB;
  const int swz = li & 15;
  for (int koff = 0; koff < K; koff += SLAB) {
  if (koff > 0) __syncthreads();  // every wave is done with the previous slab
  if constexpr (SLAB == 256) {
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane ^ (wave + 4 * j);
    voffA[j] = koff + 4 * c + 4 > lda ? 0u : 4u * koff + 16u * c;  // (past the row: any valid address; those k are >= K and zeroed below)
    voffB[j] = koff + 4 * c + 4 > ldb ? 0u : 4u * koff + 16u * c;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = wave + 4 * i;
    const char* row = reinterpret_cast<const char*>(gA + (long long)min(m0 + r, M - 1) * lda);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)(row + voffA[i & 3]),
                                     (__attribute__((address_space(3))) void*)(smem + r * 256), 16, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = wave + 4 * i;
    const char* row = reinterpret_cast<const char*>(gB + (long long)min(n0 + r, N - 1) * ldb);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)(row + voffB[i & 3]),
                                     (__attribute__((address_space(3))) void*)(smem + (32 + r) * 256), 16, 0, 0);
  }
  } else {
  // two rows of 128 floats per wave instruction: lanes 0-31 row 2 ri, lanes 32-63 row 2 ri + 1 (adjacent in LDS); position p of
  // row r holds chunk p ^ (r & 15), as above
  const int rsub = lane >> 5, pch = lane & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 2 * (wave + 4 * i) + rsub;
    const int c = pch ^ (r & 15);
    const unsigned vo = koff + 4 * c + 4 > lda ? 0u : 4u * koff + 16u * c;
    const char* row = reinterpret_cast<const char*>(gA + (long long)min(m0 + r, M - 1) * lda);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)(row + vo),
                                     (__attribute__((address_space(3))) void*)(smem + 2 * (wave + 4 * i) * 128), 16, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 2 * (wave + 4 * i) + rsub;
    const int c = pch ^ (r & 15);
    const unsigned vo = koff + 4 * c + 4 > ldb ? 0u : 4u * koff + 16u * c;
    const char* row = reinterpret_cast<const char*>(gB + (long long)min(n0 + r, N - 1) * ldb);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)(row + vo),
                                     (__attribute__((address_space(3))) void*)(smem + (32 + 2 * (wave + 4 * i)) * 128), 16, 0, 0);
  }
  }
  D32FINE(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- contraction: wave w takes its quarter of the slab's k-steps of 8
  const int nsteps = (min(K - koff, SLAB) + 7) >> 3, per = (nsteps + 3) >> 2;
  const int s0 = wave * per, s1 = min(nsteps, s0 + per);
  for (int s = s0; s < s1; ++s) {
    const int q = ((2 * s + lh) ^ swz) << 2;
    f32x4v a4 = *reinterpret_cast<const f32x4v*>(arow + q), b4 = *reinterpret_cast<const f32x4v*>(brow + q);
    const int kk = koff + 8 * s + 4 * lh;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (kk + e >= K) { a4[e] = 0.f; b4[e] = 0.f; }  // (select, not multiply: the padding may hold anything)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
    }
  }
  }
  D32FINE(2);
  __syncthreads();  // every wave is done with the operand rows: the partial tiles take their place
  float(*part)[16][64] = reinterpret_cast<float(*)[16][64]>(smem);
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][i][lane] = acc[i];
  __syncthreads();
  D32FINE(3);
  // ---- gradient element -> Keras Adam (train_kernels.h: adam_update_element) -> arena, moments, packed fp32 streams.
  // wave w finishes accumulator registers 4 w .. 4 w + 3 = rows 8 w + 4 lh + {0..3} of the tile (as gemm_nt_body)
  const int n = n0 + li, mrow = m0 + 8 * wave + 4 * lh;
  const bool nvalid = n < N;
  float wnew[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int reg = 4 * wave + e;
    const float v = (part[0][reg][lane] + part[1][reg][lane]) + (part[2][reg][lane] + part[3][reg][lane]);
    const int m = mrow + e;
    if (nvalid && m < M) {
      const long long i = arena_off + (long long)m * ldc + n;
      const float mi = pm[e] + (v - pm[e]) * omb1;
      const float vi = pv[e] + (v * v - pv[e]) * omb2;
      const float wi = pw[e] - (mi * alpha) / (sqrtf(vi) + eps);
      gC[(long long)m * ldc + n] = v;
      am_[i] = mi; av_[i] = vi; aw_[i] = wi;
      if (m < aK) wnew[e] = wi;  // (the bias row has no packed copy)
    }
  }
  if (fmt == 4 && mrow + 3 < aK) {  // (the same for the four lanes of a quad: they share wave and lh)
    // the four rows of this thread are ONE 16-byte word of the forward stream; after a 4 x 4 transpose inside the quad of
    // lanes that holds columns n & ~3 .. + 3 (n & 3 == li & 3: tiles start at multiples of 32), one word of the backward
    // stream per lane
    if (nvalid)
      *reinterpret_cast<f32x4v*>(afw + fw_off + ((((long long)(n >> 6) * KS + (mrow >> 2)) * 64 + (n & 63)) << 2)) =
          f32x4v{wnew[0], wnew[1], wnew[2], wnew[3]};
    const int j = li & 3;
    auto xchg1 = [&](float give) __attribute__((always_inline)) -> float {  // partner j ^ 1: quad_perm [1,0,3,2]
      return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
    };
    auto xchg2 = [&](float give) __attribute__((always_inline)) -> float {  // partner j ^ 2: quad_perm [2,3,0,1]
      return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0x4E, 0xf, 0xf, true));
    };
    {
      const float g0 = xchg1((j & 1) ? wnew[0] : wnew[1]), g1 = xchg1((j & 1) ? wnew[2] : wnew[3]);
      if (j & 1) { wnew[0] = g0; wnew[2] = g1; } else { wnew[1] = g0; wnew[3] = g1; }
    }
    {
      const float g0 = xchg2((j & 2) ? wnew[0] : wnew[2]), g1 = xchg2((j & 2) ? wnew[1] : wnew[3]);
      if (j & 2) { wnew[0] = g0; wnew[1] = g1; } else { wnew[2] = g0; wnew[3] = g1; }
    }
    const int mj = mrow + j, nb = n & ~3;  // now wnew[i] = W[mrow + j][nb + i]
    if (nb < N)
      *reinterpret_cast<f32x4v*>(abw + bw_off + ((((long long)(mj >> 6) * NS + (nb >> 2)) * 64 + (mj & 63)) << 2)) =
          f32x4v{wnew[0], wnew[1], wnew[2], wnew[3]};
  } else if (nvalid) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = mrow + e;
      if (m >= aK) continue;
      long long qf, qb;  // (train_kernels.h: adam_repack_element spells the two formats out)
      if (fmt == 4) {
        qf = fw_off + ((((long long)(n >> 6) * KS + (m >> 2)) * 64 + (n & 63)) << 2) + (m & 3);
        qb = bw_off + ((((long long)(m >> 6) * NS + (n >> 2)) * 64 + (m & 63)) << 2) + (n & 3);
      } else {
        qf = fw_off + ((((long long)(n >> 5) * KS + 2 * (m >> 4) + ((n >> 4) & 1)) * 64 + (n & 15) + 16 * ((m >> 2) & 3)) << 2) + (m & 3);
        qb = bw_off + ((((long long)(m >> 5) * NS + 2 * (n >> 4) + ((m >> 4) & 1)) * 64 + (m & 15) + 16 * ((n >> 2) & 3)) << 2) + (n & 3);
      }
      afw[qf] = wnew[e];
      abw[qb] = wnew[e];
    }
  }
  D32FINE(4);
#ifdef V21_CHAIN_FINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  D32FINE(5);
}
static __global__ void __launch_bounds__(256) dwadam32_kernel(const NtGroupBig grp, const NtAdamInfo ad) { dwadam32_body<256>(grp, ad, (int)blockIdx.x, -1, 0.f, -1); }
// a sweep of f32 models: blockIdx.y = model (its problem and Adam blocks in device memory), blockIdx.x = tile of the model.
// (One model per XCD -- workgroup b = tile b / models of model b % models, as train_chain32s_group_kernel deals its row
// blocks -- was slower: 59.8 against 53.3 us for 8 autoencoders of 128 .. 512 hidden units; the XCD with the largest
// model's tiles ends last.)
template <int SLAB>
static __global__ void __launch_bounds__(256) dwadam32_group_kernel(const Dw32Model* __restrict__ tab, const Dw32Step st) {
  const Dw32Model& md = tab[blockIdx.y];
  if ((int)blockIdx.x >= md.grp.first[md.grp.count]) return;
  dwadam32_body<SLAB>(md.grp, md.ad, (int)blockIdx.x, st.rows, st.alpha[blockIdx.y], st.slot);
}

}  // namespace v21
