// train_chain32.h -- the chain kernel of train_chain.h in the REFERENCE'S OWN ARITHMETIC: fp32 operands, fp32
// accumulation (v_mfma_f32_16x16x4_f32), for the default precision of the class surface (gfx950).
//
// The f32 training step used to be 14 dependent launches of ~7 us each (gemm_nt.h: every one a cold round trip, a short
// MFMA chain and an LDS meeting; hipGraph replay does not shorten them).  A row-block chain removes the launches, but
// in f32 it is bound by the MFMA rate of the CUs it occupies (256 FLOP/clk per CU): at the reference's batch of 256
// rows, 32-row blocks would use 8 CUs (>= 69 us for the autoencoder stack at the 76 % the compiled f32 forward kernel
// reaches).  Hence 16 ROWS per workgroup -- 16 workgroups at batch 256 -- and the 16 x 16 x 4 MFMA:
//
//   lane l of a wave owns batch row m = l % 16 and, of every 16-feature MFMA tile, the features 4 (l / 16) .. + 3
//   (D[i][j]: i = 4 (l / 16) + r, j = l % 16; A[i][k]: i = l % 16, k = l / 16; B likewise -- checked on the part by
//   scripts/diag/mfma16_probe.hip).  As in the 16-bit kernel the weights are the A operand (rows = output features),
//   the activations the B operand (columns = batch rows): the result goes back to LDS as ONE 16-byte write per MFMA
//   tile into the fp32 [row][feature] image the next layer reads with one ds_read_b128 per sixteen k.
//   A wave's tile is 32 features (two MFMA tiles that share the activation read); its weights come as 1-KiB fragments
//   (64 lanes x 4 floats: lane l, element e = W[16 ks + 4 (l / 16) + e][n0 + l % 16]) pre-packed by the Adam kernel in
//   the order the wave reads them, four fragments (two k16 steps x two MFMA tiles, 16 MFMAs) per chunk of the same
//   rolling prefetch as train_chain_body's.
//   The operands of the weight gradients leave as fp32 in the layout the NT GEMM already reads (feature-major,
//   batch contiguous: H^T and dZ^T, gemm_nt.h): ALL weight gradients of the step are then one grouped launch, Adam
//   another -- 3 launches instead of 14.
// Same ChainModel / ChainStep blocks as the 16-bit kernels (ChainLayer::KS / NS = fragments per tile here; ht16 / dzt16
// = the fp32 transposed buffers, ChainModel::BS = their row pitch); fwd_only = the validation pass, `out` = FORWARD mode
// (Model.predict of any stack up to 512 wide in f32).  No variational layer (that stack keeps the per-layer path).
#pragma once
#include "train_chain.h"

namespace v21 {

constexpr int kC32Rows = 16;
constexpr int kC32Pitch = kChainMaxDim + 4;  // floats; 2064 B = 16 B mod 128 B: conflict-free ds_read_b128 across 16 rows
constexpr int kC32Waves = 16;
constexpr int kC32MaskTiles = 112;
constexpr int kC32BufBytes = 2 * kC32Rows * kC32Pitch * 4;
constexpr int kC32YBytes = kC32Rows * kC32Pitch * 4;
constexpr int kC32MaskBytes = kC32MaskTiles * 64 * 2;
constexpr int kC32LdsBytes = kC32BufBytes + kC32YBytes + kC32MaskBytes + kC32Waves * kC32Rows * 4 + kC32Rows * 4 + 64;
// fragments per 32-feature tile over a contraction range of d: two per k16 step, whole chunks of two steps
__host__ __device__ constexpr int chain32_frags(int d) { return 2 * (((d + 15) / 16 + 1) / 2 * 2); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// FEAT (train_chain.h): kChainOut = this instantiation carries FORWARD mode; kChainFwd = it is forward-only (no backward
// pass compiled).  Training launches use 0, validation kChainFwd, Model.predict kChainOut | kChainFwd.
template <int FEAT>
__device__ __forceinline__ void train_chain32_body(const ChainModel& a, const ChainStep& st) {
#define GO(cond) ((FEAT & kChainOut) ? (bool)(cond) : false)
#define FWD_ONLY ((FEAT & kChainFwd) ? true : (bool)st.fwd_only)
  constexpr int NW = kC32Waves, ROWS = kC32Rows, PITCH = kC32Pitch;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  float(*buf)[ROWS * PITCH] = reinterpret_cast<float(*)[ROWS * PITCH]>(chain_smem);
  float* ystg = reinterpret_cast<float*>(chain_smem + kC32BufBytes);
  unsigned short(*masks)[64] = reinterpret_cast<unsigned short(*)[64]>(chain_smem + kC32BufBytes + kC32YBytes);
  float(*red)[ROWS] = reinterpret_cast<float(*)[ROWS]>(chain_smem + kC32BufBytes + kC32YBytes + kC32MaskBytes);
  float* rwl = reinterpret_cast<float*>(chain_smem + kC32BufBytes + kC32YBytes + kC32MaskBytes + NW * ROWS * 4);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;  // batch row of the block, feature / k quarter
  const long long first = st.sc.desc ? st.sc.desc[*st.sc.cur].first : st.first;
  const int nrb = (st.rows + ROWS - 1) / ROWS;
  const int bidx = (int)blockIdx.x - st.blk0;
  const int rb = (bidx & 7) * ((nrb + 7) >> 3) + (bidx >> 3);  // XCD-major row blocks (speed only)
  if (rb >= nrb) return;
  const int m0 = rb * ROWS;
  const int vrows = st.rows - m0;  // valid rows of this block (>= 1; < 16 only in the last block)

  const f32x4* fw = reinterpret_cast<const f32x4*>(a.fw);
  const f32x4* bw = reinterpret_cast<const f32x4*>(a.bw);

  // ---- the wave's weight stream (rolling prefetch across tiles, layers and the forward / backward turn) and, for a
  // forward tile, its 8 bias values (features 32 t + 16 s + 4 kq + r)
  struct Job { const f32x4* w; const float* b; int nb; };  // nb: features of the layer left from the tile's first one
  // A layer with ONE tile (an output <= 32 wide: the 9-wide latent; backward, an input <= 32 wide) would keep one wave
  // busy for its whole contraction while fifteen wait (r3 stamps: 8 k cycles each for 352 -> 9 forward and 352 <- 32
  // backward).  There the CONTRACTION is split instead: wave w takes chunk w of the tile's stream (two k16 steps), the
  // partial tiles meet in LDS and wave 0 finishes the tile.  A wave's unit in a layer is therefore a tile (several-tile
  // layers) or a chunk of tile 0 (single-tile layers): at most one per wave either way (<= 16 tiles, <= 16 chunks).
  auto fwd_units = [&](int l) __attribute__((always_inline)) -> int { return a.lt[l].NT == 1 ? a.lt[l].KS >> 2 : a.lt[l].NT; };
  auto bwd_units = [&](int l) __attribute__((always_inline)) -> int { return a.lt[l].KT == 1 ? a.lt[l].NS >> 2 : a.lt[l].KT; };
  auto fwd_job = [&](int l, int u) __attribute__((always_inline)) -> Job {
    const ChainLayer& ly = a.lt[l];
    if (ly.NT == 1) return Job{fw + ly.fw_off + (long long)(4 * u) * 64 + lane, a.w + ly.b_off, u == 0 ? ly.N : 0};
    return Job{fw + ly.fw_off + ((long long)u * ly.KS) * 64 + lane, a.w + ly.b_off + 32 * u, ly.N - 32 * u};
  };
  auto bwd_job = [&](int l, int u) __attribute__((always_inline)) -> Job {
    const ChainLayer& ly = a.lt[l];
    if (ly.KT == 1) return Job{bw + ly.bw_off + (long long)(4 * u) * 64 + lane, a.w, 0};
    return Job{bw + ly.bw_off + ((long long)u * ly.NS) * 64 + lane, a.w, 0};
  };
  auto bwd_from = [&](int l) __attribute__((always_inline)) -> Job {
    if (!FWD_ONLY)
      for (; l >= 1; --l)
        if (wave < bwd_units(l)) return bwd_job(l, wave);
    return Job{fw + lane, a.w, 0};  // nothing left: any valid address (the data is never used)
  };
  auto fwd_from = [&](int l) __attribute__((always_inline)) -> Job {
    for (; l < a.L; ++l)
      if (wave < fwd_units(l)) return fwd_job(l, wave);
    return bwd_from(a.L - 1);
  };
  auto load_bias = [&](const Job& j, float (&bv)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * s + 4 * kq + r;
        const float v = j.b[o < j.nb ? o : 0];  // (clamped: always a valid address; nb = 0 for backward jobs)
        bv[4 * s + r] = o < j.nb ? v : 0.f;
      }
  };
  // this wave gathers row `wave` of the block: its source row first (the oldest load in flight)
  const int mq = m0 + wave;
  long long srow = mq < st.rows ? first + mq : first + m0;  // clamped: always a valid position
  if (st.idx) srow = st.idx[srow];
  f32x4 wa[4], wb[4];
  float bnext[8];
  {
    const Job j0 = fwd_from(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[j] = j0.w[j * 64];
    load_bias(j0, bnext);
  }
  chain_stamp(a, 0);
  // buf[1] <- 0 once: padding columns must hold finite values (they meet zero weights); buf[0] is written in full below
  for (int i = tid; i < ROWS * PITCH / 4; i += 64 * NW) reinterpret_cast<f32x4*>(buf[1])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- gather: x[idx] -> buf[0] (and the target rows)
  {
    const int K0 = a.lt[0].K, DO = a.lt[a.L - 1].N;
    const float rwv = st.rw ? st.rw[srow] : 0.f;
    if (lane == 0) rwl[wave] = mq < st.rows ? rwv : 0.f;
    const int kmax = mq < st.rows ? K0 : 0;
    const float* xs = st.x + srow * st.ldx;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + 64 * i;
      float t = xs[k < K0 ? k : K0 - 1];
      if (i == 0 && GO(st.tin)) {  // par_transform on the (<= 8) input columns, as affine_in_kernel does it
        const int jc = lane < K0 ? lane : 0;
        t = par_transform_f32(t, st.tin->log_mask[jc], st.tin->zero_floor[jc], st.tin->lo[jc], st.tin->span[jc]);
      }
      v[i] = k < kmax ? t : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + 64 * i;  // <= 511 < PITCH
      buf[0][wave * PITCH + k] = v[i];
      if (!st.y && !GO(st.out)) ystg[wave * PITCH + k] = v[i];
    }
    if (st.y && !GO(st.out)) {
      const int ymax = mq < st.rows ? DO : 0;
      const float* ys = st.y + srow * st.ldy;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = lane + 64 * i;
        const float t = ys[k < DO ? k : DO - 1];
        v[i] = k < ymax ? t : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) ystg[wave * PITCH + lane + 64 * i] = v[i];
    }
  }
  chain_barrier();
  chain_stamp(a, 1);
  float lsum = 0.f;
  int cur = 0;

  // The 16 rows x F features in `act` -> the fp32 operand of the weight gradient, feature-major with the batch
  // contiguous (dst[f * BS + batch row]: what gemm_nt.h reads).  A lane takes one feature and four consecutive rows
  // (four LDS reads, conflict-free: 16 features x 4 row groups per instruction) and stores them as ONE 16-byte word: an
  // instruction = 16 features x 64 contiguous bytes.  (First cut: one float per lane, 4 features x 16 rows per
  // instruction -- four times the store instructions; the flush of a 352-wide activation by the few idle waves WAS the
  // duration of the single-tile layers that follow the big ones.)  Done by the waves without a unit in the contraction
  // that follows (tiles = 0: by everyone); rows past the batch are written as zeros.
  auto flush_t = [&](const float* act, int F, void* dst, int tiles) __attribute__((always_inline)) {
    const int ngrp = (F + 15) >> 4;
    float* d = reinterpret_cast<float*>(dst);
    const int w0 = tiles < NW ? tiles : 0;
    if (wave < w0 || FWD_ONLY) return;
    const int fq = lane >> 2, r4 = 4 * (lane & 3);
    for (int id = wave - w0; id < ngrp; id += NW - w0) {
      const int f = 16 * id + fq;
      const int fc = f < F ? f : F - 1;
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = r4 + j < vrows ? act[(r4 + j) * PITCH + fc] : 0.f;
      if (f < F) *reinterpret_cast<f32x4*>(d + (long long)f * a.BS + m0 + r4) = v;
    }
  };

  // One 32-wide tile: acc[s](rows = features 16 s + 4 kq + r of the tile, col = batch row m) over `nch` chunks of two
  // k16 steps.  Weight chunks roll as in train_chain_body (wa in use / wb in flight, roles alternate); the activation
  // words of the next k16 step are read from LDS under the 8 MFMAs of this one.
  auto contract = [&](const f32x4* wsrc, const float* act, int nch, f32x4 (&acc)[2], const Job nxt, int k16 = 0) __attribute__((always_inline)) -> bool {
    const float* ap = act + m * PITCH + 4 * kq + 16 * k16;  // (k16: first k16 step of this unit within the contraction)
    load_bias(nxt, bnext);  // (the caller has consumed this tile's values)
    f32x4 bc = *reinterpret_cast<const f32x4*>(ap), bn;
    auto chunk = [&](f32x4 (&w)[4], int kc, bool more) __attribute__((always_inline)) {
      bn = *reinterpret_cast<const f32x4*>(ap + 16 * (2 * kc + 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = mfma16(w[0][e], bc[e], acc[0]);
        acc[1] = mfma16(w[1][e], bc[e], acc[1]);
      }
      if (more) bc = *reinterpret_cast<const f32x4*>(ap + 16 * (2 * kc + 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = mfma16(w[2][e], bn[e], acc[0]);
        acc[1] = mfma16(w[3][e], bn[e], acc[1]);
      }
    };
    int c = 0;
    for (; c + 2 <= nch; c += 2) {
      {
        const f32x4* p = wsrc + (long long)(4 * (c + 1)) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = p[j * 64];
      }
      __builtin_amdgcn_sched_barrier(0);  // (keeps the first MFMA -- and the wait for the CURRENT chunk -- below these loads)
      chunk(wa, c, true);
      {
        const f32x4* p = c + 2 < nch ? wsrc + (long long)(4 * (c + 2)) * 64 : nxt.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) wa[j] = p[j * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      chunk(wb, c + 1, c + 2 < nch);
    }
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wb[j] = nxt.w[j * 64];
      __builtin_amdgcn_sched_barrier(0);
      chunk(wa, c, false);
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

  // partial tiles of a split contraction: [wave][lane][8 floats] in the image the layer writes to (it is free until the
  // finished tile goes there; 16 x 64 x 32 B = 32 KB of its 33 KB)
  auto put_partial = [&](float* out, const f32x4 (&acc)[2]) __attribute__((always_inline)) {
    f32x4* pb = reinterpret_cast<f32x4*>(out) + (wave * 64 + lane) * 2;
    pb[0] = acc[0]; pb[1] = acc[1];
  };
  auto sum_partials = [&](const float* out, int n, f32x4 (&acc)[2]) __attribute__((always_inline)) {  // waves 0 .. n-1, fixed order
    const f32x4* pb = reinterpret_cast<const f32x4*>(out) + lane * 2;
    acc[0] = pb[0]; acc[1] = pb[1];
    for (int w = 1; w < n; ++w) {
      const f32x4 p0 = pb[w * 128], p1 = pb[w * 128 + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[0][r] += p0[r]; acc[1][r] += p1[r]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every partial is in registers before the tile overwrites the area
  };

  // ---- forward
  for (int l = 0; l < a.L; ++l) {
    const ChainLayer& ly = a.lt[l];
    const bool last = l == a.L - 1;
    const float* act = buf[cur];
    float* out = buf[cur ^ 1];
    const int nch = ly.KS >> 2;
    const bool split = ly.NT == 1;
    const int units = split ? nch : ly.NT;
    // what happens to a finished tile (bias already inside): the next layer's operand / the FORWARD-mode output / the loss
    auto finish = [&](int t, f32x4 (&acc)[2]) __attribute__((always_inline)) {
      const int n0 = 32 * t;
      if (!last) {
        unsigned bits = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (ly.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[s][r] = fmaxf(acc[s][r], 0.f);
              bits |= (acc[s][r] > 0.f ? 1u : 0u) << (4 * s + r);
            }
          }
          *reinterpret_cast<f32x4*>(out + m * PITCH + n0 + 16 * s + 4 * kq) = acc[s];
        }
        if (ly.relu && ly.mask_tile >= 0) masks[ly.mask_tile + t][lane] = (unsigned short)bits;
      } else if (GO(st.out)) {
        // FORWARD mode: z * out_std + out_mean straight from the accumulators: a row's 64 consecutive bytes per MFMA tile
        typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int n = n0 + 16 * s + 4 * kq;
          f32x4 v4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v4[r] = (ly.relu ? fmaxf(acc[s][r], 0.f) : acc[s][r]) * st.out_std + (st.out_mean ? st.out_mean[n + r < ly.N ? n + r : ly.N - 1] : 0.f);
          if (m < vrows) {
            float* dst = st.out + (long long)(m0 + m) * st.ldo + n;
            if (n + 3 < ly.N) *reinterpret_cast<f32x4_u*>(dst) = v4;
            else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (n + r < ly.N) dst[r] = v4[r];
            }
          }
        }
      } else {  // loss_i = w_i sum_j (p - y)^2,  dL/dp = scale w_i (p - y)
        const float gsc = st.scale * rwl[m];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int n = n0 + 16 * s + 4 * kq;
          const f32x4 yq = *reinterpret_cast<const f32x4*>(ystg + m * PITCH + n);
          f32x4 dd;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float df = n + r < ly.N ? acc[s][r] - yq[r] : 0.f;
            lsum += df * df;
            dd[r] = gsc * df;
          }
          *reinterpret_cast<f32x4*>(out + m * PITCH + n) = dd;
        }
      }
    };
    flush_t(act, ly.K, ly.ht16, split ? 0 : units);  // this layer's input -> operand of its weight gradient (split layer: by everyone, the contraction is short)
    if (wave < units) {
      f32x4 acc[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) acc[s] = f32x4{bnext[4 * s], bnext[4 * s + 1], bnext[4 * s + 2], bnext[4 * s + 3]};
      const Job nxt = fwd_from(l + 1);
      if (!split) {
        const bool odd = contract(fw + ly.fw_off + ((long long)wave * ly.KS) * 64 + lane, act, nch, acc, nxt);
        finish(wave, acc);
        settle(odd);
      } else {  // chunk `wave` of the only tile (waves > 0 started from a zero bias: see fwd_job)
        const bool odd = contract(fw + ly.fw_off + (long long)(4 * wave) * 64 + lane, act, 1, acc, nxt, 2 * wave);
        put_partial(out, acc);
        settle(odd);
      }
    }
    if (split) {
      chain_barrier();
      if (wave == 0) {
        f32x4 acc[2];
        sum_partials(out, units, acc);
        finish(0, acc);
      }
    }
    if (last && !GO(st.out)) {  // this lane's share of the row losses (lanes m, m + 16, m + 32, m + 48 hold one row)
      lsum += __shfl_xor(lsum, 16, 64);
      lsum += __shfl_xor(lsum, 32, 64);
      if (kq == 0) red[wave][m] = lsum * rwl[m];
    }
    chain_barrier();
    cur ^= 1;
    chain_stamp(a, 2 + l);
  }
  if (GO(st.out)) return;

  // ---- loss: lanes -> rows -> workgroup (fixed order) -> one fixed-point atomic per workgroup
  if (tid < ROWS) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w][tid];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (tid == 0) atomicAdd(a.loss_acc, (unsigned long long)(long long)llrint((double)s * 4294967296.0));
  }
  chain_stamp(a, 2 + a.L);
  if (FWD_ONLY) return;

  // ---- backward: layer l consumes dZ_l (in buf[cur]) and produces dZ_{l-1}
  for (int l = a.L - 1; l >= 1; --l) {
    const ChainLayer& ly = a.lt[l];
    const ChainLayer& below = a.lt[l - 1];
    const float* act = buf[cur];
    float* out = buf[cur ^ 1];
    const int nch = ly.NS >> 2;
    const bool split = ly.KT == 1;
    const int units = split ? nch : ly.KT;
    auto finish = [&](int t, f32x4 (&acc)[2]) __attribute__((always_inline)) {
      const int k0 = 32 * t;
      const unsigned bits = below.relu ? masks[below.mask_tile + t][lane] : 0xFFu;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[s][r];
          acc[s][r] = ((bits >> (4 * s + r)) & 1u) ? v : 0.f;
        }
        *reinterpret_cast<f32x4*>(out + m * PITCH + k0 + 16 * s + 4 * kq) = acc[s];
      }
    };
    flush_t(act, ly.N, ly.dzt16, split ? 0 : units);  // dZ of this layer's output -> operand of its weight gradient
    if (wave < units) {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const Job nxt = bwd_from(l - 1);
      if (!split) {
        const bool odd = contract(bw + ly.bw_off + ((long long)wave * ly.NS) * 64 + lane, act, nch, acc, nxt);
        finish(wave, acc);
        settle(odd);
      } else {
        const bool odd = contract(bw + ly.bw_off + (long long)(4 * wave) * 64 + lane, act, 1, acc, nxt, 2 * wave);
        put_partial(out, acc);
        settle(odd);
      }
    }
    if (split) {
      chain_barrier();
      if (wave == 0) {
        f32x4 acc[2];
        sum_partials(out, units, acc);
        finish(0, acc);
      }
    }
    chain_barrier();
    cur ^= 1;
    chain_stamp(a, 3 + a.L + (a.L - 1 - l));
  }
  flush_t(buf[cur], a.lt[0].N, a.lt[0].dzt16, 0);
#undef GO
#undef FWD_ONLY
}

template <int FEAT>
__global__ void __launch_bounds__(64 * kC32Waves) train_chain32_kernel(const ChainArgs a) {
  if ((int)blockIdx.x >= a.ncons) { chain_prefetch(a, a); return; }
  train_chain32_body<FEAT>(a, a);
}
// once per step on a data-parallel rank (the loss numerator must ride in the gradient arena BEFORE the exchange);
// a single rank lets the Adam kernel do it
static __global__ void chain32_loss_kernel(unsigned long long* acc, float* out) {
  *out = (float)((double)(long long)*acc * (1.0 / 4294967296.0));
  *acc = 0ull;
}

}  // namespace v21
