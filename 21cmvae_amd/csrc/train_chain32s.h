// train_chain32s.h -- the fp32 chain (train_chain32.h) for SMALL batches: 8 batch rows per workgroup (gfx950).
//
// train_chain32_kernel carries 16 rows per workgroup on the 16 x 16 x 4 f32 MFMA: at the reference's batch of 256 rows
// that is 16 workgroups, and the launch is bound by the f32 MFMA rate of the 16 CUs they occupy (63 k cycles of matrix
// work per workgroup; measured 127 k).  With the MFMAs compiled out the same kernel takes 78 k cycles: that is the fp32
// weight stream (2.7 MB through one CU) and the per-layer latencies -- what a workgroup with LESS matrix work would be
// left with.  Hence 8 rows per workgroup (32 workgroups at batch 256, 256 at 2,048) on the 4 x 4 x 1 f32 MFMA
// (16 blocks of 4 x 4, K = 1; layout checked by scripts/diag/mfma4_probe.hip):
//
//   lane l = 4 b + j of a wave: block b <-> the features 4 b .. 4 b + 3 of a 64-FEATURE tile, j <-> batch row j of a
//   group of four rows.  A operand = W[k][n0 + l] (lane l: ONE float), B operand = H[row j][k] (the same in all 16
//   blocks: a broadcast LDS read), D register i = feature 4 b + i for row j: the result goes back to the fp32
//   [row][feature] LDS image as one 16-byte write per lane and row group.  Two row groups share every weight word (two
//   MFMAs per word: the load path, 64 B/clk, and the matrix pipe are in balance; with one group the load path would
//   lose 2 : 1).  Weights come as 1-KiB fragments (lane l, element e = W[4 f + e][n0 + l]: four k per fragment, eight
//   MFMAs), four fragments per chunk of the usual rolling prefetch.
//   A layer has <= 8 tiles of 64 features, so its CONTRACTION may be split as well: unit (tile t, part s) takes a
//   contiguous range of the tile's chunks, the partial tiles meet in LDS and the wave of part 0 finishes the tile.  Which
//   layers are split, and what every wave does in every layer, the HOST works out once per trainer (c32s_split,
//   c32s_build_jobs below): the kernel reads its row of that table with scalar loads.
// Measured on the autoencoder stack at batch 256 (scripts/train_probe.py): 16-row kernel 129 k
// cycles -> this kernel 101 k with the units derived in the kernel -> 83 k with the host's table -> 79 k with the split
// chosen per layer; the step 73.6 -> 52.5 us.  Diagnostic builds (-DV21_C32S_NOLOAD / NOMFMA / NOLDS: the kernel without
// its weight loads / MFMAs / operand reads) put the three big layers at ~14.5 k cycles each on MFMAs alone (12.3 k is
// the pipe's rate, 8.8 cycles per 4 x 4 x 1 with two waves per SIMD), ~15 k on the stream alone (11.4 k at the load
// path's 63 B/clk), 18-19 k together; the six small layers, the gather and the loss are ~25 k of fixed cost.
// Used for f32 trainers whose max_batch is <= kC32sMaxBatch (a trainer commits to ONE packed-stream format: this
// kernel's `cprec = 4` streams, or the 16-row kernel's).  Same ChainModel / ChainStep blocks (ChainLayer::KS / NS =
// fragments per 64-wide tile, NT / KT = 64-wide tiles); training and validation (fwd_only); the variational head
// (sampled latent + KL, as train_chain.h) since the end of r3; no FORWARD mode.
#pragma once
#include "train_chain32.h"

namespace v21 {
#ifdef V21_C32S_NOLOAD  // (diagnostic builds: the kernel without its weight loads / LDS reads / MFMAs -- wrong results)
#define WLOAD(dst, src) asm volatile("" : "+v"(dst))
#else
#define WLOAD(dst, src) dst = src
#endif
#ifdef V21_CHAIN_FINE  // (diagnostic build: per-wave stamps of workgroup 0, scripts/diag/chain_wave_stamps.py)
#define FINE(i) do { if (blockIdx.x == 0 && lane == 0 && a.stamps) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); a.stamps[64 + (i) * 16 + wave] = t_; } } while (0)
#else
#define FINE(i)
#endif

constexpr int kC32sRows = 8;
// Steps of up to this many rows take the 4-row form (<= 256 workgroups: one round of the chip).  Batch 256 on the
// autoencoder stack: 79.4 k -> 63.1 k cycles (the big layers 18-19 k -> 14-15 k each: what the stream alone needs with
// one chunk of prefetch), step 49.9 -> 43.3 us; 512 rows 56.5 -> 51.3; 1,024 rows 70.5 -> 69.2; 2,048 rows (two rounds
// of workgroups) 90.0 -> 114.8: stays with 8 rows.
constexpr int kC32sRows4Max = 1024;
constexpr int kC32sWaves = 16;
constexpr int kC32sMaxBatch = 2048;  // 256 workgroups: one round of the chip (2,048 rows: 92.8 us per step against the 16-row kernel's 106.8)
constexpr int kC32sMaskTiles = 120;
constexpr int kC32sBufBytes = 2 * kC32sRows * kC32Pitch * 4;
constexpr int kC32sYBytes = kC32sRows * kC32Pitch * 4;
constexpr int kC32sMaskBytes = kC32sMaskTiles * 64 * 2;
constexpr int kC32sPartBytes = kC32sWaves * 64 * 8 * 4;  // partial tiles of a split contraction: [wave][lane][8 floats]
constexpr int kC32sZPitch = 2 * kChainMaxLatent + 4;  // joint step: the encoder's latents of the block's rows (fp32, <= 64 wide)
constexpr int kC32sZOff = kC32sBufBytes + kC32sYBytes + kC32sMaskBytes + kC32sPartBytes + kC32sWaves * kC32sRows * 4 + kC32sRows * 4 + 64;
constexpr int kC32sLdsBytes = kC32sZOff + kC32sRows * kC32sZPitch * 4 + kC32sRows * 4;  // ... + kl_weight * KL per row (variational head)
static_assert(kC32sZOff % 16 == 0, "LDS areas are 16-byte aligned");
// fragments (4 k each) per 64-feature tile over a contraction range of d, whole chunks of four
__host__ __device__ constexpr int chain32s_frags(int d) { return ((d + 3) / 4 + 3) / 4 * 4; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// how a layer's contraction is cut: `tiles` 64-wide tiles x `parts` chunk ranges of `cps` chunks = units, one per wave
struct C32sSplit { int tiles, parts, cps, units; };
inline C32sSplit c32s_split(int tiles, int nch) {
  // Splitting the contraction buys balance (a SIMD's MFMA pipe is already full with two of these waves: 8.8 cycles per
  // 4 x 4 x 1 MFMA against 8.4 with four, scripts/diag/mfma4_rate_probe.hip) and costs the meeting of the partial tiles
  // (a second barrier, the sum, ~700 cycles).  Wave w sits on SIMD w % 4: the layer lasts as long as its busiest SIMD,
  // or as one wave's chunks end to end (a chunk's loads are issued one chunk ahead: ~600 cycles from issue to use).
  // (The constants are the 8-row form's; with the 4-row form's 16 MFMAs per chunk the model would also split the 8-tile
  // 352 -> 451 layer into 16 units of 11 chunks -- measured: 14.6-15.0 k cycles against 13.9 k unsplit.  Eight waves
  // already move that layer's weights at 50 B/clk; the meeting of the partial tiles is all the split adds.)
  const int kChunkCycles = 32 * 9, kChunkLatency = 600, kMeetCycles = 700;
  int max_parts = kC32sWaves / tiles;
  if (max_parts > nch) max_parts = nch;
  if (max_parts < 1) max_parts = 1;
  C32sSplit best{tiles, 1, nch, tiles};
  long long best_cost = -1;
  for (int want = 1; want <= max_parts; ++want) {
    const int cps = (nch + want - 1) / want, parts = (nch + cps - 1) / cps;
    int simd[4] = {0, 0, 0, 0};
    for (int u = 0; u < tiles * parts; ++u) {
      const int sidx = u % parts, c0 = sidx * cps;
      simd[u % 4] += std::min(cps, nch - c0);
    }
    const int busiest = std::max(std::max(simd[0], simd[1]), std::max(simd[2], simd[3]));
    const long long cost = std::max((long long)busiest * kChunkCycles, (long long)cps * kChunkLatency) + (parts > 1 ? kMeetCycles : 0);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = C32sSplit{tiles, parts, cps, tiles * parts}; }
  }
  return best;
}
// What a wave does in one step of the chain (steps 0 .. L-1: forward layers; L .. 2L-2: backward layers L-1 .. 1), worked
// out by the host.  (First cut: every wave derived its unit and its next unit in the kernel -- three integer divisions
// per split, a search over the layers for the next one: ~1,500 VALU cycles per wave and layer, and since the waves of a
// SIMD issue oldest first, the younger waves' preambles waited behind the older waves' MFMAs: their contractions started
// 2 / 6 / 12 k cycles late in an 18-k-cycle layer.)  One row per (step, wave), row 0 of the table = before step 0; read
// with scalar loads.
struct C32sJob {
  int w_off;    // first fragment of the unit (16-byte words from the stream's base, lane 0)
  int nch;      // chunks of the unit (0: this wave has no unit in this step)
  int f0;       // first fragment within the tile's contraction
  int t, s;     // tile, part
  int parts;    // parts per tile in this step
  int units;    // units of this step
  int nxt_w_off, nxt_bw;  // the wave's NEXT unit (whose first chunk rolls in under this one's last): offset, 1 = backward stream
  int nxt_b_off, nxt_nb;  // ... its bias (arena offset of the tile's first feature) and how many features have one (0: none)
  // what the step needs of its layer (kept here, not read from ChainModel::lt[l] in the kernel: a dynamically indexed
  // kernel-argument field is a scalar load where it is used, ~200 cycles each and waited for on the spot -- five such
  // round trips stood at the head of every step)
  int mask_tile;   // forward: first mask tile of this layer's ReLU (-1: a linear layer; a hidden ReLU layer always keeps
                   // its mask); backward: of the layer below, whose mask applies to the gradient this step produces
  int width;       // forward: output width N (the loss masks the padding); backward: unused
  int flush_f;     // features of the activation image this step flushes (forward: K, backward: N)
  unsigned flush_lo, flush_hi;  // ... and where to (ht16 of the layer / dzt16 of the layer)
};
static_assert(sizeof(C32sJob) == 64, "C32sJob rows are read as 16 dwords");
// rows (1 + 2 L - 1) x 16 waves; `fw_off` / `bw_off` in 16-byte words, `b_off` in floats (ChainLayer's)
inline void c32s_build_jobs(const ChainModel& a, C32sJob* tab) {
  const int L = a.L, steps = 2 * L - 1;
  for (int i = 0; i < (steps + 1) * kC32sWaves; ++i) tab[i] = C32sJob{};
  for (int i = 0; i < steps; ++i) {
    const bool fwd = i < L;
    const int l = fwd ? i : L - 1 - (i - L);
    const ChainLayer& ly = a.lt[l];
    const C32sSplit sp = fwd ? c32s_split(ly.NT, ly.KS >> 2) : c32s_split(ly.KT, ly.NS >> 2);
    const int frags = fwd ? ly.KS : ly.NS, nch = frags >> 2;
    for (int w = 0; w < kC32sWaves; ++w) {
      C32sJob& j = tab[(1 + i) * kC32sWaves + w];
      j.parts = sp.parts; j.units = sp.units;
      const ChainLayer& mk = fwd ? ly : a.lt[l - 1];
      j.mask_tile = mk.relu ? mk.mask_tile : -1;
      j.width = ly.N;
      j.flush_f = fwd ? ly.K : ly.N;
      const unsigned long long dst = (unsigned long long)(fwd ? ly.ht16 : ly.dzt16);
      j.flush_lo = (unsigned)dst; j.flush_hi = (unsigned)(dst >> 32);
      if (w >= sp.units) continue;
      j.t = w / sp.parts; j.s = w % sp.parts;
      const int c0 = j.s * sp.cps;
      j.nch = std::min(sp.cps, nch - c0);
      j.f0 = 4 * c0;
      j.w_off = (int)((fwd ? ly.fw_off : ly.bw_off) + ((long long)j.t * frags + 4 * c0) * 64);
    }
  }
  for (int w = 0; w < kC32sWaves; ++w) {  // every row's "next unit" (searched backwards)
    int nw = 0, nbw = 0, nb_off = 0, nnb = 0;  // nothing left: any valid address (the data is never used)
    for (int i = steps - 1; i >= -1; --i) {
      C32sJob& j = tab[(1 + i) * kC32sWaves + w];
      j.nxt_w_off = nw; j.nxt_bw = nbw; j.nxt_b_off = nb_off; j.nxt_nb = nnb;
      if (i >= 0 && j.nch > 0) {
        const bool fwd = i < L;
        const int l = fwd ? i : L - 1 - (i - L);
        nw = j.w_off; nbw = fwd ? 0 : 1;
        nb_off = fwd ? (int)a.lt[l].b_off + 64 * j.t : 0;
        nnb = fwd && j.s == 0 ? a.lt[l].N - 64 * j.t : 0;
      }
    }
  }
}

// Host-side check of a job table against the buffers it points into -- the kernel follows the table blindly (scalar
// loads of offsets, no range checks in the loop), so a bad row is a GPU memory fault, not an error code.  Every address
// a wave may REQUEST must lie inside the allocation: the chunks of its unit, and the first chunk of its next unit (the
// rolling prefetch requests it under this unit's last chunk whether or not it is ever used).  `fw_words` / `bw_words`:
// sizes of the two packed streams in 16-byte words (one lane's share of a fragment; a fragment is 64 words, a chunk 256);
// `arena_floats`: size of the parameter arena (biases).  Returns nullptr or a description of the first bad row (static
// buffer).  Called by build_chain32s_jobs (api_trainer.hip) when a trainer is created and by v21_debug_check_chain_jobs.
inline const char* c32s_validate_jobs(const ChainModel& a, const C32sJob* tab, long long fw_words, long long bw_words,
                                      long long arena_floats) {
  static thread_local char msg[256];
  const int L = a.L, steps = 2 * L - 1;
  auto bad = [&](int i, int w, const char* what, long long v, long long lim) {
    snprintf(msg, sizeof msg, "chain job table: step %d wave %d: %s = %lld outside [0, %lld]", i, w, what, v, lim);
    return msg;
  };
  for (int i = -1; i < steps; ++i) {
    const bool fwd = i < L;
    const int l = i < 0 ? 0 : (fwd ? i : L - 1 - (i - L));
    const ChainLayer& ly = a.lt[l];
    const int frags = fwd ? ly.KS : ly.NS, tiles = fwd ? ly.NT : ly.KT;
    const long long words = fwd ? fw_words : bw_words;
    const long long lay0 = fwd ? ly.fw_off : ly.bw_off;
    for (int w = 0; w < kC32sWaves; ++w) {
      const C32sJob& j = tab[(1 + i) * kC32sWaves + w];
      if (i >= 0 && j.nch > 0) {
        if (j.t < 0 || j.t >= tiles) return bad(i, w, "tile", j.t, tiles - 1);
        if (j.parts < 1 || j.s < 0 || j.s >= j.parts) return bad(i, w, "part", j.s, j.parts - 1);
        if (j.f0 < 0 || j.f0 + 4 * j.nch > frags) return bad(i, w, "last fragment of the unit", j.f0 + 4 * j.nch, frags);
        const long long end = (long long)j.w_off + 256ll * j.nch;
        if (j.w_off < lay0 || end > lay0 + (long long)tiles * frags * 64) return bad(i, w, "unit end (words, within its layer)", end, lay0 + (long long)tiles * frags * 64);
        if (j.w_off < 0 || end > words) return bad(i, w, "unit end (words, within the stream)", end, words);
        if (j.mask_tile >= 0 && j.mask_tile + tiles > kC32sMaskTiles) return bad(i, w, "mask tile", j.mask_tile + tiles, kC32sMaskTiles);
        if (j.flush_f > 0 && !j.flush_lo && !j.flush_hi) return bad(i, w, "flush pointer", 0, 0);
      }
      // the next unit's first chunk is requested in any case (offset 0 of the forward stream when nothing is left)
      const long long nwords = j.nxt_bw ? bw_words : fw_words;
      if (j.nxt_w_off < 0 || (long long)j.nxt_w_off + 256 > nwords) return bad(i, w, "next unit's first chunk end (words)", (long long)j.nxt_w_off + 256, nwords);
      if (j.nxt_nb > 0 && (j.nxt_b_off < 0 || (long long)j.nxt_b_off + std::min(j.nxt_nb, 64) > arena_floats))
        return bad(i, w, "next unit's bias end (floats)", (long long)j.nxt_b_off + std::min(j.nxt_nb, 64), arena_floats);
    }
  }
  return nullptr;
}

// ROWS = 8: two groups of four rows share every weight word (two MFMAs per word); ROWS = 4: one group -- twice the
// workgroups, half the matrix work in each, the same weight stream through each CU (the LDS carve-up stays the 8-row one).
// GAUSS = false: a promise of the host that the stack has no variational head -- its code folds away (with it compiled in,
// a plain autoencoder's step took 44.5 instead of 42.9 us at batch 256: 67 k against 63 k cycles of the chain).
template <int ROWS, bool GAUSS>
__device__ __forceinline__ void train_chain32s_body(const ChainModel& a, const ChainStep& st, const int bidx) {
  static_assert(ROWS == 4 || ROWS == 8, "one or two groups of four rows");
  constexpr int NW = kC32sWaves, PITCH = kC32Pitch, G = ROWS / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  float(*buf)[ROWS * PITCH] = reinterpret_cast<float(*)[ROWS * PITCH]>(chain_smem);
  float* ystg = reinterpret_cast<float*>(chain_smem + kC32sBufBytes);
  unsigned short(*masks)[64] = reinterpret_cast<unsigned short(*)[64]>(chain_smem + kC32sBufBytes + kC32sYBytes);
  f32x4* part = reinterpret_cast<f32x4*>(chain_smem + kC32sBufBytes + kC32sYBytes + kC32sMaskBytes);
  float(*red)[ROWS] = reinterpret_cast<float(*)[ROWS]>(chain_smem + kC32sBufBytes + kC32sYBytes + kC32sMaskBytes + kC32sPartBytes);
  float* rwl = reinterpret_cast<float*>(chain_smem + kC32sBufBytes + kC32sYBytes + kC32sMaskBytes + kC32sPartBytes + NW * ROWS * 4);
  // joint step (train_chain32s_joint_kernel): the encoder's latents; variational head: (mu | lv) of the block's rows
  // (the encoder pass of a variational autoencoder leaves both at once: the latents ARE z_mean)
  float* zs = reinterpret_cast<float*>(chain_smem + kC32sZOff);
  constexpr int ZP = kC32sZPitch;
  float* klb = zs + kC32sRows * ZP;  // kl_weight * KL_i

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jr = lane & 3, blk = lane >> 2;  // batch row within a group of four, feature block of the tile
  const long long first = st.sc.desc ? st.sc.desc[*st.sc.cur].first : st.first;
  const int nrb = (st.rows + ROWS - 1) / ROWS;
  const int rb = (bidx & 7) * ((nrb + 7) >> 3) + (bidx >> 3);  // XCD-major row blocks (speed only)
  if (rb >= nrb) return;
  const int m0 = rb * ROWS;
  const int vrows = st.rows - m0;  // valid rows of this block (>= 1; < 8 only in the last block)

  const f32x4* fw = reinterpret_cast<const f32x4*>(a.fw);
  const f32x4* bw = reinterpret_cast<const f32x4*>(a.bw);

  // ---- the wave's rows of the job table (uniform: scalar loads from the constant address space)
  struct Job { const f32x4* w; const float* b; int nb; };  // first fragment; bias of the tile (nb features have one)
  typedef const C32sJob __attribute__((address_space(4)))* jobptr;
  const jobptr jobs = (jobptr)(unsigned long long)a.jobs + wave;
  int gl = -1;  // the variational head (V21_ACT_GAUSS: Dense outputs [z_mean | z_log_var]) or -1
  if constexpr (GAUSS)
    for (int l = 0; l + 1 < a.L; ++l) gl = a.lt[l].gauss ? l : gl;
  // (a row is fetched one step before it is used: the scalar loads of a step's row miss the constant cache -- every
  // wave has its own 64 bytes per step -- and ~600 cycles at the head of each of nine steps were exactly that)
  auto row = [&](int r) __attribute__((always_inline)) -> C32sJob {
    const jobptr p = jobs + (r < 2 * a.L ? r : 2 * a.L - 1) * NW;
    C32sJob j;
    j.w_off = p->w_off; j.nch = p->nch; j.f0 = p->f0; j.t = p->t; j.s = p->s; j.parts = p->parts; j.units = p->units;
    j.nxt_w_off = p->nxt_w_off; j.nxt_bw = p->nxt_bw; j.nxt_b_off = p->nxt_b_off; j.nxt_nb = p->nxt_nb;
    j.mask_tile = p->mask_tile; j.width = p->width; j.flush_f = p->flush_f;
    j.flush_lo = p->flush_lo; j.flush_hi = p->flush_hi;
    return j;
  };
  auto next_of = [&](const C32sJob& j) __attribute__((always_inline)) -> Job {
    return Job{(j.nxt_bw ? bw : fw) + j.nxt_w_off + lane, a.w + j.nxt_b_off, j.nxt_nb};
  };
  auto load_bias = [&](const Job& j, f32x4& bv) __attribute__((always_inline)) {  // features 4 blk .. + 3 of the tile
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * blk + r;
      const float v = j.b[o < j.nb ? o : 0];  // (clamped: always a valid address; nb = 0 for parts > 0 and backward jobs)
      bv[r] = o < j.nb ? v : 0.f;
    }
  };
  // waves 0 .. 7 gather one row of the block each: the source row first (the oldest load in flight)
  const int mq = m0 + (wave < ROWS ? wave : 0);
  long long srow = (wave < ROWS && mq < st.rows) ? first + mq : first + m0;  // clamped: always a valid position
  if (st.idx) srow = st.idx[srow];
  f32x4 wa[4], wb[4];
  f32x4 bnext;
  {
    const Job j0 = next_of(row(0));
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[j] = j0.w[j * 64];
    load_bias(j0, bnext);
  }
  chain_stamp(a, 0);
  // buf[1] <- 0 once: padding columns must hold finite values (they meet zero weights); buf[0] is written in full below
  for (int i = tid; i < ROWS * PITCH / 4; i += 64 * NW) reinterpret_cast<f32x4*>(buf[1])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- gather: x[idx] -> buf[0] (and the target rows)
  if (wave < ROWS) {
    const int K0 = a.lt[0].K, DO = a.lt[a.L - 1].N;
    const float rwv = st.rw ? st.rw[srow] : 0.f;
    if (lane == 0) rwl[wave] = mq < st.rows ? rwv : 0.f;
    const int kmax = mq < st.rows ? K0 : 0;
    const float* xs = st.x + srow * st.ldx;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + 64 * i;
      const float t = xs[k < K0 ? k : K0 - 1];
      v[i] = k < kmax ? t : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + 64 * i;  // <= 511 < PITCH
      buf[0][wave * PITCH + k] = v[i];
      if (!st.y && !st.y_from_lds) ystg[wave * PITCH + k] = v[i];
    }
    if (st.y) {
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
  if (st.y_from_lds) {  // joint step: the targets are the latents the encoder pass of this workgroup left in `zs`
    const int DOl = a.lt[a.L - 1].N;
    for (int i = tid; i < ROWS * DOl; i += 64 * NW) ystg[(i / DOl) * PITCH + i % DOl] = zs[(i / DOl) * ZP + i % DOl];
  }
  C32sJob jnext = row(1);  // (collected by the barrier's wait, see the layer loop)
  chain_barrier();
  chain_stamp(a, 1);
  float lsum[G] = {};
  int cur = 0;

  // The 8 rows x F features in `act` -> the fp32 operand of the weight gradient (dst[f * BS + batch row], gemm_nt.h): a lane
  // takes one feature and four consecutive rows and stores them as one 16-byte word (32 features x 32 contiguous bytes per
  // instruction).  Done by the waves without a unit in the contraction that follows; rows past the batch are zeros.
  auto flush_t = [&](const float* act, int F, void* dst, int units) __attribute__((always_inline)) {
    constexpr int FPI = 64 / G;  // features per instruction
    const int ngrp = (F + FPI - 1) / FPI;
    float* d = reinterpret_cast<float*>(dst);
    const int w0 = units < NW ? units : 0;
    if (wave < w0 || st.fwd_only) return;
    const int fq = G == 2 ? lane >> 1 : lane, r4 = G == 2 ? 4 * (lane & 1) : 0;
    __builtin_amdgcn_s_setprio(3);  // (the waves without a unit are the youngest of the workgroup, and the SIMDs issue oldest first)
    for (int id = wave - w0; id < ngrp; id += NW - w0) {
      const int f = FPI * id + fq;
      const int fc = f < F ? f : F - 1;
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = r4 + j < vrows ? act[(r4 + j) * PITCH + fc] : 0.f;
      // (an address-space-1 store: `dst` comes out of the job row as an integer, and through a generic pointer this
      //  would be a FLAT store, which counts on vmcnt AND lgkmcnt -- every wait of the contraction then degrades to 0)
      typedef f32x4 __attribute__((address_space(1))) * gptr;
      if (f < F) *(gptr)(unsigned long long)(d + (long long)f * a.BS + m0 + r4) = v;
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // One unit: acc[g](register i = feature 4 blk + i of the tile, row 4 g + jr) over `nch` chunks of four fragments
  // starting at fragment f0 of the tile's contraction.  Weight chunks roll as in train_chain_body (wa in use / wb in
  // flight, roles alternate); the activation words of the next fragment are read under the 8 MFMAs of this one.
  // (Units padded to whole PAIRS of chunks, so that the two buffers never change roles and no register move waits for a
  // chunk in flight: measured slower -- +3 % of stream in the 451-wide layers, a zero chunk in the 9-wide one.
  // BOTH buffers requested before the step's barrier -- the job row names the wave's next two chunks, a buffer is refilled
  // as soon as its MFMAs are issued, after an odd unit the buffers are exchanged at the head of the wave's next unit: no
  // gain, 44.3 against 43.3 us at batch 256 (4 rows), 93.7 against 90.0 at 2,048 (8 rows).  The steady state is the same
  // sequence of requests, and what a unit saves at its start it pays where the exchange, and the bias values requested an
  // unknown number of loads earlier, make hipcc wait for everything in flight.  The big layers move their weights at
  // 47-51 B/clk; the same two-buffer loop alone reaches 56 (scripts/diag/chain_loop_probe.hip), a pure stream 63.
  // THREE buffers in the 4-row form -- two chunks in flight per wave, the job row naming the wave's next three chunks, the
  // buffers rotated back by selects at the head of a unit (under control flow hipcc kept two of them in scratch memory:
  // 335 k cycles): 126 VGPRs, nothing spilled, and slower: 74.5 k cycles against 66 k, the unsplit 8-tile layer
  // (22 chunks per wave end to end) 14.2 k against 13.9 k.  What looked like one round trip per chunk is the CU's load
  // path at ~50 B/clk, shared by the waves that stream.
  // That experiment also ABORTED the process once (r3, gpurun_out/r3ring2.log: SIGABRT inside v21_trainer_run_epoch of
  // tests/test_train_gpu.py::test_vae_step_matches_oracle, the suite's smallest f32 stack 451-64-(9|9)-32-451, every
  // larger stack before it had passed).  The diff was dropped uncommitted a minute later; what is in hand names the
  // cause: with two chunks in flight a wave requests the chunk TWO ahead of the one it computes on, and for the wave
  // that holds the last unit of a packed stream -- and for every unit shorter than the look-ahead, which is every unit of
  // that stack's 1- and 2-chunk layers -- that address lies up to 8 KiB PAST THE END OF THE STREAM.  Streams are whole
  // 4-KiB chunks (chain32s_frags pads to four fragments) and were allocated with 64 bytes of slack, so a stream ends
  // on a page boundary and the request lands on the page after the allocation: harmless while hipMalloc happens to
  // place another buffer there (all the stacks before), a memory access fault -- which the HSA runtime answers with
  // abort() -- when it does not.  The two-buffer kernel never had the problem (its one look-ahead address is the next
  // unit's first chunk, or offset 0 when nothing is left); the class of bug is "the kernel trusts offsets the host
  // wrote".  Since r4: c32s_validate_jobs checks every address a row makes a wave request against the allocated sizes
  // when the table is built (V21_ERR_STATE instead of a fault; tests/test_train_gpu.py truncates a stream on purpose), and
  // the packed streams carry kChainStreamSlack = two chunks of zeroed slack, so a deeper prefetch tried later reads
  // zeros inside the allocation instead of whatever lies behind it.)
  auto contract = [&](const f32x4* wsrc, const float* act, int nch, f32x4 (&acc)[G], const Job nxt, int f0) __attribute__((always_inline)) -> bool {
    const float* ap = act + jr * PITCH + 4 * f0;
    load_bias(nxt, bnext);  // (the caller has consumed this unit's values)
    f32x4 bq[G], nq[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bq[g] = *reinterpret_cast<const f32x4*>(ap + 4 * g * PITCH);
    auto chunk = [&](f32x4 (&w)[4], int kc, bool more) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int fn = 4 * kc + j + 1;  // the next fragment of this unit (read ahead; past the unit: a harmless re-read)
#ifndef V21_C32S_NOLDS
        if (j < 3 || more) {
#pragma unroll
          for (int g = 0; g < G; ++g) nq[g] = *reinterpret_cast<const f32x4*>(ap + 4 * g * PITCH + 4 * fn);
        }
        // (the reads stay AHEAD of this fragment's MFMAs: left to itself the scheduler now and then sinks them to just
        // before their use, and the wait that follows exposes an LDS round trip per fragment -- 18.8 vs 21.5 k cycles per
        // big layer between two builds that differed in unrelated scalar code)
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef V21_C32S_NOMFMA  // (one VALU instruction per fragment keeps the loads alive)
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[0][0]) : "v"(w[j][0]), "v"(bq[0][0]), "v"(bq[G - 1][0]));
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = mfma4(w[j][e], bq[g][e], acc[g]);
        }
#endif
#ifndef V21_C32S_NOLDS
#pragma unroll
        for (int g = 0; g < G; ++g) bq[g] = nq[g];
#endif
      }
    };
    int c = 0;
    for (; c + 2 <= nch; c += 2) {
      {
        const f32x4* p = wsrc + (long long)(4 * (c + 1)) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) WLOAD(wb[j], p[j * 64]);
      }
      __builtin_amdgcn_sched_barrier(0);  // (keeps the first MFMA -- and the wait for the CURRENT chunk -- below these loads)
      chunk(wa, c, true);
      {
        const f32x4* p = c + 2 < nch ? wsrc + (long long)(4 * (c + 2)) * 64 : nxt.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) WLOAD(wa[j], p[j * 64]);
      }
      __builtin_amdgcn_sched_barrier(0);
      chunk(wb, c + 1, c + 2 < nch);
    }
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 4; ++j) WLOAD(wb[j], nxt.w[j * 64]);
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
  auto put_partial = [&](const f32x4 (&acc)[G]) __attribute__((always_inline)) {
    f32x4* pb = part + (wave * 64 + lane) * 2;
#pragma unroll
    for (int g = 0; g < G; ++g) pb[g] = acc[g];
  };
  auto sum_partials = [&](int n, f32x4 (&acc)[G]) __attribute__((always_inline)) {  // waves wave .. wave + n - 1, fixed order
    const f32x4* pb = part + (wave * 64 + lane) * 2;
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = pb[g];
    for (int w = 1; w < n; ++w) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 p = pb[w * 128 + g];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[g][r] += p[r];
      }
    }
  };

  // ---- forward
  const int LF = st.nfwd > 0 ? st.nfwd : a.L;  // (joint step: the encoder alone)
  for (int l = 0; l < LF; ++l) {
    const bool last = l == a.L - 1;
    const float* act = buf[cur];
    float* out = buf[cur ^ 1];
    const C32sJob jb = jnext;
    const int parts = jb.parts, t = jb.t;
    auto finish = [&](int t, f32x4 (&acc)[G]) __attribute__((always_inline)) {
      const int n = 64 * t + 4 * blk;
      if (!last) {
        unsigned bits = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (jb.mask_tile >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[g][r] = fmaxf(acc[g][r], 0.f);
              bits |= (acc[g][r] > 0.f ? 1u : 0u) << (4 * g + r);
            }
          }
          *reinterpret_cast<f32x4*>(out + (4 * g + jr) * PITCH + n) = acc[g];
        }
        if (jb.mask_tile >= 0) masks[jb.mask_tile + t][lane] = (unsigned short)bits;
      } else {  // loss_i = w_i sum_j (p - y)^2,  dL/dp = scale w_i (p - y)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float gsc = st.scale * rwl[4 * g + jr];
          const f32x4 yq = *reinterpret_cast<const f32x4*>(ystg + (4 * g + jr) * PITCH + n);
          f32x4 dd;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float df = n + r < jb.width ? acc[g][r] - yq[r] : 0.f;
            lsum[g] += df * df;
            dd[r] = gsc * df;
          }
          *reinterpret_cast<f32x4*>(out + (4 * g + jr) * PITCH + n) = dd;
        }
      }
    };
    FINE(4 * l);
    flush_t(act, jb.flush_f, (void*)(((unsigned long long)jb.flush_hi << 32) | jb.flush_lo), jb.units);  // this layer's input -> operand of its weight gradient
    if (jb.nch > 0) {
      f32x4 acc[G];
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = bnext;
      FINE(4 * l + 1);
      const bool odd = contract(fw + jb.w_off + lane, act, jb.nch, acc, next_of(jb), jb.f0);
      if (parts == 1) finish(t, acc);
      else put_partial(acc);
      settle(odd);
    }
    FINE(4 * l + 2);
    if (parts > 1) {
      chain_barrier();
      FINE(4 * l + 3);
      if (jb.nch > 0 && jb.s == 0) {
        f32x4 acc[G];
        sum_partials(parts, acc);
        finish(t, acc);
      }
    }
    if (last) {  // this lane's share of the row losses (the 16 lanes with the same jr hold one row of each group)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float v = lsum[g];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        if (blk == 0) red[wave][4 * g + jr] = v * rwl[4 * g + jr];
      }
    }
    // the next step's row: requested HERE, so that the barrier's own `s_waitcnt lgkmcnt(0)` collects it.  (Requested at
    // the head of the step it was still in flight inside the contraction, and with a scalar load outstanding every wait
    // for an LDS read there became lgkmcnt(0): the operand read-ahead was gone, 18.8 -> 21.5 k cycles per big layer.)
    jnext = row(2 + l);
    chain_barrier();
    cur ^= 1;
    if (l == gl) {  // z = mu + exp(lv / 2) eps replaces mu in the image ((mu | lv) kept for the backward pass); KL_i -> the row's loss
      // one (batch row, latent dimension) per thread; the columns past z keep lv: finite values against zero weights
      const int LAT = a.lt[l].N >> 1;
      float* img = buf[cur];
      // (KL terms in the first ROWS * LAT <= 256 floats of the partial-tile area: that is wave 0's own 2 KB of it, and
      //  wave 0 -- whose first lanes sum the terms below -- writes partial tiles there only afterwards, in program order)
      float* kls = reinterpret_cast<float*>(part);
      const int row = tid & (ROWS - 1), d = tid / ROWS;
      if (d < LAT) {
        const float mu = img[row * PITCH + d], lv = img[row * PITCH + LAT + d];
        const float sd = expf(0.5f * lv);
        const float e = a.sample ? gauss_eps(a.seed, a.step + st.step_off, st.row0 + m0 + row, d) : 0.f;
        zs[row * ZP + d] = mu; zs[row * ZP + LAT + d] = lv;
        img[row * PITCH + d] = mu + sd * e;
        kls[d * ROWS + row] = -0.5f * (1.0f + lv - mu * mu - sd * sd);
      }
      chain_barrier();
      if (tid < ROWS) {
        float kl = 0.f;
        for (int dd = 0; dd < LAT; ++dd) kl += kls[dd * ROWS + tid];  // fixed order
        klb[tid] = m0 + tid < st.rows ? a.kl_weight * kl : 0.f;
      }
    }
    chain_stamp(a, 2 + l);
  }

  if (LF < a.L) {  // joint step, the encoder alone: its last layer's outputs (the image the last barrier completed) are
                   // the targets of the model that follows in this workgroup
    // (a variational head: z_mean -- what encoder.predict returns -- is in `zs` already)
    if (LF - 1 != gl) {
      const int W = a.lt[LF - 1].N;
      for (int i = tid; i < ROWS * W; i += 64 * NW) zs[(i / W) * ZP + i % W] = buf[cur][(i / W) * PITCH + i % W];
    }
    return;
  }
  // ---- loss: lanes -> rows -> workgroup (fixed order) -> one fixed-point atomic per workgroup
  if (tid < ROWS) {
    float sl = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sl += red[w][tid];
    if (gl >= 0) sl += klb[tid];
#pragma unroll
    for (int o = ROWS / 2; o > 0; o >>= 1) sl += __shfl_xor(sl, o, 64);
    if (tid == 0) atomicAdd(a.loss_acc, (unsigned long long)(long long)llrint((double)sl * 4294967296.0));
  }
  chain_stamp(a, 2 + a.L);
  if (st.fwd_only) return;

  // dL/dz (latent wide, in `img`) -> dL/d[mu | lv] in place: d mu = dz + beta mu, d lv = dz eps exp(lv / 2) / 2 +
  // beta (exp lv - 1) / 2, beta = kl_weight / B (the KL term's own gradient) -- train_chain.h: gauss_backward
  auto gauss_backward = [&](float* img) __attribute__((always_inline)) {
    const int LAT = a.lt[gl].N >> 1;
    const int row = tid & (ROWS - 1), d = tid / ROWS;
    if (d < LAT) {
      const float mu = zs[row * ZP + d], lv = zs[row * ZP + LAT + d];
      const float sd = expf(0.5f * lv);
      const float e = a.sample ? gauss_eps(a.seed, a.step + st.step_off, st.row0 + m0 + row, d) : 0.f;
      const float g = img[row * PITCH + d];
      const float kb = m0 + row < st.rows ? st.gs * a.kl_weight * st.inv_b : 0.f;
      img[row * PITCH + d] = g + kb * mu;
      img[row * PITCH + LAT + d] = g * e * 0.5f * sd + kb * 0.5f * (sd * sd - 1.0f);
    }
    chain_barrier();
  };
  // ---- backward: layer l consumes dZ_l (in buf[cur]) and produces dZ_{l-1}
  for (int l = a.L - 1; l >= 1; --l) {
    if (l == gl) gauss_backward(buf[cur]);
    const float* act = buf[cur];
    float* out = buf[cur ^ 1];
    const C32sJob jb = jnext;
    const int parts = jb.parts, t = jb.t;
    auto finish = [&](int t, f32x4 (&acc)[G]) __attribute__((always_inline)) {
      const int k = 64 * t + 4 * blk;
      const unsigned bits = jb.mask_tile >= 0 ? masks[jb.mask_tile + t][lane] : 0xFFu;
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[g][r];
          acc[g][r] = ((bits >> (4 * g + r)) & 1u) ? v : 0.f;
        }
        *reinterpret_cast<f32x4*>(out + (4 * g + jr) * PITCH + k) = acc[g];
      }
    };
    flush_t(act, jb.flush_f, (void*)(((unsigned long long)jb.flush_hi << 32) | jb.flush_lo), jb.units);  // dZ of this layer's output -> operand of its weight gradient
    if (jb.nch > 0) {
      f32x4 acc[G];
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool odd = contract(bw + jb.w_off + lane, act, jb.nch, acc, next_of(jb), jb.f0);
      if (parts == 1) finish(t, acc);
      else put_partial(acc);
      settle(odd);
    }
    if (parts > 1) {
      chain_barrier();
      if (jb.nch > 0 && jb.s == 0) {
        f32x4 acc[G];
        sum_partials(parts, acc);
        finish(t, acc);
      }
    }
    jnext = row(2 + a.L + (a.L - 1 - l));
    chain_barrier();
    cur ^= 1;
    chain_stamp(a, 3 + a.L + (a.L - 1 - l));
  }
  if (gl == 0) gauss_backward(buf[cur]);
  flush_t(buf[cur], a.lt[0].N, a.lt[0].dzt16, 0);
}

template <int ROWS, bool GAUSS = false>
__global__ void __launch_bounds__(64 * kC32sWaves) train_chain32s_kernel(const ChainArgs a) {
  if ((int)blockIdx.x >= a.ncons) { chain_prefetch(a, a); return; }
  train_chain32s_body<ROWS, GAUSS>(a, a, (int)blockIdx.x - a.blk0);
}
// joint step (train_chain.h: train_chain_joint_kernel; BASELINE configs[2]) in the reference's arithmetic: blocks
// [0, ncons) carry the autoencoder's row blocks, blocks [ncons, 2 ncons) the emulator's -- first the ENCODER alone on the
// block's rows (forward layers [0, zcap_layer]; its latents stay in LDS), then the emulator's chain with those latents as
// targets.  tab[0] = autoencoder, tab[1] = emulator (device memory); no prefetchers.  The latent layer may be the
// autoencoder's variational head (the emulator then learns z_mean); the emulator has none.
template <int ROWS, bool GAUSS = false>  // GAUSS: the autoencoder is variational (the emulator never is)
__global__ void __launch_bounds__(64 * kC32sWaves) train_chain32s_joint_kernel(const ChainModel* __restrict__ tab, const ChainStep sa, const ChainStep sb) {
  const int b = (int)blockIdx.x;
  if (b < sa.ncons) { train_chain32s_body<ROWS, GAUSS>(tab[0], sa, b); return; }
  ChainStep se = sa;
  se.fwd_only = 1; se.nfwd = tab[0].zcap_layer + 1;
  train_chain32s_body<ROWS, GAUSS>(tab[0], se, b - sa.ncons);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the latents are written; the encoder's last LDS reads precede the emulator's gather
  train_chain32s_body<ROWS, false>(tab[1], sb, b - sa.ncons);
}
// a sweep: `models` per-model blocks in device memory (as train_chain_group_kernel); no prefetchers.  Workgroup b carries
// row block b / models of model b % models: workgroups are dealt to the XCDs round-robin, so with 8 models (or a divisor
// or multiple of 8) a model's row blocks share ONE XCD and its weights are fetched into one L2 -- with blockIdx.y = model
// every XCD streamed every model's weights (8 x 2.5 MB through a 4-MB L2): 68.8 us per launch of 8 models.
template <int ROWS, bool GAUSS = false>  // GAUSS: some member of the sweep has a variational head
__global__ void __launch_bounds__(64 * kC32sWaves) train_chain32s_group_kernel(const ChainModel* __restrict__ tab, const ChainStep st, const int models) {
  const int b = (int)blockIdx.x;
  train_chain32s_body<ROWS, GAUSS>(tab[b % models], st, b / models);
}

}  // namespace v21
