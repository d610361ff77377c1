// fused_train.h -- forward pass, loss and activation-gradient chain of ONE optimizer step in the fused_fwd.h decomposition
// (r4).  The reference's step is Keras `fit` on a dense stack (emulator.py:369-378, :739-747, :756-764):
// forward, per-sample loss, backward, Adam.  The chain kernels (train_chain.h) carry 32 batch rows per workgroup through
// all layers with the activations in LDS and stream the whole weight set through one CU per 32 rows: 2 FLOP per streamed
// weight byte and row block, 63 us for 16,384 rows (two rounds of 256 workgroups).  Here, for LARGE steps:
//
//   * a workgroup = 4 waves x 32 rows = 128 batch rows; every weight fragment enters LDS once (the LDS-DMA ring of
//     fused_fwd.h) and feeds four waves -- 4x the FLOP per streamed byte;
//   * the activations stay in REGISTERS in the transposed form (features x rows: lane = one batch row, 16 features of a
//     32-feature tile in the accumulator registers), so a tile's result, packed, IS the next layer's MFMA operand;
//   * forward and backward are ONE straight-line "virtual stack" of 2 L - 1 layers: the L forward layers, then the
//     activation-gradient layers L-1 .. 1 with the transposed weights (dX^T = W dZ^T has the shape of a forward layer
//     whose weight matrix is W^T and whose bias is zero).  The layer in the middle ends in the loss instead of an
//     activation: dL/dp = scale w_i (p - y), the row losses summed as 2^-32 fixed point;
//   * ReLU masks are 16 bits per lane and tile, in registers (the whole kernel is unrolled: static indices);
//   * the operands of the weight gradients -- every layer's input H^T and gs dZ^T -- leave in the MFMA-fragment order
//     gemm_dw16_lds_kernel reads (train_chain.h: ChainLayer::ht16 / dzt16): a tile goes through 2.5 KB of LDS per wave as
//     [row][feature] and comes back through the hardware-transposing read (ds_read_b64_tr_b16) as 16-byte fragments
//     lanes: one coalesced 1-KiB store per 16 rows and tile.
// One workgroup per CU (up to 512 registers per wave: the 451-wide operands of the autoencoder alone are 116 of them,
// twice); the weight gradients, the exchange of data-parallel ranks and Adam follow as after a chain launch.
// Same ChainArgs as train_chain_kernel (a.fw = THIS kernel's packed stream: fused_train_pack in api_trainer.hip).
#pragma once
#include "fused_fwd.h"
#include "chain_types.h"   // (ChainArgs, chain_tr_read, kTrainRowsPerWg -- and nothing else of the chain kernels: csrc/jit.hip compiles this file at run time)

namespace v21 {

// one workgroup per CU, one column tile per wave, refill spread over the block being consumed
#ifndef V21_TRAIN_DEPTH
#define V21_TRAIN_DEPTH 2   // LDS read-ahead of the source, in fragments (the compiler sinks the reads to ONE fragment ahead whatever this says: fused_train.h notes, DESIGN K3-fused)
#endif
struct PrecF16t : PrecF16 { static constexpr int CT = 1, BLK = 24, RING = 4, WPS = 1, DEPTH = V21_TRAIN_DEPTH; static constexpr bool SPREAD_DMA = true; };
struct PrecBF16t : PrecBF16 { static constexpr int CT = 1, BLK = 24, RING = 4, WPS = 1, DEPTH = V21_TRAIN_DEPTH; static constexpr bool SPREAD_DMA = true; };
#ifdef V21_T_STAMPS  // diagnostic build (scripts/diag/fused_train_stamps.py): s_memtime of wave 0 of physical workgroup 0 at the start of every virtual layer
#define TSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && a.stamps) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); a.stamps[(i)] = t_; } } while (0)
// ... and of wave 0 of EVERY workgroup at its start and end (slots 64 + 2 b, 65 + 2 b of physical workgroup b < 992), in ticks of
// the constant 100 MHz clock (s_memrealtime: one time base for the whole chip, whatever the shader clock does)
#define WGSTAMP(i) do { if (threadIdx.x == 0 && a.stamps && blockIdx.x < 992) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); a.stamps[64 + 2 * blockIdx.x + (i)] = t_; } } while (0)
#else
#define TSTAMP(i)
#define WGSTAMP(i)
#endif
constexpr int kTrainStagePitch = 40;                                    // halfs per staged row (80 B: 8-byte aligned, rows 20 banks apart)
constexpr int kTrainStageBytes = 4 * 32 * kTrainStagePitch * 2;         // four waves
constexpr int kTrainMaskTiles = 48;                                     // ReLU mask tiles of a stack (16 bits per lane and tile, kept in LDS)
constexpr int kTrainMaskBytes = 4 * kTrainMaskTiles * 64 * 2;
template <class P> constexpr int fused_train_lds() { return fused_lds<P>() + kTrainStageBytes + kTrainMaskBytes; }
// (kTrainRowsPerWg = 128 rows per workgroup: chain_types.h)
#ifndef V21_TRAIN_WPS
#define V21_TRAIN_WPS 1
#endif

// the virtual stack of a real stack TA (L layers, dims[L+1], act[L]): dims d0 .. dL, d(L-1) .. d1
template <class TA> struct TrainArch {
  static constexpr int LR = TA::L;
  static constexpr int L = 2 * LR - 1;
  struct Tab { int d[34]; int a[33]; };
  static constexpr Tab make() {
    Tab t{};
    for (int v = 0; v <= LR; ++v) t.d[v] = TA::dims[v];
    for (int j = 0; j + 1 < LR; ++j) t.d[LR + 1 + j] = TA::dims[LR - 1 - j];
    for (int v = 0; v < L; ++v) t.a[v] = 0;  // (Geo does not look at activations; the kernel asks TA)
    return t;
  }
  static constexpr Tab tab = make();
  static constexpr const int* dims = tab.d;
  static constexpr const int* act = tab.a;
};

// When the pieces of a tile's epilogue run, and how many vector-memory operations a wave has issued before a given step of
// the unrolled stream: the ring's rendezvous waits are COUNTED (s_waitcnt vmcnt(N): all but the N youngest operations
// have completed), so every flush store and target load issued after a block's LDS-DMA must be in N -- a lower bound
// is safe (the wait then covers more than it must), and without them every rendezvous waited for the last flush
// stores to be acknowledged: 65.8 us per launch at 16,384 rows against 36.4 with the flush compiled out.
template <class TA, class P> struct TrainSched {
  using G = Geo<TrainArch<TA>, P>;
  static constexpr int D = P::DEPTH, NCH = 8, LR = TA::L;
  // k-step of tile g + 1 in which tile g's last epilogue chunk runs (the packed tile goes to LDS there: phase W),
  // in which it is read back transposed (R) and stored (S): spread over three k-steps where tile g + 1 has them
  static constexpr int ks_next(int g) { return G::ks_of(G::tile_at(g + 1).l); }
  static constexpr int w_ks(int g) {
    const int lim = G::spread_limit(g);
    if (lim == 0) return 0;
    const int cpk = G::chunks_per_kstep(g, NCH);
    return (NCH + cpk - 1) / cpk - 1;
  }
  static constexpr int r_ks(int g) { return w_ks(g) + 1 < ks_next(g) ? w_ks(g) + 1 : ks_next(g) - 1; }
  static constexpr int s_ks(int g) { return w_ks(g) + 2 < ks_next(g) ? w_ks(g) + 2 : ks_next(g) - 1; }
  // step of the main loop whose compute side issues tile g's two flush stores (g < n_tiles - 1)
  static constexpr int store_step(int g) {
    const typename G::Item n = G::tile_at(g + 1);
    return G::tile_base(n.l, n.nt) + 1 + s_ks(g) + D;
  }
  // target loads of loss tile nt: one 16-byte load per 8-feature group that has a feature inside the output
  static constexpr int target_loads(int nt) {
    int c = 0;
    for (int g = 0; g < 4; ++g) c += 32 * nt + 8 * g < TA::dims[LR] ? 1 : 0;
    return c;
  }
  // LDS-DMA pieces a wave issues on the load side of steps [S0, S1) (refill spread over the block being consumed)
  static constexpr int dma_between(int S0, int S1) {
    int c = 0;
    for (int S = S0; S < S1 && S < G::total(); ++S) {
      const int Bc = S / G::BLK, o = S % G::BLK;
      if (Bc >= 2 && o % G::WAVES == 3 % G::WAVES && o / G::WAVES < G::blk_glds(Bc + G::RING - 2)) ++c;
    }
    return c;
  }
  // vector-memory operations issued after the target loads of loss tile nt (load side of its aux item) and before the
  // first use of those targets (the first epilogue chunk of the tile, on the compute side of the NEXT tile's k-step 0 --
  // or, for the layer's last tile, of the next layer's first tile): LDS-DMA pieces, flush stores, the next tile's targets
  static constexpr int after_targets(int nt) {
    const int s0 = G::tile_base(LR - 1, nt);                  // aux step: loads issued on its load side (after the piece, if any)
    const int g = G::gtile(LR - 1, nt);
    const typename G::Item n = G::tile_at(g + 1);
    const int s1 = G::tile_base(n.l, n.nt) + 1 + D;           // step whose compute side runs chunk 0 (after its own load side)
    int c = dma_between(s0 + 1, s1 + 1);
    for (int q = 0; q + 1 < G::n_tiles(); ++q)
      if (store_step(q) >= s0 && store_step(q) < s1) c += 2;  // (a flush of step s1 itself comes after the chunk)
    if (nt + 1 < G::nt_of(LR - 1)) c += target_loads(nt + 1);
    return c;
  }
  static constexpr int ops_before(int S) {
    int c = 0;
    for (int g = 0; g + 1 < G::n_tiles(); ++g)
      if (store_step(g) < S) c += 2;
    for (int nt = 0; nt < G::nt_of(LR - 1); ++nt)  // the loss layer's target loads, issued with the tile's aux item
      if (G::tile_base(LR - 1, nt) < S) c += target_loads(nt);
    return c;
  }
};

template <class TA, class P>
__global__ void __launch_bounds__(64 * P::WAVES, V21_TRAIN_WPS) fused_train(const ChainArgs a) {
  using VA = TrainArch<TA>;
  using G = Geo<VA, P>;
  using frag = typename P::frag;
  using Item = typename G::Item;
  using SCH = TrainSched<TA, P>;
  constexpr int kBlkFrags = P::BLK, kRing = P::RING;
  constexpr int LR = TA::L, EPI = P::EPI, FPI = P::FPI, IPT = G::IPT;
  constexpr int KSM = G::ks_max();
  constexpr int D = P::DEPTH;
  constexpr bool SPREAD = spread_of<P>::value;
  constexpr int TOTAL = G::total();
  constexpr int NCH = 8;
  static_assert(P::CT == 1 && EPI == 8 && FPI == 16 && IPT == 2, "16-bit operands, one column tile per wave");
  static_assert(TA::act[LR - 1] == 0, "the loss is taken on a linear output layer (emulator.py:44)");
  static_assert([] { for (int l = 0; l < G::L; ++l) if (G::ks_of(l) < 1) return false; return true; }(), "every layer has a k-step");
  // the targets of loss tile G + 2 land in the registers of tile G's (yv[parity]) when the load side reaches its aux
  // fragment, D items ahead of the compute side: tile G's 8 epilogue chunks, one per k-step of tile G + 1, must be done
  static_assert(G::nt_of(LR - 1) == 1 || G::ks_of(LR - 1) - D >= NCH, "loss layer: too few k-steps per tile for the target double buffer");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  // XCD-major row blocks, as train_chain_kernel: XCD x carries a contiguous eighth of the batch, where the slices of
  // gemm_dw16_lds_kernel will look for it
  TSTAMP(0);
  WGSTAMP(0);
  const int nrb = (a.rows + kTrainRowsPerWg - 1) / kTrainRowsPerWg;
  const int rb = (int)(blockIdx.x & 7) * ((nrb + 7) >> 3) + (int)(blockIdx.x >> 3);
  if (rb >= nrb) return;
  const int m0 = rb * kTrainRowsPerWg + wave * 32;  // this wave's first batch row
  const int row = m0 + r;
  const bool ok = row < a.rows;
  const long long src = ok ? (a.idx ? (long long)a.idx[a.first + row] : a.first + row) : 0;
  const float wi = ok ? a.rw[src] : 0.f;
  const float* xr = a.x + src * a.ldx;
  const float* yr = a.y ? a.y + src * a.ldy : xr;  // y == nullptr: the autoencoder, targets = inputs
  const float gsc = a.scale * wi * a.gs;           // dL/dp scaled by the operand scale (a power of two)

  unsigned bufA[KSM][4], bufB[KSM][4];              // operand words of the two layers in flight
  constexpr int NMASK = [] { int n = 0; for (int l = 0; l + 1 < LR; ++l) n += TA::act[l] ? (TA::dims[l + 1] + 31) / 32 : 0; return n; }();
  static_assert(NMASK <= kTrainMaskTiles, "ReLU mask tiles of the stack exceed the LDS area");
  // ReLU masks: 16 bits per lane and tile, written when a forward tile's epilogue completes, read back before the first
  // chunk of the activation-gradient tile they gate.  In LDS, not in registers: the compiler spilled twelve such
  // registers to SCRATCH (long lives, few uses), and every reload -- a vector-memory load it knows of -- came with a
  // wait that ignores the LDS-DMA pieces in flight, i.e. a full drain of the ring: 43 of them, ~29 us of a 66-us launch.
  unsigned short* const mk_lds = reinterpret_cast<unsigned short*>(smem + fused_lds<P>() + kTrainStageBytes) + wave * (kTrainMaskTiles * 64) + lane;
  unsigned mcur = 0u;   // mask bits of the forward tile whose epilogue is running / of the gradient tile being gated
  float lsum = 0.f;

  // ---- a tile's 8 packed words (this lane: one batch row, features 16 i + 8 (w >> 1) + 4 h + 2 (w & 1) + {0, 1} of the
  // tile for word (i, w)) -> fragment order in HBM.  `nfeat`: features of the operand that are written (the ht16 buffers
  // keep their constant row of ones at feature K: api_trainer.hip pre-fills it, nobody writes it)
  unsigned short* stg = reinterpret_cast<unsigned short*>(smem + fused_lds<P>()) + wave * (32 * kTrainStagePitch);
  // (addresses that do not depend on the tile, computed once: with a pointer per tile the compiler hoists a hundred
  //  64-bit address computations to the top of the unrolled kernel and spills ~370 registers)
  unsigned short* const st_w = stg + r * kTrainStagePitch + 4 * h;
  const unsigned short* const st_r = stg + (8 * ((lane >> 4) >> 1) + ((lane & 15) >> 2)) * kTrainStagePitch + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const unsigned fvoff = (unsigned)(((m0 >> 4) * 64 + lane) * 16);  // this wave's 16-row groups, this lane's 16 bytes
  const unsigned tile_bytes = (unsigned)a.BS * 1024u;               // one feature tile of an operand buffer
  chain_s4 fr[4];  // a tile between its transposed read and its stores
  // phase W: this lane's row of the tile (8 packed words) into the wave's staging area as [row][feature]
  auto flush_w = [&](const unsigned (&w0)[4], const unsigned (&w1)[4]) __attribute__((always_inline)) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(st_w + 0) = u32x2{w0[0], w0[1]};
    *reinterpret_cast<u32x2*>(st_w + 8) = u32x2{w0[2], w0[3]};
    *reinterpret_cast<u32x2*>(st_w + 16) = u32x2{w1[0], w1[1]};
    *reinterpret_cast<u32x2*>(st_w + 24) = u32x2{w1[2], w1[3]};
  };
  // phase R: back through the hardware transpose -- lane = (8-row group, feature), four rows per read
  auto flush_r = [&]() __attribute__((always_inline)) {
    fr[0] = chain_tr_read(st_r); fr[1] = chain_tr_read(st_r + 4 * kTrainStagePitch);
    fr[2] = chain_tr_read(st_r + 16 * kTrainStagePitch); fr[3] = chain_tr_read(st_r + 20 * kTrainStagePitch);
  };
  // phase S: two 1-KiB fragments (rows 0-15, 16-31 of the tile) as buffer stores: base = the operand buffer, scalar offset
  // = the tile, vector offset = (16-row group, lane); lanes whose feature lies past the operand get an offset beyond the
  // buffer and are dropped by the range check (exactly two store instructions per tile: TrainSched counts on it)
  auto flush_s = [&](auto tile_, auto nfeat_, void* dst, bool live = true) __attribute__((always_inline)) {
    constexpr int tile = decltype(tile_)::value, nfeat = decltype(nfeat_)::value;
    const chain_s8 v0 = {fr[0][0], fr[0][1], fr[0][2], fr[0][3], fr[1][0], fr[1][1], fr[1][2], fr[1][3]};
    const chain_s8 v1 = {fr[2][0], fr[2][1], fr[2][2], fr[2][3], fr[3][0], fr[3][1], fr[3][2], fr[3][3]};
    constexpr int ntile = (nfeat + 31) / 32;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(ntile * tile_bytes), 0x00020000);
    unsigned vo = live ? fvoff : 0xFFFFF000u;
    if constexpr (32 * tile + 32 > nfeat) vo = (32 * tile + (lane & 31) < nfeat) ? vo : 0xFFFFF000u;
#ifdef V21_T_NOSTORE  // (diagnostic build: every lane's offset beyond the buffer -- the instructions issue, nothing is written)
    vo = 0xFFFFF000u;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rs, vo, tile * tile_bytes, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), rs, vo + 1024u, tile * tile_bytes, 0);
  };
  // (`live` false -- wave-uniform -- : the two stores are issued with every lane's offset beyond the buffer, i.e. dropped by the
  //  range check: TrainSched's operation counts stay as they are)
  auto flush_tile = [&](auto tile_, auto nfeat_, const unsigned (&w0)[4], const unsigned (&w1)[4], void* dst, bool live = true) __attribute__((always_inline)) {
    flush_w(w0, w1);
    flush_r();
    flush_s(tile_, nfeat_, dst, live);
  };

  // ---- ring prologue, FIRST: its pieces land while the input rows below are gathered (the rows come from HBM, 59 MB per launch
  // at 32,768 rows, all workgroups at once: ~18 k cycles).  Everything the input phase issues is then YOUNGER than the
  // prologue's pieces: its loads have returned by the time their data is flushed, its kInputStores buffer stores may
  // still be on their way at the first rendezvous and are added to the waits of the prologue's blocks below.
  static_for<kRing>([&](auto b) __attribute__((always_inline)) { issue_block<G, decltype(b)::value>((const unsigned char*)a.fw, smem, wave, lane); });
  constexpr int kInputStores = 2 * ((TA::dims[0] + 31) / 32);

  // ---- layer-0 operand: the gathered rows, as 16-bit operand words; flushed as the first weight-gradient operand
  {
    constexpr int K0 = TA::dims[0];
    if (a.x16) {
      // the rows as 16-bit elements (ChainStep::x16; rows zero-padded to 32 features: no bound checks): two 8-byte loads per item
      const unsigned short* xh = a.x16 + src * a.ldx16 + 4 * h;
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      static_for<G::ks_of(0)>([&](auto ks_) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_)::value;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(xh + FPI * ks), hi = *reinterpret_cast<const u32x2*>(xh + FPI * ks + 8);
        bufA[ks][0] = ok ? lo[0] : 0u; bufA[ks][1] = ok ? lo[1] : 0u; bufA[ks][2] = ok ? hi[0] : 0u; bufA[ks][3] = ok ? hi[1] : 0u;
      });
    } else
    static_for<G::ks_of(0)>([&](auto ks_) __attribute__((always_inline)) {
      constexpr int ks = decltype(ks_)::value;
      float v[8];
      typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
      if constexpr (FPI * ks + 15 < K0) {  // both halves of the item inside the row: two 16-byte loads (rows are not 16-byte aligned)
        const f32x4_u lo = *reinterpret_cast<const f32x4_u*>(xr + FPI * ks + 4 * h);
        const f32x4_u hi = *reinterpret_cast<const f32x4_u*>(xr + FPI * ks + 8 + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = ok ? lo[e] : 0.f; v[4 + e] = ok ? hi[e] : 0.f; }
      } else {
        static_for<8>([&](auto e_) __attribute__((always_inline)) {
          constexpr int e = decltype(e_)::value;
          constexpr int f0 = FPI * ks + 8 * (e >> 2) + (e & 3), f1 = f0 + 4;
          float t = 0.f;
          if constexpr (f0 < K0) {
            if (ok && (h == 0 || f1 < K0)) t = xr[f0 + 4 * h];
          }
          v[e] = t;
        });
      }
#pragma unroll
      for (int wd = 0; wd < 4; ++wd) bufA[ks][wd] = P::pack2(v[2 * wd], v[2 * wd + 1]);
    });
    // (ht16 of layer 0 == nullptr: the weight-gradient launch gathers this operand from the resident rows itself -- train_chain.h:
    //  DwXRows -- and nothing is written here; the buffer descriptor's base is then the packed stream, never dereferenced)
    const bool xlive = a.lt[0].ht16 != nullptr;
    void* const xdst = xlive ? a.lt[0].ht16 : (void*)a.fw;
    static_for<(K0 + 31) / 32>([&](auto t_) __attribute__((always_inline)) {
      constexpr int t = decltype(t_)::value;
      const unsigned z[4] = {0u, 0u, 0u, 0u};
      if constexpr (2 * t + 1 < G::ks_of(0)) flush_tile(t_, std::integral_constant<int, K0>{}, bufA[2 * t], bufA[2 * t + 1], xdst, xlive);
      else flush_tile(t_, std::integral_constant<int, K0>{}, bufA[2 * t], z, xdst, xlive);
    });
  }

  TSTAMP(1);
#ifdef V21_T_STAMPS
  unsigned long long ring_wait_cycles = 0;
#endif
  frag q[D + 1];
  f32x16 auxb[2];  // (by tile parity: a 1- or 2-k-step tile's successor reads its aux fragment before this one's k-step 0 has run)
  f32x16 acc[2];   // by tile parity: the epilogue of tile G runs under the k-steps of tile G + 1
  f32x4 yv[2][4];  // targets of the loss layer's tile in flight, by tile parity
  int ysh[2] = {0, 0};  // ... of the group the row ends in: how far its load was moved back

  // first mask tile of real layer l's ReLU
  auto mask_base = [](int l) constexpr { int n = 0; for (int i = 0; i < l; ++i) n += TA::act[i] ? (TA::dims[i + 1] + 31) / 32 : 0; return n; };

  // chunk c (accumulator registers 2c, 2c+1) of the epilogue of global tile GT
  auto epilogue_chunk = [&](auto g_, auto c_) __attribute__((always_inline)) {
    constexpr int GT = decltype(g_)::value;
    constexpr int pr = decltype(c_)::value;
    constexpr Item t = G::tile_at(GT);
    constexpr int v = t.l, nt = t.nt;
    constexpr int item = IPT * nt + (2 * pr) / EPI, e0 = (2 * pr) % EPI;
    auto& out = (v & 1) ? bufA : bufB;
    float x0 = acc[GT & 1][2 * pr], x1 = acc[GT & 1][2 * pr + 1];
    unsigned w;
    if constexpr (v < LR - 1) {  // forward hidden layer: activation, mask bits, next operand
      w = P::pack2(x0, x1);
      if constexpr (TA::act[v] != 0) {
        // ReLU and its mask on the PACKED pair, without the condition code: as a 16-bit integer a positive f16 / bf16 is
        // > 0, so max(min(w, 1), 0) is 1 in the halves that pass and 0 in the others.  (A compare writes VCC and the
        // select that reads it waits for it; with one wave per SIMD nothing covers that round trip: ~12 cycles per pair,
        // scripts/diag/mfma_issue_probe.hip -- the single-tile layers around the latent, all epilogue, run 15-25 % faster.)
        // The mask is that of the ROUNDED activation: a pre-activation that underflows to +0 in 16 bits passes neither
        // value nor gradient.
        const i16x2 z = {0, 0}, one = {1, 1};
        const i16x2 wi = __builtin_bit_cast(i16x2, w);
        const unsigned m01 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_elementwise_min(wi, one), z));
        w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(wi, z));
        constexpr int mt = mask_base(v) + nt;
        // bits pr (first value of the pair) and 16 + pr (second) of a 32-bit word; folded to 16 bits for the LDS slot
        if constexpr (pr == 0) mcur = m01;
        else mcur |= m01 << pr;
        if constexpr (pr == NCH - 1) mk_lds[mt * 64] = (unsigned short)(mcur | (mcur >> 8));  // [first values: bits 0-7 | second values: 8-15]
      }
    } else if constexpr (v == LR - 1) {  // the loss: dL/dp = scale w_i (p - y), loss_i = w_i sum (p - y)^2
      constexpr int NO = TA::dims[LR];
      if constexpr (pr == 0) {  // the tile's targets have landed once at most N younger operations are outstanding
        constexpr int N = SCH::after_targets(nt) > 63 ? 63 : SCH::after_targets(nt);
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(yv[GT & 1][0]), "+v"(yv[GT & 1][1]), "+v"(yv[GT & 1][2]), "+v"(yv[GT & 1][3]) : "n"(N));
        static_for<4>([&](auto g_) __attribute__((always_inline)) {
          constexpr int g = decltype(g_)::value;
          constexpr int f0 = 32 * nt + 8 * g;
          if constexpr (f0 < NO && f0 + 8 > NO) {  // the moved-back load: element e of the group is element e + sh of the load
            const f32x4 t = yv[GT & 1][g];
            const int sh = ysh[GT & 1];
            f32x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = sh == 0 ? t[e] : (sh == 1 ? t[(e + 1) & 3] : (sh == 2 ? t[(e + 2) & 3] : t[(e + 3) & 3]));
            yv[GT & 1][g] = u;
          }
        });
      }
      constexpr int fa = 32 * nt + 8 * (pr >> 1) + 2 * (pr & 1);  // + 4 h: this lane's features fa + 4 h, + 1
      const float y0 = yv[GT & 1][pr >> 1][2 * (pr & 1)], y1 = yv[GT & 1][pr >> 1][2 * (pr & 1) + 1];
      const float d0 = (fa + 4 * h < NO) ? x0 - y0 : 0.f, d1 = (fa + 4 * h + 1 < NO) ? x1 - y1 : 0.f;
      lsum += d0 * d0 + d1 * d1;
      w = P::pack2(gsc * d0, gsc * d1);
    } else {  // activation gradient of real layer l: dX masked by the ReLU of the layer below = dZ of that layer
      constexpr int l = 2 * LR - 1 - v;  // v = LR + j  <->  l = LR - 1 - j
      w = P::pack2(x0, x1);
      if constexpr (TA::act[l - 1] != 0) {
        constexpr int mt = mask_base(l - 1) + nt;
        if constexpr (pr == 0) {  // back to bits pr / 16 + pr
          const unsigned m16 = mk_lds[mt * 64];
          mcur = (m16 & 0xFFu) | ((m16 & 0xFF00u) << 8);
        }
        // 0xFFFF in the halves that pass: (0 / 1) * 0xFFFF as a packed 16-bit product -- again no condition code
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const u16x2 sel = __builtin_bit_cast(u16x2, (mcur >> pr) & 0x00010001u), ff = {0xFFFF, 0xFFFF};
        w &= __builtin_bit_cast(unsigned, (u16x2)(sel * ff));
      }
    }
    // the packed pair: word e0 / 2 of item `item` of the next virtual layer's operand (kept even where no layer
    // follows or the item lies past its contraction: the flush reads it from there)
    if constexpr (item < KSM) out[item][e0 / 2] = w;
  };
  // after the 8 chunks: the tile as a weight-gradient operand, in three phases (TrainSched: W, R, S)
  auto tile_flush_w = [&](auto g_) __attribute__((always_inline)) {
    constexpr int GT = decltype(g_)::value;
    constexpr Item t = G::tile_at(GT);
    constexpr int v = t.l, nt = t.nt;
    auto& out = (v & 1) ? bufA : bufB;
    const unsigned z[4] = {0u, 0u, 0u, 0u};
    if constexpr (IPT * nt + 1 < KSM) flush_w(out[IPT * nt], out[IPT * nt + 1]);
    else flush_w(out[IPT * nt], z);
  };
  auto tile_flush_s = [&](auto g_) __attribute__((always_inline)) {
    constexpr int GT = decltype(g_)::value;
    constexpr Item t = G::tile_at(GT);
    constexpr int v = t.l, nt = t.nt;
    constexpr int nfeat = v < LR - 1 ? TA::dims[v + 1] : (v == LR - 1 ? TA::dims[LR] : TA::dims[2 * LR - 1 - v]);
    void* dst;
    if constexpr (v < LR - 1) dst = a.lt[v + 1].ht16;
    else if constexpr (v == LR - 1) dst = a.lt[LR - 1].dzt16;
    else dst = a.lt[2 * LR - 1 - v - 1].dzt16;
    flush_s(std::integral_constant<int, nt>{}, std::integral_constant<int, nfeat>{}, dst);
  };
  auto epilogue_range = [&](auto g_, auto lo_, auto hi_) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
    static_for<(HI > LO ? HI - LO : 0)>([&](auto k) __attribute__((always_inline)) {
      epilogue_chunk(g_, std::integral_constant<int, LO + decltype(k)::value>{});
    });
#ifndef V21_T_NOFLUSH
    if constexpr (HI == NCH && LO < NCH) tile_flush_w(g_);
#endif
  };
  auto operand = [&](auto& buf, int ks) __attribute__((always_inline)) {
    const u32x4 wds = {buf[ks][0], buf[ks][1], buf[ks][2], buf[ks][3]};
    return __builtin_bit_cast(frag, wds);
  };

  static_for<TOTAL + D>([&](auto s_) __attribute__((always_inline)) {
    constexpr int S = decltype(s_)::value;
    // ---- load side: item S
    if constexpr (S < TOTAL) {
      // (the waits count the ring's own LDS-DMA pieces only: the flush stores and target loads issued in between make
      //  the true number of younger operations larger, so every wait is on the safe side)
      if constexpr (S % kBlkFrags == 0 && S < G::padded()) {
        constexpr int B = S / kBlkFrags;
        constexpr int last_issued = (B + kRing - 3 > kRing - 1) ? B + kRing - 3 : kRing - 1;
        constexpr int GA = [] { int n = 0; for (int i = B + 1; i <= last_issued; ++i) n += G::blk_glds(i); return n; }();
        // (the pieces of block B were issued during the consumption of block B - kRing + 2: operations issued after that
        //  whole block are younger than all of them)
        constexpr int S_issue = (B < kRing) ? 0 : (B - kRing + 3) * kBlkFrags;
#ifdef V21_T_NOFLUSH
        constexpr int SA = 0;
#else
        constexpr int SA = SCH::ops_before(S) - SCH::ops_before(S_issue);
#endif
        constexpr int IN = (B < kRing) ? kInputStores : 0;  // (issued after the prologue's pieces, before everything else)
        constexpr int N = (GA + SA + IN) > 63 ? 63 : (GA + SA + IN);
#ifdef V21_T_STAMPS  // cycles this wave spends at the ring's rendezvous (counted wait + barrier), summed over the kernel
        unsigned long long tb0_, tb1_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb0_)::"memory");
        wait_vmcnt_barrier<N>();
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb1_)::"memory");
        ring_wait_cycles += tb1_ - tb0_;
#else
        wait_vmcnt_barrier<N>();
#endif
      }
      if constexpr (SPREAD && S / kBlkFrags >= 2) {
        constexpr int Bc = S / kBlkFrags, o = S % kBlkFrags;
        if constexpr (o % G::WAVES == 3 % G::WAVES)
          issue_piece<G, Bc + kRing - 2, o / G::WAVES>((const unsigned char*)a.fw, smem, wave, lane);
      }
      constexpr Item it = G::item_at(S);
      if constexpr (it.ks >= 0) {
        q[S % (D + 1)] = *(const frag*)frag_ptr<G, S>(smem, lane);
      } else {
        constexpr int GT = G::gtile(it.l, it.nt);
        const unsigned char* aux = frag_ptr<G, S>(smem, 0);
        const f32x4* bp = (const f32x4*)(aux + h * 64);  // bias[32 nt + rho(reg) + 4 h]: the accumulator's initial value
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 t = bp[qd];
          auxb[GT & 1][4 * qd + 0] = t[0]; auxb[GT & 1][4 * qd + 1] = t[1];
          auxb[GT & 1][4 * qd + 2] = t[2]; auxb[GT & 1][4 * qd + 3] = t[3];
        }
        if constexpr (it.l == LR - 1) {
          // the loss layer: this tile's targets, a whole tile of k-steps ahead of their use.  Inline asm: a load the
          // compiler knows of gets a compiler-made wait where it is used -- counted without the LDS-DMA pieces issued in
          // between (inline asm too), i.e. a wait for THOSE to land as well, ~2 k cycles per tile; the wait for these is
          // made by hand before the tile's first epilogue chunk (TrainSched::after_targets).
          constexpr int NO = TA::dims[LR];
          static_for<4>([&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            constexpr int f0 = 32 * it.nt + 8 * g;  // + 4 h
            if constexpr (f0 >= NO) {
              yv[GT & 1][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if constexpr (f0 + 8 <= NO) {
              asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(yv[GT & 1][g]) : "v"(yr + f0 + 4 * h) : "memory");
            } else {
              // the row ends inside this group: load the last four floats that exist and shift (a lane past the end
              // gets values it never uses)
              const int f = f0 + 4 * h, fc = f + 4 <= NO ? f : NO - 4, sh = f - fc;
              f32x4 t;
              asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(yr + fc) : "memory");
              ysh[GT & 1] = sh;
              yv[GT & 1][g] = t;
            }
          });
        }
      }
    }
    // ---- compute side: item S - D
    if constexpr (S >= D) {
      constexpr int C = S - D;
      constexpr Item it = G::item_at(C);
      if constexpr (it.ks >= 0) {
        constexpr int GT = G::gtile(it.l, it.nt);
        constexpr int GP = GT > 0 ? GT - 1 : 0;
        if constexpr (it.ks == 0 && it.nt == 0) TSTAMP(2 + it.l);
        constexpr int CPK = G::chunks_per_kstep(GP, NCH);
        constexpr bool whole_first = (GT > 0) && (G::spread_limit(GP) == 0);
        if constexpr (whole_first && it.ks == 0) {
          epilogue_range(std::integral_constant<int, GP>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, NCH>{});
        }
        auto& in = (it.l & 1) ? bufB : bufA;
        const frag w = q[C % (D + 1)];
        // (ONE accumulation chain per tile.  Measured r4: two chains -- even and odd k-steps, summed after the tile's last
        //  k-step, so that with one wave per SIMD an MFMA need not wait for its predecessor's result -- cost 32 more
        //  registers, 28 of which the compiler spilled to scratch, and every reload of a spilled register drains the ring
        //  (see the ReLU masks above): 66.8 against 57.7 us per launch at 16,384 rows.)
        f32x16 c0;
        if constexpr (it.ks == 0) c0 = auxb[GT & 1];
        else c0 = acc[GT & 1];
        acc[GT & 1] = P::template mfma<false>(w, operand(in, it.ks), c0);
        if constexpr (GT > 0 && !whole_first) {
          constexpr int lo = it.ks * CPK < NCH ? it.ks * CPK : NCH;
          constexpr int hi = (it.ks + 1) * CPK < NCH ? (it.ks + 1) * CPK : NCH;
          epilogue_range(std::integral_constant<int, GP>{}, std::integral_constant<int, lo>{}, std::integral_constant<int, hi>{});
        }
#ifndef V21_T_NOFLUSH
        if constexpr (GT > 0) {
          if constexpr (it.ks == SCH::r_ks(GP)) flush_r();
          if constexpr (it.ks == SCH::s_ks(GP)) tile_flush_s(std::integral_constant<int, GP>{});
        }
#endif
      }
    }
  });
  epilogue_range(std::integral_constant<int, G::n_tiles() - 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, NCH>{});
#ifndef V21_T_NOFLUSH
  flush_r();
  tile_flush_s(std::integral_constant<int, G::n_tiles() - 1>{});
#endif

  TSTAMP(2 + 2 * LR - 1);
  WGSTAMP(1);
#ifdef V21_T_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.stamps) a.stamps[30] = ring_wait_cycles;
#endif
  // ---- batch loss: this wave's rows as 2^-32 fixed point (an integer sum does not depend on the order of arrival)
  float s = lsum * wi;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) atomicAdd(a.loss_acc, (unsigned long long)(long long)llrint((double)s * 4294967296.0));
}

}  // namespace v21
