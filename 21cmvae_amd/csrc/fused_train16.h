// fused_train16.h -- the fused training kernel (fused_train.h) on 16 batch rows per wave: v_mfma_f32_16x16x32_f16 / _bf16.
//
// fused_train.h keeps 32 rows per wave; the 451- and 352-wide operands of the reference's stacks are then 204 registers, a
// wave needs all 512 of its SIMD, and with ONE wave per SIMD every wait of that wave (an LDS read, the condition code, the
// staging round trip of the flush) is a hole in the matrix pipe: its big layers run at 63-94 cycles per MFMA where the
// pipe is busy 32 (scripts/diag/fused_train_stamps.py and the microbenchmarks beside it).
// Half the rows per wave are half the operand registers: two 64-row workgroups per CU, two waves per SIMD that cover each
// other -- what fused_fwd's x2sp form does for predict -- and a step of 16,384 rows fills the chip (256 workgroups) instead
// of half of it.
//
// Geometry.  D = A B with A = a 16-feature tile of W^T over a k-step of 32 input features (one 1-KiB fragment: lane l holds
// output feature l % 16, k-slots 8 (l / 16) + 0..7), B = H^T of the wave's 16 rows (lane = (row l % 16, group g = l / 16),
// 8 halves = 4 registers per k-step), D: lane (row, g) holds output features 4 g + 0..3 of the tile in 4 accumulators.
// Tiles 2 s and 2 s + 1 make operand item s of the next layer with no cross-lane move because every layer's fragments
// are packed in the order  kmap(s, g, j) = 32 s + 16 (j / 4) + 4 g + j % 4  (train_kernels.h: pack_stream_kernel fmt16,
// adam_repack_element), and the input rows are loaded in that order.  The bias is 16 floats per tile (one ds_read_b128
// at offset 16 g: the accumulators' initial value).  The weight-gradient operands leave in gemm_dw16_lds_kernel's format
// as in fused_train.h: a tile PAIR of 16 rows is one 1-KiB fragment ([16 rows][32 features] staged with two 8-byte
// writes per lane, two transposed reads, one 16-byte store).  ReLU masks: 4 bits per lane and tile, in registers.
// Forward + loss + activation gradients as ONE virtual stack (TrainArch), the ring / rendezvous / spread refill of
// fused_fwd.h through a Geo of the same interface.
#pragma once
#include "fused_train.h"

namespace v21 {

constexpr int kTrain16RowsPerWg = 64;
// LDS read-ahead in fragments.  An MFMA of this shape is 16 cycles of the pipe: two fragments in flight cover 32 cycles of
// the ~70 an LDS read takes under load (scripts/diag/lds_read_probe.hip) -- with D = 2 the big layers ran at 44 cycles per
// MFMA with one wave per SIMD and at 98 per wave with two (first stamps, r4)
#ifndef V21_TRAIN16_DEPTH
#define V21_TRAIN16_DEPTH 4
#endif
struct PrecF16t16 : PrecF16 {
  static constexpr int CT = 1, BLK = 16, RING = 4, WPS = 2, DEPTH = V21_TRAIN16_DEPTH;
  static constexpr bool SPREAD_DMA = true;
  static __device__ __forceinline__ f32x4 mfma16(frag w, frag x, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0); }
};
struct PrecBF16t16 : PrecBF16 {
  static constexpr int CT = 1, BLK = 16, RING = 4, WPS = 2, DEPTH = V21_TRAIN16_DEPTH;
  static constexpr bool SPREAD_DMA = true;
  static __device__ __forceinline__ f32x4 mfma16(frag w, frag x, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0); }
};
constexpr int kTrain16StageBytes = 4 * 16 * kTrainStagePitch * 2;  // four waves x 16 rows
constexpr int kTrain16MaskPairs = 44;                              // ReLU mask bytes of a stack: one per lane and tile PAIR, in LDS
constexpr int kTrain16MaskBytes = 4 * kTrain16MaskPairs * 64;
template <class P> constexpr int fused_train16_lds() { return fused_lds<P>() + kTrain16StageBytes + kTrain16MaskBytes; }

// Geo (fused_fwd.h) for 16-feature tiles and 32-feature k-steps: the members issue_block / issue_piece / frag_ptr and the
// main loop use, with the same meaning
template <class Arch, class P> struct Geo16 {
  static constexpr int L = Arch::L;
  static constexpr int BLK = P::BLK, RING = P::RING, WAVES = P::WAVES;
  static constexpr int dim(int i) { return Arch::dims[i]; }
  static constexpr int ks_of(int l) { return (dim(l) + 31) / 32; }
  static constexpr int nt_of(int l) { return (dim(l + 1) + 15) / 16; }
  static constexpr int tile_base(int l, int nt) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += nt_of(i) * (ks_of(i) + 1);
    return f + nt * (ks_of(l) + 1);
  }
  static constexpr int total() { return tile_base(L, 0); }
  static constexpr int padded() { return (total() + 7) / 8 * 8; }
  static constexpr int n_blocks() { return (padded() + BLK - 1) / BLK; }
  static constexpr int blk_glds(int b) {
    if (b < 0 || b >= n_blocks()) return 0;
    const int rem = padded() - b * BLK;
    return (rem < BLK ? rem : BLK) / WAVES;
  }
  static constexpr int ks_max() {
    int m = 0;
    for (int l = 0; l < L; ++l) m = ks_of(l) > m ? ks_of(l) : m;
    return m;
  }
  struct Item { int l, nt, ks; };  // ks == -1: the tile's aux fragment
  static constexpr Item item_at(int F) {
    for (int l = 0; l < L; ++l) {
      const int tl = ks_of(l) + 1, cnt = nt_of(l) * tl;
      if (F < cnt) return Item{l, F / tl, F % tl - 1};
      F -= cnt;
    }
    return Item{-1, -1, -1};
  }
  static constexpr int gtile(int l, int nt) {
    int g = 0;
    for (int i = 0; i < l; ++i) g += nt_of(i);
    return g + nt;
  }
  static constexpr Item tile_at(int g) {
    for (int l = 0; l < L; ++l) {
      if (g < nt_of(l)) return Item{l, g, -1};
      g -= nt_of(l);
    }
    return Item{-1, -1, -1};
  }
  static constexpr int n_tiles() { return gtile(L, 0); }
  // may tile G's epilogue run AFTER the first MFMA of tile G + 1?  Not when that MFMA already consumes it: the last tile
  // of a layer of one or two tiles (its pair is operand item 0 of the next layer), and the last tile of all
  static constexpr bool epilogue_first(int G) {
    const Item t = tile_at(G), n = tile_at(G + 1);
    if (n.l < 0) return true;
    if (n.l == t.l) return false;
    return (nt_of(t.l) - 1) / 2 == 0;
  }
};

// vector-memory operations of a wave, by step of the unrolled stream (the counted waits of the ring's rendezvous and of the
// loss layer's targets: fused_train.h, TrainSched).  A tile's epilogue runs on the compute side of the step that holds
// k-step 0 of the NEXT tile; a tile pair's flush (ONE store) follows the epilogue of its second tile (or of a layer's
// last tile); the loss layer's targets (one 16-byte load per tile) are issued on the load side of the tile's aux item.
template <class TA, class P> struct TrainSched16 {
  using G = Geo16<TrainArch<TA>, P>;
  static constexpr int D = P::DEPTH, LR = TA::L;
  static constexpr bool flushes(int g) {  // does tile g's epilogue end in a flush?
    const typename G::Item t = G::tile_at(g);
    return (t.nt & 1) == 1 || t.nt == G::nt_of(t.l) - 1;
  }
  // step whose compute side runs tile g's epilogue (g < n_tiles - 1)
  static constexpr int epi_step(int g) {
    const typename G::Item n = G::tile_at(g + 1);
    return G::tile_base(n.l, n.nt) + 1 + D;
  }
  static constexpr int dma_between(int S0, int S1) {
    int c = 0;
    for (int S = S0; S < S1 && S < G::total(); ++S) {
      const int Bc = S / G::BLK, o = S % G::BLK;
      if (Bc >= 2 && o % G::WAVES == 3 % G::WAVES && o / G::WAVES < G::blk_glds(Bc + G::RING - 2)) ++c;
    }
    return c;
  }
  // operations issued after the target load of loss tile nt and before its use (the tile's epilogue)
  static constexpr int after_targets(int nt) {
    const int s0 = G::tile_base(LR - 1, nt);
    const int g = G::gtile(LR - 1, nt);
    const int s1 = epi_step(g);
    int c = dma_between(s0 + 1, s1 + 1);
    for (int q = 0; q + 1 < G::n_tiles(); ++q)
      if (flushes(q) && epi_step(q) >= s0 && epi_step(q) < s1) c += 1;
    if (nt + 1 < G::nt_of(LR - 1) && G::tile_base(LR - 1, nt + 1) <= s1) c += 1;  // the next tile's target load
    return c;
  }
  static constexpr int ops_before(int S) {
    int c = 0;
    for (int g = 0; g + 1 < G::n_tiles(); ++g)
      if (flushes(g) && epi_step(g) < S) c += 1;
    for (int nt = 0; nt < G::nt_of(LR - 1); ++nt)
      if (G::tile_base(LR - 1, nt) < S) c += 1;
    return c;
  }
};

template <class TA, class P>
__global__ void __launch_bounds__(64 * P::WAVES, P::WPS) fused_train16(const ChainArgs a) {
  using VA = TrainArch<TA>;
  using G = Geo16<VA, P>;
  using frag = typename P::frag;
  using Item = typename G::Item;
  using SCH = TrainSched16<TA, P>;
  constexpr int kBlkFrags = P::BLK, kRing = P::RING;
  constexpr int LR = TA::L;
  constexpr int KSM = G::ks_max();
  constexpr int D = P::DEPTH;
  constexpr int TOTAL = G::total();
  static_assert(TA::act[LR - 1] == 0, "the loss is taken on a linear output layer (emulator.py:44)");
  // the aux fragment of tile G + 4 must not land in tile G's slot before tile G's first MFMA has taken it: tiles are >= 2
  // items, the load side runs D items ahead of the compute side
  constexpr int NAUX = 4;
  static_assert(D >= 1 && D <= 6 && D <= kBlkFrags, "read-ahead within the four-slot aux ring");
  // the target of loss tile G + 2 lands in the registers of tile G's (yv[parity]) when the load side reaches its aux item,
  // D items ahead of the compute side, i.e. during tile G + 1: tile G's epilogue runs at that tile's k-step 0, before
  static_assert(G::nt_of(LR - 1) == 1 || G::ks_of(LR - 1) > D, "loss layer: too few k-steps per tile for the target double buffer");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g4 = lane >> 4;
  // XCD-major row blocks, as fused_train.h
  TSTAMP(0);
  WGSTAMP(0);
  const int nrb = (a.rows + kTrain16RowsPerWg - 1) / kTrain16RowsPerWg;
  const int rb = (int)(blockIdx.x & 7) * ((nrb + 7) >> 3) + (int)(blockIdx.x >> 3);
  if (rb >= nrb) return;
  const int m0 = rb * kTrain16RowsPerWg + wave * 16;  // this wave's first batch row
  const int row = m0 + r;
  const bool ok = row < a.rows;
  const long long src = ok ? (a.idx ? (long long)a.idx[a.first + row] : a.first + row) : 0;
  const float wi = ok ? a.rw[src] : 0.f;
  const float* xr = a.x + src * a.ldx;
  const float* yr = a.y ? a.y + src * a.ldy : xr;
  const float gsc = a.scale * wi * a.gs;
  const float* const ybase = yr + 4 * g4;
  static_assert(4 * TA::dims[LR] < 4096, "the target loads' immediate offset");

  unsigned bufA[KSM][4], bufB[KSM][4];  // operand words of the two layers in flight
  // ReLU masks: 4 bits per lane and tile; a tile PAIR's byte goes to LDS when the pair is complete and comes back before the
  // pair's first activation-gradient tile (in registers the compiler spilled them to scratch, and a reload from scratch
  // is a vector-memory load whose wait drains the ring: fused_train.h).  Pair index = mask_base(layer) + nt / 2.
  constexpr int NMASK = [] { int n = 0; for (int l = 0; l + 1 < LR; ++l) n += TA::act[l] ? ((TA::dims[l + 1] + 15) / 16 + 1) / 2 : 0; return n; }();
  static_assert(NMASK <= kTrain16MaskPairs, "ReLU mask pairs of the stack exceed the LDS area");
  unsigned char* const mk_lds = smem + fused_lds<P>() + kTrain16StageBytes + wave * (kTrain16MaskPairs * 64) + lane;
  unsigned mcur = 0u;  // bits 0-1 / 16-17: the pair's first tile (values 0-1 / 2-3 ... see the epilogue), 2-3 / 18-19: its second
  auto mask_base = [](int l) constexpr { int n = 0; for (int i = 0; i < l; ++i) n += TA::act[i] ? ((TA::dims[i + 1] + 15) / 16 + 1) / 2 : 0; return n; };
  float lsum = 0.f;

  // ---- an operand item (this lane: one batch row, features 4 g + 0..3 and 16 + 4 g + 0..3 of the 32) -> fragment order in
  // HBM: [row][feature] in the wave's staging area, back through the hardware transpose, one 16-byte store
  unsigned short* stg = reinterpret_cast<unsigned short*>(smem + fused_lds<P>()) + wave * (16 * kTrainStagePitch);
  unsigned short* const st_w = stg + r * kTrainStagePitch + 4 * g4;
  const unsigned short* const st_r = stg + (8 * ((lane >> 4) >> 1) + ((lane & 15) >> 2)) * kTrainStagePitch + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const unsigned fvoff = (unsigned)(((m0 >> 4) * 64 + lane) * 16);  // this wave's 16-row group, this lane's 16 bytes
  const unsigned tile_bytes = (unsigned)a.BS * 1024u;               // one 32-feature tile of an operand buffer
  // (`live` false -- wave-uniform -- : the store is issued with every lane's offset beyond the buffer, i.e. dropped by the range
  //  check: the counted waits of the stream keep their operation counts.  The staging round trip stays -- a branch around it
  //  made the compiler duplicate and interleave the fifteen items of the input layer; it is two LDS writes and two reads)
  auto flush_item = [&](auto tile_, auto nfeat_, const unsigned (&w)[4], void* dst, bool live = true) __attribute__((always_inline)) {
    constexpr int tile = decltype(tile_)::value, nfeat = decltype(nfeat_)::value;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(st_w + 0) = u32x2{w[0], w[1]};
    *reinterpret_cast<u32x2*>(st_w + 16) = u32x2{w[2], w[3]};
    const chain_s4 f0 = chain_tr_read(st_r), f1 = chain_tr_read(st_r + 4 * kTrainStagePitch);
    const chain_s8 v0 = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
    constexpr int ntile = (nfeat + 31) / 32;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(ntile * tile_bytes), 0x00020000);
    unsigned vo = live ? fvoff : 0xFFFFF000u;
    if constexpr (32 * tile + 32 > nfeat) vo = (32 * tile + (lane & 31) < nfeat) ? vo : 0xFFFFF000u;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rs, vo, tile * tile_bytes, 0);
  };

  // ---- ring prologue first (fused_train.h): its pieces land while the input rows are gathered
  static_for<kRing>([&](auto b) __attribute__((always_inline)) { issue_block<G, decltype(b)::value>((const unsigned char*)a.fw, smem, wave, lane); });
  constexpr int kInputStores = (TA::dims[0] + 31) / 32;

  // ---- layer-0 operand: the gathered rows as operand items, flushed as the first weight-gradient operand
  {
    constexpr int K0 = TA::dims[0];
    if (a.x16) {
      // the rows as 16-bit elements (ChainStep::x16; rows zero-padded to 32 features: no bound checks): two 8-byte loads per item
      const unsigned short* xh = a.x16 + src * a.ldx16 + 4 * g4;
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      static_for<G::ks_of(0)>([&](auto ks_) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_)::value;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(xh + 32 * ks), hi = *reinterpret_cast<const u32x2*>(xh + 32 * ks + 16);
        bufA[ks][0] = ok ? lo[0] : 0u; bufA[ks][1] = ok ? lo[1] : 0u; bufA[ks][2] = ok ? hi[0] : 0u; bufA[ks][3] = ok ? hi[1] : 0u;
      });
    } else
    static_for<G::ks_of(0)>([&](auto ks_) __attribute__((always_inline)) {
      constexpr int ks = decltype(ks_)::value;
      float v[8];
      static_for<2>([&](auto half_) __attribute__((always_inline)) {
        constexpr int hf = decltype(half_)::value;
        constexpr int f0 = 32 * ks + 16 * hf;  // + 4 g + 0..3
        typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
        if constexpr (f0 + 16 <= K0) {
          const f32x4_u t = *reinterpret_cast<const f32x4_u*>(xr + f0 + 4 * g4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * hf + e] = ok ? t[e] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int f = f0 + 4 * g4 + e;
            v[4 * hf + e] = (ok && f < K0) ? xr[f < K0 ? f : 0] : 0.f;
          }
        }
      });
#pragma unroll
      for (int wd = 0; wd < 4; ++wd) bufA[ks][wd] = P::pack2(v[2 * wd], v[2 * wd + 1]);
    });
    // (ht16 of layer 0 == nullptr: the weight-gradient launch gathers this operand from the resident rows itself -- train_chain.h:
    //  DwXRows -- and nothing is written here; the buffer descriptor's base is then the packed stream, never dereferenced)
    const bool xlive = a.lt[0].ht16 != nullptr;
    void* const xdst = xlive ? a.lt[0].ht16 : (void*)a.fw;
    static_for<G::ks_of(0)>([&](auto t_) __attribute__((always_inline)) {
      flush_item(t_, std::integral_constant<int, K0>{}, bufA[decltype(t_)::value], xdst, xlive);
    });
  }

  TSTAMP(1);
  frag q[D + 1];
  f32x4 auxb[NAUX];  // by tile number mod 4 (the load side is up to three short tiles ahead)
  f32x4 acc[2];   // by tile parity: the epilogue of tile G runs beside the first k-step of tile G + 1
  f32x4 yv[2];    // targets of the loss layer's tile in flight, by tile parity
  int ysh[2] = {0, 0};

  // the epilogue of global tile GT
  auto epilogue = [&](auto g_) __attribute__((always_inline)) {
    constexpr int GT = decltype(g_)::value;
    constexpr Item t = G::tile_at(GT);
    constexpr int v = t.l, nt = t.nt;
    constexpr int NT = G::nt_of(v);
    constexpr int item = nt / 2, w0i = 2 * (nt & 1);
    auto& out = (v & 1) ? bufA : bufB;
    float x0 = acc[GT & 1][0], x1 = acc[GT & 1][1], x2 = acc[GT & 1][2], x3 = acc[GT & 1][3];
    unsigned wa, wb;
    const i16x2 z = {0, 0};
    if constexpr (v < LR - 1) {  // forward hidden layer: activation, mask bits, next operand
      wa = P::pack2(x0, x1); wb = P::pack2(x2, x3);
      if constexpr (TA::act[v] != 0) {
        // (ReLU and its mask on the packed pair without the condition code: fused_train.h)
        // (inline asm: left to itself the compiler turns max(min(w, 1), 0) back into two compares and selects per word)
        unsigned ma, mb;
        asm("v_pk_min_i16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_max_i16 %0, %0, 0 op_sel_hi:[1,0]" : "=&v"(ma) : "v"(wa));
        asm("v_pk_min_i16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_max_i16 %0, %0, 0 op_sel_hi:[1,0]" : "=&v"(mb) : "v"(wb));
        wa = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, wa), z));
        wb = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, wb), z));
        // bits {0, 16} of ma: values 0, 1 of the tile; of mb: values 2, 3
        if constexpr ((nt & 1) == 0) mcur = ma | (mb << 1);
        else mcur |= (ma | (mb << 1)) << 2;
        if constexpr ((nt & 1) == 1 || nt == NT - 1) mk_lds[(mask_base(v) + nt / 2) * 64] = (unsigned char)(mcur | (mcur >> 12));
      }
    } else if constexpr (v == LR - 1) {  // the loss: dL/dp = scale w_i (p - y), loss_i = w_i sum (p - y)^2
      constexpr int NO = TA::dims[LR];
      constexpr int N = SCH::after_targets(nt) > 63 ? 63 : SCH::after_targets(nt);
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(yv[GT & 1]) : "n"(N));
      f32x4 y = yv[GT & 1];
      if constexpr (16 * nt + 16 > NO) {  // the row ends inside this tile: element e of a group is element e + sh of its (moved-back) load
        const int sh = ysh[GT & 1];
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = sh == 0 ? y[e] : (sh == 1 ? y[(e + 1) & 3] : (sh == 2 ? y[(e + 2) & 3] : y[(e + 3) & 3]));
        y = u;
      }
      const int f = 16 * nt + 4 * g4;
      const float d0 = (f + 0 < NO) ? x0 - y[0] : 0.f, d1 = (f + 1 < NO) ? x1 - y[1] : 0.f;
      const float d2 = (f + 2 < NO) ? x2 - y[2] : 0.f, d3 = (f + 3 < NO) ? x3 - y[3] : 0.f;
      lsum += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      wa = P::pack2(gsc * d0, gsc * d1); wb = P::pack2(gsc * d2, gsc * d3);
    } else {  // activation gradient of real layer l: dX masked by the ReLU of the layer below = dZ of that layer
      constexpr int l = 2 * LR - 1 - v;
      wa = P::pack2(x0, x1); wb = P::pack2(x2, x3);
      if constexpr (TA::act[l - 1] != 0) {
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const u16x2 ff = {0xFFFF, 0xFFFF};
        if constexpr ((nt & 1) == 0) {
          const unsigned b = mk_lds[(mask_base(l - 1) + nt / 2) * 64];
          mcur = (b & 0x0Fu) | ((b & 0xF0u) << 12);
        }
        const unsigned m = mcur >> (2 * (nt & 1));
        wa &= __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, m & 0x00010001u) * ff));
        wb &= __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, (m >> 1) & 0x00010001u) * ff));
      }
    }
    if constexpr (item < KSM) {
      out[item][w0i] = wa; out[item][w0i + 1] = wb;
      if constexpr ((nt & 1) == 0 && nt == NT - 1) { out[item][2] = 0u; out[item][3] = 0u; }  // the pair's missing half
    }
    // the pair as a weight-gradient operand
    if constexpr (SCH::flushes(GT)) {
      constexpr int nfeat = v < LR - 1 ? TA::dims[v + 1] : (v == LR - 1 ? TA::dims[LR] : TA::dims[2 * LR - 1 - v]);
      void* dst;
      if constexpr (v < LR - 1) dst = a.lt[v + 1].ht16;
      else if constexpr (v == LR - 1) dst = a.lt[LR - 1].dzt16;
      else dst = a.lt[2 * LR - 1 - v - 1].dzt16;
      if constexpr (item < KSM) flush_item(std::integral_constant<int, item>{}, std::integral_constant<int, nfeat>{}, out[item], dst);
    }
  };
  auto operand = [&](auto& buf, int ks) __attribute__((always_inline)) {
    const u32x4 wds = {buf[ks][0], buf[ks][1], buf[ks][2], buf[ks][3]};
    return __builtin_bit_cast(frag, wds);
  };

  static_for<TOTAL + D>([&](auto s_) __attribute__((always_inline)) {
    constexpr int S = decltype(s_)::value;
    // ---- load side: item S
    if constexpr (S < TOTAL) {
      if constexpr (S % kBlkFrags == 0 && S < G::padded()) {
        constexpr int B = S / kBlkFrags;
        constexpr int last_issued = (B + kRing - 3 > kRing - 1) ? B + kRing - 3 : kRing - 1;
        constexpr int GA = [] { int n = 0; for (int i = B + 1; i <= last_issued; ++i) n += G::blk_glds(i); return n; }();
        constexpr int S_issue = (B < kRing) ? 0 : (B - kRing + 3) * kBlkFrags;
        constexpr int SA = SCH::ops_before(S) - SCH::ops_before(S_issue);
        constexpr int IN = (B < kRing) ? kInputStores : 0;
        constexpr int N = (GA + SA + IN) > 63 ? 63 : (GA + SA + IN);
        wait_vmcnt_barrier<N>();
      }
      if constexpr (S / kBlkFrags >= 2) {
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
        auxb[GT % NAUX] = *(const f32x4*)(aux + g4 * 16);  // bias[16 nt + 4 g + 0..3]: the accumulators' initial value
        if constexpr (it.l == LR - 1) {
          // the loss layer: this tile's targets, a whole tile of k-steps ahead of their use (inline asm + a hand-counted wait
          // before the tile's epilogue: fused_train.h says why)
          constexpr int NO = TA::dims[LR];
          constexpr int f0 = 16 * it.nt;  // + 4 g
          if constexpr (f0 + 16 <= NO) {
            // (one base address per lane, the tile as the instruction's immediate offset: with an address per tile the compiler
            //  computed them all ahead and spilled them)
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(yv[GT & 1]) : "v"(ybase), "n"(4 * f0) : "memory");
          } else {
            // the row ends inside this tile: load the last four floats that exist and shift (a lane whose group lies past
            // the end gets values it never uses)
            const int f = f0 + 4 * g4, fc = f + 4 <= NO ? f : NO - 4, sh = f - fc;
            f32x4 t;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(yr + fc) : "memory");
            ysh[GT & 1] = sh;
            yv[GT & 1] = t;
          }
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
        constexpr bool first = (GT > 0) && G::epilogue_first(GP);
        if constexpr (it.ks == 0 && it.nt == 0) TSTAMP(2 + it.l);
        if constexpr (first && it.ks == 0) epilogue(std::integral_constant<int, GP>{});
        auto& in = (it.l & 1) ? bufB : bufA;
        const frag w = q[C % (D + 1)];
        f32x4 c0;
        if constexpr (it.ks == 0) c0 = auxb[GT % NAUX];
        else c0 = acc[GT & 1];
        acc[GT & 1] = P::mfma16(w, operand(in, it.ks), c0);
        if constexpr (GT > 0 && !first && it.ks == 0) epilogue(std::integral_constant<int, GP>{});
      }
    }
  });
  epilogue(std::integral_constant<int, G::n_tiles() - 1>{});
  TSTAMP(2 + 2 * LR - 1);
  WGSTAMP(1);

  // ---- batch loss: this wave's rows as 2^-32 fixed point (an integer sum does not depend on the order of arrival)
  float s = lsum * wi;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) atomicAdd(a.loss_acc, (unsigned long long)(long long)llrint((double)s * 4294967296.0));
}

}  // namespace v21
