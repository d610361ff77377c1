// fused_fwd16.h -- the fused dense stack (see fused_fwd.h) on the 16x16x32 MFMA shape.
//
// Same algorithm as fused_fwd<>: transposed activations resident in registers, weights
// streamed once through an LDS ring by LDS-DMA, everything compile-time unrolled.  What
// changes is the matrix instruction: v_mfma_f32_16x16x32_{f16,bf16} instead of 32x32x16.
// Under MFMA-dense load the chip is power-limited and holds a higher clock on the
// 16x16x32 shape at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7), and
// the 16-wide output tile halves the accumulator registers per tile.
//
// Layout facts used (cdna_hip_programming.md section 3):
//   A/B operand: lane l holds k = 8*(l>>4) + e, e = 0..7, of row/column l & 15;
//   C/D:         lane l holds column l & 15, rows 4*(l>>4) + r, r = 0..3.
// A wave owns CB = 2 column blocks of 16 signals.  Output tile nt (16 features) of a layer
// leaves features 16nt + 4g + r in register r of lane group g = l>>4; two consecutive tiles
// (2u, 2u+1) packed to f16 are exactly the B fragment of k-step u of the next layer when the
// weight fragment is packed with the same k-permutation:
//   element e of lane group g  <->  feature 32u + 16(e>>2) + 4g + (e&3).
// All layers, including the last, use the transposed orientation; the output is stored as
// one 16-byte store per (tile, column block): 4 consecutive bins of one signal per lane.
#pragma once
#include "fused_fwd.h"

namespace v21 {

struct PrecF16s16 {
  using frag = f16x8;
  static constexpr int CB = 2, BLK = 16, RING = 5, WPS = 2, DEPTH = 2;
  static constexpr int CT = 1;  // 32 signals per wave (launcher geometry)
  static __device__ __forceinline__ unsigned pack2(float a, float b) { return PrecF16::pack2(a, b); }
  static __device__ __forceinline__ f32x4 mfma(frag w, frag x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0);
  }
};
struct PrecBF16s16 {
  using frag = bf16x8;
  static constexpr int CB = 2, BLK = 16, RING = 5, WPS = 2, DEPTH = 2;
  static constexpr int CT = 1;
  static __device__ __forceinline__ unsigned pack2(float a, float b) { return PrecBF16::pack2(a, b); }
  static __device__ __forceinline__ f32x4 mfma(frag w, frag x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
  }
};

// ---- compile-time geometry: n-tiles of 16 outputs, k-steps of 32 features ---------------
template <class Arch, class P> struct Geo16 {
  static constexpr int L = Arch::L;
  static constexpr int BLK = P::BLK, RING = P::RING, CB = P::CB, WAVES = kWaves;
  static constexpr int NCH = CB * 2;  // epilogue chunks per tile: (column block, register pair)
  static constexpr int dim(int i) { return Arch::dims[i]; }
  static constexpr int act(int l) { return Arch::act[l]; }
  static constexpr int ks_of(int l) { return (dim(l) + 31) / 32; }
  static constexpr int nt_of(int l) { return (dim(l + 1) + 15) / 16; }
  static constexpr int tile_base(int l, int nt) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += nt_of(i) * (ks_of(i) + 1);
    return f + nt * (ks_of(l) + 1);
  }
  static constexpr int total() { return tile_base(L, 0); }
  static constexpr int padded() { return (total() + 3) / 4 * 4; }
  static constexpr int n_blocks() { return (padded() + BLK - 1) / BLK; }
  static constexpr int blk_glds(int b) {
    if (b < 0 || b >= n_blocks()) return 0;
    const int rem = padded() - b * BLK;
    return (rem < BLK ? rem : BLK) / kWaves;
  }
  static constexpr int ks_max() {
    int m = 0;
    for (int l = 0; l < L; ++l) m = ks_of(l) > m ? ks_of(l) : m;
    return m;
  }
  struct Item { int l, nt, ks; };
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
  // k-steps of tile G+1 over which tile G's epilogue may be spread (0: run it whole first)
  static constexpr int spread_limit(int G) {
    const Item t = tile_at(G), n = tile_at(G + 1);
    if (n.l < 0) return 0;
    if (n.l == t.l) return ks_of(n.l);
    const int first_use = (nt_of(t.l) - 1) / 2;  // k-step of the next layer that reads this tile
    return first_use < ks_of(n.l) ? first_use : ks_of(n.l);
  }
  static constexpr int chunks_per_kstep(int G) {
    const int lim = spread_limit(G);
    return lim <= 0 ? NCH : (NCH + lim - 1) / lim;
  }
  // output tile nt: all 16 bins valid -> one 16-byte store per column block, else 4 dwords
  static constexpr bool full_out_tile(int nt) { return 16 * nt + 16 <= dim(L); }
  static constexpr int chunk_stores(int nt, int c) { return (c % 2 == 0) ? (full_out_tile(nt) ? 1 : 4) : 0; }
  // output-layer store instructions issued by one wave in steps < S (see fused_fwd.h)
  static constexpr int stores_before_step(int S, int D) {
    int s = 0;
    const int l = L - 1;
    for (int nt = 0; nt + 1 < nt_of(l); ++nt) {
      const int cpk = chunks_per_kstep(gtile(l, nt));
      const int nb = tile_base(l, nt + 1);
      for (int c = 0; c < NCH; ++c)
        if (nb + 1 + c / cpk + D < S) s += chunk_stores(nt, c);
    }
    return s;
  }
};

template <class G, int D, int S>
__device__ __forceinline__ void ring_boundary16(const unsigned char* gstream, unsigned char* smem, int wave, int lane) {
  constexpr int kBlkFrags = G::BLK, kRing = G::RING;
  if constexpr (S % kBlkFrags == 0 && S < G::padded()) {
    constexpr int B = S / kBlkFrags;
    constexpr int last_issued = (B + kRing - 3 > kRing - 1) ? B + kRing - 3 : kRing - 1;
    constexpr int GA = [] {
      int s = 0;
      for (int i = B + 1; i <= last_issued; ++i) s += G::blk_glds(i);
      return s;
    }();
    constexpr int S_issue = (B < kRing) ? 0 : (B - kRing + 2) * kBlkFrags;
    constexpr int SA = G::stores_before_step(S, D) - G::stores_before_step(S_issue, D);
    constexpr int N = (GA + SA) > 63 ? 63 : (GA + SA);
    wait_vmcnt_barrier<N>();
    if constexpr (B >= 2) issue_block<G, B + kRing - 2>(gstream, smem, wave, lane);
  }
}

// grid.x = ceil(n_rows / 128); block = 256 threads; dynamic LDS RING*BLK KiB; 2 workgroups/CU
template <class Arch, class P>
__global__ void __launch_bounds__(256, P::WPS) fused_fwd16(const FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = Geo16<Arch, P>;
  using frag = typename P::frag;
  using Item = typename G::Item;
  constexpr int L = G::L, CB = P::CB, KSM = G::ks_max(), D = P::DEPTH, TOTAL = G::total(), NOUT = G::dim(L);
  constexpr int NCH = G::NCH;
  static_assert(G::act(L - 1) == 0, "output layer must be linear");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const long long wg_row0 = (long long)blockIdx.x * (kWaves * CB * 16);
  const long long row0 = wg_row0 + wave * (CB * 16);

  unsigned bufA[CB][KSM][4], bufB[CB][KSM][4];  // operand words of the two layers in flight

  // ---- layer-0 operand: element e of lane group g <-> feature 32u + 16(e>>2) + 4g + (e&3)
  {
    constexpr int K0 = G::dim(0);
    static_for<CB>([&](auto cb_) __attribute__((always_inline)) {
      constexpr int cb = decltype(cb_)::value;
      const long long row = row0 + cb * 16 + j;
      const bool ok = row < a.n_rows;
      const float* xr = a.x + (ok ? row : 0) * a.ldx;
      static_for<G::ks_of(0)>([&](auto u_) __attribute__((always_inline)) {
        constexpr int u = decltype(u_)::value;
        float v[8];
        static_for<8>([&](auto e_) __attribute__((always_inline)) {
          constexpr int e = decltype(e_)::value;
          constexpr int fb = 32 * u + 16 * (e >> 2) + (e & 3);  // + 4g
          float t = 0.f;
          if constexpr (fb < K0) {
            const int f = fb + 4 * g;
            if (ok && f < K0) {
              t = xr[f];
              if (a.in_transform) {  // K0 <= 8: only lane groups 0 and 1 reach here (f = fb or fb + 4)
                constexpr int g0 = fb < 8 ? fb : 7, g1 = fb + 4 < 8 ? fb + 4 : 7;
                const float zf = g ? a.tin.zero_floor[g1] : a.tin.zero_floor[g0];
                const int lm = g ? a.tin.log_mask[g1] : a.tin.log_mask[g0];
                const float lo = g ? a.tin.lo[g1] : a.tin.lo[g0];
                const float sc = g ? a.tin.scale[g1] : a.tin.scale[g0];
                if (zf > 0.f && t == 0.f) t = zf;
                if (lm) t = __log10f(t);
                t = (t - lo) * sc - 1.0f;
              }
            }
          }
          v[e] = t;
        });
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) bufA[cb][u][wd] = P::pack2(v[2 * wd], v[2 * wd + 1]);
      });
    });
  }

  static_for<G::RING>([&](auto b) __attribute__((always_inline)) {
    issue_block<G, decltype(b)::value>(a.stream, smem, wave, lane);
  });

  long long wg_rows = a.n_rows - wg_row0;
  if (wg_rows > kWaves * CB * 16) wg_rows = kWaves * CB * 16;
  if (wg_rows < 0) wg_rows = 0;
  const unsigned out_bytes = (unsigned)(wg_rows * a.ldy * 4);
  __amdgpu_buffer_rsrc_t orsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + wg_row0 * a.ldy), 0, out_bytes, 0x00020000);
  const unsigned ldy_b = (unsigned)a.ldy * 4u;
  // per-lane part of the store offset: this lane's signal (row) and its 4-bin group
  const unsigned ovoff = (unsigned)(wave * (CB * 16) + j) * ldy_b + (unsigned)(4 * g) * 4u;

  frag q[D + 1];
  f32x4 auxb[2], auxm[2];  // bias (accumulator init) and, for the output layer, mean
  f32x4 acc[2][CB];

  auto epilogue_chunk = [&](auto g_, auto c_) __attribute__((always_inline)) {
    constexpr int GT = decltype(g_)::value;
    constexpr int c = decltype(c_)::value;
    constexpr Item t = G::tile_at(GT);
    constexpr int l = t.l, nt = t.nt;
    constexpr int cb = c / 2, half = c % 2;
    if constexpr (l < L - 1) {
      constexpr int u = nt / 2, wd = 2 * (nt & 1) + half;
      if constexpr (u < G::ks_of(l + 1)) {
        auto& out = (l & 1) ? bufA : bufB;
        unsigned w = P::pack2(acc[GT & 1][cb][2 * half], acc[GT & 1][cb][2 * half + 1]);
        if constexpr (G::act(l) != 0) {
          const i16x2 z = {0, 0};  // ReLU on the packed pair: signed 16-bit max with 0
          w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), z));
        }
        out[cb][u][wd] = w;
        // an odd tile count leaves the second half of the last k-step without a producer
        if constexpr (nt == G::nt_of(l) - 1 && (nt & 1) == 0) out[cb][u][2 + half] = 0u;
      }
    } else if constexpr (half == 0) {
      // bias rode in as the accumulator's initial value; unpreproc: y * std + mean
      const f32x4 mean = auxm[GT & 1];
      float y[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(acc[GT & 1][cb][r], a.out_std, mean[r] * a.out_mean_scale);
      const unsigned voff = ovoff + (unsigned)(cb * 16) * ldy_b + (unsigned)(16 * nt) * 4u;
      if constexpr (G::full_out_tile(nt)) {
        u32x4 pk = {__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]),
                    __builtin_bit_cast(unsigned, y[2]), __builtin_bit_cast(unsigned, y[3])};
        __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, voff, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          unsigned off = voff + 4u * r;
          if (16 * nt + 4 * g + r >= NOUT) off = 0xFFFFFFF0u;  // bin past out_dim -> dropped
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[r]), orsrc, off, 0, 0);
        }
      }
    }
  };
  auto epilogue_range = [&](auto g_, auto lo_, auto hi_) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
    static_for<(HI > LO ? HI - LO : 0)>([&](auto k) __attribute__((always_inline)) {
      epilogue_chunk(g_, std::integral_constant<int, LO + decltype(k)::value>{});
    });
  };
  auto operand = [&](auto& buf, int cb, int u) __attribute__((always_inline)) {
    const u32x4 wds = {buf[cb][u][0], buf[cb][u][1], buf[cb][u][2], buf[cb][u][3]};
    return __builtin_bit_cast(frag, wds);
  };

  static_for<TOTAL + D>([&](auto s_) __attribute__((always_inline)) {
    constexpr int S = decltype(s_)::value;
    if constexpr (S < TOTAL) {
      ring_boundary16<G, D, S>(a.stream, smem, wave, lane);
      constexpr Item it = G::item_at(S);
      if constexpr (it.ks >= 0) {
        q[S % (D + 1)] = *(const frag*)frag_ptr<G, S>(smem, lane);
      } else {
        constexpr int GT = G::gtile(it.l, it.nt);
        const unsigned char* aux = frag_ptr<G, S>(smem, 0);
        auxb[GT & 1] = *(const f32x4*)(aux + g * 16);            // bias[16nt + 4g + r]
        if constexpr (it.l == L - 1) auxm[GT & 1] = *(const f32x4*)(aux + 64 + g * 16);  // mean
      }
    }
    if constexpr (S >= D) {
      constexpr int C = S - D;
      constexpr Item it = G::item_at(C);
      if constexpr (it.ks >= 0) {
        constexpr int GT = G::gtile(it.l, it.nt);
        constexpr int GP = GT > 0 ? GT - 1 : 0;
        constexpr int CPK = G::chunks_per_kstep(GP);
        constexpr bool whole_first = (GT > 0) && (G::spread_limit(GP) == 0);
        if constexpr (whole_first && it.ks == 0)
          epilogue_range(std::integral_constant<int, GP>{}, std::integral_constant<int, 0>{},
                         std::integral_constant<int, NCH>{});
        auto& in = (it.l & 1) ? bufB : bufA;
        const frag w = q[C % (D + 1)];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          f32x4 c0;
          if constexpr (it.ks == 0) c0 = auxb[GT & 1];
          else c0 = acc[GT & 1][cb];
          acc[GT & 1][cb] = P::mfma(w, operand(in, cb, it.ks), c0);
        }
        if constexpr (GT > 0 && !whole_first) {
          constexpr int lo = it.ks * CPK < NCH ? it.ks * CPK : NCH;
          constexpr int hi = (it.ks + 1) * CPK < NCH ? (it.ks + 1) * CPK : NCH;
          epilogue_range(std::integral_constant<int, GP>{}, std::integral_constant<int, lo>{},
                         std::integral_constant<int, hi>{});
        }
      }
    }
  });
  epilogue_range(std::integral_constant<int, G::n_tiles() - 1>{}, std::integral_constant<int, 0>{},
                 std::integral_constant<int, NCH>{});
}

}  // namespace v21
