// fused_fwd.h -- K1: the whole dense stack of one emulator in ONE launch (gfx950).
//
// Replaces the Keras predict loop behind emulator.py:402 / :789-790 of the reference
// (batch-of-32 MatMul+BiasAdd+Relu per layer) with a register-resident chain:
//
//   * a wave owns CT column tiles of 32 signals and keeps the TRANSPOSED activation
//     H^T (features x signals) in registers across all layers: the f32 accumulator
//     tile of layer l is, after ReLU (+ pack to f16/bf16), directly the B operand of
//     layer l+1 ("accumulator tile as the next MFMA's operand").  Hidden layers compute
//     H_{l+1}^T = W^T H_l^T (W^T fragment = A operand); the LAST layer flips to
//     Y = H W (activation = A operand) so the 32 lanes of a half-wave hold 32
//     consecutive output bins and rows are stored as 128-byte segments.
//   * activations never touch LDS or HBM; the only streamed operand is the weight
//     set, pre-packed on the host into 1-KiB MFMA fragments in consumption order
//     ("the stream", ensure_stream() in api_forward.hip).  The 4 waves of a workgroup share
//     it through a 4-slot LDS ring filled by LDS-DMA (global_load_lds_dwordx4) two
//     blocks ahead, with one counted vmcnt + s_barrier per 24-fragment block.
//   * bias enters as the accumulator's initial value (hidden layers) or in the
//     epilogue together with preprocess.unpreproc (last layer); the optional
//     prologue is preprocess.par_transform with cached training-set statistics.
//
// Everything (layer, tile, k-step, fragment and ring-slot indices, wait counts) is a
// compile-time constant: the kernel is straight-line code per architecture.
#pragma once
#ifndef __HIPCC_RTC__  // (hiprtc brings its own runtime header: csrc/jit.hip compiles this file at run time)
#include <hip/hip_runtime.h>
#endif

#include <type_traits>
#include <utility>

#include "../../include/v21_types.h"
#include "par_transform.h"

namespace v21 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kFragBytes = 1024;  // one MFMA operand fragment: 64 lanes x 16 B
constexpr int kWaves = 4;         // waves per workgroup
// ring geometry comes from the precision/variant traits: P::BLK fragments per block,
// P::RING slots; LDS per workgroup = RING * BLK KiB
template <class P> constexpr int fused_lds() { return P::RING * P::BLK * kFragBytes; }
#ifdef V21_FUSED_STAMP
template <class P> constexpr int fused_lds_alloc() { return fused_lds<P>() + 16384; }  // + stamp area
#else
template <class P> constexpr int fused_lds_alloc() { return fused_lds<P>(); }
#endif

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>),
// expanded in chunks of 256 so that fold expressions stay below clang's nesting limit
template <int Base, int N, class F>
__device__ __forceinline__ void static_for_chunk(F& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    (f(std::integral_constant<int, Base + I>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  constexpr int CH = 256;
  if constexpr (N <= CH) {
    static_for_chunk<0, N>(f);
  } else {
    [&]<int... C>(std::integer_sequence<int, C...>) __attribute__((always_inline)) {
      (static_for_chunk<C * CH, (N - C * CH < CH ? N - C * CH : CH)>(f), ...);
    }(std::make_integer_sequence<int, (N + CH - 1) / CH>{});
  }
}

struct FusedArgs {
  const float* x;
  long long ldx;
  float* y;
  long long ldy;
  long long n_rows;
  const unsigned char* stream;  // packed fragments, padded to 4 KiB
  float out_std;                // 1.0 when no output transform
  float out_mean_scale;         // 1.0 / 0.0: output transform on / off
  int in_transform;
  v21_affine_in tin;
  unsigned long long* dbg;      // diagnostic builds only (V21_FUSED_STAMP): cycle stamps
};

// ---- precision traits ------------------------------------------------------------
// FPI = features per stream item (one 1-KiB fragment = one k-step); EPI = operand
// elements per lane per item.  Accumulator register i of tile nt becomes element
// i % EPI of item (16/EPI)*nt + i / EPI of the next layer's operand, and element e of
// item ku held by lane half h is feature  FPI*ku + 8*(e>>2) + 4*h + (e&3).
struct PrecF16 {
  static constexpr int BLK = 24, RING = 4, WPS = 1, DEPTH = 2, WAVES = 4;  // 96 KiB ring, one wave per SIMD
  using frag = f16x8;
  using elem = _Float16;
  static constexpr int FPI = 16, EPI = 8, CT = 2;
  static constexpr int WPI = 4;  // 32-bit operand words per lane per item
  // two f32 -> one packed word (round to nearest even)
  static __device__ __forceinline__ unsigned pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
  }
  template <bool SWAP> static __device__ __forceinline__ f32x16 mfma(frag w, frag x, f32x16 c) {
    if constexpr (!SWAP) return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, c, 0, 0, 0);
  }
};
struct PrecBF16 {
  static constexpr int BLK = 24, RING = 4, WPS = 1, DEPTH = 2, WAVES = 4;  // 96 KiB ring, one wave per SIMD
  using frag = bf16x8;
  using elem = __bf16;
  static constexpr int FPI = 16, EPI = 8, CT = 2;
  static constexpr int WPI = 4;
  static __device__ __forceinline__ unsigned pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  }
  template <bool SWAP> static __device__ __forceinline__ f32x16 mfma(frag w, frag x, f32x16 c) {
    if constexpr (!SWAP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, w, c, 0, 0, 0);
  }
};
// exact f32: v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain (no reduced precision)
struct PrecF32 {
  static constexpr int BLK = 24, RING = 4, WPS = 1, DEPTH = 2, WAVES = 4;
  using frag = f32x4;
  using elem = float;
  static constexpr int FPI = 8, EPI = 4, CT = 1;
  static constexpr int WPI = 4;
  template <bool SWAP> static __device__ __forceinline__ f32x16 mfma(frag w, frag x, f32x16 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (!SWAP) c = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], x[e], c, 0, 0, 0);
      else c = __builtin_amdgcn_mfma_f32_32x32x2f32(x[e], w[e], c, 0, 0, 0);
    }
    return c;
  }
};

// Variant "x2": one column tile per wave (128 signals per workgroup), <= 256 registers,
// 80 KiB ring -> TWO workgroups per CU, i.e. two waves per SIMD that cover each other's
// stalls (LDS latency, DMA issue, epilogue VALU, output-layer stores).
#ifdef V21_FUSED_STAMP  // diagnostic build: one ring slot less makes room for the stamp area at 2 workgroups/CU
struct PrecF16x2 : PrecF16 { static constexpr int CT = 1, BLK = 16, RING = 4, WPS = 2; };
struct PrecBF16x2 : PrecBF16 { static constexpr int CT = 1, BLK = 16, RING = 4, WPS = 2; };
#else
struct PrecF16x2 : PrecF16 { static constexpr int CT = 1, BLK = 16, RING = 5, WPS = 2; };
struct PrecBF16x2 : PrecBF16 { static constexpr int CT = 1, BLK = 16, RING = 5, WPS = 2; };
#endif
// (A/B-tested and dropped in round 1, see DESIGN.md: one wave per SIMD with two column tiles, a shared ring
// for eight waves, the 16x16x32 MFMA shape, read-ahead depth 1/3, ring geometries 20x4 / 12x6, burst refill)

// Variant "x2sp": x2 with the ring refill spread over the block being consumed (one DMA per wave every 4
// k-steps, issued by every wave -- no wave-dependent branch) instead of a burst of 4 at each rendezvous.
struct PrecF16x2sp : PrecF16x2 { static constexpr bool SPREAD_DMA = true; };
struct PrecBF16x2sp : PrecBF16x2 { static constexpr bool SPREAD_DMA = true; };
// r5: the same two kernels with the clock stamps of VERDICT r4 item 2 -- wave 0 of EVERY workgroup reads the shader-clock
// counter (s_memtime), the constant 100 MHz counter (s_memrealtime) and its XCD when it starts and when it ends, and leaves
// them in FusedArgs::dbg (five 64-bit words per workgroup).  A separate instantiation (bench.py runs it in a separate,
// untimed repeat of the K launches through v21_debug_forward_clocked): the shipped kernels' ISA is untouched.
struct PrecF16x2spClk : PrecF16x2sp { static constexpr bool CLOCK_STAMPS = true; };
struct PrecBF16x2spClk : PrecBF16x2sp { static constexpr bool CLOCK_STAMPS = true; };
template <class P, class = void> struct clk_of { static constexpr bool value = false; };
template <class P> struct clk_of<P, std::void_t<decltype(P::CLOCK_STAMPS)>> { static constexpr bool value = P::CLOCK_STAMPS; };
template <class P, class = void> struct spread_of { static constexpr bool value = false; };
template <class P> struct spread_of<P, std::void_t<decltype(P::SPREAD_DMA)>> { static constexpr bool value = P::SPREAD_DMA; };

// ---- compile-time geometry of (architecture, precision) ---------------------------
// Arch::L layers, Arch::dims[L+1], Arch::act[L] (1 = ReLU).  The last layer is the
// "output orientation" layer and must be linear.
template <class Arch, class P> struct Geo {
  static constexpr int L = Arch::L;
  static constexpr int FPI = P::FPI;
  static constexpr int BLK = P::BLK, RING = P::RING, WAVES = P::WAVES;
  static constexpr int IPT = 16 / P::EPI;  // operand items produced per 32-wide tile
  static constexpr int dim(int i) { return Arch::dims[i]; }
  static constexpr int act(int l) { return Arch::act[l]; }
  static constexpr int ks_of(int l) { return (dim(l) + FPI - 1) / FPI; }
  static constexpr int nt_of(int l) { return (dim(l + 1) + 31) / 32; }
  // stream index of the aux fragment of tile (l, nt); its k-steps follow it
  static constexpr int tile_base(int l, int nt) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += nt_of(i) * (ks_of(i) + 1);
    return f + nt * (ks_of(l) + 1);
  }
  static constexpr int total() { return tile_base(L, 0); }
  static constexpr int padded() { return (total() + 7) / 8 * 8; }
  static constexpr int n_blocks() { return (padded() + BLK - 1) / BLK; }
  static constexpr int blk_glds(int b) {  // LDS-DMA instructions per wave in block b
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
  static constexpr int gtile(int l, int nt) {  // global tile counter
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

  // Program order is a sequence of steps S = 0 .. total+D-1.  Step S does, in order:
  // [ring rendezvous if S is the first item of a block] -> LDS read of item S ->
  // MFMAs of item S-D -> a slice ("chunks") of the PREVIOUS tile's epilogue.
  // A tile's epilogue is cut into chunks of one (column tile, register pair) each and
  // spread over the first k-steps of the next tile, so that its VALU work (hidden
  // layers: ReLU + pack) or its stores (output layer) issue in the gaps between that
  // tile's MFMAs instead of stalling the matrix pipe at the tile seam.
  //
  // k-steps of tile G+1 over which tile G's epilogue may be spread (0 = it must run
  // whole before the first k-step of G+1, which already consumes it)
  static constexpr int spread_limit(int G) {
    const Item t = tile_at(G), n = tile_at(G + 1);
    if (n.l < 0) return 0;
    // Output layer: a tile's epilogue reads its bias / mean from auxb[tile parity], and the aux fragment of tile G + 2
    // (same parity) lands there when the LOAD side reaches it, DEPTH items ahead of the compute side, i.e. during
    // k-step ks - DEPTH of tile G + 1: the chunks must be done by then.  (Found by the run-time instantiations of r4:
    // the compiled-in stacks have 14-28 k-steps per output tile and finish their 8 chunks long before; an output layer
    // fed by 128 features has 8 k-steps in f16, and its rows 24-31 -- chunks 6 and 7 -- left with the NEXT tile's bias.)
    if (n.l == t.l && t.l == L - 1) return ks_of(n.l) > P::DEPTH ? ks_of(n.l) - P::DEPTH : 0;
    if (n.l == t.l) return ks_of(n.l);
    const int first_use = IPT * (nt_of(t.l) - 1);
    return first_use < ks_of(n.l) ? first_use : ks_of(n.l);
  }
  static constexpr int chunks_per_kstep(int G, int nchunks) {
    const int lim = spread_limit(G);
    return lim <= 0 ? nchunks : (nchunks + lim - 1) / lim;
  }
  // output-layer store instructions issued by one wave in steps < S
  static constexpr int stores_before_step(int S, int CT, int D) {
    int s = 0;
    const int l = L - 1, nch = CT * 8;
    for (int nt = 0; nt + 1 < nt_of(l); ++nt) {
      const int cpk = chunks_per_kstep(gtile(l, nt), nch);
      const int nb = tile_base(l, nt + 1);  // aux item of the next tile
      for (int c = 0; c < nch; ++c)
        if (nb + 1 + c / cpk + D < S) s += 2;
    }
    return s;
  }
};

template <int N> __device__ __forceinline__ void wait_vmcnt_barrier() {
  static_assert(N >= 0 && N <= 63, "vmcnt is 6 bits");
  // one statement: nothing that touches memory may move across the rendezvous
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// One LDS-DMA: 64 lanes x 16 B, global [sbase + voff] -> LDS [dst, dst + 1 KiB).
// Issued through inline asm ON PURPOSE: with the builtin form hipcc (ROCm 7.2) treats
// the DMA as a pending LDS event and degrades every later LDS-read wait in the kernel
// from a counted lgkmcnt(N) to lgkmcnt(0).  The statement has no VGPR destination, so
// it is register-safe; completion is tracked by hand (wait_vmcnt_barrier).  M0 (the
// LDS destination) is written in the same statement that reads it; nothing else in
// this kernel uses M0 (checked in the .s: no m0 outside ASMSTART/ASMEND).
__device__ __forceinline__ void glds16(const unsigned char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}

// LDS-DMA of block B of the stream into ring slot B % kRing: wave w moves
// fragments w, w+4, ... (1 KiB each, lane-linear both sides).
template <class G, int B>
__device__ __forceinline__ void issue_block(const unsigned char* gstream, unsigned char* smem,
                                            int wave, int lane) {
  if constexpr (B >= 0 && B < G::n_blocks()) {
    constexpr int NG = G::blk_glds(B);
    const unsigned char* g = gstream + (size_t)B * G::BLK * kFragBytes;  // wave-uniform
    const unsigned voff = wave * kFragBytes + lane * 16;
    const unsigned s = lds_addr(smem) + (B % G::RING) * G::BLK * kFragBytes + wave * kFragBytes;
    static_for<NG>([&](auto i) __attribute__((always_inline)) {
      constexpr int I = decltype(i)::value;
      glds16(g + I * G::WAVES * kFragBytes, voff, s + I * G::WAVES * kFragBytes);
    });
  }
}

// One piece (fragment I*4 + wave) of block B, issued by the wave whose turn it is.
// Spreading a block's DMA over the k-steps of the block being consumed -- one wave per
// step -- keeps each 1-KiB DMA's issue (tens of cycles, during which the in-order wave
// cannot issue MFMAs) inside the shadow of the MFMAs already in flight; issuing all of
// them right after the rendezvous stalled all four SIMDs at once.
template <class G, int B, int I>
__device__ __forceinline__ void issue_piece(const unsigned char* gstream, unsigned char* smem,
                                            int wave, int lane) {
  if constexpr (B >= 0 && B < G::n_blocks()) {
    if constexpr (I < G::blk_glds(B)) {
      const unsigned char* g = gstream + (size_t)B * G::BLK * kFragBytes + I * G::WAVES * kFragBytes;
      const unsigned voff = wave * kFragBytes + lane * 16;
      const unsigned s = lds_addr(smem) + (B % G::RING) * G::BLK * kFragBytes + wave * kFragBytes +
                         I * G::WAVES * kFragBytes;
      glds16(g, voff, s);
    }
  }
}

// Before the LDS read of item S (S = first item of block B): wait until this wave's
// share of block B has landed, rendezvous (now every wave's share has), then refill
// the slot of block B-2, which every wave has finished consuming (D <= kBlkFrags).
// The prologue issued blocks 0..kRing-1; boundary B' >= 2 issues block B'+kRing-2.
template <class G, int CT, int D, int S, bool SPREAD>
__device__ __forceinline__ void ring_boundary(const unsigned char* gstream, unsigned char* smem,
                                              int wave, int lane, unsigned long long* g_dbg = nullptr) {
  constexpr int kBlkFrags = G::BLK, kRing = G::RING;
  if constexpr (S % kBlkFrags == 0 && S < G::padded()) {
    constexpr int B = S / kBlkFrags;
    constexpr int last_issued = (B + kRing - 3 > kRing - 1) ? B + kRing - 3 : kRing - 1;
    constexpr int GA = [] {
      int s = 0;
      for (int i = B + 1; i <= last_issued; ++i) s += G::blk_glds(i);
      return s;
    }();
    // stores younger than block B's DMA.  With SPREAD the pieces of block B were issued
    // during the consumption of block B-kRing+2; counting only stores issued after that
    // whole block (a lower bound of the true number) keeps the wait on the safe side.
    constexpr int S_issue = (B < kRing) ? 0 : (B - kRing + 2 + (SPREAD ? 1 : 0)) * kBlkFrags;
    constexpr int SA = G::stores_before_step(S, CT, D) - G::stores_before_step(S_issue, CT, D);
    constexpr int N = (GA + SA) > 63 ? 63 : (GA + SA);
#ifdef V21_FUSED_STAMP
    { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      ((unsigned long long*)(smem + G::RING * G::BLK * kFragBytes))[wave * 512 + 2 * B] = t; }
#endif
    wait_vmcnt_barrier<N>();
#ifdef V21_FUSED_STAMP
    { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      ((unsigned long long*)(smem + G::RING * G::BLK * kFragBytes))[wave * 512 + 2 * B + 1] = t; }
#endif
    if constexpr (B >= 2 && !SPREAD) issue_block<G, B + kRing - 2>(gstream, smem, wave, lane);
  }
}

template <class G, int F> __device__ __forceinline__ const unsigned char* frag_ptr(const unsigned char* smem, int lane) {
  constexpr int B = F / G::BLK;
  return smem + ((B % G::RING) * G::BLK + (F % G::BLK)) * kFragBytes + lane * 16;
}

// ---- the kernel -------------------------------------------------------------------
// grid.x = ceil(n_rows / (WAVES*CT*32)); block = 64*WAVES threads; dynamic LDS fused_lds<P>().
template <class Arch, class P>
__global__ void __launch_bounds__(64 * P::WAVES, P::WPS) fused_fwd(const FusedArgs a) {
  constexpr int kWaves = P::WAVES;
  constexpr int kBlkFrags = P::BLK, kRing = P::RING, kFusedLds = fused_lds<P>();
  (void)kFusedLds;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = Geo<Arch, P>;
  using frag = typename P::frag;
  using Item = typename G::Item;
  constexpr int L = G::L, CT = P::CT, EPI = P::EPI, FPI = P::FPI, IPT = G::IPT;
  constexpr int KSM = G::ks_max();
#ifndef V21_FUSED_D
#define V21_FUSED_D 2
#endif
#ifndef V21_FUSED_SPREAD
#define V21_FUSED_SPREAD 0
#endif
  constexpr int D = P::DEPTH;  // LDS read-ahead, in fragments
  constexpr bool SPREAD = (V21_FUSED_SPREAD != 0) || spread_of<P>::value;
  constexpr int TOTAL = G::total();
  constexpr int NOUT = G::dim(L);
  constexpr int NCH = CT * 8;  // epilogue chunks per tile
  static_assert(G::act(L - 1) == 0, "output layer must be linear");
  // (see Geo::spread_limit: with fewer k-steps than this the next-but-one tile's aux fragment overtakes the epilogue)
  static_assert(G::nt_of(L - 1) == 1 || G::ks_of(L - 1) > P::DEPTH, "output layer: too few k-steps per tile for the aux double buffer");
  static_assert(D <= kBlkFrags, "read-ahead must stay within one block");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  unsigned long long clk_t0 = 0, clk_r0 = 0;
  if constexpr (clk_of<P>::value) {  // (every wave reads: scalar registers, no divergence; wave 0 reports)
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk_t0), "=s"(clk_r0)::"memory");
  }
  const long long wg_row0 = (long long)blockIdx.x * (kWaves * CT * 32);
  const long long row0 = wg_row0 + wave * (CT * 32);

  // Operand registers of the two layers in flight, as 32-bit words (4 per item).
  unsigned bufA[CT][KSM][4], bufB[CT][KSM][4];

  // ---- layer-0 operand: x rows -> operand registers; optional fused par_transform
  {
    constexpr int K0 = G::dim(0);
    static_for<CT>([&](auto ct_) __attribute__((always_inline)) {
      constexpr int ct = decltype(ct_)::value;
      const long long row = row0 + ct * 32 + r;
      const bool ok = row < a.n_rows;
      const float* xr = a.x + (ok ? row : 0) * a.ldx;
      static_for<G::ks_of(0)>([&](auto ks_) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_)::value;
        float v[EPI];
        static_for<EPI>([&](auto e_) __attribute__((always_inline)) {
          constexpr int e = decltype(e_)::value;
          constexpr int f0 = FPI * ks + 8 * (e >> 2) + (e & 3);  // lanes h = 0
          constexpr int f1 = f0 + 4;                             // lanes h = 1
          float t = 0.f;
          if constexpr (f0 < K0) {
            const bool valid = ok && (h == 0 || f1 < K0);
            if (valid) {
              t = xr[f0 + 4 * h];
              if (a.in_transform) {
                constexpr int g1 = f1 < 8 ? f1 : 7;
                constexpr int g0 = f0 < 8 ? f0 : 7;
                const double zf = h ? a.tin.zero_floor[g1] : a.tin.zero_floor[g0];
                const int lm = h ? a.tin.log_mask[g1] : a.tin.log_mask[g0];
                const double lo = h ? a.tin.lo[g1] : a.tin.lo[g0];
                const double sp = h ? a.tin.span[g1] : a.tin.span[g0];
                t = par_transform_f32(t, lm, zf, lo, sp);  // (train_kernels.h: float64 log10 and map)
              }
            }
          }
          v[e] = t;
        });
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
          if constexpr (EPI == 8) bufA[ct][ks][wd] = P::pack2(v[2 * wd], v[2 * wd + 1]);
          else bufA[ct][ks][wd] = __builtin_bit_cast(unsigned, v[wd]);
        }
      });
    });
  }

  // ---- ring prologue: blocks 0..kRing-1 in flight
  static_for<kRing>([&](auto b) __attribute__((always_inline)) {
    issue_block<G, decltype(b)::value>(a.stream, smem, wave, lane);
  });

  // output addressing (last layer): per-workgroup buffer resource so that rows past
  // n_rows and columns past out_dim are dropped by the hardware range check while the
  // store instruction still issues (keeps the vmcnt arithmetic of ring_boundary exact).
  long long wg_rows = a.n_rows - wg_row0;
  if (wg_rows > kWaves * CT * 32) wg_rows = kWaves * CT * 32;
  if (wg_rows < 0) wg_rows = 0;
  const unsigned out_bytes = (unsigned)(wg_rows * a.ldy * 4);
  __amdgpu_buffer_rsrc_t orsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + wg_row0 * a.ldy), 0, out_bytes, 0x00020000);
  const unsigned ldy_b = (unsigned)a.ldy * 4u;
  // per-lane part of the store offset: row 4h of this wave's rows, column r
  const unsigned ovoff = (unsigned)(wave * (CT * 32) + 4 * h) * ldy_b + (unsigned)r * 4u;

  frag q[D + 1];      // fragments in flight LDS -> registers
  f32x16 auxb[2];     // aux fragment of the tile being started (by tile parity)
  f32x16 acc[2][CT];  // accumulators, double-buffered by tile parity

  // chunk c = (ct, register pair) of the epilogue of global tile GT (compile-time):
  // hidden layer -> pack (+ ReLU) two values into the next layer's operand registers;
  // output layer -> Dense bias + preprocess.unpreproc + two row-segment stores.
  auto epilogue_chunk = [&](auto g_, auto c_) __attribute__((always_inline)) {
    constexpr int GT = decltype(g_)::value;
    constexpr int c = decltype(c_)::value;
    constexpr Item t = G::tile_at(GT);
    constexpr int l = t.l, nt = t.nt;
    constexpr int ct = c / 8, pr = c % 8;  // accumulator registers 2pr, 2pr+1
    if constexpr (l < L - 1) {
      constexpr int item = IPT * nt + (2 * pr) / EPI, e0 = (2 * pr) % EPI;
      if constexpr (item < G::ks_of(l + 1)) {
        auto& out = (l & 1) ? bufA : bufB;
        const float x0 = acc[GT & 1][ct][2 * pr], x1 = acc[GT & 1][ct][2 * pr + 1];
        if constexpr (EPI == 8) {
          // pack first, then ReLU on the packed pair as a signed 16-bit max with 0
          // (negative f16/bf16 are negative integers): 2 VALU ops per 2 values
          unsigned w = P::pack2(x0, x1);
          if constexpr (G::act(l) != 0) {
            const i16x2 z = {0, 0};
            w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), z));
          }
          out[ct][item][e0 / 2] = w;
        } else {
          int b0 = __builtin_bit_cast(int, x0), b1 = __builtin_bit_cast(int, x1);
          if constexpr (G::act(l) != 0) { b0 = max(b0, 0); b1 = max(b1, 0); }
          out[ct][item][e0] = (unsigned)b0;
          out[ct][item][e0 + 1] = (unsigned)b1;
        }
      }
    } else {
      const float obias = auxb[GT & 1][0], omean = auxb[GT & 1][1] * a.out_mean_scale;
      unsigned voff = ovoff + (unsigned)(32 * nt) * 4u;
      if (32 * nt + r >= NOUT) voff = 0xFFFFFFF0u;  // column past out_dim -> dropped
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = 2 * pr + u;
        const unsigned soff = (unsigned)(ct * 32 + (i & 3) + 8 * (i >> 2)) * ldy_b;  // wave-uniform
        float y;
        if constexpr (EPI == 8) {
          y = __builtin_fmaf(acc[GT & 1][ct][i], a.out_std, __builtin_fmaf(obias, a.out_std, omean));
        } else {
          // exact mode: (acc + bias) * std + mean, each rounded to f32 as numpy does
          y = (acc[GT & 1][ct][i] + obias) * a.out_std + omean;
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), orsrc, voff, soff, 0);
      }
    }
  };
  auto epilogue_range = [&](auto g_, auto lo_, auto hi_) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
    static_for<(HI > LO ? HI - LO : 0)>([&](auto k) __attribute__((always_inline)) {
      epilogue_chunk(g_, std::integral_constant<int, LO + decltype(k)::value>{});
    });
  };
  auto operand = [&](auto& buf, int ct, int ks) __attribute__((always_inline)) {
    const u32x4 wds = {buf[ct][ks][0], buf[ct][ks][1], buf[ct][ks][2], buf[ct][ks][3]};
    return __builtin_bit_cast(frag, wds);
  };

  static_for<TOTAL + D>([&](auto s_) __attribute__((always_inline)) {
    constexpr int S = decltype(s_)::value;
    // ---- load side: item S
    if constexpr (S < TOTAL) {
      ring_boundary<G, CT, D, S, SPREAD>(a.stream, smem, wave, lane, a.dbg);
      if constexpr (SPREAD && S / kBlkFrags >= 2) {
        // refill of slot (B-2), spread over the block being consumed: EVERY wave issues its piece o/W of block
        // B+kRing-2 at offsets o = 0, W, 2W, ... (W = waves) -- one DMA per W k-steps instead of a burst of
        // BLK/W at the rendezvous, and no wave-dependent branch in the unrolled stream
        constexpr int Bc = S / kBlkFrags, o = S % kBlkFrags;
#ifndef V21_SP_PHASE
#define V21_SP_PHASE 3  // measured r1: the last step of each group of W beats the first (which coincides with the rendezvous)
#endif
        if constexpr (o % G::WAVES == V21_SP_PHASE % G::WAVES)
          issue_piece<G, Bc + kRing - 2, o / G::WAVES>(a.stream, smem, wave, lane);
      }
      constexpr Item it = G::item_at(S);
      if constexpr (it.ks >= 0) {
        q[S % (D + 1)] = *(const frag*)frag_ptr<G, S>(smem, lane);
      } else {
        constexpr int GT = G::gtile(it.l, it.nt);
        const unsigned char* aux = frag_ptr<G, S>(smem, 0);
        if constexpr (it.l < L - 1) {
          // bias[32nt + rho(reg) + 4h]: the accumulator's initial value
          const f32x4* bp = (const f32x4*)(aux + h * 64);
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const f32x4 t = bp[qd];
            auxb[GT & 1][4 * qd + 0] = t[0]; auxb[GT & 1][4 * qd + 1] = t[1];
            auxb[GT & 1][4 * qd + 2] = t[2]; auxb[GT & 1][4 * qd + 3] = t[3];
          }
        } else {
          auxb[GT & 1][0] = ((const float*)aux)[r];       // bias[32nt + c]
          auxb[GT & 1][1] = ((const float*)aux)[32 + r];  // mean[32nt + c]
        }
      }
    }
    // ---- compute side: item S - D
    if constexpr (S >= D) {
      constexpr int C = S - D;
      constexpr Item it = G::item_at(C);
      if constexpr (it.ks >= 0) {
        constexpr int GT = G::gtile(it.l, it.nt);
        constexpr int GP = GT > 0 ? GT - 1 : 0;  // previous tile: its epilogue is pending
        constexpr int CPK = G::chunks_per_kstep(GP, NCH);
        constexpr bool whole_first = (GT > 0) && (G::spread_limit(GP) == 0);
        if constexpr (whole_first && it.ks == 0) {
          epilogue_range(std::integral_constant<int, GP>{}, std::integral_constant<int, 0>{},
                         std::integral_constant<int, NCH>{});
        }
#ifdef V21_FUSED_STAMP
        if constexpr (it.ks == 0) {
          unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
          ((unsigned long long*)(smem + kFusedLds))[wave * 512 + 128 + GT] = t; }
#endif
        auto& in = (it.l & 1) ? bufB : bufA;
        const frag w = q[C % (D + 1)];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          f32x16 c0;
          if constexpr (it.ks == 0) {
            if constexpr (it.l < L - 1) c0 = auxb[GT & 1];
            else
#pragma unroll
              for (int i = 0; i < 16; ++i) c0[i] = 0.f;
          } else {
            c0 = acc[GT & 1][ct];
          }
          acc[GT & 1][ct] = P::template mfma<(it.l == L - 1)>(w, operand(in, ct, it.ks), c0);
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
  if constexpr (clk_of<P>::value) {
    // after the wave's last output store has been ISSUED (the stores drain behind it; the next launch's start stamp
    // cannot precede them on an in-order stream)
    unsigned long long t1, r1;
    unsigned xcc, hwid;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %3, hwreg(HW_REG_HW_ID)\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(t1), "=s"(r1), "=s"(xcc), "=s"(hwid)::"memory");
    if (a.dbg && wave == 0 && lane == 0) {
      unsigned long long* gd = a.dbg + (size_t)blockIdx.x * 5;
      // word 4: XCD in bits 0-3, HW_REG_HW_ID (wave / SIMD / CU / shader array / shader engine of wave 0) in bits 8-39
      gd[0] = clk_t0; gd[1] = clk_r0; gd[2] = t1; gd[3] = r1; gd[4] = (unsigned long long)(xcc & 0xF) | ((unsigned long long)hwid << 8);
    }
  }
#ifdef V21_FUSED_STAMP
  if (a.dbg) {  // stamps were kept in LDS so that they add no VMEM operation to the counted waits
    unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    unsigned long long* sd = (unsigned long long*)(smem + kFusedLds) + wave * 512;
    sd[127] = t;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    unsigned long long* gd = a.dbg + ((size_t)blockIdx.x * kWaves + wave) * 512;
#pragma unroll
    for (int k = 0; k < 8; ++k) gd[64 * k + lane] = sd[64 * k + lane];
  }
#endif
}

}  // namespace v21
