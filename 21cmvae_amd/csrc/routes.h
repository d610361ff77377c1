// routes.h -- WHICH kernels a call takes, decided in ONE place (r5).
//
// The library has several kernels for the same arithmetic (VERDICT r4 "route and switch sprawl": five forward routes, seven
// training routes, chosen from precision, row count, stack and rank count).  Until r5 every dispatch site computed its
// own conditions; now the dispatch sites of api_forward.hip / api_trainer.hip / api_sweep.hip / api_joint.hip ask the
// pure functions below and switch on the answer, and the same functions answer the queries of include/v21.h
// (v21_route_forward / v21_route_train: no GPU needed), which is what tests/test_routes.py checks against the table of
// INTEGRATION.md section 6 -- on the CPU against the decision, on the GPU against what was launched
// (v21_trainer_last_route / v21_mlp_last_route are written by the launch sites themselves).
//
// Every environment switch that can change a route is read HERE (RouteEnv) and listed in INTEGRATION.md section 6; the
// defaults are what ships, the switches exist for the tests (which force every kernel onto small cases) and for A/B
// measurements.  Replaces nothing in the reference: Keras has one path (emulator.py:369-378, 402).
#pragma once
#include <cstdlib>
// internal forward flag (never part of include/v21.h; v21_mlp_forward masks caller flags to 0xFF): see decide_forward
#define V21_FWD_RT_LATCH_OFF 0x200

namespace v21 {

// ---- forward (v21_mlp_forward_dev)
enum FwdRoute {
  FWD_NONE = 0,
  FWD_SMALL = 1,     // few rows: one latency-oriented NT launch per layer (api_forward.hip: forward_small)
  FWD_FUSED = 2,     // fused_fwd<Arch, Prec> compiled into the library (archs.h S1-S4)
  FWD_FUSED_RT = 3,  // the same kernel instantiated at run time for this stack (jit.hip)
  FWD_TABLE = 4,     // table-driven one-launch forward: the chain kernels in FORWARD mode (train_chain.h / train_chain32.h)
  FWD_GENERIC = 5,   // per-layer K-loop GEMM (gemm.h): stacks wider than 512, f32 variational stacks
};
// ---- one optimizer step: the launch that carries forward pass, loss and activation gradients ...
enum TrainFwdRoute {
  TR_NONE = 0,
  TR_PER_LAYER = 1,   // one NT launch per layer and direction (gemm_nt.h)
  TR_CHAIN16 = 2,     // train_chain_kernel: 32-row blocks, activations in LDS, f16 / bf16
  TR_FUSED128 = 3,    // fused_train<Arch, Prec>: 128-row workgroups, 32 rows per wave (fused_train.h)
  TR_FUSED64 = 4,     // fused_train16<Arch, Prec>: 64-row workgroups, 16 rows per wave, two per CU (fused_train16.h)
  TR_CHAIN32 = 5,     // train_chain32_kernel: 16-row blocks, fp32 (train_chain32.h)
  TR_CHAIN32S_8 = 6,  // train_chain32s_kernel<8>: 8-row blocks, fp32, host-built job table (train_chain32s.h)
  TR_CHAIN32S_4 = 7,  // train_chain32s_kernel<4>
};
// ---- ... and what forms the weight gradients and applies Adam
enum TrainUpdRoute {
  UP_NONE = 0,
  UP_PER_LAYER = 1,     // NT launches per layer (+ slab sum) + [exchange] + adam_repack_kernel
  UP_DW16_ADAM = 2,     // dw16_adam_kernel: gradients + Adam + packed copies in ONE launch (dw_adam.h; one rank)
  UP_DW16_SPLITK = 3,   // gemm_dw16[_lds]_kernel grouped over the layers (+ slab sum) + [exchange] + adam_repack_kernel
  UP_DWADAM32 = 4,      // dwadam32_kernel: fp32 operands through LDS in whole rows, Adam in the epilogue (dw_adam32.h; one rank)
  UP_NT_DWADAM = 5,     // gemm_nt_dwadam_kernel: register operands, Adam in the epilogue (gemm_nt.h; one rank)
  UP_NT_SLICED = 6,     // grouped NT launch over batch slices (+ slab sum) + [exchange] + adam_repack_kernel (fp32 chain)
};
struct StepRoute { int fwd = TR_NONE, upd = UP_NONE; };

// Every environment switch that changes a route.  Read per decision (a getenv is ~0.1 us; the tests set them per case).
struct RouteEnv {
  int train_chain = 1;        // V21_TRAIN_CHAIN=0: per-layer training path for every trainer
  int fused_train = 1;        // V21_FUSED_TRAIN=0: no fused training kernel
  int fused_train16 = -1;     // V21_FUSED_TRAIN16=0 / 1: which fused kernel a trainer commits to (-1: by max_batch)
  int fused_train_rows = -1;  // V21_FUSED_TRAIN_ROWS=n: steps of >= n rows take the fused kernel (-1: 8,193 / 16,384)
  int dw_split_rows = 8192;   // V21_DW_SPLIT_ROWS=n: one-rank 16-bit steps of >= n rows take split-K + Adam
  int chain32s = -1;          // V21_CHAIN32S=0 / 1: the small-batch fp32 chain (-1: by max_batch)
  int c32s_rows = 0;          // V21_C32S_ROWS=4 / 8: rows per workgroup of the small-batch fp32 chain (0: by rows)
  int dw32_lds = 1;           // V21_DW32_LDS=0: no LDS-staged fp32 gradient launch
  int dw32_adam = 1;          // V21_DW32_ADAM=0: no Adam in the fp32 gradient launch's epilogue
  int jit = 1;                // V21_JIT=0: no run-time compilation (cached code objects are still used)
  int dw_xrows = 0;           // V21_DW_XROWS=1: a fused large step does not flush its layer-0 operand, the gradient launch gathers the resident rows (fewer bytes, measured slower)
  int sweep_streams = 2;      // V21_SWEEP_STREAMS=1: a sweep's grouped launches on one stream (default 2: two half-groups, api_sweep.hip)
  static RouteEnv read() {
    RouteEnv e;
    auto flag = [](const char* n, int dflt) { const char* v = getenv(n); return v ? (v[0] == '0' ? 0 : 1) : dflt; };
    auto num = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
    e.train_chain = flag("V21_TRAIN_CHAIN", 1);
    e.fused_train = flag("V21_FUSED_TRAIN", 1);
    if (const char* v = getenv("V21_FUSED_TRAIN16")) e.fused_train16 = v[0] == '1' ? 1 : 0;
    e.fused_train_rows = num("V21_FUSED_TRAIN_ROWS", -1);
    e.dw_split_rows = num("V21_DW_SPLIT_ROWS", 8192);
    if (const char* v = getenv("V21_CHAIN32S")) e.chain32s = v[0] == '1' ? 1 : 0;
    e.c32s_rows = num("V21_C32S_ROWS", 0);
    e.dw32_lds = flag("V21_DW32_LDS", 1);
    e.dw32_adam = flag("V21_DW32_ADAM", 1);
    e.jit = flag("V21_JIT", 1);
    e.dw_xrows = flag("V21_DW_XROWS", 0);
    e.sweep_streams = num("V21_SWEEP_STREAMS", 2) == 1 ? 1 : 2;
    return e;
  }
};

// What a trainer commits to when it is created (buffers and packed stream formats depend on it).
struct TrainerKind {
  bool chain = false;     // 16-bit chain path (train_chain.h) -- otherwise per-layer
  bool chain32 = false;   // fp32 chain path
  bool chain32s = false;  // ... its small-batch kernel and stream format
  int train_arch = -1;    // index of the compiled fused training kernel (archs.h T1..), or -1
  bool train_rt = false;  // the fused training kernel may be instantiated at run time for this stack (jit.hip)
  bool train16 = false;   // the fused training kernel on 16 rows per wave
  int gl = -1;            // the V21_ACT_GAUSS layer or -1
};
// defined in api_trainer.hip: the registry of compiled fused training kernels / eligibility for run-time instantiation
int fused_train_arch_of(int L, const int* dims, const int* act);
bool fused_train_rt_eligible(int L, const int* dims, const int* act);

inline int route_nw(const int* dims, const int* act, int l) { return act[l] == V21_ACT_GAUSS ? 2 * dims[l + 1] : dims[l + 1]; }

inline TrainerKind decide_trainer_kind(int L, const int* dims, const int* act, int precision, int max_batch, const RouteEnv& e) {
  TrainerKind k;
  for (int l = 0; l < L; ++l)
    if (act[l] == V21_ACT_GAUSS) k.gl = l;
  bool narrow = true;
  for (int l = 0; l <= L; ++l) narrow = narrow && dims[l] <= kChainMaxDim;
  if (precision != V21_PREC_F32) {
    bool ok = e.train_chain && narrow && (k.gl < 0 || dims[k.gl + 1] <= kChainMaxLatent);
    int mask_tiles = 0;
    for (int l = 0; l + 1 < L; ++l) mask_tiles += act[l] == V21_ACT_RELU ? (dims[l + 1] + 31) / 32 : 0;
    k.chain = ok && mask_tiles <= kChainMaskTiles;
    if (k.chain && k.gl < 0 && e.fused_train) {
      k.train_arch = fused_train_arch_of(L, dims, act);
      // (only the 16-rows-per-wave kernel is instantiated at run time: a stack outside archs.h takes it whatever max_batch is)
      k.train_rt = k.train_arch < 0 && e.fused_train16 != 0 && fused_train_rt_eligible(L, dims, act);
      // 16 rows per wave (64-row workgroups, two per CU) for trainers of fewer than 24,576 rows per step: a step of
      // 8,193 .. 24,575 rows does not fill the chip with 128-row workgroups; above, both take the same time and the
      // 128-row form has the evener workgroups
      if (k.train_arch >= 0) k.train16 = e.fused_train16 >= 0 ? e.fused_train16 == 1 : max_batch < 24576;
      if (k.train_rt) k.train16 = true;
    }
  } else {
    bool ok = e.train_chain && narrow;
    k.chain32s = e.chain32s >= 0 ? e.chain32s == 1 : max_batch <= kC32sMaxBatch;
    if (k.gl >= 0) ok = ok && k.chain32s && dims[k.gl + 1] <= kChainMaxLatent;
    int mask_tiles = 0;
    for (int l = 0; l + 1 < L; ++l)
      mask_tiles += act[l] == V21_ACT_RELU ? (k.chain32s ? (dims[l + 1] + 63) / 64 : (dims[l + 1] + 31) / 32) : 0;
    ok = ok && mask_tiles <= (k.chain32s ? kC32sMaskTiles : kC32MaskTiles);
    k.chain32 = ok;
    if (!ok) k.chain32s = false;
  }
  return k;
}

// One optimizer step of `rows` rows on this rank.  `fused_ready`: a fused training kernel can be launched now (compiled
// in, or its run-time code object has arrived); `capturing`: the step is being recorded into a hipGraph.
inline StepRoute decide_step(const TrainerKind& k, int L, const int* dims, const int* act, int rows, int nranks, bool capturing,
                             bool fused_ready, const RouteEnv& e) {
  StepRoute r;
  if (k.chain32) {
    if (k.chain32s) {
      const int rpw = e.c32s_rows == 4 || e.c32s_rows == 8 ? e.c32s_rows : (rows <= kC32sRows4Max ? 4 : 8);
      r.fwd = rpw == 4 ? TR_CHAIN32S_4 : TR_CHAIN32S_8;
    } else {
      r.fwd = TR_CHAIN32;
    }
    const bool single = nranks == 1;
    long long work = 0;
    for (int l = 0; l < L; ++l) work += (long long)((dims[l] + 1 + 63) / 64) * ((route_nw(dims, act, l) + 63) / 64);
    const bool dw32 = single && e.dw32_adam && e.dw32_lds && L <= kNtMaxGroup && work < 192 && rows <= kDw32MaxRows;
    int nslice = dw32 ? 1 : (rows + kNtMaxKPerWg - 1) / kNtMaxKPerWg;
    if (nslice < 1) nslice = 1;
    const int k_chunk = ((rows + nslice - 1) / nslice + 15) / 16 * 16;
    nslice = (rows + k_chunk - 1) / k_chunk;
    if (single && nslice <= 1 && L <= kNtMaxGroup && e.dw32_adam)
      r.upd = work >= 192 ? UP_NT_DWADAM : (rows <= kDw32MaxRows && e.dw32_lds ? UP_DWADAM32 : UP_NT_DWADAM);
    else
      r.upd = UP_NT_SLICED;
    return r;
  }
  if (!k.chain) { r.fwd = TR_PER_LAYER; r.upd = UP_PER_LAYER; return r; }
  // 16-bit chain trainers.  Steps of >= fused_rows rows of a stack with a fused training kernel take it: 8,193 rows for a
  // trainer on the 16-rows-per-wave kernel (the chain's second round of 256 workgroups starts there: 9,216 rows 67 against
  // 80 us, 12,288 rows 72 against 83, 16,384 rows 78-82 against 90-92), 16,384 rows for one on the 128-row kernel
  // (24,576 rows 101 against 127 us, 32,768 rows 123-130 against 152-160; autoencoder stack, f16, whole steps, r4)
  const int fused_rows = e.fused_train_rows >= 0 ? e.fused_train_rows : (k.train16 ? 8193 : 16384);
  const bool fused = (k.train_arch >= 0 || k.train_rt) && fused_ready && rows >= fused_rows && !capturing;
  r.fwd = fused ? (k.train16 ? TR_FUSED64 : TR_FUSED128) : TR_CHAIN16;
  // one rank, nothing to exchange: gradients, Adam and the packed copies in one launch -- up to the batch where its
  // 32 x 32 tiles, each pulling its operands over the WHOLE batch through one CU, lose to the 128 x 128 LDS-staged
  // split-K kernel + an Adam launch that sums the slabs
  r.upd = (nranks == 1 && (rows < e.dw_split_rows || capturing) && !fused) ? UP_DW16_ADAM : UP_DW16_SPLITK;
  return r;
}

// ---- forward.  jit: 0 = no run-time kernel for this stack (not eligible / switched off / failed), 1 = being compiled,
// 2 = ready.  `fused_compiled`: archs.h holds this stack.  Mirrors v21_mlp_forward_dev.
struct FwdQuery {
  int L; const int* dims; const int* act;
  bool fused_compiled; int jit; int precision; long long n; int flags; long long ldy;
};
inline bool route_chain_fwd_eligible(const FwdQuery& q) {
  if (q.flags & V21_FWD_FORCE_GENERIC) return false;
  for (int l = 0; l <= q.L; ++l)
    if (q.dims[l] > kChainMaxDim) return false;
  for (int l = 0; l < q.L; ++l)
    if (q.act[l] == V21_ACT_GAUSS && (q.precision == V21_PREC_F32 || q.dims[l + 1] > kChainMaxLatent || l == q.L - 1)) return false;
  if ((q.flags & V21_FWD_IN_TRANSFORM) && q.dims[0] > 8) return false;
  return true;
}
inline int decide_forward(const FwdQuery& q) {
  int maxdim = 0;
  for (int l = 0; l <= q.L; ++l) maxdim = q.dims[l] > maxdim ? q.dims[l] : maxdim;
  const bool tin_ok = !(q.flags & V21_FWD_IN_TRANSFORM) || q.dims[0] <= 8;
  const bool ldy_ok = q.ldy < (1ll << 21);
  const int force = q.flags & (V21_FWD_FORCE_GENERIC | V21_FWD_FORCE_CHAIN | V21_FWD_FORCE_JIT);
  const bool fused = q.fused_compiled && !force && tin_ok && ldy_ok;
  const bool fused_small = q.fused_compiled && !(q.flags & V21_FWD_FORCE_GENERIC) && tin_ok;
  const bool small = q.n <= V21_SMALL_BATCH_ROWS && !(q.flags & V21_FWD_NO_SMALL) && !force &&
                     (q.precision == V21_PREC_F32 || !fused_small) && tin_ok && maxdim <= kNtMaxKPerWg && ldy_ok;
  if (small) return FWD_SMALL;
  const bool jit_path = (!q.fused_compiled || (q.flags & V21_FWD_FORCE_JIT)) && !(q.flags & (V21_FWD_FORCE_GENERIC | V21_FWD_FORCE_CHAIN)) &&
                        ldy_ok && tin_ok && !(q.flags & V21_FWD_RT_LATCH_OFF);
  if (jit_path && q.jit == 2) return FWD_FUSED_RT;
  if (fused) return FWD_FUSED;
  if (route_chain_fwd_eligible(q) && q.n < (1ll << 30)) return FWD_TABLE;
  return FWD_GENERIC;
}

}  // namespace v21
