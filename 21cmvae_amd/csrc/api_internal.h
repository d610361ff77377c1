// api_internal.h -- what the translation units of the C ABI share (r4: v21_api.hip, one 3,300-line file until then, is
// now api_base.hip (errors, contexts, memory, events, communicator, diagnostics), api_forward.hip (dense stacks and the
// forward routes), api_trainer.hip (trainers, the step machinery, captured steps), api_sweep.hip and api_joint.hip --
// all behind the unchanged include/v21.h).  Kernels live in the headers included below; a kernel template is
// instantiated by the unit that launches it, non-template kernels have internal linkage.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// the library is built with -fvisibility=hidden: only what include/v21.h declares is exported
#pragma GCC visibility push(default)
#include "../../include/v21.h"
#pragma GCC visibility pop
#include "archs.h"
#include "fused_fwd.h"
#include "jit.h"
#include "gemm.h"
#include "gemm_nt.h"
#include "train_kernels.h"
#include "train_chain.h"
#include "train_chain32.h"
#include "train_chain32s.h"
#include "dw_adam32.h"
#if defined(V21_CHAIN_FINE) || defined(V21_T_STAMPS)
constexpr int kStampSlots = 2048;  // (diagnostic builds: per-wave / per-workgroup stamps)
#else
constexpr int kStampSlots = 64;
#endif
#include "dw_adam.h"
#include "routes.h"

using namespace v21;

// ---- errors: v21_last_error() returns the calling thread's last message (api_base.hip)
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(V21_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                  __FILE__, __LINE__);                                                    \
  } while (0)
#define CHK(expr)            \
  do {                       \
    int r_ = (expr);         \
    if (r_ != V21_OK) return r_; \
  } while (0)
static inline long long p16(int d) { return (d + 15) & ~15; }  // row pitch: whole 16-float groups
// Zeroed bytes behind every packed weight stream of the chain kernels: two 4-KiB chunks.  A stream is whole chunks, so
// its end is a page boundary; the rolling prefetch requests addresses AHEAD of what it uses, and a request must never
// leave the allocation (train_chain32s.h: the r3 abort).
constexpr size_t kChainStreamSlack = 8192;
// floats behind the P parameters of an arena: the loss slot, then room to round P + 1 up to whole shards of up to
// 64 ranks (sharded data-parallel Adam works on nranks * ceil((P + 1) / nranks) elements in place)
constexpr size_t kArenaPad = 4 + 64;

// ---- context (api_base.hip)
typedef void* nccl_comm;  // (rccl.h is not included: librccl is dlopen'ed by api_base.hip)
struct v21_ctx {
  int device = 0;
  hipStream_t own = nullptr, stream = nullptr;
  nccl_comm comm = nullptr;
  int nranks = 1, rank = 0;
  // host-staged collectives (v21_comm_init_host): the same data-parallel logic over any transport the host has
  bool host_comm = false;
  v21_comm_host_ops host{};
  float* h_stage = nullptr;
  size_t h_stage_n = 0;
  int sharded = 0;  // 1: reduce-scatter -> Adam on this rank's shard -> all-gather (v21_comm_set_sharded)
  // r5: a communicator WITHOUT a transport (v21_comm_init_null): this rank computes its share of every global batch and
  // takes the N > 1 step structure (operands -> gradients -> [exchange: nothing] -> Adam), so that the compute side of a
  // data-parallel step can be timed on one GPU (bench.py: dp_compute_only)
  bool null_comm = false;
  // r5: the gradient exchange in `buckets` messages (v21_comm_set_buckets; 1 = one message after all weight gradients,
  // 2 = the upper layers' half leaves on comm_stream while the lower layers' gradients are still being formed)
  int buckets = 1;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_bucket[2] = {nullptr, nullptr}, ev_comm_done = nullptr;
  // v21_mlp_forward on many rows: results leave over PCIe on a second stream, slice by slice, while the next slice
  // is being computed (created on first use)
  hipStream_t copy_stream = nullptr;
  hipEvent_t slice_done[2] = {nullptr, nullptr};
  // v21_debug_clock_probe_*: the sampling wave runs on its own stream beside the kernels under test
  hipStream_t probe_stream = nullptr;
  unsigned long long* d_probe = nullptr;
  int probe_cap = 0;
};
int use(v21_ctx* c);

// ---- dense stack (api_forward.hip)
struct v21_mlp {
  v21_ctx* ctx = nullptr;
  int L = 0;
  std::vector<int> dims, act;
  std::vector<long long> w_off, b_off;
  size_t nparams = 0;
  float* d_w = nullptr;  // nparams (+4 pad) floats
  int fused_id = -1;
  unsigned char* d_stream[3] = {nullptr, nullptr, nullptr};
  bool stream_ok[3] = {false, false, false};
  bool has_tin = false, has_tout = false;
  v21_affine_in tin{};
  float out_std = 1.f;
  float* d_mean = nullptr;
  // generic path scratch
  float* d_act[2] = {nullptr, nullptr};
  long long act_rows = 0;
  // host-API staging
  float *d_xs = nullptr, *d_ys = nullptr;
  double* d_xs64 = nullptr;  // float64 rows of v21_mlp_forward awaiting the float64 par_transform
  long long stage_rows = 0;
  int maxdim = 0;
  bool wpad_ok = false;  // false after the arena was rewritten from outside a trainer (set_weights)
  // small-batch latency path: fp32 W^T copies + two padded activation images
  float* d_wt = nullptr;
  std::vector<long long> wt_off;
  bool wt_ok = false;
  float* d_small[2] = {nullptr, nullptr};
  float* d_xpad = nullptr;  // host-API staging of zero-padded input rows
  long long stage_pad_rows = 0;
  // one-launch forward of ANY stack up to 512 wide in f16 / bf16 (train_chain.h, FORWARD mode): the packed forward
  // weight stream per precision (+ the backward stream the packing kernel writes beside it), rebuilt lazily
  void* d_cfw[3] = {nullptr, nullptr, nullptr};
  void* d_cbw[3] = {nullptr, nullptr, nullptr};
  bool cfw_ok[3] = {false, false, false};
  std::vector<long long> cfw_off[2], cbw_off[2];  // element offsets per layer; [0]: 16-bit streams, [1]: fp32 (train_chain32.h)
  long long cfw_bytes[2] = {0, 0}, cbw_bytes[2] = {0, 0};
  v21_affine_in* d_tin = nullptr;           // device copy of the input transform
  // fused_fwd<this stack, precision> instantiated at run time (jit.h) for stacks outside archs.h; requested on the
  // first large forward call, used once its code object is there
  v21::JitKernel* jit[3] = {nullptr, nullptr, nullptr};
  bool jit_asked[3] = {false, false, false};
  // routes.h: the route of the last v21_mlp_forward_dev call and how many calls took each (v21_mlp_last_route)
  int last_route = 0;
  long long route_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long* clk_stamps = nullptr;  // set for the duration of v21_debug_forward_clocked
  // width of layer l's Dense output: dims[l+1], or 2*dims[l+1] = [z_mean | z_log_var] for V21_ACT_GAUSS
  int nw(int l) const { return act[l] == V21_ACT_GAUSS ? 2 * dims[l + 1] : dims[l + 1]; }
};

// ---- trainer (api_trainer.hip)
struct v21_trainer {
  v21_mlp* mlp = nullptr;
  v21_ctx* ctx = nullptr;
  int prec = 0, max_batch = 0;
  v21_adam adam{1e-3f, 0.9f, 0.999f, 1e-7f};
  long long iter = 0;
  size_t P = 0;
  float *d_g = nullptr, *d_m = nullptr, *d_v = nullptr;  // P + 4 floats; d_g[P] = loss slot
  float* d_x[2] = {nullptr, nullptr};
  unsigned short* d_x16 = nullptr; long long ldx16 = 0;  // the training inputs as 16-bit operand elements (fused training kernels: ChainStep::x16)
  float* d_y[2] = {nullptr, nullptr};
  float* d_rw[2] = {nullptr, nullptr};
  long long n[2] = {0, 0};
  bool y_is_x[2] = {false, false};
  int* d_perm = nullptr;
  long long perm_cap = 0;
  long long Bp = 0;  // row pitch of the transposed buffers (batch padded to 32, + slack)
  std::vector<float*> d_h, d_ht, d_dz, d_dzt;
  float *d_wt = nullptr, *d_wp = nullptr;
  std::vector<long long> wt_off, wp_off;
  bool copies_ok = false;
  bool nt_ok = false;  // the fp32 W^T / padded-W copies of the per-layer path are fresh (chain steps skip them)
  float* d_yb = nullptr;
  float* d_wb = nullptr;
  float* d_rowloss = nullptr;
  float* d_steploss = nullptr;
  long long steploss_cap = 0;
  float* d_evalsum = nullptr;
  float* d_slab = nullptr;  // split-K partial gradients: max_slices x (P + 4)
  int max_slices = 1;
  // variational latent layer (V21_ACT_GAUSS, A13): gl = its index or -1
  int gl = -1;
  float *d_zs = nullptr, *d_dzs = nullptr, *d_dzst = nullptr;  // [z_mean | z_log_var], its gradient, transposed
  float* d_klrow = nullptr;
  float kl_weight = 0.f;
  int sample = 1;
  unsigned long long seed = 0;
  // one-kernel forward + activation-gradient chain (train_chain.h; f16 / bf16 stacks up to 512 wide)
  bool chain = false;
  // the same chain in fp32 (train_chain32.h): f32 stacks up to 512 wide without a variational layer; d_fw / d_bw then
  // hold fp32 fragments, fw_off / bw_off count floats, and the weight-gradient operands are d_ht / d_dzt
  bool chain32 = false;
  bool chain32s = false;  // ... with the 8-row kernel and its stream format (train_chain32s.h): trainers of small batches
  int* d_jobs = nullptr;  // train_chain32s.h: C32sJob rows
  int c32_frags(int d) const { return chain32s ? chain32s_frags(d) : chain32_frags(d); }
  int c32_tiles(int d) const { return chain32s ? (d + 63) / 64 : (d + 31) / 32; }
  int loss_slot_pending = -2;  // f32 chain step on one rank: the Adam launch publishes the loss (-2: nothing pending)
  void *d_fw = nullptr, *d_bw = nullptr;
  long long fw_bytes = 0, bw_bytes = 0;
  std::vector<long long> fw_off, bw_off;  // element offsets per layer
  float* d_partial = nullptr;
  unsigned* d_ticket = nullptr;
  std::vector<void*> d_ht16, d_dzt16;  // fragment-ordered weight-gradient operands (train_chain.h)
  // large steps of f16 / bf16 trainers whose stack has a compiled fused training kernel (fused_train.h; archs.h: T1 ..):
  // index into the registry of api_trainer.hip or -1, and that kernel's packed stream (rebuilt before every launch)
  int train_arch = -1;
  v21::JitKernel* train_jit = nullptr;  // r5: ... or its run-time instantiation for a stack outside archs.h (jit.hip; asked for at creation)
  bool train16 = false;  // the fused training kernel on 16 rows per wave (fused_train16.h): the stream below is in ITS format
  unsigned char* d_tstream = nullptr;
  int tstream_total = 0, tstream_padded = 0;
  std::vector<int> ts_first;    // first fragment of every virtual layer (2 L - 1 of them)
  bool ts_write = false;        // the Adam pass that ends the current step also rewrites d_tstream (the step took the fused kernel)
  long long n_chain_steps = 0, n_fused_steps = 0, n_stream_packs = 0, n_stream_adam = 0;  // v21_debug_trainer_counters
  // routes.h: what the trainer committed to at creation, the route of its last eager step (written where the kernels are
  // launched) and how many steps took each (v21_trainer_last_route)
  TrainerKind kind;
  StepRoute last_route;
  long long fwd_count[8] = {0, 0, 0, 0, 0, 0, 0, 0}, upd_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool tstream_fresh = false;   // d_tstream holds the arena's current weights (cleared by every ensure_copies: any other step, eval, sweep, joint)
  int* d_dworder = nullptr;            // dw_adam.h: tile order per XCD (two-dimensional blocks per layer)
  int dw_xper = 0;
  long long BS = 0;                    // batch steps of 16 per feature tile
  unsigned long long* d_stamps = nullptr;
  bool stamps_on = false;  // v21_trainer_enable_stamps: a stamp costs the stamping wave ~600 cycles (s_memtime + its wait), eleven per launch
  // ---- replayed steps (hipGraph).  One optimizer step is captured once per (rows, global rows, data pointers)
  // and replayed; what differs between steps comes from a device table of StepDesc (train_kernels.h) that the
  // host fills for the steps ahead: an epoch's steps in run_epoch, the next kDescRing steps in step_dev.
  int graph_mode = 0;         // 0: off (default, see graph_eligible), 1: asked for (v21_trainer_use_graph)
  bool capturing = false;     // train_on_rows is being recorded, not run
  StepDesc* d_desc = nullptr; StepDesc* h_desc = nullptr;  // device table, page-locked staging copy
  long long desc_cap = 0;
  int* d_cur = nullptr;       // index of the next step's descriptor
  long long desc_next = 0, desc_count = 0;  // host mirror of *d_cur, entries valid in the table
  long long desc_iter0 = -1; float desc_lr = -1.f; bool desc_epoch = false;  // what the table was built for
  // r5: HIP-event stamps of an eager step (v21_trainer_phase_timing / v21_trainer_phase_times): two events per step -- its
  // start and ONE cut point (1 after the chain launch, 2 after the last weight-gradient launch, 3 after the exchange has
  // been joined, 4 after Adam) -- for up to phase_cap steps after they were switched on
  bool phase_on = false;
  std::vector<hipEvent_t> phase_ev;
  int phase_steps = 0, phase_cap = 0, phase_seen = 0, phase_cut = 4;
  struct StepGraph { int rows, brows; const void *x, *y, *rw, *idx; long long row0; hipGraph_t graph; hipGraphExec_t exec; };
  std::vector<StepGraph> graphs;
  int graph_misses = 0;
};
constexpr long long kDescRing = 1024;
static inline StepCtx step_ctx(const v21_trainer* t) { return t->capturing ? StepCtx{t->d_desc, t->d_cur} : StepCtx{nullptr, nullptr}; }
static inline int zalloc(float** p, size_t nfloat, hipStream_t st) {
  HIPCHK(hipMalloc((void**)p, nfloat * sizeof(float)));
  HIPCHK(hipMemsetAsync(*p, 0, nfloat * sizeof(float), st));
  return V21_OK;
}

// ---- shared between the units
// one grouped launch of the latency-oriented NT GEMM (gemm_nt.h); instantiated per unit and group type
template <class GROUP>
static int launch_nt(int prec, GROUP& grp, hipStream_t st) {
  // 64x64 workgroup tiles once the problems are large enough to fill the chip with them
  long long work = 0;
  for (int i = 0; i < grp.count; ++i) work += (long long)((grp.p[i].M + 63) / 64) * ((grp.p[i].N + 63) / 64) * std::max(1, grp.p[i].nz);
  const int T = work >= 192 ? 2 : 1;
  int blocks = 0;
  for (int i = 0; i < grp.count; ++i) {
    NtArgs& g = grp.p[i];
    g.tile = 32 * T;
    g.nx = (g.N + g.tile - 1) / g.tile; g.ny = (g.M + g.tile - 1) / g.tile;
    if (g.nz < 1) g.nz = 1;
    if (g.a_scale == 0.f) g.a_scale = 1.f;
    if (g.b_scale == 0.f) g.b_scale = 1.f;
    if (g.out_scale == 0.f) g.out_scale = 1.f;
    if (g.nz == 1) { g.k_chunk = g.K > 0 ? g.K : 1; g.slab_stride = 0; }
    // (kNtMaxKPerWg = 512 is the range a workgroup keeps in flight at once -- the split of the weight gradient's batch
    //  contraction aims at it; a longer range, e.g. a layer fed by 600 features, is walked in rounds by the same loop)
    grp.first[i] = blocks;
    blocks += g.nx * g.ny * g.nz;
  }
  grp.first[grp.count] = blocks;
  if (blocks <= 0) return V21_OK;
#define V21_NT(PT) \
  do { if (T == 2) hipLaunchKernelGGL((gemm_nt_kernel<PT, 2, GROUP>), dim3(blocks), dim3(256), 0, st, grp); \
       else hipLaunchKernelGGL((gemm_nt_kernel<PT, 1, GROUP>), dim3(blocks), dim3(256), 0, st, grp); } while (0)
  switch (prec) {
    case V21_PREC_F32: V21_NT(PrecF32); break;
    case V21_PREC_F16: V21_NT(PrecF16); break;
    default: V21_NT(PrecBF16); break;
  }
#undef V21_NT
  HIPCHK(hipGetLastError());
  return V21_OK;
}
float adam_alpha(const v21_adam& a, long long t);  // api_trainer.hip
AdamArgs adam_args(v21_trainer* t, bool do_adam, float alpha, bool skip_nt = false);  // api_trainer.hip
int chain_attr(int prec);  // api_trainer.hip
ChainModel chain_model(v21_trainer* t);  // api_trainer.hip
ChainModel chain_model32(v21_trainer* t);  // api_trainer.hip
int chain_prefetchers(int ncons, int models);  // api_trainer.hip
ChainStep chain_step(const float* x, long long ldx, const float* y, long long ldy, const float* rw, const int* d_idx, long long first, int rows, int brows, int dout, const v21_trainer* vae = nullptr, long long row0 = 0);  // api_trainer.hip
void destroy_graphs(v21_trainer* t);  // api_trainer.hip
void dw16_problems(v21_trainer* t, int rows, int brows, int* nslice_out, std::vector<Dw16Args>& probs, float* loss_out2 = nullptr);  // api_trainer.hip
// every entry of an epoch's row table names a row of the training set (an entry outside it is a GPU memory fault in the
// gather of whichever kernel takes the step: checked on the host, one pass over n ints per epoch)
int check_row_table(const int32_t* perm, long long n);
int ensure_copies(v21_trainer* t, bool need_nt = true);  // api_trainer.hip
int gather_batch(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy_src, const float* rw, const int* d_idx, long long first, int rows);  // api_trainer.hip
float grad_opscale(int brows, int dout);  // api_trainer.hip
void invalidate_streams(v21_mlp* m);  // api_forward.hip
int launch_chain32_args(ChainArgs& a, hipStream_t st, bool small = false, int rows_per_wg = 0 /* 4 / 8: the caller decided (routes.h); 0: by the row count */);  // api_trainer.hip
int launch_dw16(int prec, const std::vector<Dw16Args>& probs, hipStream_t st, const DwXRows* xr = nullptr);  // api_trainer.hip
int launch_dw32_group(const std::vector<v21_trainer*>& trs, const Dw32Model* d_tab, int rows, long long step_index, int max_blocks, hipStream_t st, bool small_slabs = false);  // api_sweep.hip
int launch_dw_adam_group(const std::vector<v21_trainer*>& tr, const DwAdamModel* d_tab, const std::vector<DwAdamModel>& h_tab, int rows, int brows, long long slot, hipStream_t st);  // api_trainer.hip
void launch_joint32_kernel(int rpw, bool gauss, dim3 grid, dim3 block, hipStream_t st, const ChainModel* tab, const ChainStep& sa, const ChainStep& sb);  // api_trainer.hip
void launch_joint_kernel(int prec, bool gauss, dim3 grid, dim3 block, hipStream_t st, const ChainModel* tab, const ChainStep& sa, const ChainStep& sb);  // api_trainer.hip
int reduce_and_update(v21_trainer* t, bool chain_copies, int fold, bool exchanged = false);  // api_trainer.hip
// api_base.hip: the all-reduce of d_buf[0, n) on a stream of the caller's choice (RCCL: enqueued there; host-staged
// transport: blocking, staged through that stream; null transport / one rank: nothing)
int comm_allreduce_on(v21_ctx* c, float* d_buf, size_t n, hipStream_t st);
int refresh_dw32_table(const std::vector<v21_trainer*>& trs, Dw32Model* d_tab, std::vector<Dw32Model>& h_tab, int* max_blocks, bool* ok, hipStream_t st);  // api_sweep.hip
int refresh_dw_adam_table(const std::vector<v21_trainer*>& tr, DwAdamModel** d_tab, std::vector<DwAdamModel>& h_tab, hipStream_t st);  // api_trainer.hip
int train_on_rows_chain32(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy, const float* rw, const int* d_idx, long long first, int rows, int brows, float* loss_out, long long row0, bool chain_done = false /* the joint step: the chain of this model ran in the joint launch */);  // api_trainer.hip
int launch_nt_many(int prec, std::vector<NtArgs>& probs, hipStream_t st);  // api_trainer.hip
void launch_chain_forward_mode(int prec, dim3 grid, dim3 block, hipStream_t st, const ChainArgs& a);  // api_trainer.hip
