// api_trainer.hip -- v21_trainer_*: the optimizer step on its three routes (per-layer NT kernels, the 16-bit chain, the
// fp32 chains), the exchange of data-parallel ranks, captured steps (hipGraph), the Keras-fit()-shaped epoch driver.
#include "api_internal.h"

// ---- compiled fused training kernels (fused_train.h, one translation unit each: fused_train_inst.hip)
namespace v21 {
#define V21_TDECL(a)                                                             \
  hipError_t launch_fused_train_##a##_F16t(const ChainArgs&, hipStream_t);      \
  hipError_t launch_fused_train_##a##_BF16t(const ChainArgs&, hipStream_t);     \
  hipError_t launch_fused_train16_##a##_F16t16(const ChainArgs&, hipStream_t);  \
  hipError_t launch_fused_train16_##a##_BF16t16(const ChainArgs&, hipStream_t);
V21_TRAIN_ARCH_LIST(V21_TDECL)
#undef V21_TDECL
}  // namespace v21
typedef hipError_t (*train_launcher)(const ChainArgs&, hipStream_t);
struct TrainEntry { int L; const int* dims; const int* act; train_launcher fn[2]; /* f16, bf16 */ train_launcher fn16[2]; /* 16 rows per wave */ };
#define V21_TENTRY(a) {Arch##a::L, Arch##a::dims, Arch##a::act, {launch_fused_train_##a##_F16t, launch_fused_train_##a##_BF16t}, \
                       {launch_fused_train16_##a##_F16t16, launch_fused_train16_##a##_BF16t16}},
static const TrainEntry g_train[] = {V21_TRAIN_ARCH_LIST(V21_TENTRY)};
#undef V21_TENTRY
// routes.h: which compiled fused training kernel serves this stack (-1: none)
int v21::fused_train_arch_of(int L, const int* dims, const int* act) {
  for (size_t e = 0; e < sizeof(g_train) / sizeof(g_train[0]); ++e) {
    bool same = g_train[e].L == L;
    for (int l = 0; same && l <= L; ++l) same = g_train[e].dims[l] == dims[l];
    for (int l = 0; same && l < L; ++l) same = g_train[e].act[l] == act[l];
    if (same) return (int)e;
  }
  return -1;
}
bool v21::fused_train_rt_eligible(int L, const int* dims, const int* act) { return v21::jit_train_eligible(L, dims, act, nullptr); }

// ---------------------------------------------------------------------------------
// trainer (NT path: gemm_nt.h).  Every contraction of a step reads operands whose
// contraction index is contiguous; the producers write the transposed copies.
//   h[l]   (batch x p16(dims[l]))      activations, row-major          (forward A operand, ReLU mask)
//   ht[l]  ((dims[l]+1) x Bp)          activations transposed + a row of ones (weight-gradient A operand)
//   dz[l]  (batch x p16(dims[l]))      gradient w.r.t. pre-activation of layer l-1's output (backward A operand)
//   dzt[l] (dims[l] x Bp)              its transpose                    (weight-gradient B operand)
//   wt[l]  (N x p16(K)) = W^T          forward B operand;   wp[l] (K x p16(N)) = row-padded W: backward B operand
// ---------------------------------------------------------------------------------

void destroy_graphs(v21_trainer* t) {
  for (auto& g : t->graphs) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); }
  t->graphs.clear();
  t->desc_count = 0; t->desc_next = 0; t->desc_iter0 = -1;
}

// routes.h: the route of a step of `rows` rows of this trainer now, and the record of the step that takes it
// a fused training kernel can be launched now: compiled in (archs.h), or instantiated at run time and arrived (jit.hip)
static bool fused_train_ready(const v21_trainer* t) {
  return t->train_arch >= 0 || (t->train_jit && v21::jit_state(t->train_jit) == v21::JIT_READY);
}
static StepRoute step_route(const v21_trainer* t, int rows) {
  const v21_mlp* m = t->mlp;
  return decide_step(t->kind, m->L, m->dims.data(), m->act.data(), rows, t->ctx->nranks, t->capturing, fused_train_ready(t), RouteEnv::read());
}
static void note_route(v21_trainer* t, const StepRoute& r) {
  if (t->capturing) return;
  t->last_route = r; t->fwd_count[r.fwd & 7] += 1; t->upd_count[r.upd & 7] += 1;
}
// r5: HIP-event stamps around the phases of an eager step (include/v21.h: v21_trainer_phase_timing)
static void phase_mark(v21_trainer* t, int idx) {
  // TWO markers per stamped step -- its start and ONE cut point (v21_trainer_phase_timing: `cut`) -- because a HIP event is
  // a barrier packet of its own: five per step (r5's first form) cost 13 us of a 44-us step, an empty interval between two
  // of them read 4.6-5.2 us.  The phases come out as differences of the cumulative times of separate runs, in which the
  // one marker's cost cancels.
  if (!t->phase_on || t->capturing || t->phase_steps >= t->phase_cap || (idx != 0 && idx != t->phase_cut)) return;
  if (idx == 0) t->phase_seen = 0;
  (void)hipEventRecord(t->phase_ev[(size_t)t->phase_steps * 2 + (idx ? 1 : 0)], t->ctx->stream);
  t->phase_seen |= idx ? 2 : 1;
  if (idx && t->phase_seen == 3) t->phase_steps += 1;  // (a step that did not pass both marks -- the joint step's members -- is not counted)
}
// r5: the gradient exchange of an all-reduce step in TWO messages (v21_comm_set_buckets(ctx, 2)): the arena is
// [W0 b0 | W1 b1 | ... | W(L-1) b(L-1) | loss]; bucket 1 = the UPPER layers k .. L-1 and the loss slot (their weight
// gradients are formed first, from the operands the chain launch left), bucket 2 = layers 0 .. k-1.  Bucket 1's all-reduce
// is issued on the communicator stream as soon as its gradients exist and runs while the second weight-gradient launch
// forms bucket 2.  Every rank takes the same decision from the same numbers (the split depends on the stack only), also
// a rank whose share of a batch is empty.  k = the split with the most even parameter counts.
static bool dp_bucketed(const v21_trainer* t) {
  const v21_ctx* c = t->ctx;
  return c->nranks > 1 && !c->sharded && c->buckets == 2 && t->mlp->L >= 2 && !t->capturing;
}
static int dp_split_layer(const v21_mlp* m) {
  int best = 1;
  long long bd = -1;
  for (int k = 1; k < m->L; ++k) {
    const long long lower = m->w_off[k], upper = (long long)m->nparams - lower;
    const long long d = lower > upper ? lower - upper : upper - lower;
    if (bd < 0 || d < bd) { bd = d; best = k; }
  }
  return best;
}
// bucket b (0: arena [lo, hi) just became final on the main stream): its all-reduce goes to the communicator stream
static int dp_exchange_bucket(v21_trainer* t, int b, size_t lo, size_t hi) {
  v21_ctx* c = t->ctx;
  if (c->host_comm || c->null_comm) return comm_allreduce_on(c, t->d_g + lo, hi - lo, c->stream);  // (blocking / nothing: no second stream needed)
  HIPCHK(hipEventRecord(c->ev_bucket[b], c->stream));
  HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_bucket[b], 0));
  return comm_allreduce_on(c, t->d_g + lo, hi - lo, c->comm_stream);
}
// the main stream continues once both buckets have been reduced
static int dp_exchange_join(v21_trainer* t) {
  v21_ctx* c = t->ctx;
  if (c->host_comm || c->null_comm) return V21_OK;
  HIPCHK(hipEventRecord(c->ev_comm_done, c->comm_stream));
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm_done, 0));
  return V21_OK;
}
static int reduce_slabs_range(v21_trainer* t, int nslice, long long lo, long long hi) {
  if (hi <= lo) return V21_OK;
  const long long n4 = (hi - (lo & ~3ll) + 3) / 4;
  hipLaunchKernelGGL(reduce_slabs_range_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, t->ctx->stream, t->d_g,
                     (const float*)t->d_slab, nslice, (long long)t->P + 4, lo, hi);
  HIPCHK(hipGetLastError());
  return V21_OK;
}

static int build_chain32s_jobs(v21_trainer* t);
extern "C" int v21_trainer_create(v21_mlp* m, int precision, int max_batch, v21_trainer** out) {
  if (!m || !out) return fail(V21_ERR_ARG, "null argument");
  if (precision < 0 || precision > 2) return fail(V21_ERR_ARG, "precision %d unknown", precision);
  if (max_batch < 1 || max_batch > (1 << 20)) return fail(V21_ERR_ARG, "max_batch %d out of range", max_batch);
  CHK(use(m->ctx));
  hipStream_t st = m->ctx->stream;
  v21_trainer* t = new v21_trainer();
  t->mlp = m; t->ctx = m->ctx; t->prec = precision; t->max_batch = max_batch; t->P = m->nparams;
  const int L = m->L;
  for (int l = 0; l < L; ++l)
    if (m->act[l] == V21_ACT_GAUSS) t->gl = l;
  if (t->gl == L - 1) { delete t; return fail(V21_ERR_UNSUPPORTED, "a V21_ACT_GAUSS layer cannot be the last layer of a trained stack"); }
  if (m->act[L - 1] != V21_ACT_LINEAR) {  // (the loss gradient is taken w.r.t. the Dense output: no output non-linearity is differentiated)
    delete t;
    return fail(V21_ERR_UNSUPPORTED, "the output layer of a trained stack must be linear (the reference's output Dense has no activation, emulator.py:44)");
  }
  CHK(zalloc(&t->d_g, t->P + kArenaPad, st));
  CHK(zalloc(&t->d_m, t->P + kArenaPad, st));
  CHK(zalloc(&t->d_v, t->P + kArenaPad, st));
  t->Bp = ((long long)max_batch + 31) / 32 * 32 + 32;
  t->d_h.assign(L + 1, nullptr); t->d_ht.assign(L + 1, nullptr);
  t->d_dz.assign(L + 1, nullptr); t->d_dzt.assign(L + 1, nullptr);
  std::vector<float> ones((size_t)t->Bp, 1.0f);
  for (int l = 0; l <= L; ++l) {
    CHK(zalloc(&t->d_h[l], (size_t)(max_batch + 32) * p16(m->dims[l]), st));
    if (l < L) {  // the output activation is never a weight-gradient operand
      CHK(zalloc(&t->d_ht[l], (size_t)(m->dims[l] + 1 + 32) * t->Bp, st));
      HIPCHK(hipMemcpyAsync(t->d_ht[l] + (size_t)m->dims[l] * t->Bp, ones.data(), (size_t)t->Bp * sizeof(float),
                            hipMemcpyHostToDevice, st));  // the row of ones -> bias gradient
    }
    if (l >= 1) {
      CHK(zalloc(&t->d_dz[l], (size_t)(max_batch + 32) * p16(m->dims[l]), st));
      CHK(zalloc(&t->d_dzt[l], (size_t)(m->dims[l] + 32) * t->Bp, st));
    }
  }
  HIPCHK(hipStreamSynchronize(st));  // `ones` is a host temporary
  long long ot = 0, op = 0;
  for (int l = 0; l < L; ++l) {
    t->wt_off.push_back(ot); ot += (long long)(m->nw(l) + 32) * p16(m->dims[l]);
    t->wp_off.push_back(op); op += (long long)(m->dims[l] + 32) * p16(m->nw(l));
  }
  if (t->gl >= 0) {
    const int W2 = m->nw(t->gl);
    CHK(zalloc(&t->d_zs, (size_t)(max_batch + 32) * p16(W2), st));
    CHK(zalloc(&t->d_dzs, (size_t)(max_batch + 32) * p16(W2), st));
    CHK(zalloc(&t->d_dzst, (size_t)(W2 + 32) * t->Bp, st));
    CHK(zalloc(&t->d_klrow, (size_t)max_batch + 32, st));
  }
  CHK(zalloc(&t->d_wt, (size_t)ot + 64, st));
  CHK(zalloc(&t->d_wp, (size_t)op + 64, st));
  CHK(zalloc(&t->d_yb, (size_t)(max_batch + 32) * p16(m->dims[L]), st));
  CHK(zalloc(&t->d_wb, (size_t)max_batch + 32, st));
  CHK(zalloc(&t->d_rowloss, (size_t)max_batch + 32, st));
  CHK(zalloc(&t->d_evalsum, 4, st));
  // what this trainer commits to: csrc/routes.h (the same function answers v21_route_train)
  t->kind = decide_trainer_kind(L, m->dims.data(), m->act.data(), precision, max_batch, RouteEnv::read());
  {  // the 16-bit chain kernel
    const bool ok = t->kind.chain;
    if (ok) {
      long long of = 0, ob = 0;
      for (int l = 0; l < L; ++l) {
        const int K = m->dims[l], N = m->nw(l);
        t->fw_off.push_back(of); of += (long long)((N + 31) / 32) * chain_steps(K) * 512;
        t->bw_off.push_back(ob); ob += (long long)((K + 31) / 32) * chain_steps(N) * 512;
      }
      t->fw_bytes = of * 2; t->bw_bytes = ob * 2;
      HIPCHK(hipMalloc(&t->d_fw, (size_t)of * 2 + kChainStreamSlack)); HIPCHK(hipMemsetAsync(t->d_fw, 0, (size_t)of * 2 + kChainStreamSlack, st));
      HIPCHK(hipMalloc(&t->d_bw, (size_t)ob * 2 + kChainStreamSlack)); HIPCHK(hipMemsetAsync(t->d_bw, 0, (size_t)ob * 2 + kChainStreamSlack, st));
      CHK(zalloc(&t->d_partial, (size_t)(max_batch + 31) / 32 + 4, st));
      HIPCHK(hipMalloc((void**)&t->d_ticket, 16)); HIPCHK(hipMemsetAsync(t->d_ticket, 0, 16, st));
      HIPCHK(hipMalloc((void**)&t->d_stamps, kStampSlots * 8)); HIPCHK(hipMemsetAsync(t->d_stamps, 0, kStampSlots * 8, st));
      // 16-row groups per feature tile of the operand buffers, for whole 128-row blocks: the fused training kernel
      // (fused_train.h) stores every group of a workgroup's 128 rows, pad rows included, and a group past BS would land in
      // the NEXT feature tile's first rows (r4: with BS from 32-row blocks, a ragged step whose last 128-row block reached
      // past it -- 777 rows of max_batch 777 -- had rows 0-63 of the following tile overwritten by whichever workgroup
      // finished last; launch_fused_train checks the bound)
      t->BS = ((long long)max_batch + 127) / 128 * 8 + 2;
      t->d_ht16.assign(L + 1, nullptr); t->d_dzt16.assign(L + 1, nullptr);
      const unsigned short one = precision == V21_PREC_F16 ? 0x3C00 : 0x3F80;
      for (int l = 0; l < L; ++l) {
        const int K = m->dims[l], N = m->nw(l);
        const size_t na = (size_t)((K + 1 + 31) / 32) * t->BS * 512, nb = (size_t)((N + 31) / 32) * t->BS * 512;
        HIPCHK(hipMalloc(&t->d_ht16[l], na * 2)); HIPCHK(hipMemsetAsync(t->d_ht16[l], 0, na * 2, st));
        HIPCHK(hipMalloc(&t->d_dzt16[l + 1], nb * 2)); HIPCHK(hipMemsetAsync(t->d_dzt16[l + 1], 0, nb * 2, st));
        // feature K of the input operand: the constant row of ones that turns [dW; db] into one contraction
        std::vector<unsigned short> tile((size_t)t->BS * 512, 0);
        for (long long b = 0; b < t->BS * 16; ++b)
          tile[(size_t)((b >> 4) * 64 + ((b >> 3) & 1) * 32 + (K & 31)) * 8 + (b & 7)] = one;
        // (only element f%32 == K%32 of the last feature tile is set; the chain kernel writes features < K only)
        HIPCHK(hipMemcpyAsync((char*)t->d_ht16[l] + (size_t)(K >> 5) * t->BS * 1024, tile.data(), tile.size() * 2, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
      }
      t->chain = true;
      // a compiled fused training kernel for this stack (large steps; no variational head), and which of the two: the
      // packed weight stream is in the kernel's own format, so the choice is fixed here (routes.h: decide_trainer_kind)
      t->train_arch = t->kind.train_arch;
      if (t->kind.train_rt) {
        // r5: no compiled kernel, but the template can be instantiated for this stack at run time (jit.hip: fused_train16 only).
        // Asked for here when the trainer is large enough to ever take it; until the code object arrives -- and if it never
        // does (no libhiprtc, a stack that spills) -- the steps stay on the chain kernel.  v21_trainer_jit waits for it.
        const RouteEnv e = RouteEnv::read();
        const int fused_rows = e.fused_train_rows >= 0 ? e.fused_train_rows : 8193;
        if (max_batch >= fused_rows) t->train_jit = v21::jit_request(L, m->dims.data(), m->act.data(), precision | v21::kJitTrain16);
        if (!t->train_jit) t->kind.train_rt = false;
      }
      if (t->train_arch >= 0 || t->kind.train_rt) {
        t->train16 = t->kind.train16;
        int total = 0;
        for (int v = 0; v < 2 * L - 1; ++v) {
          const int l = v < L ? v : 2 * L - 1 - v;
          const int K = v < L ? m->dims[l] : m->dims[l + 1], N = v < L ? m->dims[l + 1] : m->dims[l];
          t->ts_first.push_back(total);
          total += t->train16 ? ((N + 15) / 16) * ((K + 31) / 32 + 1) : ((N + 31) / 32) * ((K + 15) / 16 + 1);
        }
        t->tstream_total = total;
        t->tstream_padded = (total + 7) / 8 * 8;
        HIPCHK(hipMalloc((void**)&t->d_tstream, (size_t)t->tstream_padded * 1024 + kChainStreamSlack));
        HIPCHK(hipMemsetAsync(t->d_tstream, 0, (size_t)t->tstream_padded * 1024 + kChainStreamSlack, st));
      }
      if (!(getenv("V21_DW_BLOCKS") && getenv("V21_DW_BLOCKS")[0] == '0')) {
        // tile order of dw16_adam_kernel: per layer the R x C tile grid in 8 blocks (rb x cb = 8, the shape with the least
        // operand rows per block), the blocks handed to the XCDs largest first onto the least loaded XCD
        std::vector<std::vector<int>> per(8);
        int first = 0;
        for (int l = 0; l < L; ++l) {
          const int R = (m->dims[l] + 1 + 31) / 32, C = (m->nw(l) + 31) / 32;
          int brb = 8, bcb = 1;
          double best = 1e30;
          for (int rb : {1, 2, 4, 8}) {
            const int cb = 8 / rb;
            const double cost = std::ceil((double)R / rb) + std::ceil((double)C / cb);
            if (cost < best) { best = cost; brb = rb; bcb = cb; }
          }
          std::vector<std::vector<int>> blocks;
          for (int i = 0; i < brb; ++i)
            for (int j = 0; j < bcb; ++j) {
              std::vector<int> b;
              for (int ti = R * i / brb; ti < R * (i + 1) / brb; ++ti)
                for (int tj = C * j / bcb; tj < C * (j + 1) / bcb; ++tj) b.push_back(first + ti * C + tj);
              blocks.push_back(b);
            }
          std::sort(blocks.begin(), blocks.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() > b.size(); });
          for (auto& b : blocks) {
            int xmin = 0;
            for (int x = 1; x < 8; ++x) if (per[x].size() < per[xmin].size()) xmin = x;
            per[xmin].insert(per[xmin].end(), b.begin(), b.end());
          }
          first += R * C;
        }
        size_t xper = 0;
        for (auto& v : per) xper = std::max(xper, v.size());
        std::vector<int> order(8 * xper, -1);
        for (int x = 0; x < 8; ++x) std::copy(per[x].begin(), per[x].end(), order.begin() + x * xper);
        HIPCHK(hipMalloc((void**)&t->d_dworder, order.size() * sizeof(int) + 16));
        HIPCHK(hipMemcpyAsync(t->d_dworder, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        t->dw_xper = (int)xper;
      }
    }
  }
  {  // the fp32 chain kernels (train_chain32.h; trainers of small batches -- the reference's 256 rows -- the 8-row kernel
     // of train_chain32s.h, which also carries a variational head)
    const bool ok = t->kind.chain32;
    t->chain32s = t->kind.chain32s;
    if (ok) {
      long long of = 0, ob = 0;  // floats
      for (int l = 0; l < L; ++l) {
        const int K = m->dims[l], N = m->nw(l);
        t->fw_off.push_back(of); of += (long long)t->c32_tiles(N) * t->c32_frags(K) * 256;
        t->bw_off.push_back(ob); ob += (long long)t->c32_tiles(K) * t->c32_frags(N) * 256;
      }
      t->fw_bytes = of * 4; t->bw_bytes = ob * 4;
      HIPCHK(hipMalloc(&t->d_fw, (size_t)of * 4 + kChainStreamSlack)); HIPCHK(hipMemsetAsync(t->d_fw, 0, (size_t)of * 4 + kChainStreamSlack, st));
      HIPCHK(hipMalloc(&t->d_bw, (size_t)ob * 4 + kChainStreamSlack)); HIPCHK(hipMemsetAsync(t->d_bw, 0, (size_t)ob * 4 + kChainStreamSlack, st));
      HIPCHK(hipMalloc((void**)&t->d_ticket, 16)); HIPCHK(hipMemsetAsync(t->d_ticket, 0, 16, st));
      HIPCHK(hipMalloc((void**)&t->d_stamps, kStampSlots * 8)); HIPCHK(hipMemsetAsync(t->d_stamps, 0, kStampSlots * 8, st));
      t->chain32 = true;
    }
  }
  t->max_slices = std::max(1, (max_batch + 127) / 128);  // weight-gradient slices down to 8 batch steps
  CHK(zalloc(&t->d_slab, (size_t)t->max_slices * (t->P + 4), st));
  if (t->chain32s) CHK(build_chain32s_jobs(t));
  *out = t;
  return V21_OK;
}
extern "C" int v21_trainer_destroy(v21_trainer* t) {
  if (!t) return V21_OK;
  hipSetDevice(t->ctx->device);
  hipStreamSynchronize(t->ctx->stream);
  hipFree(t->d_g); hipFree(t->d_m); hipFree(t->d_v);
  for (int i = 0; i < 2; ++i) {
    if (t->d_x[i]) hipFree(t->d_x[i]);
    if (t->d_y[i] && !t->y_is_x[i]) hipFree(t->d_y[i]);
    if (t->d_rw[i]) hipFree(t->d_rw[i]);
  }
  if (t->d_perm) hipFree(t->d_perm);
  for (auto* v : {&t->d_h, &t->d_ht, &t->d_dz, &t->d_dzt})
    for (float* p : *v) if (p) hipFree(p);
  hipFree(t->d_wt); hipFree(t->d_wp);
  hipFree(t->d_yb); hipFree(t->d_wb); hipFree(t->d_rowloss); hipFree(t->d_evalsum);
  destroy_graphs(t);
  for (hipEvent_t e : t->phase_ev) hipEventDestroy(e);
  if (t->d_desc) hipFree(t->d_desc);
  if (t->h_desc) hipHostFree(t->h_desc);
  if (t->d_cur) hipFree(t->d_cur);
  if (t->d_steploss) hipFree(t->d_steploss);
  if (t->d_slab) hipFree(t->d_slab);
  if (t->d_zs) { hipFree(t->d_zs); hipFree(t->d_dzs); hipFree(t->d_dzst); hipFree(t->d_klrow); }
  if (t->d_dworder) hipFree(t->d_dworder);
  if (t->d_tstream) hipFree(t->d_tstream);
  if (t->d_x16) hipFree(t->d_x16);
  if (t->chain32) { hipFree(t->d_fw); hipFree(t->d_bw); hipFree(t->d_ticket); hipFree(t->d_stamps); if (t->d_jobs) hipFree(t->d_jobs); }
  if (t->chain) { hipFree(t->d_fw); hipFree(t->d_bw); hipFree(t->d_partial); hipFree(t->d_ticket); hipFree(t->d_stamps);
    for (void* p : t->d_ht16) if (p) hipFree(p);
    for (void* p : t->d_dzt16) if (p) hipFree(p); }
  delete t;
  return V21_OK;
}
extern "C" int v21_trainer_set_adam(v21_trainer* t, const v21_adam* cfg) {
  if (!t || !cfg) return fail(V21_ERR_ARG, "null argument");
  if (!(cfg->lr >= 0.f) || !(cfg->beta1 >= 0.f && cfg->beta1 < 1.f) || !(cfg->beta2 >= 0.f && cfg->beta2 < 1.f) || !(cfg->eps >= 0.f))
    return fail(V21_ERR_ARG, "bad Adam hyper-parameters");
  if (cfg->beta1 != t->adam.beta1 || cfg->beta2 != t->adam.beta2 || cfg->eps != t->adam.eps)
    destroy_graphs(t);  // (lr only enters through the step descriptors)
  t->adam = *cfg;
  return V21_OK;
}
extern "C" int v21_trainer_set_lr(v21_trainer* t, float lr) { if (!t) return fail(V21_ERR_ARG, "null"); t->adam.lr = lr; return V21_OK; }
extern "C" int v21_trainer_get_lr(v21_trainer* t, float* lr) { if (!t || !lr) return fail(V21_ERR_ARG, "null"); *lr = t->adam.lr; return V21_OK; }

extern "C" int v21_trainer_set_data(v21_trainer* t, int which, const float* x, const float* y, const float* rw, int64_t n) {
  if (!t || !x || !rw) return fail(V21_ERR_ARG, "null argument");
  if (which < 0 || which > 1) return fail(V21_ERR_ARG, "which must be 0 (train) or 1 (val)");
  if (n < 1) return fail(V21_ERR_ARG, "need at least one row");
  CHK(use(t->ctx));
  v21_mlp* m = t->mlp;
  const int din = m->dims[0], dout = m->dims[m->L];
  if (!y && din != dout) return fail(V21_ERR_ARG, "y == NULL (y = x) needs in_dim == out_dim");
  hipStream_t st = t->ctx->stream;
  if (which == 0) { HIPCHK(hipStreamSynchronize(st)); destroy_graphs(t); }  // captured steps hold the old pointers
  if (t->d_x[which]) { HIPCHK(hipFree(t->d_x[which])); t->d_x[which] = nullptr; }
  if (t->d_y[which] && !t->y_is_x[which]) HIPCHK(hipFree(t->d_y[which]));
  t->d_y[which] = nullptr;
  if (t->d_rw[which]) { HIPCHK(hipFree(t->d_rw[which])); t->d_rw[which] = nullptr; }
  HIPCHK(hipMalloc((void**)&t->d_x[which], (size_t)n * din * sizeof(float)));
  HIPCHK(hipMemcpyAsync(t->d_x[which], x, (size_t)n * din * sizeof(float), hipMemcpyHostToDevice, st));
  if (which == 0) {  // the fused training kernels gather the training rows as 16-bit elements (ChainStep::x16)
    if (t->d_x16) { HIPCHK(hipFree(t->d_x16)); t->d_x16 = nullptr; }
    if ((t->train_arch >= 0 || t->kind.train_rt) && !(getenv("V21_TRAIN_X16") && getenv("V21_TRAIN_X16")[0] == '0')) {
      t->ldx16 = (din + 31) / 32 * 32;
      const long long tot = (long long)n * t->ldx16;
      HIPCHK(hipMalloc((void**)&t->d_x16, (size_t)tot * 2));
      hipLaunchKernelGGL(rows_to_half_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float*)t->d_x[0], din, (long long)n,
                         t->d_x16, t->ldx16, t->prec == V21_PREC_BF16 ? 1 : 0);
      HIPCHK(hipGetLastError());
    }
  }
  if (y) {
    HIPCHK(hipMalloc((void**)&t->d_y[which], (size_t)n * dout * sizeof(float)));
    HIPCHK(hipMemcpyAsync(t->d_y[which], y, (size_t)n * dout * sizeof(float), hipMemcpyHostToDevice, st));
    t->y_is_x[which] = false;
  } else {
    t->d_y[which] = t->d_x[which];
    t->y_is_x[which] = true;
  }
  HIPCHK(hipMalloc((void**)&t->d_rw[which], (size_t)n * sizeof(float)));
  HIPCHK(hipMemcpyAsync(t->d_rw[which], rw, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  t->n[which] = n;
  return V21_OK;
}

// power-of-two scale that lifts dL/dz ~ 2 w_i (p - y) / B, w_i ~ 1/D, into the f16 normal range
float grad_opscale(int brows, int dout) {
  const double s = (double)brows * (double)dout / 16.0;
  return (float)std::ldexp(1.0, std::max(0, std::min(24, (int)std::lround(std::log2(std::max(1.0, s))))));
}
float adam_alpha(const v21_adam& a, long long t) {
  // [K] alpha_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in float32
  const float b1p = powf(a.beta1, (float)t), b2p = powf(a.beta2, (float)t);
  return a.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
}




// Adam (do_adam) and/or refresh of the W^T / padded-W copies from the arena
static int adam_and_copies(v21_trainer* t, bool do_adam, float alpha, bool skip_nt = false, int nslab = 1) {
  AdamArgs a = adam_args(t, do_adam, alpha, skip_nt);
  if (nslab > 1) { a.gw = t->d_g; a.slab = t->d_slab; a.nslab = nslab; a.slab_stride = (long long)t->P + 4; }
  hipLaunchKernelGGL(adam_repack_kernel, dim3((unsigned)((t->P + 255) / 256)), dim3(256), 0, t->ctx->stream, a);
  HIPCHK(hipGetLastError());
  t->copies_ok = true;
  t->nt_ok = !skip_nt;
  t->tstream_fresh = a.ts != nullptr;
  if (a.ts) t->n_stream_adam += 1;
  return V21_OK;
}
AdamArgs adam_args(v21_trainer* t, bool do_adam, float alpha, bool skip_nt) {
  v21_mlp* m = t->mlp;
  AdamArgs a{};
  a.w = m->d_w; a.m = t->d_m; a.v = t->d_v; a.g = t->d_g; a.wt = t->d_wt; a.wp = t->d_wp;
  a.n = (long long)t->P; a.alpha = alpha; a.omb1 = 1.0f - t->adam.beta1; a.omb2 = 1.0f - t->adam.beta2;
  a.eps = t->adam.eps; a.do_adam = do_adam ? 1 : 0; a.L = m->L;
  for (int l = 0; l < m->L; ++l) {
    AdamLayer& al = a.lt[l];
    al.w_off = m->w_off[l]; al.wt_off = t->wt_off[l]; al.wp_off = t->wp_off[l];
    al.K = m->dims[l]; al.N = m->nw(l); al.ldwt = p16(al.K); al.ldwp = p16(al.N);
    if (t->chain) {
      al.fw_off = t->fw_off[l]; al.bw_off = t->bw_off[l];
      al.KS = chain_steps(al.K); al.NS = chain_steps(al.N);
    } else if (t->chain32) {
      al.fw_off = t->fw_off[l]; al.bw_off = t->bw_off[l];
      al.KS = t->c32_frags(al.K); al.NS = t->c32_frags(al.N);
    }
  }
  if (t->chain) { a.fw = t->d_fw; a.bw = t->d_bw; a.cprec = t->prec == V21_PREC_F16 ? 1 : 2; }
  if (t->chain32) { a.fw = t->d_fw; a.bw = t->d_bw; a.cprec = t->chain32s ? 4 : 3; }
  if (t->ts_write && t->d_tstream) {  // fused_train.h's stream from the same pass (api_trainer.hip: train_on_rows)
    a.ts = t->d_tstream; a.ts_bf16 = t->prec == V21_PREC_BF16; a.ts_fmt16 = t->train16 ? 1 : 0;
    const int L = m->L, kstep = t->train16 ? 32 : 16;
    for (int l = 0; l < L; ++l) {
      AdamLayer& al = a.lt[l];
      al.tsf = t->ts_first[l]; al.tkf = (m->dims[l] + kstep - 1) / kstep;
      al.tsb = l >= 1 ? t->ts_first[2 * L - 1 - l] : -1; al.tkb = (m->dims[l + 1] + kstep - 1) / kstep;
    }
  }
  a.skip_nt = (skip_nt && (t->chain || t->chain32)) ? 1 : 0;
  if (do_adam && t->chain32 && t->loss_slot_pending > -2) {  // a single-rank f32 chain step: this launch publishes its loss
    a.loss_acc = (unsigned long long*)t->d_ticket; a.loss_out = t->d_g + t->P; a.loss_out2 = t->d_steploss;
    a.loss_slot = t->loss_slot_pending;
  }
  a.sc = step_ctx(t);
  return a;
}
// The end of every eager optimizer step: gradients (and the loss numerator in slot P) summed over the ranks, Adam,
// refreshed weight copies.  Two data-parallel forms:
//   all-reduce:  every rank receives the whole summed arena and applies the identical Adam update;
//   sharded   :  reduce-scatter -> each rank updates ONLY its 1/R slice of (w, m, v) -> all-gather of the
//                updated weights -> every rank rebuilds its packed copies.  Same bytes on the wire, 1/R of the
//                Adam traffic, and on the full xGMI mesh both halves are direct exchanges.  The loss numerator
//                rides in slot P: summed by the reduce-scatter, it is copied into the weight arena's first pad
//                float by its owner, so that the all-gather hands it to everyone.
// `fold` > 1 (single rank): Adam sums that many split-K slabs itself.
int reduce_and_update(v21_trainer* t, bool chain_copies, int fold, bool exchanged) {
  v21_ctx* c = t->ctx;
  hipStream_t st = c->stream;
  const size_t P = t->P;
  if (c->nranks > 1 && c->sharded) {
    const int R = c->nranks;
    const size_t S = (P + 1 + R - 1) / R;  // elements per rank (the last ranks' tails are padding)
    if (R > 64) return fail(V21_ERR_UNSUPPORTED, "sharded Adam: at most 64 ranks");
    if (S * R > P + 1) HIPCHK(hipMemsetAsync(t->d_g + P + 1, 0, (S * R - P - 1) * sizeof(float), st));
    CHK(v21_comm_reduce_scatter_f32(c, t->d_g, S));
    t->iter += 1;
    const size_t lo = std::min(P, (size_t)c->rank * S), hi = std::min(P, lo + S);
    if (hi > lo) {
      AdamArgs a = adam_args(t, true, adam_alpha(t->adam, t->iter), chain_copies);
      a.i0 = (long long)lo; a.n = (long long)(hi - lo); a.no_pack = 1;
      hipLaunchKernelGGL(adam_repack_kernel, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, st, a);
      HIPCHK(hipGetLastError());
    }
    float* w = t->mlp->d_w;
    if (P / S == (size_t)c->rank) HIPCHK(hipMemcpyAsync(w + P, t->d_g + P, sizeof(float), hipMemcpyDeviceToDevice, st));
    CHK(v21_comm_allgather_f32(c, w, S));
    phase_mark(t, 3);  // (sharded form: the slice's Adam pass sits between the two halves of the exchange and is counted with it)
    HIPCHK(hipMemcpyAsync(t->d_g + P, w + P, sizeof(float), hipMemcpyDeviceToDevice, st));  // the loss slot, on every rank
    CHK(adam_and_copies(t, false, 0.f, chain_copies));  // packed copies from the gathered arena
    phase_mark(t, 4);
    return V21_OK;
  }
  if (!exchanged) CHK(v21_comm_allreduce_f32(c, t->d_g, P + 1));  // (exchanged: the step reduced its two buckets itself)
  phase_mark(t, 3);
  t->iter += 1;
  CHK(adam_and_copies(t, true, adam_alpha(t->adam, t->iter), chain_copies, fold));
  phase_mark(t, 4);
  return V21_OK;
}

// need_nt: the caller reads the fp32 copies (per-layer forward/backward); chain steps do not
int ensure_copies(v21_trainer* t, bool need_nt) {
  t->tstream_fresh = false;  // (every step, evaluation, sweep or joint step passes here: only a fused-route step's Adam pass sets it again)
  // the arena may have been rewritten behind our back (set_weights): wpad_ok doubles as the dirty flag
  if (t->copies_ok && t->mlp->wpad_ok && (t->nt_ok || !need_nt)) return V21_OK;
  CHK(adam_and_copies(t, false, 0.f));
  t->mlp->wpad_ok = true;
  return V21_OK;
}

static GaussArgs gauss_args(v21_trainer* t, int rows, bool sample, long long row0) {
  v21_mlp* m = t->mlp;
  const int l = t->gl;
  GaussArgs a{};
  a.zs = t->d_zs; a.ldz = p16(m->nw(l)); a.L = m->dims[l + 1]; a.n = rows;
  a.h = t->d_h[l + 1]; a.ldh = p16(m->dims[l + 1]); a.ht = nullptr; a.ldt = t->Bp;
  a.klrow = t->d_klrow;
  a.dz = t->d_dz[l + 1]; a.lddz = p16(m->dims[l + 1]);
  a.dzs = t->d_dzs; a.lddzs = p16(m->nw(l)); a.dzst = t->d_dzst;
  a.beta = t->kl_weight;
  a.sample = (sample && t->sample) ? 1 : 0;
  a.seed = t->seed; a.step = (unsigned long long)t->iter; a.row0 = (unsigned long long)row0;
  return a;
}

// forward through the stack; h[0] / ht[0] hold the batch.  `sample`: draw eps at the
// variational layer (training); row0 = position of this rank's first row in the global batch
static int trainer_forward(v21_trainer* t, int rows, bool want_t, bool sample = false, long long row0 = 0) {
  v21_mlp* m = t->mlp;
  for (int l = 0; l < m->L; ++l) {
    const bool gauss = m->act[l] == V21_ACT_GAUSS;
    NtGroup grp{};
    grp.count = 1;
    NtArgs& g = grp.p[0];
    g.A = t->d_h[l]; g.lda = p16(m->dims[l]);
    g.B = t->d_wt + t->wt_off[l]; g.ldb = p16(m->dims[l]);
    g.C = gauss ? t->d_zs : t->d_h[l + 1]; g.ldc = p16(m->nw(l));
    g.CT = (want_t && l + 1 < m->L && !gauss) ? t->d_ht[l + 1] : nullptr; g.ldct = t->Bp;
    g.M = rows; g.N = m->nw(l); g.K = m->dims[l];
    g.bias = m->d_w + m->b_off[l];
    g.ep = m->act[l] == V21_ACT_RELU ? NT_FWD_RELU : NT_FWD;
    g.nz = 1;
    CHK(launch_nt(t->prec, grp, t->ctx->stream));
    if (gauss) {  // z = z_mean + exp(z_log_var / 2) eps -> h[l+1] (and its transpose), kl_weight * KL_i -> klrow
      GaussArgs a = gauss_args(t, rows, sample, row0);
      a.ht = want_t ? t->d_ht[l + 1] : nullptr;
      hipLaunchKernelGGL(gauss_sample_kernel, dim3((rows + 3) / 4), dim3(256), 0, t->ctx->stream, a);
      HIPCHK(hipGetLastError());
    }
  }
  return V21_OK;
}

// one optimizer step on the batch already gathered into h[0]/ht[0], yb, wb
static int trainer_step(v21_trainer* t, const float* yb, long long ldy, int rows, int brows, float* loss_out,
                        long long row0) {
  v21_mlp* m = t->mlp;
  hipStream_t st = t->ctx->stream;
  const int L = m->L, dout = m->dims[L];
  if (rows > t->max_batch) return fail(V21_ERR_ARG, "batch of %d rows exceeds max_batch %d", rows, t->max_batch);
  // single rank, an epoch's per-step slot: the sum kernel writes it itself (a device-to-device copy per step is a launch)
  const bool in_table = t->ctx->nranks == 1 && rows > 0 && loss_out && t->d_steploss && loss_out >= t->d_steploss &&
                        loss_out < t->d_steploss + t->steploss_cap;
  phase_mark(t, 0);
  if (rows > 0) {
    note_route(t, step_route(t, rows));
    CHK(ensure_copies(t));
    CHK(trainer_forward(t, rows, true, true, row0));
    const int wpb = 4;  // waves (rows) per block
    hipLaunchKernelGGL(loss_grad_t_kernel, dim3((rows + wpb - 1) / wpb), dim3(64 * wpb), 0, st, t->d_h[L], p16(dout), yb,
                       ldy, t->d_wb, t->d_dz[L], p16(dout), t->d_dzt[L], t->Bp, t->d_rowloss, rows, dout,
                       2.0f / (float)brows, (const float*)t->d_klrow);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, t->d_rowloss, rows, t->d_g + t->P, 0, t->d_steploss,
                       step_ctx(t), in_table ? (int)(loss_out - t->d_steploss) : -1);
    HIPCHK(hipGetLastError());
    // weight gradients contract over the batch: slices of <= kNtMaxKPerWg rows -> slabs
    int nslice = (rows + kNtMaxKPerWg - 1) / kNtMaxKPerWg;
    const int k_chunk = ((rows + nslice - 1) / nslice + 15) / 16 * 16;
    nslice = (rows + k_chunk - 1) / k_chunk;
    const long long slab_stride = (long long)t->P + 4;
    const float gs = grad_opscale(brows, dout);
    for (int l = L - 1; l >= 0; --l) {
      const int K = m->dims[l], N = m->nw(l);
      const bool gauss = l == t->gl;  // gradient w.r.t. this layer's Dense output: dzs / dzst instead of dz[l+1]
      if (gauss) {  // dz[l+1] = dL/dz  ->  dL/d[z_mean | z_log_var] (+ the KL term's own gradient)
        GaussArgs a = gauss_args(t, rows, true, row0);
        a.beta = t->kl_weight / (float)brows;
        hipLaunchKernelGGL(gauss_sample_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
        HIPCHK(hipGetLastError());
      }
      NtGroup grp{};
      NtArgs& g = grp.p[0];  // [dW; db] = [H^T; 1^T] dZ
      g.A = t->d_ht[l]; g.lda = t->Bp;
      g.B = gauss ? t->d_dzst : t->d_dzt[l + 1]; g.ldb = t->Bp;
      g.C = (nslice > 1 ? t->d_slab : t->d_g) + m->w_off[l]; g.ldc = N;
      g.M = K + 1; g.N = N; g.K = rows;
      g.ep = NT_DW; g.nz = nslice; g.k_chunk = k_chunk; g.slab_stride = slab_stride;
      g.b_scale = gs; g.out_scale = 1.0f / gs;
      grp.count = 1;
      if (l > 0) {  // dH = dZ W^T, masked by the ReLU of the layer below -> dz[l], dzt[l]
        NtArgs& d = grp.p[1];
        d.A = gauss ? t->d_dzs : t->d_dz[l + 1]; d.lda = p16(N);
        d.B = t->d_wp + t->wp_off[l]; d.ldb = p16(N);
        d.C = t->d_dz[l]; d.ldc = p16(K);
        d.CT = t->d_dzt[l]; d.ldct = t->Bp;
        d.M = rows; d.N = K; d.K = N;
        d.mask = t->d_h[l]; d.ldmask = p16(K);
        d.ep = m->act[l - 1] == V21_ACT_RELU ? NT_DX_MASK : NT_DX;
        d.nz = 1;
        d.a_scale = gs; d.out_scale = 1.0f / gs;
        grp.count = 2;
      }
      CHK(launch_nt(t->prec, grp, st));
    }
    if (nslice > 1) {
      const long long n4 = ((long long)t->P + 3) / 4;
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, t->d_g,
                         (const float*)t->d_slab, nslice, slab_stride, (long long)t->P);
      HIPCHK(hipGetLastError());
    }
  } else {
    HIPCHK(hipMemsetAsync(t->d_g, 0, (t->P + 1) * sizeof(float), st));
  }
  if (t->capturing) {  // recorded, not run: iteration count, step size and loss slot come from the descriptors
    CHK(adam_and_copies(t, true, 0.f));
    return V21_OK;
  }
  phase_mark(t, 1); phase_mark(t, 2);  // (per-layer path: forward, loss and every backward launch are reported as the first phase)
  CHK(reduce_and_update(t, false, 1));
  if (loss_out && !in_table) HIPCHK(hipMemcpyAsync(loss_out, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
  invalidate_streams(m);
  m->wpad_ok = true;  // ... but our own copies were just refreshed
  return V21_OK;
}


// forward + loss + activation gradients of the chain path: ONE launch (train_chain.h)
ChainModel chain_model(v21_trainer* t) {
  v21_mlp* m = t->mlp;
  const int L = m->L;
  ChainModel a{};
  a.L = L;
  int mt = 0;
  for (int l = 0; l < L; ++l) {
    ChainLayer& c = a.lt[l];
    c.K = m->dims[l]; c.N = m->nw(l);
    c.gauss = m->act[l] == V21_ACT_GAUSS;
    c.KS = chain_steps(c.K); c.NT = (c.N + 31) / 32;
    c.NS = chain_steps(c.N); c.KT = (c.K + 31) / 32;
    c.relu = m->act[l] == V21_ACT_RELU;
    c.mask_tile = -1;
    if (c.relu && l + 1 < L) { c.mask_tile = mt; mt += c.NT; }
    c.fw_off = t->fw_off[l] / 8; c.bw_off = t->bw_off[l] / 8;
    c.b_off = m->b_off[l];
    c.ht16 = t->d_ht16[l]; c.dzt16 = t->d_dzt16[l + 1];
  }
  a.fw = t->d_fw; a.bw = t->d_bw; a.w = m->d_w;
  a.fw_bytes = t->fw_bytes; a.bw_bytes = t->bw_bytes;
  a.BS = t->BS;
  a.loss_acc = (unsigned long long*)t->d_ticket;
  a.stamps = t->stamps_on ? t->d_stamps : nullptr;
  a.zcap_layer = -1;
  if (t->gl >= 0) { a.kl_weight = t->kl_weight; a.sample = t->sample; a.seed = t->seed; a.step = (unsigned long long)t->iter; }
  return a;
}
ChainStep chain_step(const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                            const int* d_idx, long long first, int rows, int brows, int dout,
                            const v21_trainer* vae, long long row0) {
  ChainStep st{};
  st.x = x; st.ldx = ldx; st.y = y; st.ldy = ldy; st.rw = rw; st.idx = d_idx; st.first = first;
  st.rows = rows;
  st.scale = 2.0f / (float)brows;
  st.gs = grad_opscale(brows, dout);
  st.inv_b = 1.0f / (float)brows;
  st.row0 = (unsigned long long)row0;
  if (vae) st.sc = step_ctx(vae);
  return st;
}
// every instantiation of the 16-bit chain kernels (train_chain.h: FEAT) needs the dynamic-LDS attribute once per device
template <class P>
static int chain_attr_of() {
  const void* fs[] = {(const void*)train_chain_kernel<P, 0>, (const void*)train_chain_kernel<P, kChainGauss>,
                      (const void*)train_chain_kernel<P, kChainFwd>, (const void*)train_chain_kernel<P, kChainFwd | kChainGauss>,
                      (const void*)train_chain_kernel<P, kChainFwd | kChainOut | kChainGauss>,
                      (const void*)train_chain_group_kernel<P, false>, (const void*)train_chain_group_kernel<P, true>,
                      (const void*)train_chain_joint_kernel<P, false>, (const void*)train_chain_joint_kernel<P, true>};
  for (const void* f : fs) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kChainLdsBytes));
  return V21_OK;
}
void launch_joint32_kernel(int rpw, bool gauss, dim3 grid, dim3 block, hipStream_t st, const ChainModel* tab, const ChainStep& sa,
                                  const ChainStep& sb) {
  if (rpw == 4) {
    if (gauss) hipLaunchKernelGGL((train_chain32s_joint_kernel<4, true>), grid, block, kC32sLdsBytes, st, tab, sa, sb);
    else hipLaunchKernelGGL((train_chain32s_joint_kernel<4, false>), grid, block, kC32sLdsBytes, st, tab, sa, sb);
  } else {
    if (gauss) hipLaunchKernelGGL((train_chain32s_joint_kernel<8, true>), grid, block, kC32sLdsBytes, st, tab, sa, sb);
    else hipLaunchKernelGGL((train_chain32s_joint_kernel<8, false>), grid, block, kC32sLdsBytes, st, tab, sa, sb);
  }
}
void launch_joint_kernel(int prec, bool gauss, dim3 grid, dim3 block, hipStream_t st, const ChainModel* tab, const ChainStep& sa,
                                const ChainStep& sb) {
  if (prec == V21_PREC_F16) {
    if (gauss) hipLaunchKernelGGL((train_chain_joint_kernel<PrecF16, true>), grid, block, kChainLdsBytes, st, tab, sa, sb);
    else hipLaunchKernelGGL((train_chain_joint_kernel<PrecF16, false>), grid, block, kChainLdsBytes, st, tab, sa, sb);
  } else {
    if (gauss) hipLaunchKernelGGL((train_chain_joint_kernel<PrecBF16, true>), grid, block, kChainLdsBytes, st, tab, sa, sb);
    else hipLaunchKernelGGL((train_chain_joint_kernel<PrecBF16, false>), grid, block, kChainLdsBytes, st, tab, sa, sb);
  }
}
// one model's chain launch: training / validation (FEAT 0 or kChainGauss) or FORWARD mode (kChainOut | kChainGauss)
template <class P>
static void launch_chain_kernel(int feat, dim3 grid, dim3 block, hipStream_t st, const ChainArgs& a) {
  if (feat == 0) hipLaunchKernelGGL((train_chain_kernel<P, 0>), grid, block, kChainLdsBytes, st, a);
  else if (feat == kChainGauss) hipLaunchKernelGGL((train_chain_kernel<P, kChainGauss>), grid, block, kChainLdsBytes, st, a);
  else if (feat == kChainFwd) hipLaunchKernelGGL((train_chain_kernel<P, kChainFwd>), grid, block, kChainLdsBytes, st, a);
  else if (feat == (kChainFwd | kChainGauss)) hipLaunchKernelGGL((train_chain_kernel<P, kChainFwd | kChainGauss>), grid, block, kChainLdsBytes, st, a);
  else hipLaunchKernelGGL((train_chain_kernel<P, kChainFwd | kChainOut | kChainGauss>), grid, block, kChainLdsBytes, st, a);
}
// FORWARD mode of the 16-bit chain kernel for api_forward.hip (forward_chain): a stack without a trainer
void launch_chain_forward_mode(int prec, dim3 grid, dim3 block, hipStream_t st, const ChainArgs& a) {
  if (prec == V21_PREC_F16) launch_chain_kernel<PrecF16>(kChainFwd | kChainOut | kChainGauss, grid, block, st, a);
  else launch_chain_kernel<PrecBF16>(kChainFwd | kChainOut | kChainGauss, grid, block, st, a);
}
int chain_attr(int prec) {
  static bool done_dev[64][3] = {};  // per (device, precision): function attributes are per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool* done = done_dev[dev & 63];
  if (done[prec]) return V21_OK;
  if (prec == V21_PREC_F32) {
    for (const void* f : {(const void*)train_chain32_kernel<0>, (const void*)train_chain32_kernel<kChainFwd>, (const void*)train_chain32_kernel<kChainFwd | kChainOut>})
      HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kC32LdsBytes));
    for (const void* f : {(const void*)train_chain32s_kernel<8, false>, (const void*)train_chain32s_kernel<4, false>,
                          (const void*)train_chain32s_kernel<8, true>, (const void*)train_chain32s_kernel<4, true>,
                          (const void*)train_chain32s_group_kernel<8, false>, (const void*)train_chain32s_group_kernel<4, false>,
                          (const void*)train_chain32s_group_kernel<8, true>, (const void*)train_chain32s_group_kernel<4, true>,
                          (const void*)train_chain32s_joint_kernel<8, false>, (const void*)train_chain32s_joint_kernel<4, false>,
                          (const void*)train_chain32s_joint_kernel<8, true>, (const void*)train_chain32s_joint_kernel<4, true>})
      HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kC32sLdsBytes));
  } else if (prec == V21_PREC_F16) {
    CHK(chain_attr_of<PrecF16>());
  } else {
    CHK(chain_attr_of<PrecBF16>());
  }
  done[prec] = true;
  return V21_OK;
}
// prefetcher workgroups per XCD for a launch of `models` x `ncons` row-block workgroups: the CUs the row blocks leave idle
int chain_prefetchers(int ncons, int models) {
  if (models > 1) return 0;  // a sweep: measured slower with them (8 models, 24 prefetchers each: 106 k -> 95 k model-steps/s)
  const int idle = 256 - ncons;
  static const char* env = getenv("V21_CHAIN_PREF");  // (diagnosis: prefetcher workgroups per XCD, 0 = none)
  if (env) return std::max(0, std::min(atoi(env), idle / 8));
  // (per XCD: none 45.6 us per f16 step at 4,096 rows, 2-4 43.4-43.6, 8 43.8-43.9, 16 44.2; f32 at batch 256: 46.4 / 42.6-42.8 / 42.9 / 43.4)
  return idle >= 8 ? std::min(4, idle / 8) : 0;
}
// Large steps (fused_train.h): the packed stream of the virtual stack -- forward layers, then the activation-gradient
// layers with the transposed weights -- is rebuilt from the arena (the previous step's Adam moved it), then one launch
// carries 128 rows per workgroup through forward pass, loss and activation gradients.
// `xr` (or nullptr): filled when the step's rows lie in the resident training set -- the kernel then does NOT flush its layer-0
// operand and the weight-gradient launch gathers the set's 16-bit rows instead (train_chain.h: DwXRows)
static int launch_fused_train(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                              const int* d_idx, long long first, int rows, int brows, long long row0, bool stream_fresh,
                              DwXRows* xr = nullptr) {
  v21_mlp* m = t->mlp;
  const int L = m->L;
  hipStream_t st = t->ctx->stream;
  // (the previous step's Adam pass wrote the stream itself when that step took this route too: adam_repack_element)
  if ((long long)((rows + kTrainRowsPerWg - 1) / kTrainRowsPerWg) * (kTrainRowsPerWg / 16) > t->BS)
    return fail(V21_ERR_STATE, "fused training step of %d rows: its 128-row blocks reach past the operand buffers (%lld groups of 16)", rows, t->BS);
  t->n_fused_steps += 1;
  if (!stream_fresh) {
  t->n_stream_packs += 1;
  PackArgs pa{};
  pa.w = m->d_w; pa.mean = nullptr; pa.stream = t->d_tstream;
  pa.L = 2 * L - 1; pa.total = t->tstream_total; pa.padded = t->tstream_padded;
  pa.fpi = 16; pa.epi = 8; pa.esize = 2; pa.is_bf16 = t->prec == V21_PREC_BF16; pa.all_hidden = 1; pa.fmt16 = t->train16 ? 1 : 0;
  int f = 0;
  for (int v = 0; v < 2 * L - 1; ++v) {
    const int l = v < L ? v : 2 * L - 1 - v;
    PackLayer& pl = pa.lt[v];
    pl.K = v < L ? m->dims[l] : m->dims[l + 1]; pl.N = v < L ? m->dims[l + 1] : m->dims[l];
    if (t->train16) { pl.ks = (pl.K + 31) / 32; pl.nt = (pl.N + 15) / 16; }
    else { pl.ks = (pl.K + 15) / 16; pl.nt = (pl.N + 31) / 32; }
    pl.w_off = m->w_off[l]; pl.b_off = m->b_off[l];
    pl.flags = v < L ? 0 : 3;  // activation-gradient layer: transposed weights, no bias
    pl.first = f;
    f += pl.nt * (pl.ks + 1);
  }
  hipLaunchKernelGGL(pack_stream_kernel, dim3((pa.padded + 3) / 4), dim3(256), 0, st, pa);
  HIPCHK(hipGetLastError());
  }
  ChainArgs a{};
  static_cast<ChainModel&>(a) = chain_model(t);
  static_cast<ChainStep&>(a) = chain_step(x, ldx, y, ldy, rw, d_idx, first, rows, brows, m->dims[L], nullptr, row0);
  a.stamps = t->stamps_on ? t->d_stamps : nullptr;  // (written by diagnostic builds only: -DV21_T_STAMPS)
  a.fw = t->d_tstream; a.fw_bytes = (long long)t->tstream_padded * 1024;
  // rows of the resident training set: the kernels gather their 16-bit copy
  if (t->d_x16 && ldx == m->dims[0] && x >= t->d_x[0] && x < t->d_x[0] + (size_t)t->n[0] * ldx && (x - t->d_x[0]) % ldx == 0) {
    a.x16 = t->d_x16 + (size_t)((x - t->d_x[0]) / ldx) * t->ldx16; a.ldx16 = t->ldx16;
    if (xr) {
      xr->x16 = a.x16; xr->ld = a.ldx16; xr->idx = d_idx; xr->first = first; xr->rows = rows;
      a.lt[0].ht16 = nullptr;
    }
  }
  if (t->train_arch < 0) {  // instantiated at run time (jit.hip)
    const hipError_t e = v21::jit_launch_train(t->train_jit, t->ctx->device, a, st);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      std::string why;
      v21::jit_state(t->train_jit, &why);
      t->n_fused_steps -= 1;
      return fail(V21_ERR_UNSUPPORTED, "run-time fused training kernel: %s (%s)", why.c_str(), hipGetErrorString(e));
    }
    return V21_OK;
  }
  if (t->train16) HIPCHK(g_train[t->train_arch].fn16[t->prec == V21_PREC_F16 ? 0 : 1](a, st));
  else HIPCHK(g_train[t->train_arch].fn[t->prec == V21_PREC_F16 ? 0 : 1](a, st));
  return V21_OK;
}
static int launch_chain(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                        const int* d_idx, long long first, int rows, int brows, long long row0) {
  ChainArgs a{};
  static_cast<ChainModel&>(a) = chain_model(t);
  static_cast<ChainStep&>(a) = chain_step(x, ldx, y, ldy, rw, d_idx, first, rows, brows, t->mlp->dims[t->mlp->L], t, row0);
  CHK(chain_attr(t->prec));
  a.ncons = ((rows + 31) / 32 + 7) / 8 * 8;  // whole rounds of the 8 XCDs
  a.npref = chain_prefetchers(a.ncons, 1);
  const dim3 grid(a.ncons + 8 * a.npref), block(64 * kChainWaves);
  // (train_chain.h: FEAT -- a trainer's launch never needs the joint step's or FORWARD mode's code, and the variational
  //  head's only when the stack has one; V21_CHAIN_PLAIN=0: everything through the variational instantiation)
  static const bool plain_ok = !(getenv("V21_CHAIN_PLAIN") && getenv("V21_CHAIN_PLAIN")[0] == '0');
  const int feat = plain_ok && t->gl < 0 ? 0 : kChainGauss;
  if (t->prec == V21_PREC_F16) launch_chain_kernel<PrecF16>(feat, grid, block, t->ctx->stream, a);
  else launch_chain_kernel<PrecBF16>(feat, grid, block, t->ctx->stream, a);
  HIPCHK(hipGetLastError());
  return V21_OK;
}

// ---- the fp32 chain (train_chain32.h)
ChainModel chain_model32(v21_trainer* t) {
  v21_mlp* m = t->mlp;
  const int L = m->L;
  ChainModel a{};
  a.L = L;
  int mt = 0;
  for (int l = 0; l < L; ++l) {
    ChainLayer& c = a.lt[l];
    c.K = m->dims[l]; c.N = m->nw(l);
    c.KS = t->c32_frags(c.K); c.NT = t->c32_tiles(c.N);   // fragments per tile (32 wide; 64 in the 8-row kernel)
    c.NS = t->c32_frags(c.N); c.KT = t->c32_tiles(c.K);
    c.relu = m->act[l] == V21_ACT_RELU;
    c.mask_tile = -1;
    if (c.relu && l + 1 < L) { c.mask_tile = mt; mt += c.NT; }
    c.fw_off = t->fw_off[l] / 4; c.bw_off = t->bw_off[l] / 4;  // units of one lane's 16 bytes
    c.b_off = m->b_off[l];
    c.ht16 = t->d_ht[l]; c.dzt16 = t->d_dzt[l + 1];           // fp32, feature-major, batch contiguous (pitch Bp)
  }
  a.fw = t->d_fw; a.bw = t->d_bw; a.w = m->d_w;
  a.fw_bytes = t->fw_bytes; a.bw_bytes = t->bw_bytes;
  a.BS = t->Bp;
  a.loss_acc = (unsigned long long*)t->d_ticket;
  a.stamps = t->stamps_on ? t->d_stamps : nullptr;
  a.zcap_layer = -1;
  a.jobs = t->d_jobs;
  if (t->gl >= 0) {
    a.lt[t->gl].gauss = 1;
    a.kl_weight = t->kl_weight; a.sample = t->sample; a.seed = t->seed; a.step = (unsigned long long)t->iter;
  }
  return a;
}
// the 8-row kernel's job table (once per trainer: it depends on the layer widths only)
static int build_chain32s_jobs(v21_trainer* t) {
  const ChainModel a = chain_model32(t);
  std::vector<C32sJob> tab((size_t)2 * a.L * kC32sWaves);
  c32s_build_jobs(a, tab.data());
  // the kernel follows these rows without range checks: every chunk and bias a row names must lie inside the buffers
  // allocated above, or the trainer is not created (the alternative is a GPU memory fault in the first step)
  if (const char* why = c32s_validate_jobs(a, tab.data(), t->fw_bytes / 16, t->bw_bytes / 16, (long long)t->P))
    return fail(V21_ERR_STATE, "%s", why);
  HIPCHK(hipMalloc((void**)&t->d_jobs, tab.size() * sizeof(C32sJob)));
  HIPCHK(hipMemcpyAsync(t->d_jobs, tab.data(), tab.size() * sizeof(C32sJob), hipMemcpyHostToDevice, t->ctx->stream));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  return V21_OK;
}
int launch_chain32_args(ChainArgs& a, hipStream_t st, bool small, int rows_per_wg) {
  CHK(chain_attr(V21_PREC_F32));
  if (small) {  // the 8-row kernel (train_chain32s.h), or its 4-row form
    // (routes.h decides for a trainer's steps; evaluation passes and other callers: by the row count, V21_C32S_ROWS forces)
    const int force_rows = rows_per_wg ? rows_per_wg : RouteEnv::read().c32s_rows;
    const int rpw = force_rows == 4 || force_rows == 8 ? force_rows : (a.rows <= kC32sRows4Max ? 4 : 8);
    a.ncons = (int)((((long long)a.rows + rpw - 1) / rpw + 7) / 8 * 8);
    a.npref = chain_prefetchers(a.ncons, 1);
    bool gauss = false;  // (train_chain32s.h: GAUSS -- the variational head's code only where the stack has one)
    for (int l = 0; l < a.L; ++l) gauss = gauss || a.lt[l].gauss;
    const dim3 grid(a.ncons + 8 * a.npref), block(64 * kC32sWaves);
    if (rpw == 4) {
      if (gauss) hipLaunchKernelGGL((train_chain32s_kernel<4, true>), grid, block, kC32sLdsBytes, st, a);
      else hipLaunchKernelGGL((train_chain32s_kernel<4, false>), grid, block, kC32sLdsBytes, st, a);
    } else {
      if (gauss) hipLaunchKernelGGL((train_chain32s_kernel<8, true>), grid, block, kC32sLdsBytes, st, a);
      else hipLaunchKernelGGL((train_chain32s_kernel<8, false>), grid, block, kC32sLdsBytes, st, a);
    }
    HIPCHK(hipGetLastError());
    return V21_OK;
  }
  a.ncons = (int)((((long long)a.rows + kC32Rows - 1) / kC32Rows + 7) / 8 * 8);  // whole rounds of the 8 XCDs
  a.npref = chain_prefetchers(a.ncons, 1);
  const dim3 grid(a.ncons + 8 * a.npref), block(64 * kC32Waves);
  // (train_chain32.h: FEAT -- FORWARD mode, validation, training)
  if (a.out) hipLaunchKernelGGL((train_chain32_kernel<kChainFwd | kChainOut>), grid, block, kC32LdsBytes, st, a);
  else if (a.fwd_only) hipLaunchKernelGGL((train_chain32_kernel<kChainFwd>), grid, block, kC32LdsBytes, st, a);
  else hipLaunchKernelGGL((train_chain32_kernel<0>), grid, block, kC32LdsBytes, st, a);
  HIPCHK(hipGetLastError());
  return V21_OK;
}


// weight gradients of chain-mode trainers: problems in groups of <= 16 per launch
void dw16_problems(v21_trainer* t, int rows, int brows, int* nslice_out, std::vector<Dw16Args>& probs,
                          float* loss_out2) {
  v21_mlp* m = t->mlp;
  const int L = m->L;
  const int steps = (rows + 15) / 16;
  // large batches: 8 slices, one per XCD (train_chain_kernel leaves slice z's operands in XCD z's L2)
  int nslice = steps >= 64 ? 8 : (steps + 31) / 32;
  const int sps = (steps + nslice - 1) / nslice;
  nslice = (steps + sps - 1) / sps;
  const float gs = grad_opscale(brows, m->dims[L]);
  for (int l = 0; l < L; ++l) {
    Dw16Args g{};
    g.A = t->d_ht16[l]; g.B = t->d_dzt16[l + 1];
    g.C = (nslice > 1 ? t->d_slab : t->d_g) + m->w_off[l]; g.ldc = m->nw(l);
    g.M = m->dims[l] + 1; g.N = m->nw(l);
    g.nx = (g.N + 63) / 64; g.ny = (g.M + 63) / 64; g.nz = nslice;
    g.steps = steps; g.steps_per_slice = sps; g.BS = t->BS;
    g.slab_stride = (long long)t->P + 4;
    g.out_scale = 1.0f / gs;
    if (l == 0) {
      g.loss_acc = (unsigned long long*)t->d_ticket; g.loss_out = t->d_g + t->P; g.loss_out2 = loss_out2;
      if (t->capturing) { g.loss_out2 = t->d_steploss; g.sc = step_ctx(t); }
    }
    probs.push_back(g);
  }
  *nslice_out = nslice;
}
static int dw16_attr(int prec) {  // (function attributes are per device; set outside any stream capture)
  static bool attr_done_dev[64][3] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool* attr_done = attr_done_dev[dev & 63];
  if (!attr_done[prec]) {
    if (prec == V21_PREC_F16)
      HIPCHK(hipFuncSetAttribute((const void*)gemm_dw16_lds_kernel<PrecF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kDwLdsTotal));
    else
      HIPCHK(hipFuncSetAttribute((const void*)gemm_dw16_lds_kernel<PrecBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kDwLdsTotal));
    attr_done[prec] = true;
  }
  return V21_OK;
}
int launch_dw16(int prec, const std::vector<Dw16Args>& probs, hipStream_t st, const DwXRows* xr) {
  // large batches: 128x128 tiles staged through LDS (half the bytes pulled into a CU per MFMA)
  const bool big = !probs.empty() && probs[0].steps >= 64;
  for (const Dw16Args& p : probs)
    if (!p.A && !(big && xr && xr->x16)) return fail(V21_ERR_STATE, "weight-gradient problem without an operand");
  if (big) CHK(dw16_attr(prec));
  for (size_t o = 0; o < probs.size(); o += kNtMaxGroup) {
    Dw16Group grp{};
    grp.count = (int)std::min<size_t>(kNtMaxGroup, probs.size() - o);
    int blocks = 0;
    for (int i = 0; i < grp.count; ++i) {
      grp.p[i] = probs[o + i];
      if (big) { grp.p[i].nx = (grp.p[i].N + 127) / 128; grp.p[i].ny = (grp.p[i].M + 127) / 128; }
      grp.first[i] = blocks;  // tiles; every problem of a step has the same slice count
      blocks += grp.p[i].nx * grp.p[i].ny;
    }
    grp.first[grp.count] = blocks;
    if (xr) grp.xr = *xr;
    blocks *= grp.p[0].nz;
    if (blocks <= 0) continue;
    const dim3 grid((blocks + 7) / 8 * 8);  // whole rounds of the 8 XCDs (the kernels remap block ids XCD-wise)
    if (big) {
      if (prec == V21_PREC_F16) hipLaunchKernelGGL(gemm_dw16_lds_kernel<PrecF16>, grid, dim3(kDwThreads), kDwLdsTotal, st, grp);
      else hipLaunchKernelGGL(gemm_dw16_lds_kernel<PrecBF16>, grid, dim3(kDwThreads), kDwLdsTotal, st, grp);
    } else {
      if (prec == V21_PREC_F16) hipLaunchKernelGGL(gemm_dw16_kernel<PrecF16>, grid, dim3(256), 0, st, grp);
      else hipLaunchKernelGGL(gemm_dw16_kernel<PrecBF16>, grid, dim3(256), 0, st, grp);
    }
    HIPCHK(hipGetLastError());
  }
  return V21_OK;
}

// ---- single rank: weight gradients + Adam + packed copies in ONE launch (dw_adam.h)
static void dw_adam_model(v21_trainer* t, DwAdamModel& md) {
  v21_mlp* m = t->mlp;
  memset(&md, 0, sizeof(md));  // (the device tables are compared bytewise: padding included)
  md.L = m->L;
  md.omb1 = 1.0f - t->adam.beta1; md.omb2 = 1.0f - t->adam.beta2; md.eps = t->adam.eps;
  md.cprec = t->prec == V21_PREC_F16 ? 1 : 2;
  int nb = 0;
  for (int l = 0; l < m->L; ++l) {
    DwAdamLayer& d = md.lt[l];
    d.A = t->d_ht16[l]; d.B = t->d_dzt16[l + 1]; d.BS = t->BS;
    d.w = m->d_w + m->w_off[l]; d.m = t->d_m + m->w_off[l]; d.v = t->d_v + m->w_off[l]; d.g = t->d_g + m->w_off[l];
    d.fw = t->d_fw; d.bw = t->d_bw; d.fw_off = t->fw_off[l]; d.bw_off = t->bw_off[l];
    d.K = m->dims[l]; d.N = m->nw(l);
    d.KS = chain_steps(d.K); d.NS = chain_steps(d.N);
    d.nt = (d.N + 31) / 32;
    d.first = nb;
    nb += ((d.K + 1 + 31) / 32) * d.nt;
    if (l == 0) {
      d.loss_acc = (unsigned long long*)t->d_ticket; d.loss_out = t->d_g + t->P;
      d.loss_out2 = t->d_steploss;  // (may be null: then no step asks for a slot)
    }
  }
  md.nblk = nb;
}
static int launch_dw_adam(v21_trainer* t, int rows, int brows, float alpha, int slot = -1) {
  DwAdamModel md;
  dw_adam_model(t, md);
  // (from ~2k rows on: below that the operands are small and the contiguous runs balance the XCDs better --
  //  r3, autoencoder stack, f16: 4,096 rows 46.9 -> 45.7 us per step, 1,024 rows 36.3 -> 37.4)
  if (rows >= 2048) { md.order = t->d_dworder; md.xper = t->dw_xper; }
#ifdef V21_CHAIN_FINE
  md.dbg = t->stamps_on ? t->d_stamps + 1024 : nullptr;
#endif
  DwAdamStep st{};
  st.steps = (rows + 15) / 16;
  st.slot = slot;
  st.alpha[0] = alpha;
  st.out_scale[0] = 1.0f / grad_opscale(brows, t->mlp->dims[t->mlp->L]);
  st.sc = step_ctx(t);
  const dim3 grid(md.order ? 8 * md.xper : (md.nblk + 7) / 8 * 8);
  if (t->prec == V21_PREC_F16) hipLaunchKernelGGL(dw16_adam_kernel<PrecF16>, grid, dim3(64 * kDwAdamWaves), 0, t->ctx->stream, md, st);
  else hipLaunchKernelGGL(dw16_adam_kernel<PrecBF16>, grid, dim3(64 * kDwAdamWaves), 0, t->ctx->stream, md, st);
  HIPCHK(hipGetLastError());
  t->copies_ok = true;
  t->nt_ok = false;
  return V21_OK;
}

// group form (sweep, joint step): per-model blocks in a device table, refreshed when anything in them changed
int refresh_dw_adam_table(const std::vector<v21_trainer*>& tr, DwAdamModel** d_tab, std::vector<DwAdamModel>& h_tab,
                                 hipStream_t st) {
  std::vector<DwAdamModel> tab(tr.size());
  for (size_t k = 0; k < tr.size(); ++k) dw_adam_model(tr[k], tab[k]);
  if (!*d_tab) HIPCHK(hipMalloc((void**)d_tab, tab.size() * sizeof(DwAdamModel)));
  if (tab.size() != h_tab.size() || memcmp(tab.data(), h_tab.data(), tab.size() * sizeof(DwAdamModel)) != 0) {
    h_tab = tab;
    HIPCHK(hipMemcpyAsync(*d_tab, h_tab.data(), tab.size() * sizeof(DwAdamModel), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return V21_OK;
}
// every model takes one Adam step (iter advanced here) on the operands its chain launch left; slot: see DwAdamStep
int launch_dw_adam_group(const std::vector<v21_trainer*>& tr, const DwAdamModel* d_tab,
                                const std::vector<DwAdamModel>& h_tab, int rows, int brows, long long slot, hipStream_t st) {
  DwAdamStep stp{};
  stp.steps = (rows + 15) / 16;
  stp.slot = (int)slot;
  int maxblk = 0;
  for (size_t k = 0; k < tr.size(); ++k) {
    v21_trainer* t = tr[k];
    t->iter += 1;
    stp.alpha[k] = adam_alpha(t->adam, t->iter);
    stp.out_scale[k] = 1.0f / grad_opscale(brows, t->mlp->dims[t->mlp->L]);
    maxblk = std::max(maxblk, h_tab[k].nblk);
  }
  const dim3 grid((maxblk + 7) / 8 * 8, (unsigned)tr.size());
  const bool small = stp.steps <= 32;  // (dw_adam.h: U)
  const bool f16 = tr[0]->prec == V21_PREC_F16;
  if (small) {
    if (f16) hipLaunchKernelGGL((dw16_adam_group_kernel<PrecF16, 2>), grid, dim3(64 * kDwAdamWaves), 0, st, d_tab, stp);
    else hipLaunchKernelGGL((dw16_adam_group_kernel<PrecBF16, 2>), grid, dim3(64 * kDwAdamWaves), 0, st, d_tab, stp);
  } else {
    if (f16) hipLaunchKernelGGL((dw16_adam_group_kernel<PrecF16, kDwAdamInFlight>), grid, dim3(64 * kDwAdamWaves), 0, st, d_tab, stp);
    else hipLaunchKernelGGL((dw16_adam_group_kernel<PrecBF16, kDwAdamInFlight>), grid, dim3(64 * kDwAdamWaves), 0, st, d_tab, stp);
  }
  HIPCHK(hipGetLastError());
  for (v21_trainer* t : tr) {
    t->copies_ok = true; t->nt_ok = false;
    invalidate_streams(t->mlp);
    t->mlp->wpad_ok = true;
  }
  return V21_OK;
}

static int trainer_step(v21_trainer* t, const float* yb, long long ldy, int rows, int brows, float* loss_out,
                        long long row0);
// one optimizer step on rows [first, first+rows) (through d_idx when given) of (x, y, rw)
// launch `probs` in groups of <= kNtMaxGroup
int launch_nt_many(int prec, std::vector<NtArgs>& probs, hipStream_t st) {
  for (size_t o = 0; o < probs.size(); o += kNtMaxGroup) {
    NtGroupBig grp{};
    grp.count = (int)std::min<size_t>(kNtMaxGroup, probs.size() - o);
    for (int i = 0; i < grp.count; ++i) grp.p[i] = probs[o + i];
    CHK(launch_nt(prec, grp, st));
  }
  return V21_OK;
}
// one f32 optimizer step in THREE launches (train_chain32.h): the chain over this rank's rows, every layer's weight
// gradient in one grouped NT launch on the fp32 operands the chain left, Adam (which also rebuilds the packed fp32
// streams and, on a single rank, publishes the batch loss)
int train_on_rows_chain32(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                                 const int* d_idx, long long first, int rows, int brows, float* loss_out, long long row0,
                                 bool chain_done) {
  v21_mlp* m = t->mlp;
  hipStream_t st = t->ctx->stream;
  const int L = m->L, dout = m->dims[L];
  if (rows > t->max_batch) return fail(V21_ERR_ARG, "batch of %d rows exceeds max_batch %d", rows, t->max_batch);
  const bool single = t->ctx->nranks == 1;
  const bool in_table = single && rows > 0 && loss_out && t->d_steploss && loss_out >= t->d_steploss &&
                        loss_out < t->d_steploss + t->steploss_cap;
  int fold = 1;
  if (rows > 0) {
    // which kernels: csrc/routes.h (decide_step).  One rank, a step of <= kDw32MaxRows rows, 32 x 32 tiles: gradients,
    // Adam, packed streams and batch loss in ONE launch whose workgroups walk the whole batch in slabs of 256 rows
    // (dw_adam32.h: UP_DWADAM32); one contraction slice with 64 x 64 tiles: gemm_nt_dwadam_kernel (UP_NT_DWADAM);
    // larger steps and data-parallel ranks: sliced gradients, [slab sum, exchange], Adam (UP_NT_SLICED)
    const StepRoute route = step_route(t, rows);
    if (!chain_done) {
      phase_mark(t, 0);
      CHK(ensure_copies(t, false));
      ChainArgs a{};
      static_cast<ChainModel&>(a) = chain_model32(t);
      static_cast<ChainStep&>(a) = chain_step(x, ldx, y, ldy, rw, d_idx, first, rows, brows, dout, t, row0);
      a.gs = 1.0f;  // fp32 operands: no scaling of the gradients
      CHK(launch_chain32_args(a, st, t->chain32s, route.fwd == TR_CHAIN32S_4 ? 4 : 8));
      phase_mark(t, 1);
    }
    note_route(t, route);  // (the joint step ran this model's chain in its own launch: the update route is what is recorded)
    long long work = 0;
    for (int l = 0; l < L; ++l) work += (long long)((m->dims[l] + 1 + 63) / 64) * ((m->nw(l) + 63) / 64);
    const bool dw32 = route.upd == UP_DWADAM32;
    int nslice = dw32 ? 1 : (rows + kNtMaxKPerWg - 1) / kNtMaxKPerWg;
    const int k_chunk = ((rows + nslice - 1) / nslice + 15) / 16 * 16;
    nslice = (rows + k_chunk - 1) / k_chunk;
    std::vector<NtArgs> probs;
    for (int l = 0; l < L; ++l) {  // [dW; db] = [H^T; 1^T] dZ
      NtArgs g{};
      g.A = t->d_ht[l]; g.lda = t->Bp;
      g.B = t->d_dzt[l + 1]; g.ldb = t->Bp;
      g.C = (nslice > 1 ? t->d_slab : t->d_g) + m->w_off[l]; g.ldc = m->nw(l);
      g.M = m->dims[l] + 1; g.N = m->nw(l); g.K = rows;
      g.ep = NT_DW; g.nz = nslice; g.k_chunk = k_chunk; g.slab_stride = (long long)t->P + 4;
      probs.push_back(g);
    }
    // (one contraction slice = up to kNtMaxKPerWg rows.  Letting one workgroup walk 1,024 or 2,048 rows instead of the
    //  sliced three-launch path below: 71.3 against 70.6 us and 96.7 against 92.8 us per step -- no gain.)
    if (route.upd == UP_DWADAM32 || route.upd == UP_NT_DWADAM) {
      // one rank, the batch is one contraction slice: gradients, Adam, the packed fp32 streams and the batch loss in ONE
      // launch (gemm_nt.h: NtAdamInfo) -- the step is 2 launches
      NtGroupBig grp{};
      grp.count = L;
      const int T = work >= 192 ? 2 : 1;
      int blocks = 0;
      NtAdamInfo ad{};
      for (int l = 0; l < L; ++l) {
        NtArgs& g = grp.p[l];
        g = probs[l];
        g.tile = 32 * T;
        g.nx = (g.N + g.tile - 1) / g.tile; g.ny = (g.M + g.tile - 1) / g.tile; g.nz = 1;
        g.a_scale = g.b_scale = g.out_scale = 1.f;
        g.k_chunk = g.K; g.slab_stride = 0;
        grp.first[l] = blocks;
        blocks += g.nx * g.ny;
        ad.lt[l] = NtAdamLayer{m->w_off[l], t->fw_off[l], t->bw_off[l], m->dims[l], t->c32_frags(m->dims[l]), t->c32_frags(m->nw(l))};
      }
      grp.first[L] = blocks;
      if (!t->capturing) t->iter += 1;
      ad.w = m->d_w; ad.m = t->d_m; ad.v = t->d_v; ad.fw = (float*)t->d_fw; ad.bw = (float*)t->d_bw;
      ad.alpha = t->capturing ? 0.f : adam_alpha(t->adam, t->iter);
      ad.omb1 = 1.0f - t->adam.beta1; ad.omb2 = 1.0f - t->adam.beta2; ad.eps = t->adam.eps;
      ad.sc = step_ctx(t);
      ad.loss_acc = (unsigned long long*)t->d_ticket; ad.loss_out = t->d_g + t->P; ad.loss_out2 = t->d_steploss;
      ad.loss_slot = in_table ? (int)(loss_out - t->d_steploss) : -1;
      ad.fmt = t->chain32s ? 4 : 3;
#ifdef V21_CHAIN_FINE
      ad.dbg = t->stamps_on ? t->d_stamps + 1024 : nullptr;
#endif
      phase_mark(t, 2); phase_mark(t, 3);  // (gradients + Adam are ONE launch here: reported under the Adam phase)
      if (T == 2) hipLaunchKernelGGL(gemm_nt_dwadam_kernel<2>, dim3(blocks), dim3(256), 0, st, grp, ad);
      else if (route.upd == UP_DWADAM32) hipLaunchKernelGGL(dwadam32_kernel, dim3(blocks), dim3(256), 0, st, grp, ad);  // operands through LDS in whole rows (dw_adam32.h)
      else hipLaunchKernelGGL(gemm_nt_dwadam_kernel<1>, dim3(blocks), dim3(256), 0, st, grp, ad);
      HIPCHK(hipGetLastError());
      phase_mark(t, 4);
      t->copies_ok = true;
      t->nt_ok = false;
      if (t->capturing) return V21_OK;
      if (loss_out && !in_table) HIPCHK(hipMemcpyAsync(loss_out, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
      invalidate_streams(m);
      m->wpad_ok = true;
      return V21_OK;
    }
    CHK(launch_nt_many(t->prec, probs, st));
    fold = nslice > 1 && single ? nslice : 1;  // single rank: Adam sums the slabs itself
    if (nslice > 1 && fold == 1) {
      const long long n4 = ((long long)t->P + 3) / 4;
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, t->d_g,
                         (const float*)t->d_slab, nslice, (long long)t->P + 4, (long long)t->P);
      HIPCHK(hipGetLastError());
    }
    if (!single) {  // the loss numerator rides in slot P of the arena: it must be there before the exchange
      hipLaunchKernelGGL(chain32_loss_kernel, dim3(1), dim3(1), 0, st, (unsigned long long*)t->d_ticket, t->d_g + t->P);
      HIPCHK(hipGetLastError());
    }
    phase_mark(t, 2);
  } else {
    HIPCHK(hipMemsetAsync(t->d_g, 0, (t->P + 1) * sizeof(float), st));
    phase_mark(t, 0); phase_mark(t, 1); phase_mark(t, 2);
  }
  t->loss_slot_pending = (single && rows > 0) ? (in_table ? (int)(loss_out - t->d_steploss) : -1) : -2;
  int r;
  if (t->capturing) r = adam_and_copies(t, true, 0.f, true, fold);  // recorded, not run: step size and slot come from the descriptors
  else r = reduce_and_update(t, true, fold);
  t->loss_slot_pending = -2;
  CHK(r);
  if (t->capturing) return V21_OK;
  if (loss_out && !in_table) HIPCHK(hipMemcpyAsync(loss_out, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
  invalidate_streams(m);
  m->wpad_ok = true;
  return V21_OK;
}

static int train_on_rows(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                         const int* d_idx, long long first, int rows, int brows, float* loss_out, long long row0) {
  v21_mlp* m = t->mlp;
  const int L = m->L, din = m->dims[0], dout = m->dims[L];
  if (t->chain32) return train_on_rows_chain32(t, x, ldx, y, ldy, rw, d_idx, first, rows, brows, loss_out, row0);
  if (!t->chain) {
    if (rows > 0) CHK(gather_batch(t, x, ldx, y, ldy, rw, d_idx, first, rows));
    const float* yb = y ? t->d_yb : t->d_h[0];
    return trainer_step(t, yb, y ? p16(dout) : p16(din), rows, brows, loss_out, row0);
  }
  hipStream_t st = t->ctx->stream;
  int fold = 1;
  if (rows > t->max_batch) return fail(V21_ERR_ARG, "batch of %d rows exceeds max_batch %d", rows, t->max_batch);
  bool fused_step = false;
  DwXRows xrows{};  // x16 != nullptr: this step's layer-0 gradient operand is gathered from the resident rows
  const bool bucketed = dp_bucketed(t);
  const int ksplit = bucketed ? dp_split_layer(m) : 0;
  const size_t lo1 = bucketed ? (size_t)m->w_off[ksplit] : 0;  // bucket 1 = arena [lo1, P + 1), bucket 2 = [0, lo1)
  phase_mark(t, 0);
  if (rows > 0) {
    const bool ts_fresh = t->tstream_fresh && t->copies_ok && t->mlp->wpad_ok;  // (read before ensure_copies clears it)
    CHK(ensure_copies(t, false));
    // steps of >= V21_FUSED_TRAIN_ROWS rows of a stack with a compiled fused training kernel take it; below, the 32-row chain.
    // Default: 8,193 rows for a trainer on the 16-rows-per-wave kernel (fused_train16.h; max_batch < 24,576: the chain's
    // second round of 256 workgroups starts there -- 9,216 rows 67 against 80 us, 10,240 rows 69 against 81, 12,288 rows 72
    // against 83, 16,384 rows 78-82 against 90-92, 20,480 rows 94 against 119), 16,384 rows for one on the 128-row kernel
    // (fused_train.h: 24,576 rows 101 against 127, 32,768 rows 123-130 against 152-160).  The kernel's weight stream is
    // written by the previous step's Adam pass (AdamArgs::ts), so a step is 3 launches.  (Autoencoder stack, f16, whole
    // steps, r4; read per step: the tests force it.)
    const StepRoute route = step_route(t, rows);  // csrc/routes.h: decide_step
    const bool fused = route.fwd == TR_FUSED64 || route.fwd == TR_FUSED128;
    note_route(t, route);
    fused_step = fused;
    bool fused_done = false;
    if (fused) {
      // (the LDS-staged gradient kernel only: launch_dw16 takes it from 64 batch steps of 16 rows on)
      const bool want_xr = route.upd == UP_DW16_SPLITK && (rows + 15) / 16 >= 64 && RouteEnv::read().dw_xrows;
      const int fr = launch_fused_train(t, x, ldx, y, ldy, rw, d_idx, first, rows, brows, row0, ts_fresh, want_xr ? &xrows : nullptr);
      if (fr != V21_OK) xrows = DwXRows{};
      if (fr == V21_OK) fused_done = true;
      else if (t->train_arch >= 0) return fr;
      else {  // the run-time kernel could not be loaded (it spills: marked failed): this step and every later one take the chain
        t->kind.train_rt = false; t->train_jit = nullptr;
        fused_step = false;
        t->last_route.fwd = TR_CHAIN16; t->fwd_count[TR_CHAIN16] += 1; t->fwd_count[route.fwd & 7] -= 1;
      }
    }
    if (!fused_done) { CHK(launch_chain(t, x, ldx, y, ldy, rw, d_idx, first, rows, brows, row0)); if (!t->capturing) t->n_chain_steps += 1; }
    phase_mark(t, 1);
    // Single rank, nothing to exchange: gradients, Adam and the packed copies in one launch (dw_adam.h) -- up to the
    // batch where its 32 x 32 tiles, each pulling its operands over the WHOLE batch through one CU, lose to the
    // 128 x 128 LDS-staged split-K kernel + an Adam launch that sums the slabs (V21_DW_SPLIT_ROWS overrides the
    // threshold; measured r3, autoencoder stack, f16: see DESIGN.md section 3)
    if (route.upd == UP_DW16_ADAM) {
      if (!t->capturing) t->iter += 1;
      // an epoch's per-step loss slot is written by the kernel itself (a device-to-device copy per step is a launch)
      const bool in_table = loss_out && t->d_steploss && loss_out >= t->d_steploss && loss_out < t->d_steploss + t->steploss_cap;
      phase_mark(t, 2); phase_mark(t, 3);  // (gradients + Adam are ONE launch here: reported under the Adam phase)
      CHK(launch_dw_adam(t, rows, brows, t->capturing ? 0.f : adam_alpha(t->adam, t->iter),
                         in_table ? (int)(loss_out - t->d_steploss) : -1));
      phase_mark(t, 4);
      if (t->capturing) return V21_OK;
      if (loss_out && !in_table) HIPCHK(hipMemcpyAsync(loss_out, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
      invalidate_streams(m);
      m->wpad_ok = true;
      return V21_OK;
    }
    int nslice = 1;
    std::vector<Dw16Args> probs;
    dw16_problems(t, rows, brows, &nslice, probs);  // every weight gradient in one launch: [dW; db] = [H^T; 1^T] dZ
    const DwXRows* xr = xrows.x16 ? &xrows : nullptr;
    if (xr) probs[0].A = nullptr;
    if (bucketed) {
      // two launches, the upper layers first (their bucket also carries the loss slot: the batch loss is published by
      // whichever problem holds loss_acc -- the first of THIS launch); each problem's tiles are the same workgroups doing
      // the same sums as in the one-launch form: bit-identical gradients
      Dw16Args& p0 = probs[0]; Dw16Args& pk = probs[ksplit];
      pk.loss_acc = p0.loss_acc; pk.loss_out = p0.loss_out; pk.loss_out2 = p0.loss_out2; pk.sc = p0.sc;
      p0.loss_acc = nullptr; p0.loss_out = nullptr; p0.loss_out2 = nullptr;
      const std::vector<Dw16Args> upper(probs.begin() + ksplit, probs.end()), lower(probs.begin(), probs.begin() + ksplit);
      CHK(launch_dw16(t->prec, upper, st, xr));
      if (nslice > 1) CHK(reduce_slabs_range(t, nslice, (long long)lo1, (long long)t->P));
      CHK(dp_exchange_bucket(t, 0, lo1, t->P + 1));
      CHK(launch_dw16(t->prec, lower, st, xr));
      if (nslice > 1) CHK(reduce_slabs_range(t, nslice, 0, (long long)lo1));
      phase_mark(t, 2);
      CHK(dp_exchange_bucket(t, 1, 0, lo1));
      CHK(dp_exchange_join(t));
    } else {
      CHK(launch_dw16(t->prec, probs, st, xr));
      fold = nslice > 1 && t->ctx->nranks == 1 ? nslice : 1;  // single rank: Adam sums the slabs itself
      if (nslice > 1 && fold == 1) {
        const long long n4 = ((long long)t->P + 3) / 4;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, t->d_g,
                           (const float*)t->d_slab, nslice, (long long)t->P + 4, (long long)t->P);
        HIPCHK(hipGetLastError());
      }
      phase_mark(t, 2);
    }
  } else {
    HIPCHK(hipMemsetAsync(t->d_g, 0, (t->P + 1) * sizeof(float), st));
    phase_mark(t, 1); phase_mark(t, 2);
    if (bucketed) {  // a rank without rows takes part in the same two messages
      CHK(dp_exchange_bucket(t, 0, lo1, t->P + 1));
      CHK(dp_exchange_bucket(t, 1, 0, lo1));
      CHK(dp_exchange_join(t));
    }
  }
  if (t->capturing) {
    CHK(adam_and_copies(t, true, 0.f, true, fold));
    return V21_OK;
  }
  t->ts_write = fused_step;  // the next step probably takes the fused kernel too: its stream comes out of this Adam pass
  const int ru = reduce_and_update(t, true, fold, bucketed);
  t->ts_write = false;
  CHK(ru);
  if (loss_out) HIPCHK(hipMemcpyAsync(loss_out, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
  invalidate_streams(m);
  m->wpad_ok = true;
  return V21_OK;
}

int gather_batch(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy_src,
                        const float* rw, const int* d_idx, long long first, int rows) {
  v21_mlp* m = t->mlp;
  const int din = m->dims[0], dout = m->dims[m->L];
  const int wpb = 4;
  hipLaunchKernelGGL(gather_batch_kernel, dim3((rows + wpb - 1) / wpb), dim3(64 * wpb), 0, t->ctx->stream, x, din,
                     t->d_h[0], p16(din), t->d_ht[0], t->Bp, y, dout, t->d_yb, p16(dout), rw, t->d_wb, d_idx, first,
                     rows, ldx, ldy_src, step_ctx(t));
  HIPCHK(hipGetLastError());
  return V21_OK;
}

// ---------------------------------------------------------------------------------
// replayed steps (hipGraph): SURVEY 7.1 step 6.  A captured step costs the host one
// hipGraphLaunch (7 us per step instead of 43 us for the 14 launches of an f32 step).
// OPT-IN (v21_trainer_use_graph), because measured on MI355X in r2 it does not pay: the
// steps are bound by the GPU, not by the host -- an f32 batch-256 step is 14 dependent
// kernels of ~7 us each (kernel boundary + a cold-L2 round trip + a short MFMA chain):
// 96 us eager, 102 us replayed; the 3-launch f16 step 45 us eager, 50 us replayed (the
// cursor-tick node and the boundary between two graph launches cost more than they save).
// ---------------------------------------------------------------------------------
static bool graph_eligible(const v21_trainer* t) {
  return t->graph_mode == 1 && t->ctx->nranks == 1 && t->gl < 0;
}
static int ensure_desc(v21_trainer* t, long long n) {
  if (t->desc_cap >= n) return V21_OK;
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  if (t->d_desc) HIPCHK(hipFree(t->d_desc));
  if (t->h_desc) HIPCHK(hipHostFree(t->h_desc));
  HIPCHK(hipMalloc((void**)&t->d_desc, (size_t)n * sizeof(StepDesc)));
  HIPCHK(hipHostMalloc((void**)&t->h_desc, (size_t)n * sizeof(StepDesc), hipHostMallocDefault));
  if (!t->d_cur) HIPCHK(hipMalloc((void**)&t->d_cur, 16));
  t->desc_cap = n;
  destroy_graphs(t);  // captured steps hold the old table
  return V21_OK;
}
// upload descriptors [0, count) from h_desc and point the device cursor at the first
static int publish_desc(v21_trainer* t, long long count) {
  hipStream_t st = t->ctx->stream;
  HIPCHK(hipMemcpyAsync(t->d_desc, t->h_desc, (size_t)count * sizeof(StepDesc), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(t->d_cur, 0, 4, st));
  t->desc_count = count; t->desc_next = 0;
  return V21_OK;
}
static int train_on_rows(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                         const int* d_idx, long long first, int rows, int brows, float* loss_out, long long row0);
// the captured step for this batch geometry and these pointers (captured on first use); nullptr if capture failed
static int step_graph(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                      const int* d_idx, int rows, int brows, long long row0, hipGraphExec_t* out) {
  *out = nullptr;
  for (auto& g : t->graphs)
    if (g.rows == rows && g.brows == brows && g.x == x && g.y == y && g.rw == rw && g.idx == d_idx && g.row0 == row0) {
      // The lazy refresh of the packed weight copies is NOT part of the captured step (it was a no-op while the
      // step was recorded): an arena rewritten between two replays (v21_mlp_set_weights, a loaded file) must reach
      // the copies before the replayed kernels read them.
      CHK(ensure_copies(t, !t->chain && !t->chain32));  // (the chain trainers never read the NT copies: after_replay's nt_ok)
      *out = g.exec;
      return V21_OK;
    }
  if (t->graphs.size() >= 8) {  // callers that pass new pointers every step would re-capture every step
    if (++t->graph_misses > 16) { t->graph_mode = 0; destroy_graphs(t); return V21_OK; }
    hipGraphExecDestroy(t->graphs.front().exec); hipGraphDestroy(t->graphs.front().graph);
    t->graphs.erase(t->graphs.begin());
  }
  hipStream_t st = t->ctx->stream;
  // everything that may not happen inside a capture: lazy refreshes, function attributes
  CHK(ensure_copies(t, !t->chain && !t->chain32));
  if (t->chain) { CHK(chain_attr(t->prec)); CHK(dw16_attr(t->prec)); }
  if (t->chain32) CHK(chain_attr(V21_PREC_F32));
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { (void)hipGetLastError(); t->graph_mode = 0; return V21_OK; }  // e.g. the legacy stream: run eagerly
  t->capturing = true;
  int r = train_on_rows(t, x, ldx, y, ldy, rw, d_idx, 0, rows, brows, nullptr, row0);
  if (r == V21_OK) {
    hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(1), 0, st, t->d_cur);
    if (hipGetLastError() != hipSuccess) r = V21_ERR_HIP;
  }
  t->capturing = false;
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(st, &graph);
  if (r != V21_OK || e != hipSuccess || !graph) {
    (void)hipGetLastError();
    if (graph) hipGraphDestroy(graph);
    t->graph_mode = 0;
    return r != V21_OK ? r : V21_OK;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGetLastError(); hipGraphDestroy(graph); t->graph_mode = 0; return V21_OK; }
  t->graphs.push_back(v21_trainer::StepGraph{rows, brows, x, y, rw, d_idx, row0, graph, exec});
  *out = exec;
  return V21_OK;
}
// bookkeeping train_on_rows does after a step, for a replayed one
static void after_replay(v21_trainer* t) {
  t->iter += 1;
  t->desc_next += 1;
  t->copies_ok = true;
  t->nt_ok = !t->chain && !t->chain32;
  invalidate_streams(t->mlp);
  t->mlp->wpad_ok = true;
}

int check_row_table(const int32_t* perm, long long n) {
  for (long long i = 0; i < n; ++i)
    if (perm[i] < 0 || perm[i] >= n)
      return fail(V21_ERR_ARG, "row table: entry %lld = %d is outside the training set's %lld rows (the table must hold one entry per row)",
                  i, (int)perm[i], n);
  return V21_OK;
}
extern "C" int v21_trainer_run_epoch(v21_trainer* t, const int32_t* perm, int batch, double* loss) {
  if (!t || !loss) return fail(V21_ERR_ARG, "null argument");
  if (t->n[0] < 1) return fail(V21_ERR_STATE, "no training data set");
  CHK(use(t->ctx));
  hipStream_t st = t->ctx->stream;
  v21_mlp* m = t->mlp;
  const long long n = t->n[0];
  const int R = t->ctx->nranks, rk = t->ctx->rank;
  if (batch < 1) return fail(V21_ERR_ARG, "batch must be >= 1");
  if ((batch + R - 1) / R > t->max_batch) return fail(V21_ERR_ARG, "per-rank batch %d exceeds max_batch %d", (batch + R - 1) / R, t->max_batch);
  const int* d_idx = nullptr;
  if (perm) {
    CHK(check_row_table(perm, n));
    if (t->perm_cap < n) {
      if (t->d_perm) HIPCHK(hipFree(t->d_perm));
      HIPCHK(hipMalloc((void**)&t->d_perm, (size_t)n * sizeof(int)));
      t->perm_cap = n;
    }
    HIPCHK(hipMemcpyAsync(t->d_perm, perm, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    d_idx = t->d_perm;
  }
  const long long steps = (n + batch - 1) / batch;
  if (t->steploss_cap < std::max<long long>(steps, kDescRing)) {
    HIPCHK(hipStreamSynchronize(st));
    destroy_graphs(t);  // captured steps hold the old pointer
    if (t->d_steploss) HIPCHK(hipFree(t->d_steploss));
    t->steploss_cap = std::max<long long>(steps, kDescRing);
    HIPCHK(hipMalloc((void**)&t->d_steploss, (size_t)t->steploss_cap * sizeof(float)));
  }
  const int din = m->dims[0], dout = m->dims[m->L];
  bool replay = graph_eligible(t);
  if (replay) {  // one descriptor per step of this epoch
    CHK(ensure_desc(t, std::max<long long>(steps, kDescRing)));
    HIPCHK(hipStreamSynchronize(st));  // a preceding step_dev's copy of the staging table may still be in flight
    for (long long s = 0; s < steps; ++s) t->h_desc[s] = StepDesc{s * batch, adam_alpha(t->adam, t->iter + s + 1), (int)s};
    CHK(publish_desc(t, steps));
    t->desc_epoch = true;
  }
  for (long long s = 0; s < steps; ++s) {
    const long long first = s * batch;
    const int brows = (int)std::min<long long>(batch, n - first);  // rows of the global batch
    const long long lo = first + (long long)brows * rk / R, hi = first + (long long)brows * (rk + 1) / R;
    const int rows = (int)(hi - lo);
    const float* yy = t->y_is_x[0] ? nullptr : t->d_y[0];
    if (replay) {
      hipGraphExec_t exec = nullptr;
      CHK(step_graph(t, t->d_x[0], din, yy, dout, t->d_rw[0], d_idx, rows, brows, 0, &exec));
      if (exec) {
        HIPCHK(hipGraphLaunch(exec, st));
        after_replay(t);
        continue;
      }
      // capture is not possible here: the rest of the epoch runs eagerly; the steps replayed so far are unaffected
      replay = false;
    }
    CHK(train_on_rows(t, t->d_x[0], din, yy, dout, t->d_rw[0], d_idx, lo, rows, brows, t->d_steploss + s, lo - first));
  }
  std::vector<float> h(steps);
  HIPCHK(hipMemcpyAsync(h.data(), t->d_steploss, (size_t)steps * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  double tot = 0.0;
  for (float v : h) tot += (double)v;  // each entry = batch_loss * n_b  ([K] epoch loss)
  *loss = tot / (double)n;
  return V21_OK;
}

extern "C" int v21_trainer_eval(v21_trainer* t, int which, int batch, double* loss) {
  if (!t || !loss) return fail(V21_ERR_ARG, "null argument");
  if (which < 0 || which > 1 || t->n[which] < 1) return fail(V21_ERR_STATE, "no data set for split %d", which);
  CHK(use(t->ctx));
  hipStream_t st = t->ctx->stream;
  v21_mlp* m = t->mlp;
  const long long n = t->n[which];
  const int din = m->dims[0], dout = m->dims[m->L];
  if (batch < 1) return fail(V21_ERR_ARG, "batch must be >= 1");
  if (t->chain32) {  // the same in fp32 (train_chain32.h)
    if (n > (1ll << 30)) return fail(V21_ERR_ARG, "too many rows for one validation launch");
    CHK(ensure_copies(t, false));
    ChainArgs a{};
    static_cast<ChainModel&>(a) = chain_model32(t);
    a.sample = 0;  // a variational head evaluates z = z_mean (include/v21.h)
    static_cast<ChainStep&>(a) = chain_step(t->d_x[which], din, t->y_is_x[which] ? nullptr : t->d_y[which], dout,
                                            t->d_rw[which], nullptr, 0, (int)n, (int)n, dout);
    a.fwd_only = 1;
    CHK(launch_chain32_args(a, st, t->chain32s));
    long long acc = 0;
    HIPCHK(hipMemcpyAsync(&acc, t->d_ticket, sizeof acc, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(t->d_ticket, 0, sizeof acc, st));
    HIPCHK(hipStreamSynchronize(st));
    *loss = (double)acc * (1.0 / 4294967296.0) / (double)n;
    return V21_OK;
  }
  if (t->chain) {
    // ONE forward-only launch of the chain kernel over all n rows (csrc/train_chain.h: fwd_only) instead of 8 launches
    // per batch of the per-layer path: the same arithmetic as the training loss of this precision, no noise drawn
    if (n > (1ll << 30)) return fail(V21_ERR_ARG, "too many rows for one validation launch");
    CHK(ensure_copies(t, false));
    ChainArgs a{};
    static_cast<ChainModel&>(a) = chain_model(t);
    a.sample = 0;
    static_cast<ChainStep&>(a) = chain_step(t->d_x[which], din, t->y_is_x[which] ? nullptr : t->d_y[which], dout,
                                            t->d_rw[which], nullptr, 0, (int)n, (int)n, dout);
    a.fwd_only = 1;
    a.ncons = (int)(((n + 31) / 32 + 7) / 8 * 8);
    a.npref = chain_prefetchers(a.ncons, 1);
    CHK(chain_attr(t->prec));
    const dim3 grid(a.ncons + 8 * a.npref), block(64 * kChainWaves);
    static const bool plain_ok = !(getenv("V21_CHAIN_PLAIN") && getenv("V21_CHAIN_PLAIN")[0] == '0');  // (see launch_chain)
    const int feat = kChainFwd | (t->gl < 0 && plain_ok ? 0 : kChainGauss);
    if (t->prec == V21_PREC_F16) launch_chain_kernel<PrecF16>(feat, grid, block, st, a);
    else launch_chain_kernel<PrecBF16>(feat, grid, block, st, a);
    HIPCHK(hipGetLastError());
    long long acc = 0;  // 2^-32 fixed point (order-independent sum over the workgroups)
    HIPCHK(hipMemcpyAsync(&acc, t->d_ticket, sizeof acc, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(t->d_ticket, 0, sizeof acc, st));
    HIPCHK(hipStreamSynchronize(st));
    *loss = (double)acc * (1.0 / 4294967296.0) / (double)n;
    return V21_OK;
  }
  const int b = std::min(batch, t->max_batch);
  CHK(ensure_copies(t));
  HIPCHK(hipMemsetAsync(t->d_evalsum, 0, 16, st));
  for (long long first = 0; first < n; first += b) {
    const int rows = (int)std::min<long long>(b, n - first);
    CHK(gather_batch(t, t->d_x[which], din, t->y_is_x[which] ? nullptr : t->d_y[which], dout, t->d_rw[which], nullptr,
                     first, rows));
    CHK(trainer_forward(t, rows, false));
    const float* yb = t->y_is_x[which] ? t->d_h[0] : t->d_yb;
    const int wpb = 4;
    hipLaunchKernelGGL(loss_grad_kernel<false>, dim3((rows + wpb - 1) / wpb), dim3(64 * wpb), 0, st, t->d_h[m->L],
                       p16(dout), yb, t->y_is_x[which] ? p16(din) : p16(dout), t->d_wb, (float*)nullptr, 0ll,
                       t->d_rowloss, rows, dout, 0.f, (const float*)t->d_klrow);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, t->d_rowloss, rows, t->d_evalsum, 1);
    HIPCHK(hipGetLastError());
  }
  float s = 0.f;
  HIPCHK(hipMemcpyAsync(&s, t->d_evalsum, sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *loss = (double)s / (double)n;
  return V21_OK;
}

extern "C" int v21_trainer_get_data_dev(v21_trainer* t, int which, const float** x, const float** y, const float** rw, int64_t* n) {
  if (!t || which < 0 || which > 1) return fail(V21_ERR_ARG, "bad argument");
  if (t->n[which] < 1) return fail(V21_ERR_STATE, "no data set for this split");
  if (x) *x = t->d_x[which];
  if (y) *y = t->d_y[which];
  if (rw) *rw = t->d_rw[which];
  if (n) *n = t->n[which];
  return V21_OK;
}
extern "C" int v21_trainer_step_dev(v21_trainer* t, const float* d_x, const float* d_y, const float* d_rw, int n_rows,
                                    int global_rows) {
  if (!t || !d_x || !d_rw) return fail(V21_ERR_ARG, "null argument");
  if (n_rows < 0 || global_rows < std::max(n_rows, 1)) return fail(V21_ERR_ARG, "bad row counts");
  if (n_rows > t->max_batch) return fail(V21_ERR_ARG, "batch of %d rows exceeds max_batch %d", n_rows, t->max_batch);
  CHK(use(t->ctx));
  v21_mlp* m = t->mlp;
  const int din = m->dims[0], dout = m->dims[m->L];
  if (!d_y && din != dout) return fail(V21_ERR_ARG, "d_y == NULL (y = x) needs in_dim == out_dim");
  if (graph_eligible(t) && n_rows > 0) {
    // descriptors for the next kDescRing steps (first = 0: the caller's pointers are the batch); rebuilt when
    // they run out, after an epoch used the table, or when lr / the iteration count changed behind them
    CHK(ensure_desc(t, kDescRing));
    if (t->steploss_cap < kDescRing) {
      HIPCHK(hipStreamSynchronize(t->ctx->stream));
      destroy_graphs(t);
      if (t->d_steploss) HIPCHK(hipFree(t->d_steploss));
      HIPCHK(hipMalloc((void**)&t->d_steploss, (size_t)kDescRing * sizeof(float)));
      t->steploss_cap = kDescRing;
    }
    if (t->desc_epoch || t->desc_next >= t->desc_count || t->desc_lr != t->adam.lr ||
        t->desc_iter0 + t->desc_next != t->iter) {
      HIPCHK(hipStreamSynchronize(t->ctx->stream));  // the staging copy may still be in flight
      for (long long i = 0; i < kDescRing; ++i) t->h_desc[i] = StepDesc{0, adam_alpha(t->adam, t->iter + i + 1), (int)i};
      CHK(publish_desc(t, kDescRing));
      t->desc_epoch = false; t->desc_lr = t->adam.lr; t->desc_iter0 = t->iter;
    }
    hipGraphExec_t exec = nullptr;
    CHK(step_graph(t, d_x, din, d_y, dout, d_rw, nullptr, n_rows, global_rows, 0, &exec));
    if (exec) {
      HIPCHK(hipGraphLaunch(exec, t->ctx->stream));
      after_replay(t);
      return V21_OK;
    }
  }
  return train_on_rows(t, d_x, din, d_y, dout, d_rw, nullptr, 0, n_rows, global_rows, nullptr,
                       (long long)t->ctx->rank * t->max_batch);
}
extern "C" int v21_trainer_last_step_loss(v21_trainer* t, double* loss) {
  if (!t || !loss) return fail(V21_ERR_ARG, "null argument");
  CHK(use(t->ctx));
  float s = 0.f;
  HIPCHK(hipMemcpyAsync(&s, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToHost, t->ctx->stream));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  *loss = (double)s;  // sum_i w_i sum_j (p-y)^2 over the global batch
  return V21_OK;
}
extern "C" int v21_trainer_get_state(v21_trainer* t, int64_t* iter, float* mm, float* vv, size_t n) {
  if (!t) return fail(V21_ERR_ARG, "null trainer");
  CHK(use(t->ctx));
  if (iter) *iter = t->iter;
  if ((mm || vv) && n != t->P) return fail(V21_ERR_ARG, "state size %zu != %zu", n, t->P);
  if ((mm || vv) && t->ctx->nranks > 1 && t->ctx->sharded) {  // each rank holds its slice of the moments: a collective call
    const size_t S = (t->P + 1 + t->ctx->nranks - 1) / t->ctx->nranks;
    CHK(v21_comm_allgather_f32(t->ctx, t->d_m, S));
    CHK(v21_comm_allgather_f32(t->ctx, t->d_v, S));
  }
  if (mm) HIPCHK(hipMemcpyAsync(mm, t->d_m, n * sizeof(float), hipMemcpyDeviceToHost, t->ctx->stream));
  if (vv) HIPCHK(hipMemcpyAsync(vv, t->d_v, n * sizeof(float), hipMemcpyDeviceToHost, t->ctx->stream));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  return V21_OK;
}
extern "C" int v21_trainer_set_state(v21_trainer* t, int64_t iter, const float* mm, const float* vv, size_t n) {
  if (!t) return fail(V21_ERR_ARG, "null trainer");
  CHK(use(t->ctx));
  if (iter < 0) return fail(V21_ERR_ARG, "negative iteration count");
  if ((mm || vv) && n != t->P) return fail(V21_ERR_ARG, "state size %zu != %zu", n, t->P);
  t->iter = iter;
  if (mm) HIPCHK(hipMemcpyAsync(t->d_m, mm, n * sizeof(float), hipMemcpyHostToDevice, t->ctx->stream));
  if (vv) HIPCHK(hipMemcpyAsync(t->d_v, vv, n * sizeof(float), hipMemcpyHostToDevice, t->ctx->stream));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  return V21_OK;
}
extern "C" int v21_trainer_get_grad(v21_trainer* t, float* g, size_t n) {
  if (!t || !g) return fail(V21_ERR_ARG, "null argument");
  if (n != t->P) return fail(V21_ERR_ARG, "grad size %zu != %zu", n, t->P);
  CHK(use(t->ctx));
  HIPCHK(hipMemcpyAsync(g, t->d_g, n * sizeof(float), hipMemcpyDeviceToHost, t->ctx->stream));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  return V21_OK;
}


extern "C" int v21_trainer_set_vae(v21_trainer* t, float kl_weight, int sample, uint64_t seed) {
  if (!t) return fail(V21_ERR_ARG, "null trainer");
  if (t->gl < 0) return fail(V21_ERR_STATE, "the stack has no V21_ACT_GAUSS layer");
  if (!(kl_weight >= 0.f)) return fail(V21_ERR_ARG, "kl_weight must be >= 0");
  t->kl_weight = kl_weight; t->sample = sample ? 1 : 0; t->seed = (unsigned long long)seed;
  return V21_OK;
}
extern "C" int v21_trainer_enable_stamps(v21_trainer* t, int enable) {
  if (!t) return fail(V21_ERR_ARG, "null trainer");
  if (!t->chain && !t->chain32) return fail(V21_ERR_STATE, "this trainer does not use the chain kernel");
  if (t->capturing) return fail(V21_ERR_STATE, "not while steps are being recorded");
  t->stamps_on = enable != 0;
  return V21_OK;
}
extern "C" int v21_trainer_chain_stamps(v21_trainer* t, uint64_t* out, int n) {
  if (!t || !out) return fail(V21_ERR_ARG, "null argument");
  if (n < 1 || n > kStampSlots) return fail(V21_ERR_ARG, "n must be in [1,%d]", kStampSlots);
  if (!t->chain && !t->chain32) return fail(V21_ERR_STATE, "this trainer does not use the chain kernel");
  if (!t->stamps_on) return fail(V21_ERR_STATE, "stamps are off (v21_trainer_enable_stamps)");
  CHK(use(t->ctx));
  HIPCHK(hipMemcpyAsync(out, t->d_stamps, (size_t)n * 8, hipMemcpyDeviceToHost, t->ctx->stream));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  return V21_OK;
}

// diagnostics: the check build_chain32s_jobs makes at creation, repeated against stream sizes the CALLER names (bytes;
// < 0 = the real ones) -- a test hands in a truncated stream and expects V21_ERR_STATE, not a launch
extern "C" int v21_debug_check_chain_jobs(v21_trainer* t, long long fw_bytes, long long bw_bytes) {
  if (!t) return fail(V21_ERR_ARG, "null trainer");
  if (!t->chain32s) return fail(V21_ERR_UNSUPPORTED, "the trainer does not run the small-batch f32 chain (no job table)");
  const ChainModel a = chain_model32(t);
  std::vector<C32sJob> tab((size_t)2 * a.L * kC32sWaves);
  c32s_build_jobs(a, tab.data());
  if (const char* why = c32s_validate_jobs(a, tab.data(), (fw_bytes < 0 ? t->fw_bytes : fw_bytes) / 16,
                                           (bw_bytes < 0 ? t->bw_bytes : bw_bytes) / 16, (long long)t->P))
    return fail(V21_ERR_STATE, "%s", why);
  return V21_OK;
}
extern "C" int v21_trainer_jit(v21_trainer* t, int wait_ms, int* status) {
  if (!t || !status) return fail(V21_ERR_ARG, "null argument");
  *status = -1;
  if (t->train_arch >= 0) { *status = 1; return V21_OK; }  // compiled into the library (archs.h)
  if (!t->train_jit) {
    std::string why = "the trainer was created with fewer rows per step than the kernel's threshold, or run-time compilation is switched off (V21_JIT=0)";
    if (t->prec == V21_PREC_F32) why = "f32 trainers have no fused training kernel";
    else if (!t->chain) why = "the trainer is not on the chain path";
    else v21::jit_train_eligible(t->mlp->L, t->mlp->dims.data(), t->mlp->act.data(), &why);
    return fail(V21_ERR_UNSUPPORTED, "no fused training kernel for this trainer: %s", why.c_str());
  }
  int s = v21::jit_state(t->train_jit);
  if (s == v21::JIT_COMPILING && wait_ms != 0) s = v21::jit_wait(t->train_jit, wait_ms);
  *status = s;
  if (s == v21::JIT_FAILED) {
    std::string why;
    v21::jit_state(t->train_jit, &why);
    return fail(V21_ERR_UNSUPPORTED, "fused training kernel of this stack: %s", why.c_str());
  }
  return V21_OK;
}
extern "C" int v21_route_train(int n_layers, const int* dims, const int* act, int precision, int max_batch, int rows, int nranks,
                               int rt_ready, int* fwd, int* upd) {
  if (!dims || !act || !fwd || !upd) return fail(V21_ERR_ARG, "null argument");
  if (n_layers < 1 || n_layers > 16) return fail(V21_ERR_ARG, "n_layers %d out of range", n_layers);
  if (precision < 0 || precision > 2) return fail(V21_ERR_ARG, "precision %d unknown", precision);
  if (max_batch < 1 || rows < 1 || rows > max_batch || nranks < 1) return fail(V21_ERR_ARG, "need 1 <= rows <= max_batch and nranks >= 1");
  const RouteEnv e = RouteEnv::read();
  const TrainerKind k = decide_trainer_kind(n_layers, dims, act, precision, max_batch, e);
  const StepRoute r = decide_step(k, n_layers, dims, act, rows, nranks, false, k.train_arch >= 0 || (k.train_rt && rt_ready), e);
  *fwd = r.fwd; *upd = r.upd;
  return V21_OK;
}
extern "C" int v21_trainer_last_route(v21_trainer* t, int* fwd, int* upd, long long fwd_counts[8], long long upd_counts[8]) {
  if (!t || !fwd || !upd) return fail(V21_ERR_ARG, "null argument");
  *fwd = t->last_route.fwd; *upd = t->last_route.upd;
  if (fwd_counts) for (int i = 0; i < 8; ++i) fwd_counts[i] = t->fwd_count[i];
  if (upd_counts) for (int i = 0; i < 8; ++i) upd_counts[i] = t->upd_count[i];
  return V21_OK;
}
extern "C" int v21_trainer_phase_timing(v21_trainer* t, int steps, int cut) {
  if (!t) return fail(V21_ERR_ARG, "null trainer");
  if (steps < 0 || steps > 4096) return fail(V21_ERR_ARG, "steps %d not in [0, 4096]", steps);
  if (steps > 0 && (cut < 1 || cut > 4)) return fail(V21_ERR_ARG, "cut %d not in [1, 4]", cut);
  CHK(use(t->ctx));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  while ((int)t->phase_ev.size() < 2 * steps) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    t->phase_ev.push_back(e);
  }
  t->phase_cap = steps; t->phase_steps = 0; t->phase_on = steps > 0; t->phase_cut = cut;
  return V21_OK;
}
extern "C" int v21_trainer_phase_times(v21_trainer* t, double* ms, int* steps) {
  if (!t || !ms || !steps) return fail(V21_ERR_ARG, "null argument");
  CHK(use(t->ctx));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  *ms = 0.0;
  *steps = t->phase_steps;
  std::vector<float> el((size_t)t->phase_steps);
  for (int s = 0; s < t->phase_steps; ++s)
    HIPCHK(hipEventElapsedTime(&el[s], t->phase_ev[(size_t)s * 2], t->phase_ev[(size_t)s * 2 + 1]));
  if (!el.empty()) {  // the MEDIAN: one stalled step (the host descheduled between two enqueues) moves a mean of 50 by microseconds
    std::sort(el.begin(), el.end());
    *ms = el.size() % 2 ? el[el.size() / 2] : 0.5 * (el[el.size() / 2 - 1] + el[el.size() / 2]);
  }
  t->phase_steps = 0;
  return V21_OK;
}
extern "C" int v21_debug_trainer_counters(v21_trainer* t, long long out[4]) {
  if (!t || !out) return fail(V21_ERR_ARG, "null argument");
  out[0] = t->n_chain_steps; out[1] = t->n_fused_steps; out[2] = t->n_stream_packs; out[3] = t->n_stream_adam;
  return V21_OK;
}
extern "C" int v21_trainer_use_graph(v21_trainer* t, int enable) {
  if (!t) return fail(V21_ERR_ARG, "null trainer");
  CHK(use(t->ctx));
  if (enable && (t->ctx->nranks > 1 || t->gl >= 0))
    return fail(V21_ERR_UNSUPPORTED, "captured steps need one rank and a stack without a variational layer");
  if (!enable) { HIPCHK(hipStreamSynchronize(t->ctx->stream)); destroy_graphs(t); }
  t->graph_mode = enable ? 1 : 0;
  return V21_OK;
}


