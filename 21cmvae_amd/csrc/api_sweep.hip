// api_sweep.hip -- v21_sweep_*: several models stepped in lock step with grouped launches (BASELINE configs[4]).
#include "api_internal.h"

// ---------------------------------------------------------------------------------
// sweep: G independent models, ONE shared batch stream, one launch per phase for all of them
// (BASELINE configs[4]: "64 concurrent latent-dim/hidden-width configs packed as batched
// GEMM", 8 per GPU; no reference code -- the reference trains one model at a time,
// emulator.py:739-747).  The models share depth, activations and in/out width; hidden and
// latent widths differ.  Phase k of a step is the same kind of kernel for every model, so it
// becomes one grouped launch (gemm_nt.h: NtGroup; train_kernels.h: *_group_kernel).
// ---------------------------------------------------------------------------------
struct v21_sweep {
  v21_ctx* ctx = nullptr;
  std::vector<v21_trainer*> tr;
  AdamArgs* d_adam = nullptr;
  std::vector<AdamArgs> h_adam;  // what d_adam holds
  bool chain = false;            // every member runs the chain kernel: one grouped launch of it per step
  ChainModel* d_chain = nullptr;
  std::vector<ChainModel> h_chain;
  DwAdamModel* d_dwadam = nullptr;  // single rank: gradients + Adam in one grouped launch (dw_adam.h)
  std::vector<DwAdamModel> h_dwadam;
  // f32 members on the small-batch chain (train_chain32s.h): one grouped chain launch + one grouped gradient / Adam launch
  // (dw_adam32.h) per step on a single rank, steps of <= kDw32MaxRows rows
  bool chain32s = false;
  Dw32Model* d_dw32 = nullptr;
  std::vector<Dw32Model> h_dw32;
  // r5: TWO half-groups on two streams (one rank, grouped chain launches).  A group step is two launches of complementary
  // character -- every member's chain (latency-bound per workgroup, little HBM traffic) and every member's gradients + Adam
  // (bound by the optimizer state's bytes: 390 MB per step of 32 members at 3.85 TB/s) -- and the members are independent
  // models: half B's chain runs while half A's Adam launch waits for HBM.  Same kernels on the same data per member:
  // bit-identical results.  Measured (one MI355X, f16, batch 256; gpurun_out/r5_sweep_f16_c.txt against r5_b5.json): 16 members
  // 159-163 k -> 167 k model-steps/s, 64 members 170-173 k -> 180 k, 8 and 32 members within noise -- the two streams drift back
  // into lock step (two Adam launches that share the HBM end together, then both chains start together).  Forcing the
  // anti-phase order with events (A's Adam launch, then B's, then A's ...) was measured SLOWER (32 members: 183 against 168 us
  // per group step: every cross-stream dependency is a ~5-us hand-over); two host threads on two contexts reached 148 us
  // (scripts/diag/sweep_two_streams_probe.py) -- the kernels' own sum (44.6 + 101 us): what is left is enqueue jitter, not overlap.
  hipStream_t s2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_off = nullptr, ev_join = nullptr;
  bool two_streams = false;  // this epoch
};

extern "C" int v21_sweep_create(v21_trainer** trainers, int count, v21_sweep** out) {
  if (!trainers || !out) return fail(V21_ERR_ARG, "null argument");
  if (count < 1 || count > kSweepMax) return fail(V21_ERR_ARG, "count %d not in [1,%d]", count, kSweepMax);
  v21_trainer* t0 = trainers[0];
  if (!t0) return fail(V21_ERR_ARG, "null trainer");
  const v21_mlp* m0 = t0->mlp;
  for (int k = 0; k < count; ++k) {
    v21_trainer* t = trainers[k];
    if (!t) return fail(V21_ERR_ARG, "null trainer");
    const v21_mlp* m = t->mlp;
    if (t->ctx != t0->ctx || t->prec != t0->prec || t->max_batch != t0->max_batch)
      return fail(V21_ERR_ARG, "model %d: context, precision and max_batch must match model 0", k);
    if (m->L != m0->L || m->act != m0->act || m->dims[0] != m0->dims[0] || m->dims[m->L] != m0->dims[m0->L])
      return fail(V21_ERR_ARG, "model %d: depth, activations and in/out width must match model 0", k);
    if (t->gl >= 0 && !t->chain && !(t->chain32 && t->chain32s))
      return fail(V21_ERR_UNSUPPORTED, "variational stacks are swept on the chain kernels only (f16 / bf16, or f32 with max_batch <= %d; latent <= %d)",
                  kC32sMaxBatch, kChainMaxLatent);
    for (int j = 0; j < k; ++j)
      if (trainers[j] == t) return fail(V21_ERR_ARG, "trainer %d listed twice", k);
  }
  CHK(use(t0->ctx));
  v21_sweep* s = new v21_sweep();
  s->ctx = t0->ctx;
  s->tr.assign(trainers, trainers + count);
  HIPCHK(hipMalloc((void**)&s->d_adam, (size_t)count * sizeof(AdamArgs)));
  s->chain = true;
  for (int k = 0; k < count; ++k) s->chain = s->chain && trainers[k]->chain;
  for (int k = 0; k < count; ++k) s->h_adam.push_back(adam_args(trainers[k], true, 0.f, s->chain));
  HIPCHK(hipMemcpyAsync(s->d_adam, s->h_adam.data(), s->h_adam.size() * sizeof(AdamArgs), hipMemcpyHostToDevice, s->ctx->stream));
  s->chain32s = !s->chain;
  for (int k = 0; k < count; ++k) s->chain32s = s->chain32s && trainers[k]->chain32s && trainers[k]->mlp->L <= kNtMaxGroup;
  if (s->chain || s->chain32s) HIPCHK(hipMalloc((void**)&s->d_chain, (size_t)count * sizeof(ChainModel)));
  if (s->chain32s) HIPCHK(hipMalloc((void**)&s->d_dw32, (size_t)count * sizeof(Dw32Model)));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  *out = s;
  return V21_OK;
}
extern "C" int v21_sweep_destroy(v21_sweep* s) {
  if (!s) return V21_OK;
  hipSetDevice(s->ctx->device);
  hipStreamSynchronize(s->ctx->stream);
  hipFree(s->d_adam);
  if (s->d_chain) hipFree(s->d_chain);
  if (s->d_dwadam) hipFree(s->d_dwadam);
  if (s->d_dw32) hipFree(s->d_dw32);
  if (s->s2) { hipStreamSynchronize(s->s2); hipStreamDestroy(s->s2); }
  for (hipEvent_t e : {s->ev_fork, s->ev_off, s->ev_join}) if (e) hipEventDestroy(e);
  delete s;
  return V21_OK;
}


// one optimizer step of every model on the batch gathered into model 0's h[0]/ht[0]/yb/wb
static int sweep_step(v21_sweep* s, const float* yb, long long ldy, int rows, int brows, long long step_index) {
  v21_trainer* t0 = s->tr[0];
  hipStream_t st = s->ctx->stream;
  const int G = (int)s->tr.size(), L = t0->mlp->L, dout = t0->mlp->dims[L];
  if (rows > t0->max_batch) return fail(V21_ERR_ARG, "batch of %d rows exceeds max_batch %d", rows, t0->max_batch);
  if (rows > 0) {
    for (v21_trainer* t : s->tr) CHK(ensure_copies(t));
    std::vector<NtArgs> probs;
    for (int l = 0; l < L; ++l) {  // forward, layer l of every model
      probs.clear();
      for (v21_trainer* t : s->tr) {
        v21_mlp* m = t->mlp;
        NtArgs g{};
        g.A = l == 0 ? t0->d_h[0] : t->d_h[l]; g.lda = p16(m->dims[l]);
        g.B = t->d_wt + t->wt_off[l]; g.ldb = p16(m->dims[l]);
        g.C = t->d_h[l + 1]; g.ldc = p16(m->dims[l + 1]);
        g.CT = l + 1 < L ? t->d_ht[l + 1] : nullptr; g.ldct = t->Bp;
        g.M = rows; g.N = m->dims[l + 1]; g.K = m->dims[l];
        g.bias = m->d_w + m->b_off[l];
        g.ep = m->act[l] == V21_ACT_RELU ? NT_FWD_RELU : NT_FWD;
        g.nz = 1;
        probs.push_back(g);
      }
      CHK(launch_nt_many(t0->prec, probs, st));
    }
    LossGroup lg{};
    SumGroup sg{};
    for (int k = 0; k < G; ++k) {
      v21_trainer* t = s->tr[k];
      lg.p[k] = t->d_h[L]; lg.ldp[k] = p16(dout);
      lg.dz[k] = t->d_dz[L]; lg.lddz[k] = p16(dout);
      lg.dzt[k] = t->d_dzt[L]; lg.rowloss[k] = t->d_rowloss;
      sg.v[k] = t->d_rowloss; sg.out[k] = t->d_g + t->P;
      sg.out2[k] = (s->ctx->nranks == 1 && step_index >= 0) ? t->d_steploss + step_index : nullptr;
    }
    lg.y = yb; lg.ldy = ldy; lg.w = t0->d_wb; lg.ldt = t0->Bp; lg.n = rows; lg.d = dout;
    lg.scale = 2.0f / (float)brows;
    sg.n = rows;
    hipLaunchKernelGGL(loss_grad_t_group_kernel, dim3((rows + 3) / 4, G), dim3(256), 0, st, lg);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sum_group_kernel, dim3(G), dim3(256), 0, st, sg);
    HIPCHK(hipGetLastError());
    int nslice = (rows + kNtMaxKPerWg - 1) / kNtMaxKPerWg;
    const int k_chunk = ((rows + nslice - 1) / nslice + 15) / 16 * 16;
    nslice = (rows + k_chunk - 1) / k_chunk;
    const float gs = grad_opscale(brows, dout);
    for (int l = L - 1; l >= 0; --l) {  // backward, layer l of every model: dW (and dX below the top)
      probs.clear();
      for (v21_trainer* t : s->tr) {
        v21_mlp* m = t->mlp;
        const int K = m->dims[l], N = m->dims[l + 1];
        NtArgs g{};
        g.A = l == 0 ? t0->d_ht[0] : t->d_ht[l]; g.lda = t->Bp;
        g.B = t->d_dzt[l + 1]; g.ldb = t->Bp;
        g.C = (nslice > 1 ? t->d_slab : t->d_g) + m->w_off[l]; g.ldc = N;
        g.M = K + 1; g.N = N; g.K = rows;
        g.ep = NT_DW; g.nz = nslice; g.k_chunk = k_chunk; g.slab_stride = (long long)t->P + 4;
        g.b_scale = gs; g.out_scale = 1.0f / gs;
        probs.push_back(g);
        if (l > 0) {
          NtArgs d{};
          d.A = t->d_dz[l + 1]; d.lda = p16(N);
          d.B = t->d_wp + t->wp_off[l]; d.ldb = p16(N);
          d.C = t->d_dz[l]; d.ldc = p16(K);
          d.CT = t->d_dzt[l]; d.ldct = t->Bp;
          d.M = rows; d.N = K; d.K = N;
          d.mask = t->d_h[l]; d.ldmask = p16(K);
          d.ep = m->act[l - 1] == V21_ACT_RELU ? NT_DX_MASK : NT_DX;
          d.nz = 1;
          d.a_scale = gs; d.out_scale = 1.0f / gs;
          probs.push_back(d);
        }
      }
      CHK(launch_nt_many(t0->prec, probs, st));
    }
    if (nslice > 1)
      for (v21_trainer* t : s->tr) {
        const long long n4 = ((long long)t->P + 3) / 4;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, t->d_g,
                           (const float*)t->d_slab, nslice, (long long)t->P + 4, (long long)t->P);
        HIPCHK(hipGetLastError());
      }
  } else {
    for (v21_trainer* t : s->tr) HIPCHK(hipMemsetAsync(t->d_g, 0, (t->P + 1) * sizeof(float), st));
  }
  AlphaGroup al{};
  size_t maxP = 0;
  for (int k = 0; k < G; ++k) {
    v21_trainer* t = s->tr[k];
    CHK(v21_comm_allreduce_f32(t->ctx, t->d_g, t->P + 1));
    if (s->ctx->nranks > 1 && step_index >= 0)
      HIPCHK(hipMemcpyAsync(t->d_steploss + step_index, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
    t->iter += 1;
    al.a[k] = adam_alpha(t->adam, t->iter);
    maxP = std::max(maxP, t->P);
  }
  hipLaunchKernelGGL(adam_repack_group_kernel, dim3((unsigned)((maxP + 255) / 256), G), dim3(256), 0, st,
                     (const AdamArgs*)s->d_adam, al);
  HIPCHK(hipGetLastError());
  for (v21_trainer* t : s->tr) {
    t->copies_ok = true; t->nt_ok = true;
    invalidate_streams(t->mlp);
    t->mlp->wpad_ok = true;
  }
  return V21_OK;
}

// chain form of a sweep step: ONE launch carries every model's rows through forward, loss and the
// activation-gradient chain (blockIdx.y = model); then all weight gradients, then all Adam updates
// members [g0, g1) of the group on stream `st`: the chain of each, and (one rank) their gradients + Adam
static int sweep_chain_launch(v21_sweep* s, const ChainStep& csp, int g0, int g1, hipStream_t st) {
  v21_trainer* t0 = s->tr[0];
  const int G = g1 - g0;
  // one-dimensional grid: a model's row blocks share an XCD label (train_chain.h: train_chain_group_kernel)
  const dim3 grid(8 * csp.ncons * ((G + 7) / 8)), block(64 * kChainWaves);
  bool gauss = false;  // (train_chain.h: FEAT)
  for (int k = g0; k < g1; ++k) gauss = gauss || s->tr[k]->gl >= 0;
  const ChainModel* tab = (const ChainModel*)s->d_chain + g0;
  if (t0->prec == V21_PREC_F16) {
    if (gauss) hipLaunchKernelGGL((train_chain_group_kernel<PrecF16, true>), grid, block, kChainLdsBytes, st, tab, csp, G);
    else hipLaunchKernelGGL((train_chain_group_kernel<PrecF16, false>), grid, block, kChainLdsBytes, st, tab, csp, G);
  } else {
    if (gauss) hipLaunchKernelGGL((train_chain_group_kernel<PrecBF16, true>), grid, block, kChainLdsBytes, st, tab, csp, G);
    else hipLaunchKernelGGL((train_chain_group_kernel<PrecBF16, false>), grid, block, kChainLdsBytes, st, tab, csp, G);
  }
  HIPCHK(hipGetLastError());
  return V21_OK;
}
static int sweep_dwadam_launch(v21_sweep* s, int g0, int g1, int rows, int brows, long long step_index, hipStream_t st) {
  const std::vector<v21_trainer*> part(s->tr.begin() + g0, s->tr.begin() + g1);
  const std::vector<DwAdamModel> hpart(s->h_dwadam.begin() + g0, s->h_dwadam.begin() + g1);
  return launch_dw_adam_group(part, s->d_dwadam + g0, hpart, rows, brows, step_index, st);
}
static int sweep_step_chain(v21_sweep* s, const ChainStep& cs, int brows, long long step_index) {
  v21_trainer* t0 = s->tr[0];
  hipStream_t st = s->ctx->stream;
  const int G = (int)s->tr.size(), rows = cs.rows;
  if (rows > t0->max_batch) return fail(V21_ERR_ARG, "batch of %d rows exceeds max_batch %d", rows, t0->max_batch);
  if (rows > 0) {
    for (v21_trainer* t : s->tr) CHK(ensure_copies(t, false));
    CHK(chain_attr(t0->prec));
    ChainStep csp = cs;
    csp.ncons = ((rows + 31) / 32 + 7) / 8 * 8;
    csp.npref = 0;  // (no prefetcher workgroups in a sweep: measured slower in r2)
    if (s->two_streams) {  // one rank: half A on the context's stream, half B on the second one, B one launch behind A
      const int GA = (G + 1) / 2;
      CHK(sweep_chain_launch(s, csp, 0, GA, st));
      if (step_index == 0) {  // (the offset that makes B's chain meet A's Adam launch, not A's chain)
        HIPCHK(hipEventRecord(s->ev_off, st));
        HIPCHK(hipStreamWaitEvent(s->s2, s->ev_off, 0));
      }
      CHK(sweep_chain_launch(s, csp, GA, G, s->s2));
      CHK(sweep_dwadam_launch(s, 0, GA, rows, brows, step_index, st));
      return sweep_dwadam_launch(s, GA, G, rows, brows, step_index, s->s2);
    }
    CHK(sweep_chain_launch(s, csp, 0, G, st));
    if (s->ctx->nranks == 1)  // nothing to exchange: all gradients, all Adam updates, all packed copies in one launch
      return launch_dw_adam_group(s->tr, s->d_dwadam, s->h_dwadam, rows, brows, step_index, st);
    int nslice = 1;
    std::vector<Dw16Args> probs;
    for (v21_trainer* t : s->tr)
      dw16_problems(t, rows, brows, &nslice, probs,
                    (s->ctx->nranks == 1 && step_index >= 0) ? t->d_steploss + step_index : nullptr);
    CHK(launch_dw16(t0->prec, probs, st));
    if (nslice > 1)
      for (v21_trainer* t : s->tr) {
        const long long n4 = ((long long)t->P + 3) / 4;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, t->d_g,
                           (const float*)t->d_slab, nslice, (long long)t->P + 4, (long long)t->P);
        HIPCHK(hipGetLastError());
      }
  } else {
    for (v21_trainer* t : s->tr) HIPCHK(hipMemsetAsync(t->d_g, 0, (t->P + 1) * sizeof(float), st));
  }
  AlphaGroup al{};
  size_t maxP = 0;
  for (int k = 0; k < G; ++k) {
    v21_trainer* t = s->tr[k];
    CHK(v21_comm_allreduce_f32(t->ctx, t->d_g, t->P + 1));
    if ((s->ctx->nranks > 1 || rows == 0) && step_index >= 0)
      HIPCHK(hipMemcpyAsync(t->d_steploss + step_index, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
    t->iter += 1;
    al.a[k] = adam_alpha(t->adam, t->iter);
    maxP = std::max(maxP, t->P);
  }
  hipLaunchKernelGGL(adam_repack_group_kernel, dim3((unsigned)((maxP + 255) / 256), G), dim3(256), 0, st,
                     (const AdamArgs*)s->d_adam, al);
  HIPCHK(hipGetLastError());
  for (v21_trainer* t : s->tr) {
    t->copies_ok = true; t->nt_ok = false;
    invalidate_streams(t->mlp);
    t->mlp->wpad_ok = true;
  }
  return V21_OK;
}

// f32 members (train_chain32s.h): the problem and Adam blocks of one model's gradient launch, as train_on_rows_chain32
// builds them per step -- without what changes from step to step (contraction length, step size, loss slot: Dw32Step).
// false: this model's gradient launch would take 64 x 64 tiles (gemm_nt_dwadam_kernel<2>): the sweep then keeps the
// per-layer path, so that a member trains bit for bit as it would on its own.
static bool build_dw32_model(v21_trainer* t, Dw32Model& md, int& blocks) {
  v21_mlp* m = t->mlp;
  const int L = m->L;
  md = Dw32Model{};
  blocks = 0;
  if (L > kNtMaxGroup) return false;
  long long work = 0;
  md.grp.count = L;
  for (int l = 0; l < L; ++l) {
    NtArgs& g = md.grp.p[l];
    g.A = t->d_ht[l]; g.lda = t->Bp;
    g.B = t->d_dzt[l + 1]; g.ldb = t->Bp;
    g.C = t->d_g + m->w_off[l]; g.ldc = m->nw(l);
    g.M = m->dims[l] + 1; g.N = m->nw(l);
    g.ep = NT_DW; g.nz = 1; g.tile = 32;
    g.nx = (g.N + 31) / 32; g.ny = (g.M + 31) / 32;
    g.a_scale = g.b_scale = g.out_scale = 1.f;
    work += (long long)((g.M + 63) / 64) * ((g.N + 63) / 64);
    md.grp.first[l] = blocks;
    blocks += g.nx * g.ny;
    md.ad.lt[l] = NtAdamLayer{m->w_off[l], t->fw_off[l], t->bw_off[l], m->dims[l], t->c32_frags(m->dims[l]), t->c32_frags(m->nw(l))};
  }
  md.grp.first[L] = blocks;
  if (work >= 192) return false;
  NtAdamInfo& ad = md.ad;
  ad.w = m->d_w; ad.m = t->d_m; ad.v = t->d_v; ad.fw = (float*)t->d_fw; ad.bw = (float*)t->d_bw;
  ad.omb1 = 1.0f - t->adam.beta1; ad.omb2 = 1.0f - t->adam.beta2; ad.eps = t->adam.eps;
  ad.loss_acc = (unsigned long long*)t->d_ticket; ad.loss_out = t->d_g + t->P; ad.loss_out2 = t->d_steploss;
  ad.loss_slot = -1;
  ad.fmt = 4;
  return true;
}
// every weight gradient + Adam + packed streams + batch loss of several f32 models in one launch (dw_adam32.h)
// small_slabs: the 128-row instantiation (dw_adam32.h: SLAB) -- decided by the caller for the WHOLE sweep, not per launch: a
// member must see the same sums whether its half-group or the whole group is launched
int launch_dw32_group(const std::vector<v21_trainer*>& trs, const Dw32Model* d_tab, int rows, long long step_index, int max_blocks,
                             hipStream_t st, bool small_slabs) {
  const int G = (int)trs.size();
  Dw32Step ds{};
  ds.rows = rows; ds.slot = (int)step_index;
  for (int k = 0; k < G; ++k) {
    v21_trainer* t = trs[k];
    t->iter += 1;
    ds.alpha[k] = adam_alpha(t->adam, t->iter);
  }
  if (small_slabs) hipLaunchKernelGGL(dwadam32_group_kernel<128>, dim3(max_blocks, G), dim3(256), 0, st, d_tab, ds);
  else hipLaunchKernelGGL(dwadam32_group_kernel<256>, dim3(max_blocks, G), dim3(256), 0, st, d_tab, ds);
  HIPCHK(hipGetLastError());
  for (v21_trainer* t : trs) {
    t->copies_ok = true; t->nt_ok = false;
    invalidate_streams(t->mlp);
    t->mlp->wpad_ok = true;
  }
  return V21_OK;
}
// builds / refreshes the device table of launch_dw32_group; *ok = false: a member's gradient launch takes 64 x 64 tiles
int refresh_dw32_table(const std::vector<v21_trainer*>& trs, Dw32Model* d_tab, std::vector<Dw32Model>& h_tab, int* max_blocks, bool* ok,
                              hipStream_t st) {
  std::vector<Dw32Model> dtab(trs.size());
  *max_blocks = 0; *ok = true;
  for (size_t k = 0; k < trs.size(); ++k) {
    int blocks = 0;
    *ok = *ok && build_dw32_model(trs[k], dtab[k], blocks);
    *max_blocks = std::max(*max_blocks, blocks);
  }
  if (!*ok) return V21_OK;
  if (dtab.size() != h_tab.size() || memcmp(dtab.data(), h_tab.data(), dtab.size() * sizeof(Dw32Model)) != 0) {
    HIPCHK(hipStreamSynchronize(st));  // (a step in flight may still read the old table)
    h_tab = dtab;
    HIPCHK(hipMemcpyAsync(d_tab, h_tab.data(), dtab.size() * sizeof(Dw32Model), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return V21_OK;
}
// one optimizer step of every f32 member in TWO launches: the chain of every model (blockIdx.y = model), then every
// weight gradient + Adam + packed streams + batch loss
static int sweep_chain32_launch(v21_sweep* s, const ChainStep& csp, int rpw, int g0, int g1, hipStream_t st) {
  const int G = g1 - g0;
  const dim3 grid(csp.ncons * G), block(64 * kC32sWaves);
  bool gauss = false;
  for (int k = g0; k < g1; ++k) gauss = gauss || s->tr[k]->gl >= 0;
  const ChainModel* tab = (const ChainModel*)s->d_chain + g0;
  if (rpw == 4) {
    if (gauss) hipLaunchKernelGGL((train_chain32s_group_kernel<4, true>), grid, block, kC32sLdsBytes, st, tab, csp, G);
    else hipLaunchKernelGGL((train_chain32s_group_kernel<4, false>), grid, block, kC32sLdsBytes, st, tab, csp, G);
  } else {
    if (gauss) hipLaunchKernelGGL((train_chain32s_group_kernel<8, true>), grid, block, kC32sLdsBytes, st, tab, csp, G);
    else hipLaunchKernelGGL((train_chain32s_group_kernel<8, false>), grid, block, kC32sLdsBytes, st, tab, csp, G);
  }
  HIPCHK(hipGetLastError());
  return V21_OK;
}
static int sweep_step_chain32(v21_sweep* s, const ChainStep& cs, long long step_index, int max_blocks) {
  hipStream_t st = s->ctx->stream;
  const int G = (int)s->tr.size(), rows = cs.rows;
  for (v21_trainer* t : s->tr) CHK(ensure_copies(t, false));
  CHK(chain_attr(V21_PREC_F32));
  // 4 rows per workgroup while every model's row blocks fit the chip in one round (train_chain32s.h)
  const int force_rows = RouteEnv::read().c32s_rows;
  const int rpw = force_rows == 4 || force_rows == 8 ? force_rows : ((long long)G * ((rows + 3) / 4) <= 256 ? 4 : 8);
  ChainStep csp = cs;
  csp.ncons = ((rows + rpw - 1) / rpw + 7) / 8 * 8;
  csp.npref = 0;
  // (dw_adam32.h: SLAB -- from 8 members on the gradient launch is deep enough for four or five workgroups per CU to pay: r5,
  //  f32, batch 256, 8 / 16 / 32 / 64 members 60 / 70-72 / 70 / 64-67 k -> 62-66 / 79 / 77-79 / 72-73 k model-steps/s)
  const bool small_slabs = G >= 8;
  if (s->two_streams) {  // (as sweep_step_chain: half B one launch behind half A)
    const int GA = (G + 1) / 2;
    const std::vector<v21_trainer*> pa(s->tr.begin(), s->tr.begin() + GA), pb(s->tr.begin() + GA, s->tr.end());
    CHK(sweep_chain32_launch(s, csp, rpw, 0, GA, st));
    if (step_index == 0) {
      HIPCHK(hipEventRecord(s->ev_off, st));
      HIPCHK(hipStreamWaitEvent(s->s2, s->ev_off, 0));
    }
    CHK(sweep_chain32_launch(s, csp, rpw, GA, G, s->s2));
    CHK(launch_dw32_group(pa, s->d_dw32, rows, step_index, max_blocks, st, small_slabs));
    return launch_dw32_group(pb, s->d_dw32 + GA, rows, step_index, max_blocks, s->s2, small_slabs);
  }
  CHK(sweep_chain32_launch(s, csp, rpw, 0, G, st));
  return launch_dw32_group(s->tr, s->d_dw32, rows, step_index, max_blocks, st, small_slabs);
}

extern "C" int v21_sweep_run_epoch(v21_sweep* s, const int32_t* perm, int batch, double* losses) {
  if (!s || !losses) return fail(V21_ERR_ARG, "null argument");
  v21_trainer* t0 = s->tr[0];
  if (t0->n[0] < 1) return fail(V21_ERR_STATE, "model 0 holds the training set of the sweep: none set");
  CHK(use(s->ctx));
  hipStream_t st = s->ctx->stream;
  v21_mlp* m = t0->mlp;
  const long long n = t0->n[0];
  const int R = s->ctx->nranks, rk = s->ctx->rank;
  if (batch < 1) return fail(V21_ERR_ARG, "batch must be >= 1");
  if ((batch + R - 1) / R > t0->max_batch) return fail(V21_ERR_ARG, "per-rank batch %d exceeds max_batch %d", (batch + R - 1) / R, t0->max_batch);
  const int* d_idx = nullptr;
  if (perm) {
    CHK(check_row_table(perm, n));
    if (t0->perm_cap < n) {
      if (t0->d_perm) HIPCHK(hipFree(t0->d_perm));
      HIPCHK(hipMalloc((void**)&t0->d_perm, (size_t)n * sizeof(int)));
      t0->perm_cap = n;
    }
    HIPCHK(hipMemcpyAsync(t0->d_perm, perm, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    d_idx = t0->d_perm;
  }
  const long long steps = (n + batch - 1) / batch;
  for (v21_trainer* t : s->tr)
    if (t->steploss_cap < steps) {
      HIPCHK(hipStreamSynchronize(st));
      destroy_graphs(t);  // captured steps of this trainer hold the old pointer (as v21_trainer_run_epoch does)
      if (t->d_steploss) HIPCHK(hipFree(t->d_steploss));
      HIPCHK(hipMalloc((void**)&t->d_steploss, (size_t)steps * sizeof(float)));
      t->steploss_cap = steps;
    }
  // the Adam hyper-parameters may have changed since create (set_adam / set_lr): refresh the device table
  std::vector<AdamArgs> tab;
  for (v21_trainer* t : s->tr) tab.push_back(adam_args(t, true, 0.f, s->chain));
  if (memcmp(tab.data(), s->h_adam.data(), tab.size() * sizeof(AdamArgs)) != 0) {
    s->h_adam = tab;
    HIPCHK(hipMemcpyAsync(s->d_adam, s->h_adam.data(), tab.size() * sizeof(AdamArgs), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  const int din = m->dims[0], dout = m->dims[m->L];
  if (s->chain) {
    std::vector<ChainModel> tab;
    for (v21_trainer* t : s->tr) tab.push_back(chain_model(t));
    if (tab.size() != s->h_chain.size() || memcmp(tab.data(), s->h_chain.data(), tab.size() * sizeof(ChainModel)) != 0) {
      s->h_chain = tab;
      HIPCHK(hipMemcpyAsync(s->d_chain, s->h_chain.data(), tab.size() * sizeof(ChainModel), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    if (R == 1) CHK(refresh_dw_adam_table(s->tr, &s->d_dwadam, s->h_dwadam, st));
  }
  bool group32 = s->chain32s && R == 1 && batch <= kDw32MaxRows && !(getenv("V21_SWEEP32_GROUP") && getenv("V21_SWEEP32_GROUP")[0] == '0');
  int max_blocks32 = 0;
  if (group32) CHK(refresh_dw32_table(s->tr, s->d_dw32, s->h_dw32, &max_blocks32, &group32, st));
  if (!s->chain && !group32)
    for (v21_trainer* t : s->tr)
      if (t->gl >= 0)
        return fail(V21_ERR_UNSUPPORTED, "a sweep of variational f32 models takes the grouped chain launches only: one rank, batches of <= %d rows",
                    kDw32MaxRows);
  if (group32) {
    std::vector<ChainModel> tab;
    for (v21_trainer* t : s->tr) {
      tab.push_back(chain_model32(t));
      tab.back().stamps = nullptr;
    }
    if (tab.size() != s->h_chain.size() || memcmp(tab.data(), s->h_chain.data(), tab.size() * sizeof(ChainModel)) != 0) {
      HIPCHK(hipStreamSynchronize(st));  // (a step in flight may still read the old table)
      s->h_chain = tab;
      HIPCHK(hipMemcpyAsync(s->d_chain, s->h_chain.data(), tab.size() * sizeof(ChainModel), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
  }
  // r5: two half-groups on two streams (v21_sweep: s2) for the grouped chain launches of one rank; V21_SWEEP_STREAMS=1: one stream
  s->two_streams = R == 1 && (s->chain || group32) && s->tr.size() >= 16 && RouteEnv::read().sweep_streams == 2;  // (below 16 members the second stream's hand-overs cost more than they hide: tiny members, 8 per group: 17.4 -> 23.6 us per step)
  if (s->two_streams) {
    if (!s->s2) {
      HIPCHK(hipStreamCreateWithFlags(&s->s2, hipStreamNonBlocking));
      for (hipEvent_t* e : {&s->ev_fork, &s->ev_off, &s->ev_join}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    for (v21_trainer* t : s->tr) CHK(ensure_copies(t, false));  // (a stale member's refresh runs on the context's stream: before the fork)
    HIPCHK(hipEventRecord(s->ev_fork, st));                        // the row table, the tables and the copies are in place
    HIPCHK(hipStreamWaitEvent(s->s2, s->ev_fork, 0));
  }
  for (long long sidx = 0; sidx < steps; ++sidx) {
    const long long first = sidx * batch;
    const int brows = (int)std::min<long long>(batch, n - first);
    const long long lo = first + (long long)brows * rk / R, hi = first + (long long)brows * (rk + 1) / R;
    const int rows = (int)(hi - lo);
    if (s->chain) {
      ChainStep cs = chain_step(t0->d_x[0], din, t0->y_is_x[0] ? nullptr : t0->d_y[0], dout, t0->d_rw[0], d_idx, lo, rows,
                                brows, dout, nullptr, lo - first);
      cs.step_off = (unsigned long long)sidx;  // the table holds every model's step counter as of the epoch's start
      CHK(sweep_step_chain(s, cs, brows, sidx));
      continue;
    }
    if (group32) {
      ChainStep cs = chain_step(t0->d_x[0], din, t0->y_is_x[0] ? nullptr : t0->d_y[0], dout, t0->d_rw[0], d_idx, lo, rows,
                                brows, dout, nullptr, lo - first);
      cs.gs = 1.0f;  // fp32 operands: no scaling of the gradients
      cs.step_off = (unsigned long long)sidx;  // the table holds every model's step counter as of the epoch's start (noise key)
      CHK(sweep_step_chain32(s, cs, sidx, max_blocks32));
      continue;
    }
    if (rows > 0)
      CHK(gather_batch(t0, t0->d_x[0], din, t0->y_is_x[0] ? nullptr : t0->d_y[0], dout, t0->d_rw[0], d_idx, lo, rows));
    const float* yb = t0->y_is_x[0] ? t0->d_h[0] : t0->d_yb;
    CHK(sweep_step(s, yb, t0->y_is_x[0] ? p16(din) : p16(dout), rows, brows, sidx));
  }
  if (s->two_streams) {  // half B's launches end before the losses are read (and before anything else touches its members)
    HIPCHK(hipEventRecord(s->ev_join, s->s2));
    HIPCHK(hipStreamWaitEvent(st, s->ev_join, 0));
  }
  std::vector<float> h((size_t)steps * s->tr.size());
  for (size_t k = 0; k < s->tr.size(); ++k)
    HIPCHK(hipMemcpyAsync(h.data() + k * steps, s->tr[k]->d_steploss, (size_t)steps * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (size_t k = 0; k < s->tr.size(); ++k) {
    double tot = 0.0;
    for (long long i = 0; i < steps; ++i) tot += (double)h[k * steps + i];
    losses[k] = tot / (double)n;
  }
  return V21_OK;
}


