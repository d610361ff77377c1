// api_joint.hip -- v21_joint_*: autoencoder and latent emulator stepping on the same rows (BASELINE configs[2]).
#include "api_internal.h"

// ---------------------------------------------------------------------------------
// joint step (BASELINE configs[2]: "encoder+decoder+emulator joint train"): the autoencoder
// (signals -> signals, emulator.py:739-747) and the latent emulator (parameters -> latent, :756-764) take one
// optimizer step each on the SAME rows of every batch, and the emulator's targets are the latents the
// encoder produces for those rows in that very step (stop-gradient) instead of the reference's
// encoder.predict() of the finished autoencoder (:753-754).  With the autoencoder frozen (lr = 0) it is
// exactly the reference's phase 2.  One launch carries a row block through both models
// (train_chain_joint_kernel), one grouped launch forms all weight gradients and applies Adam to both models (dw_adam.h).
// ---------------------------------------------------------------------------------
struct v21_joint {
  v21_trainer *ae = nullptr, *em = nullptr;
  v21_ctx* ctx = nullptr;  // (kept: destroying the joint object must not look into trainers that may already be gone)
  int latent_layer = 0;
  ChainModel* d_tab = nullptr;
  std::vector<ChainModel> h_tab;
  DwAdamModel* d_dwadam = nullptr;  // gradients + Adam of both models in one grouped launch (dw_adam.h)
  std::vector<DwAdamModel> h_dwadam;
  bool f32 = false;  // both trainers on the small-batch f32 chain (train_chain32s.h: train_chain32s_joint_kernel)
  Dw32Model* d_dw32 = nullptr;  // ... and, on a single rank with steps of <= kDw32MaxRows rows, both models' gradients + Adam in one launch
  std::vector<Dw32Model> h_dw32;
};
extern "C" int v21_joint_create(v21_trainer* ae, v21_trainer* em, int latent_layer, v21_joint** out) {
  if (!ae || !em || !out) return fail(V21_ERR_ARG, "null argument");
  if (ae == em || ae->ctx != em->ctx || ae->prec != em->prec || ae->max_batch != em->max_batch)
    return fail(V21_ERR_ARG, "the two trainers must be distinct and share context, precision and max_batch");
  const bool f32 = ae->chain32s && em->chain32s && em->gl < 0;
  if (!f32 && (!ae->chain || !em->chain || em->gl >= 0))
    return fail(V21_ERR_UNSUPPORTED, "the joint step runs on the chain kernels: f16 / bf16 (widths <= %d, no variational layer in the emulator), or "
                "f32 with max_batch <= %d (variational head: latent <= %d)", kChainMaxDim, kC32sMaxBatch, kChainMaxLatent);
  const v21_mlp* ma = ae->mlp;
  const v21_mlp* me = em->mlp;
  // the latent layer: linear, or the variational head (V21_ACT_GAUSS) -- the emulator then learns z_mean, what
  // encoder.predict returns (emulator.py:753-754)
  if (latent_layer < 0 || latent_layer >= ma->L - 1 || ma->act[latent_layer] == V21_ACT_RELU || (ae->gl >= 0 && ae->gl != latent_layer))
    return fail(V21_ERR_ARG, "latent_layer %d must be the linear (or variational) layer below the autoencoder's output", latent_layer);
  if (ma->dims[latent_layer + 1] != me->dims[me->L] || ma->dims[latent_layer + 1] > 2 * kChainMaxLatent)
    return fail(V21_ERR_ARG, "latent width %d (autoencoder) vs emulator output %d (at most %d)", ma->dims[latent_layer + 1],
                me->dims[me->L], 2 * kChainMaxLatent);
  if (ma->dims[0] != ma->dims[ma->L]) return fail(V21_ERR_ARG, "the first trainer must be an autoencoder (in == out width)");
  CHK(use(ae->ctx));
  v21_joint* j = new v21_joint();
  j->ae = ae; j->em = em; j->ctx = ae->ctx; j->latent_layer = latent_layer; j->f32 = f32;
  hipError_t e = hipMalloc((void**)&j->d_tab, 2 * sizeof(ChainModel));
  if (e != hipSuccess) { delete j; return fail(V21_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
  *out = j;
  return V21_OK;
}
extern "C" int v21_joint_destroy(v21_joint* j) {
  if (!j) return V21_OK;
  hipSetDevice(j->ctx->device);
  hipStreamSynchronize(j->ctx->stream);
  hipFree(j->d_tab);
  if (j->d_dwadam) hipFree(j->d_dwadam);
  if (j->d_dw32) hipFree(j->d_dw32);
  delete j;
  return V21_OK;
}
// one epoch: the autoencoder trainer holds the signals (set_data(0, signals, NULL, w)), the emulator trainer the
// parameters of the SAME rows (set_data(0, params, any (n, latent) array, w_mse)); losses[0] = autoencoder,
// losses[1] = emulator (Keras epoch losses)
extern "C" int v21_joint_run_epoch(v21_joint* j, const int32_t* perm, int batch, double* losses) {
  if (!j || !losses) return fail(V21_ERR_ARG, "null argument");
  v21_trainer *ta = j->ae, *te = j->em;
  if (ta->n[0] < 1 || te->n[0] != ta->n[0]) return fail(V21_ERR_STATE, "both trainers need training sets of the same row count");
  if (!ta->y_is_x[0]) return fail(V21_ERR_STATE, "the autoencoder's targets must be its inputs (y == NULL)");
  CHK(use(ta->ctx));
  hipStream_t st = ta->ctx->stream;
  const long long n = ta->n[0];
  const int R = ta->ctx->nranks, rk = ta->ctx->rank;
  if (batch < 1 || (batch + R - 1) / R > ta->max_batch) return fail(V21_ERR_ARG, "per-rank batch %d not in [1, max_batch %d]", (batch + R - 1) / R, ta->max_batch);
  const int* d_idx = nullptr;
  if (perm) {
    CHK(check_row_table(perm, n));
    if (ta->perm_cap < n) {
      if (ta->d_perm) HIPCHK(hipFree(ta->d_perm));
      HIPCHK(hipMalloc((void**)&ta->d_perm, (size_t)n * sizeof(int)));
      ta->perm_cap = n;
    }
    HIPCHK(hipMemcpyAsync(ta->d_perm, perm, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    d_idx = ta->d_perm;
  }
  const long long steps = (n + batch - 1) / batch;
  for (v21_trainer* t : {ta, te})
    if (t->steploss_cap < steps) {
      HIPCHK(hipStreamSynchronize(st));
      destroy_graphs(t);
      if (t->d_steploss) HIPCHK(hipFree(t->d_steploss));
      HIPCHK(hipMalloc((void**)&t->d_steploss, (size_t)steps * sizeof(float)));
      t->steploss_cap = steps;
    }
  {
    std::vector<ChainModel> tab = j->f32 ? std::vector<ChainModel>{chain_model32(ta), chain_model32(te)}
                                         : std::vector<ChainModel>{chain_model(ta), chain_model(te)};
    tab[0].zcap_layer = j->latent_layer;
    if (j->f32) tab[0].stamps = tab[1].stamps = nullptr;
    if (tab.size() != j->h_tab.size() || memcmp(tab.data(), j->h_tab.data(), 2 * sizeof(ChainModel)) != 0) {
      HIPCHK(hipStreamSynchronize(st));
      j->h_tab = tab;
      HIPCHK(hipMemcpyAsync(j->d_tab, j->h_tab.data(), 2 * sizeof(ChainModel), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
  }
  if (R == 1 && !j->f32) CHK(refresh_dw_adam_table({ta, te}, &j->d_dwadam, j->h_dwadam, st));
  bool group32 = j->f32 && R == 1 && batch <= kDw32MaxRows;
  int max_blocks32 = 0;
  if (group32) {
    if (!j->d_dw32) HIPCHK(hipMalloc((void**)&j->d_dw32, 2 * sizeof(Dw32Model)));
    CHK(refresh_dw32_table({ta, te}, j->d_dw32, j->h_dw32, &max_blocks32, &group32, st));
  }
  CHK(chain_attr(ta->prec));
  const int dsig = ta->mlp->dims[0], dpar = te->mlp->dims[0], dlat = te->mlp->dims[te->mlp->L];
  for (long long s = 0; s < steps; ++s) {
    const long long first = s * batch;
    const int brows = (int)std::min<long long>(batch, n - first);  // rows of the global batch
    const long long lo = first + (long long)brows * rk / R, hi = first + (long long)brows * (rk + 1) / R;
    const int rows = (int)(hi - lo);                                // this rank's share (data parallel: SURVEY 8e)
    for (v21_trainer* t : {ta, te}) CHK(ensure_copies(t, false));
    if (j->f32) {
      // the reference's arithmetic: one joint chain launch (train_chain32s_joint_kernel), then each model's gradients and
      // Adam as after a chain step of its own (train_on_rows_chain32: one launch on a single rank; the exchange otherwise)
      if (rows > 0) {
        ChainStep sa = chain_step(ta->d_x[0], dsig, nullptr, dsig, ta->d_rw[0], d_idx, lo, rows, brows, dsig, nullptr, lo - first);
        ChainStep sb = chain_step(te->d_x[0], dpar, nullptr, dlat, te->d_rw[0], d_idx, lo, rows, brows, dlat, nullptr, lo - first);
        sa.gs = sb.gs = 1.0f;
        sa.step_off = (unsigned long long)s;  // the table holds the autoencoder's step counter as of the epoch's start (noise key)
        sb.y_from_lds = 1;
        const char* er = getenv("V21_C32S_ROWS");
        const int force_rows = er ? atoi(er) : 0;
        const int rpw = force_rows == 4 || force_rows == 8 ? force_rows : (2 * ((rows + 3) / 4) <= 256 ? 4 : 8);
        sa.ncons = sb.ncons = ((rows + rpw - 1) / rpw + 7) / 8 * 8;
        const dim3 grid(2 * sa.ncons), block(64 * kC32sWaves);
        launch_joint32_kernel(rpw, ta->gl >= 0, grid, block, st, (const ChainModel*)j->d_tab, sa, sb);
        HIPCHK(hipGetLastError());
      }
      if (group32) {
        CHK(launch_dw32_group({ta, te}, j->d_dw32, rows, s, max_blocks32, st));
        continue;
      }
      for (v21_trainer* t : {ta, te}) {
        CHK(train_on_rows_chain32(t, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, rows, brows, t->d_steploss + s, 0, true));
      }
      continue;
    }
    if (rows > 0) {
      ChainStep sa = chain_step(ta->d_x[0], dsig, nullptr, dsig, ta->d_rw[0], d_idx, lo, rows, brows, dsig, nullptr, lo - first);
      ChainStep sb = chain_step(te->d_x[0], dpar, nullptr, dlat, te->d_rw[0], d_idx, lo, rows, brows, dlat, nullptr, lo - first);
      sa.step_off = (unsigned long long)s;  // the table holds the autoencoder's step counter as of the epoch's start (noise key)
      sb.y_from_lds = 1;
      sa.ncons = sb.ncons = ((rows + 31) / 32 + 7) / 8 * 8;
      sb.blk0 = sa.ncons;                   // the emulator's row blocks follow the autoencoder's in the grid
      sa.npref = sb.npref = chain_prefetchers(2 * sa.ncons, 1);
      const dim3 grid(2 * sa.ncons + 8 * sa.npref), block(64 * kChainWaves);
      launch_joint_kernel(ta->prec, ta->gl >= 0, grid, block, st, (const ChainModel*)j->d_tab, sa, sb);
      HIPCHK(hipGetLastError());
    }
    if (R == 1) {
      CHK(launch_dw_adam_group({ta, te}, j->d_dwadam, j->h_dwadam, rows, brows, s, st));
      continue;
    }
    // data parallel: each model's weight gradients (this rank's rows), summed over the ranks, then Adam -- the same
    // exchange as a plain step (reduce_and_update: all-reduce, or reduce-scatter + sharded Adam + all-gather)
    for (v21_trainer* t : {ta, te}) {
      int fold = 1;
      if (rows > 0) {
        int nslice = 1;
        std::vector<Dw16Args> probs;
        dw16_problems(t, rows, brows, &nslice, probs);
        CHK(launch_dw16(t->prec, probs, st));
        if (nslice > 1) {
          const long long n4 = ((long long)t->P + 3) / 4;
          hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, t->d_g,
                             (const float*)t->d_slab, nslice, (long long)t->P + 4, (long long)t->P);
          HIPCHK(hipGetLastError());
        }
      } else {
        HIPCHK(hipMemsetAsync(t->d_g, 0, (t->P + 1) * sizeof(float), st));
      }
      CHK(reduce_and_update(t, true, fold));
      HIPCHK(hipMemcpyAsync(t->d_steploss + s, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
      invalidate_streams(t->mlp);
      t->mlp->wpad_ok = true;
    }
  }
  std::vector<float> h((size_t)steps * 2);
  HIPCHK(hipMemcpyAsync(h.data(), ta->d_steploss, (size_t)steps * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(h.data() + steps, te->d_steploss, (size_t)steps * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int k = 0; k < 2; ++k) {
    double tot = 0.0;
    for (long long i = 0; i < steps; ++i) tot += (double)h[(size_t)k * steps + i];
    losses[k] = tot / (double)n;
  }
  return V21_OK;
}

// validation of both models in ONE launch: the autoencoder's loss on its validation signals, and the emulator's loss on
// the validation parameters against the latents the CURRENT encoder produces for the validation signals (what the
// reference gets from encoder.predict(signal_val), emulator.py:754) -- no host round trip for the latents
extern "C" int v21_joint_eval(v21_joint* j, double* losses) {
  if (!j || !losses) return fail(V21_ERR_ARG, "null argument");
  v21_trainer *ta = j->ae, *te = j->em;
  if (ta->n[1] < 1 || te->n[1] != ta->n[1]) return fail(V21_ERR_STATE, "both trainers need validation sets of the same row count");
  if (!ta->y_is_x[1]) return fail(V21_ERR_STATE, "the autoencoder's validation targets must be its inputs (y == NULL)");
  CHK(use(ta->ctx));
  hipStream_t st = ta->ctx->stream;
  const long long n = ta->n[1];
  if (n > (1ll << 30)) return fail(V21_ERR_ARG, "too many rows for one validation launch");
  for (v21_trainer* t : {ta, te}) CHK(ensure_copies(t, false));
  {
    std::vector<ChainModel> tab = j->f32 ? std::vector<ChainModel>{chain_model32(ta), chain_model32(te)}
                                         : std::vector<ChainModel>{chain_model(ta), chain_model(te)};
    tab[0].zcap_layer = j->latent_layer;
    if (j->f32) tab[0].stamps = tab[1].stamps = nullptr;
    tab[0].sample = 0;  // evaluation passes draw no noise
    if (tab.size() != j->h_tab.size() || memcmp(tab.data(), j->h_tab.data(), 2 * sizeof(ChainModel)) != 0) {
      HIPCHK(hipStreamSynchronize(st));
      j->h_tab = tab;
      HIPCHK(hipMemcpyAsync(j->d_tab, j->h_tab.data(), 2 * sizeof(ChainModel), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
  }
  CHK(chain_attr(ta->prec));
  const int dsig = ta->mlp->dims[0], dpar = te->mlp->dims[0], dlat = te->mlp->dims[te->mlp->L];
  ChainStep sa = chain_step(ta->d_x[1], dsig, nullptr, dsig, ta->d_rw[1], nullptr, 0, (int)n, (int)n, dsig);
  ChainStep sb = chain_step(te->d_x[1], dpar, nullptr, dlat, te->d_rw[1], nullptr, 0, (int)n, (int)n, dlat);
  sa.fwd_only = sb.fwd_only = 1;
  sb.y_from_lds = 1;
  if (j->f32) {
    // (the row blocks v21_trainer_eval's launch takes for n rows: the same rows meet in the same partial sums, and the
    //  autoencoder's validation loss is bit for bit the one it reports alone)
    const char* er = getenv("V21_C32S_ROWS");
    const int force_rows = er ? atoi(er) : 0;
    const int rpw = force_rows == 4 || force_rows == 8 ? force_rows : (n <= kC32sRows4Max ? 4 : 8);
    sa.gs = sb.gs = 1.0f;
    sa.ncons = sb.ncons = (int)(((n + rpw - 1) / rpw + 7) / 8 * 8);
    const dim3 grid(2 * sa.ncons), block(64 * kC32sWaves);
    launch_joint32_kernel(rpw, ta->gl >= 0, grid, block, st, (const ChainModel*)j->d_tab, sa, sb);
  } else {
    sa.ncons = sb.ncons = (int)(((n + 31) / 32 + 7) / 8 * 8);
    sb.blk0 = sa.ncons;
    sa.npref = sb.npref = chain_prefetchers(2 * sa.ncons, 1);
    const dim3 grid(2 * sa.ncons + 8 * sa.npref), block(64 * kChainWaves);
    launch_joint_kernel(ta->prec, ta->gl >= 0, grid, block, st, (const ChainModel*)j->d_tab, sa, sb);
  }
  HIPCHK(hipGetLastError());
  long long acc[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(&acc[0], ta->d_ticket, sizeof(long long), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&acc[1], te->d_ticket, sizeof(long long), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemsetAsync(ta->d_ticket, 0, sizeof(long long), st));
  HIPCHK(hipMemsetAsync(te->d_ticket, 0, sizeof(long long), st));
  HIPCHK(hipStreamSynchronize(st));
  losses[0] = (double)acc[0] * (1.0 / 4294967296.0) / (double)n;
  losses[1] = (double)acc[1] * (1.0 / 4294967296.0) / (double)n;
  return V21_OK;
}


