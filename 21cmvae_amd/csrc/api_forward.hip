// api_forward.hip -- dense stacks (v21_mlp_*) and the forward routes of include/v21.h: the fused kernels compiled into
// the library (fused_inst.hip) or instantiated at run time (jit.h), the table-driven one-launch forward (the chain
// kernels in FORWARD mode), the few-row latency route and the per-layer route.
#include "api_internal.h"

// ---------------------------------------------------------------------------------
// fused-kernel registry
// ---------------------------------------------------------------------------------
namespace v21 {
#define V21_DECL(a)                                                      \
  hipError_t launch_fused_##a##_F32(const FusedArgs&, hipStream_t);      \
  hipError_t launch_fused_##a##_F16x2sp(const FusedArgs&, hipStream_t);  \
  hipError_t launch_fused_##a##_BF16x2sp(const FusedArgs&, hipStream_t);
V21_ARCH_LIST(V21_DECL)
#undef V21_DECL
// the headline stack's 16-bit kernels with per-workgroup clock stamps (fused_fwd.h: CLOCK_STAMPS; v21_debug_forward_clocked)
hipError_t launch_fused_S1_F16x2spClk(const FusedArgs&, hipStream_t);
hipError_t launch_fused_S1_BF16x2spClk(const FusedArgs&, hipStream_t);
}  // namespace v21

typedef hipError_t (*fused_launcher)(const FusedArgs&, hipStream_t);
struct FusedEntry {
  int L;
  const int* dims;
  const int* act;
  // per precision: f32 = one wave per SIMD on the exact f32 MFMA; f16 / bf16 = two workgroups per CU,
  // one column tile per wave, ring refill spread over the block being consumed (fused_fwd.h: "x2sp")
  fused_launcher fn[3];
};
#define V21_ENTRY(a) \
  {Arch##a::L, Arch##a::dims, Arch##a::act, {launch_fused_##a##_F32, launch_fused_##a##_F16x2sp, launch_fused_##a##_BF16x2sp}},
static const FusedEntry g_fused[] = {V21_ARCH_LIST(V21_ENTRY)};
#undef V21_ENTRY

// ---------------------------------------------------------------------------------
// dense stack
// ---------------------------------------------------------------------------------

static int fpi_of(int prec) { return prec == V21_PREC_F32 ? 8 : 16; }
static void stream_geometry(const v21_mlp* m, int prec, int* total, int* padded) {
  int f = 0;
  for (int l = 0; l < m->L; ++l) f += ((m->dims[l + 1] + 31) / 32) * ((m->dims[l] + fpi_of(prec) - 1) / fpi_of(prec) + 1);
  *total = f;
  *padded = (f + 7) / 8 * 8;  // whole DMA rounds of the 4- and 8-wave kernels (fused_fwd.h: Geo::padded)
}

// archs.h: the compiled fused forward kernel of this stack, or -1
static int fused_id_of(int n_layers, const int* dims, const int* act) {
  for (size_t i = 0; i < sizeof(g_fused) / sizeof(g_fused[0]); ++i) {
    const FusedEntry& fe = g_fused[i];
    if (fe.L != n_layers) continue;
    bool same = true;
    for (int k = 0; k <= n_layers && same; ++k) same = fe.dims[k] == dims[k];
    for (int k = 0; k < n_layers && same; ++k) same = fe.act[k] == act[k];
    if (same) return (int)i;
  }
  return -1;
}
// ---- routes (csrc/routes.h) through the C ABI: pure host logic, no GPU needed
static int check_stack_desc(int n_layers, const int* dims, const int* act) {
  if (!dims || !act) return fail(V21_ERR_ARG, "null argument");
  if (n_layers < 1 || n_layers > 16) return fail(V21_ERR_ARG, "n_layers %d out of range", n_layers);
  for (int l = 0; l <= n_layers; ++l)
    if (dims[l] < 1) return fail(V21_ERR_ARG, "dims[%d] = %d", l, dims[l]);
  for (int l = 0; l < n_layers; ++l)
    if (act[l] < 0 || act[l] > V21_ACT_GAUSS) return fail(V21_ERR_ARG, "act[%d] = %d unknown", l, act[l]);
  return V21_OK;
}
extern "C" int v21_route_forward(int n_layers, const int* dims, const int* act, int precision, int64_t n, int flags, int rt_ready, int* route) {
  if (!route) return fail(V21_ERR_ARG, "null argument");
  CHK(check_stack_desc(n_layers, dims, act));
  if (precision < 0 || precision > 2) return fail(V21_ERR_ARG, "precision %d unknown", precision);
  const int out = dims[n_layers];
  FwdQuery q{n_layers, dims, act, fused_id_of(n_layers, dims, act) >= 0, rt_ready ? 2 : 0, precision, (long long)n, flags & 0xFF, (long long)out};
  if (q.jit == 2) {
    std::string why;
    if (!v21::jit_eligible(n_layers, dims, act, &why)) q.jit = 0;
  }
  *route = decide_forward(q);
  return V21_OK;
}
extern "C" int v21_mlp_last_route(v21_mlp* m, int* route, long long counts[8]) {
  if (!m || !route) return fail(V21_ERR_ARG, "null argument");
  *route = m->last_route;
  if (counts) for (int i = 0; i < 8; ++i) counts[i] = m->route_count[i];
  return V21_OK;
}
extern "C" const char* v21_route_name(int kind, int route) {
  static const char* fwd[] = {"none", "small (one NT launch per layer)", "fused_fwd (compiled)", "fused_fwd (run-time instantiated)",
                              "table-driven chain kernel, FORWARD mode", "generic per-layer GEMM"};
  static const char* tr[] = {"none", "per-layer NT", "train_chain_kernel (16-bit, 32-row blocks)", "fused_train (128-row workgroups)",
                             "fused_train16 (64-row workgroups)", "train_chain32_kernel (fp32, 16-row blocks)",
                             "train_chain32s_kernel<8> (fp32)", "train_chain32s_kernel<4> (fp32)"};
  static const char* up[] = {"none", "per-layer NT + adam_repack", "dw16_adam_kernel (one launch)", "gemm_dw16[_lds] split-K + adam_repack",
                             "dwadam32_kernel (one launch)", "gemm_nt_dwadam_kernel (one launch)", "sliced NT + adam_repack"};
  if (kind == 0 && route >= 0 && route <= FWD_GENERIC) return fwd[route];
  if (kind == 1 && route >= 0 && route <= TR_CHAIN32S_4) return tr[route];
  if (kind == 2 && route >= 0 && route <= UP_NT_SLICED) return up[route];
  return "?";
}

extern "C" int v21_mlp_create(v21_ctx* ctx, int n_layers, const int* dims, const int* act, v21_mlp** out) {
  CHK(use(ctx));
  if (!dims || !act || !out) return fail(V21_ERR_ARG, "null argument");
  if (n_layers < 1 || n_layers > 16) return fail(V21_ERR_ARG, "n_layers %d not in [1,16]", n_layers);
  for (int i = 0; i <= n_layers; ++i)
    if (dims[i] < 1 || dims[i] > 65536) return fail(V21_ERR_ARG, "dims[%d] = %d out of range", i, dims[i]);
  for (int i = 0; i < n_layers; ++i)
    if (act[i] != V21_ACT_LINEAR && act[i] != V21_ACT_RELU && act[i] != V21_ACT_GAUSS)
      return fail(V21_ERR_ARG, "act[%d] = %d unknown", i, act[i]);
  int n_gauss = 0;
  for (int i = 0; i < n_layers; ++i) n_gauss += act[i] == V21_ACT_GAUSS;
  if (n_gauss > 1) return fail(V21_ERR_UNSUPPORTED, "at most one V21_ACT_GAUSS layer per stack");
  v21_mlp* m = new v21_mlp();
  m->ctx = ctx;
  m->L = n_layers;
  m->dims.assign(dims, dims + n_layers + 1);
  m->act.assign(act, act + n_layers);
  long long o = 0;
  for (int l = 0; l < n_layers; ++l) {
    m->w_off.push_back(o); o += (long long)dims[l] * m->nw(l);
    m->b_off.push_back(o); o += m->nw(l);
  }
  m->nparams = (size_t)o;
  m->maxdim = *std::max_element(m->dims.begin(), m->dims.end());
  hipError_t e = hipMalloc((void**)&m->d_w, (m->nparams + kArenaPad) * sizeof(float));
  if (e != hipSuccess) { delete m; return fail(V21_ERR_HIP, "hipMalloc weights: %s", hipGetErrorString(e)); }
  hipMemsetAsync(m->d_w, 0, (m->nparams + kArenaPad) * sizeof(float), ctx->stream);
  m->fused_id = fused_id_of(n_layers, dims, act);
  *out = m;
  return V21_OK;
}
extern "C" int v21_mlp_destroy(v21_mlp* m) {
  if (!m) return V21_OK;
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  hipFree(m->d_w);
  for (int i = 0; i < 3; ++i) if (m->d_stream[i]) hipFree(m->d_stream[i]);
  if (m->d_mean) hipFree(m->d_mean);
  for (int i = 0; i < 2; ++i) if (m->d_act[i]) hipFree(m->d_act[i]);
  for (int i = 0; i < 2; ++i) if (m->d_small[i]) hipFree(m->d_small[i]);
  if (m->d_wt) hipFree(m->d_wt);
  if (m->d_xpad) hipFree(m->d_xpad);
  for (int i = 0; i < 3; ++i) { if (m->d_cfw[i]) hipFree(m->d_cfw[i]); if (m->d_cbw[i]) hipFree(m->d_cbw[i]); }
  if (m->d_tin) hipFree(m->d_tin);
  if (m->d_xs) hipFree(m->d_xs);
  if (m->d_xs64) hipFree(m->d_xs64);
  if (m->d_ys) hipFree(m->d_ys);
  delete m;
  return V21_OK;
}
extern "C" int v21_mlp_num_params(const v21_mlp* m, size_t* n) {
  if (!m || !n) return fail(V21_ERR_ARG, "null argument");
  *n = m->nparams;
  return V21_OK;
}
void invalidate_streams(v21_mlp* m) {
  for (int i = 0; i < 3; ++i) m->stream_ok[i] = false;
  for (int i = 0; i < 3; ++i) m->cfw_ok[i] = false;
  m->wpad_ok = false;
  m->wt_ok = false;
}

extern "C" int v21_mlp_set_weights(v21_mlp* m, const float* flat, size_t n) {
  if (!m || !flat) return fail(V21_ERR_ARG, "null argument");
  if (n != m->nparams) return fail(V21_ERR_ARG, "set_weights: got %zu floats, stack has %zu", n, m->nparams);
  CHK(use(m->ctx));
  HIPCHK(hipMemcpyAsync(m->d_w, flat, n * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
  HIPCHK(hipStreamSynchronize(m->ctx->stream));
  invalidate_streams(m);
  return V21_OK;
}
extern "C" int v21_mlp_get_weights(v21_mlp* m, float* flat, size_t n) {
  if (!m || !flat) return fail(V21_ERR_ARG, "null argument");
  if (n != m->nparams) return fail(V21_ERR_ARG, "get_weights: got room for %zu floats, stack has %zu", n, m->nparams);
  CHK(use(m->ctx));
  HIPCHK(hipMemcpyAsync(flat, m->d_w, n * sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
  HIPCHK(hipStreamSynchronize(m->ctx->stream));
  return V21_OK;
}
extern "C" int v21_mlp_set_input_transform(v21_mlp* m, const v21_affine_in* t) {
  if (!m) return fail(V21_ERR_ARG, "null mlp");
  if (!t) { m->has_tin = false; return V21_OK; }
  if (t->n != m->dims[0] || t->n > 8) return fail(V21_ERR_ARG, "input transform: n = %d, stack input = %d (max 8)", t->n, m->dims[0]);
  for (int j = 0; j < t->n; ++j)
    if (!(t->span[j] == t->span[j]) || !(t->lo[j] == t->lo[j]))
      return fail(V21_ERR_ARG, "input transform: column %d has a NaN minimum or span", j);
  m->tin = *t;
  m->has_tin = true;
  CHK(use(m->ctx));
  if (!m->d_tin) HIPCHK(hipMalloc((void**)&m->d_tin, sizeof(v21_affine_in)));
  HIPCHK(hipMemcpyAsync(m->d_tin, &m->tin, sizeof(v21_affine_in), hipMemcpyHostToDevice, m->ctx->stream));
  HIPCHK(hipStreamSynchronize(m->ctx->stream));
  return V21_OK;
}
extern "C" int v21_mlp_set_output_transform(v21_mlp* m, const v21_affine_out* t) {
  if (!m) return fail(V21_ERR_ARG, "null mlp");
  CHK(use(m->ctx));
  invalidate_streams(m);
  if (!t) { m->has_tout = false; return V21_OK; }
  if (t->n != m->dims[m->L] || !t->mean) return fail(V21_ERR_ARG, "output transform: n = %d, stack output = %d", t->n, m->dims[m->L]);
  if (!m->d_mean) HIPCHK(hipMalloc((void**)&m->d_mean, (size_t)t->n * sizeof(float)));
  HIPCHK(hipMemcpyAsync(m->d_mean, t->mean, (size_t)t->n * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
  HIPCHK(hipStreamSynchronize(m->ctx->stream));
  m->out_std = t->std;
  m->has_tout = true;
  return V21_OK;
}
extern "C" int v21_mlp_has_fused(const v21_mlp* m, int precision, int* yes) {
  if (!m || !yes) return fail(V21_ERR_ARG, "null argument");
  if (precision < 0 || precision > 2) return fail(V21_ERR_ARG, "precision %d unknown", precision);
  *yes = m->fused_id >= 0 ? 1 : 0;
  return V21_OK;
}

static int ensure_stream(v21_mlp* m, int prec) {
  if (m->stream_ok[prec]) return V21_OK;
  int total, padded;
  stream_geometry(m, prec, &total, &padded);
  unsigned char*& dst = m->d_stream[prec];
  if (!dst) HIPCHK(hipMalloc((void**)&dst, (size_t)padded * 1024));
  PackArgs pa{};
  pa.w = m->d_w;
  pa.mean = m->has_tout ? m->d_mean : nullptr;
  pa.stream = dst;
  pa.L = m->L;
  pa.total = total;
  pa.padded = padded;
  pa.fpi = fpi_of(prec);
  pa.epi = prec == V21_PREC_F32 ? 4 : 8;
  pa.esize = prec == V21_PREC_F32 ? 4 : 2;
  pa.is_bf16 = prec == V21_PREC_BF16;
  int f = 0;
  for (int l = 0; l < m->L; ++l) {
    PackLayer& pl = pa.lt[l];
    pl.K = m->dims[l]; pl.N = m->dims[l + 1];
    pl.ks = (pl.K + pa.fpi - 1) / pa.fpi; pl.nt = (pl.N + 31) / 32;
    pl.w_off = m->w_off[l]; pl.b_off = m->b_off[l];
    pl.first = f;
    f += pl.nt * (pl.ks + 1);
  }
  hipLaunchKernelGGL(pack_stream_kernel, dim3((padded + 3) / 4), dim3(256), 0, m->ctx->stream, pa);
  HIPCHK(hipGetLastError());
  m->stream_ok[prec] = true;
  return V21_OK;
}

template <class P, int EP>
static int launch_gemm(GemmArgs g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return V21_OK;
  dim3 grid((g.N + kBN - 1) / kBN, (g.M + kBM - 1) / kBM);
  const bool akc = g.sa_k == 1, bkc = g.sb_k == 1;
  if (!akc && g.sa_m != 1) return fail(V21_ERR_ARG, "gemm: A must be contiguous along m or k");
  if (!bkc && g.sb_n != 1) return fail(V21_ERR_ARG, "gemm: B must be contiguous along k or n");
  if (akc && bkc) hipLaunchKernelGGL((gemm_kernel<P, EP, true, true>), grid, dim3(256), 0, st, g);
  else if (akc) hipLaunchKernelGGL((gemm_kernel<P, EP, true, false>), grid, dim3(256), 0, st, g);
  else if (bkc) hipLaunchKernelGGL((gemm_kernel<P, EP, false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_kernel<P, EP, false, false>), grid, dim3(256), 0, st, g);
  HIPCHK(hipGetLastError());
  return V21_OK;
}
template <int EP>
static int launch_gemm_prec(int prec, const GemmArgs& g, hipStream_t st) {
  switch (prec) {
    case V21_PREC_F32: return launch_gemm<PrecF32, EP>(g, st);
    case V21_PREC_F16: return launch_gemm<PrecF16, EP>(g, st);
    case V21_PREC_BF16: return launch_gemm<PrecBF16, EP>(g, st);
  }
  return fail(V21_ERR_ARG, "precision %d unknown", prec);
}

// one dense layer: out = act(in W + b)
static int dense_forward(v21_mlp* m, int l, const float* in, long long ldin, float* out, long long ldout,
                         int rows, int prec, hipStream_t st) {
  GemmArgs g{};
  g.A = in; g.sa_m = ldin; g.sa_k = 1;
  g.B = m->d_w + m->w_off[l]; g.sb_k = m->nw(l); g.sb_n = 1;  // V21_ACT_GAUSS: the z_mean columns only (z = z_mean)
  g.C = out; g.ldc = ldout;
  g.M = rows; g.N = m->dims[l + 1]; g.K = m->dims[l];
  g.bias = m->d_w + m->b_off[l];
  return m->act[l] == V21_ACT_RELU ? launch_gemm_prec<EP_BIAS_RELU>(prec, g, st) : launch_gemm_prec<EP_BIAS>(prec, g, st);
}

static int forward_generic(v21_mlp* m, const float* d_x, long long ldx, long long n, float* d_y, long long ldy,
                           int prec, int flags) {
  hipStream_t st = m->ctx->stream;
  const long long chunk = 8192;
  if (m->act_rows < chunk) {
    for (int i = 0; i < 2; ++i) {
      if (m->d_act[i]) HIPCHK(hipFree(m->d_act[i]));
      HIPCHK(hipMalloc((void**)&m->d_act[i], (size_t)chunk * m->maxdim * sizeof(float)));
    }
    m->act_rows = chunk;
  }
  for (long long r0 = 0; r0 < n; r0 += chunk) {
    const int rows = (int)std::min(chunk, n - r0);
    const float* in = d_x + r0 * ldx;
    long long ldin = ldx;
    int cur = 0;
    if ((flags & V21_FWD_IN_TRANSFORM) && m->has_tin) {
      const long long tot = (long long)rows * m->dims[0];
      hipLaunchKernelGGL(affine_in_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m->d_act[0],
                         (long long)m->dims[0], in, ldx, (long long)rows, m->tin);
      HIPCHK(hipGetLastError());
      in = m->d_act[0]; ldin = m->dims[0]; cur = 1;
    }
    for (int l = 0; l < m->L; ++l) {
      const bool last = l == m->L - 1;
      float* out = last ? d_y + r0 * ldy : m->d_act[cur];
      const long long ldo = last ? ldy : m->dims[l + 1];
      CHK(dense_forward(m, l, in, ldin, out, ldo, rows, prec, st));
      in = out; ldin = ldo; cur ^= 1;
    }
    if ((flags & V21_FWD_OUT_TRANSFORM) && m->has_tout) {
      const long long tot = (long long)rows * m->dims[m->L];
      hipLaunchKernelGGL(affine_out_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_y + r0 * ldy,
                         ldy, (long long)rows, m->dims[m->L], m->out_std, m->d_mean);
      HIPCHK(hipGetLastError());
    }
  }
  return V21_OK;
}

static int forward_small(v21_mlp* m, const float* d_x, long long ldx, long long n, float* d_y, long long ldy, int prec,
                         int flags);
// Any stack up to 512 wide in f16 / bf16: the whole forward pass in ONE launch of the chain kernel in FORWARD mode
// (train_chain.h) -- what every `_gen_model` output without a compiled fused kernel gets (custom `hidden_dims`,
// emulator.py:12-48; the members of a sweep).  r3, 65,536 rows: the per-layer K-loop path took 0.85 ms on the
// headline stack (15x the fused kernel) and 0.35 ms on the sample notebook's 7 -> [64, 128] -> 451 model.
static int forward_chain(v21_mlp* m, const float* d_x, long long ldx, long long n, float* d_y, long long ldy, int prec,
                         int flags);
// internal: d_x rows are already zero-padded to a multiple of 16 floats (ldx) in a buffer with slack --
// the small-batch path reads them in place (set by v21_mlp_forward, which pads on the host)
#define V21_FWD_X_PADDED 0x100
// internal: this device call belongs to a host call that found no run-time kernel when it started -- do not pick one up
// half way (one result array must not hold two roundings); decide_forward (routes.h) knows the flag
static_assert(V21_FWD_RT_LATCH_OFF == 0x200, "routes.h and api_forward.hip agree on the internal flag");
// the run-time kernel of this stack: asked for on first need; true when its code object can be launched now
static bool jit_ready_now(v21_mlp* m, int precision) {
  if (m->fused_id >= 0) return false;
  if (!m->jit_asked[precision]) {
    m->jit[precision] = v21::jit_request(m->L, m->dims.data(), m->act.data(), precision);
    m->jit_asked[precision] = true;
  }
  return m->jit[precision] && v21::jit_state(m->jit[precision]) == v21::JIT_READY;
}

extern "C" int v21_mlp_forward_dev(v21_mlp* m, const float* d_x, int64_t ldx, int64_t n, float* d_y, int64_t ldy,
                                   int precision, int flags) {
  if (!m || !d_x || !d_y) return fail(V21_ERR_ARG, "null argument");
  if (precision < 0 || precision > 2) return fail(V21_ERR_ARG, "precision %d unknown", precision);
  if (n < 0 || ldx < m->dims[0] || ldy < m->dims[m->L]) return fail(V21_ERR_ARG, "bad shape: n=%lld ldx=%lld ldy=%lld", (long long)n, (long long)ldx, (long long)ldy);
  if (n == 0) return V21_OK;
  CHK(use(m->ctx));
  if ((flags & V21_FWD_IN_TRANSFORM) && !m->has_tin) return fail(V21_ERR_STATE, "input transform requested but not set");
  if ((flags & V21_FWD_OUT_TRANSFORM) && !m->has_tout) return fail(V21_ERR_STATE, "output transform requested but not set");
  // which route: csrc/routes.h (decide_forward) -- the same function answers v21_route_forward
  FwdQuery q{m->L, m->dims.data(), m->act.data(), m->fused_id >= 0, 0, precision, (long long)n, flags, (long long)ldy};
  // few rows: one latency-oriented launch per layer beats one wave walking the whole stack in f32
  // (and the K-loop GEMM of the generic path in any precision)
  if (decide_forward(q) == FWD_SMALL) {
    m->last_route = FWD_SMALL; m->route_count[FWD_SMALL] += 1;
    return forward_small(m, d_x, ldx, n, d_y, ldy, precision, flags);
  }
  // a stack outside archs.h: the same fused kernel, instantiated for it at run time (jit.h).  The first call asks for
  // it; the calls that arrive before its code object does take the table-driven routes below.  (V21_FWD_RT_LATCH_OFF:
  // a host call that found the kernel missing when it started keeps the table-driven route for ALL its slices.)
  v21::JitKernel* jk = nullptr;
  const bool force_jit = (flags & V21_FWD_FORCE_JIT) != 0;
  if ((m->fused_id < 0 || force_jit) && !(flags & (V21_FWD_FORCE_GENERIC | V21_FWD_FORCE_CHAIN | V21_FWD_RT_LATCH_OFF)) && ldy < (1ll << 21) &&
      (!(flags & V21_FWD_IN_TRANSFORM) || m->dims[0] <= 8)) {
    if (!m->jit_asked[precision]) {
      m->jit[precision] = v21::jit_request(m->L, m->dims.data(), m->act.data(), precision);
      m->jit_asked[precision] = true;
    }
    if (force_jit && m->jit[precision]) v21::jit_wait(m->jit[precision], -1);  // (diagnostics: this route or an error)
    if (m->jit[precision] && v21::jit_state(m->jit[precision]) == v21::JIT_READY) jk = m->jit[precision];
    if (force_jit && !jk) {
      std::string why = "not eligible, or V21_JIT=0";
      if (m->jit[precision]) v21::jit_state(m->jit[precision], &why);
      return fail(V21_ERR_UNSUPPORTED, "V21_FWD_FORCE_JIT: no run-time kernel for this stack: %s", why.c_str());
    }
  }
  q.jit = jk ? 2 : 0;
  const int route = decide_forward(q);
  m->last_route = route; m->route_count[route] += 1;
  if (route == FWD_TABLE) return forward_chain(m, d_x, ldx, n, d_y, ldy, precision, flags);
  if (route == FWD_GENERIC) return forward_generic(m, d_x, ldx, n, d_y, ldy, precision, flags);
  CHK(ensure_stream(m, precision));
  FusedArgs a{};
  a.x = d_x; a.ldx = ldx; a.y = d_y; a.ldy = ldy; a.n_rows = n;
  a.stream = m->d_stream[precision];
  const bool tout = (flags & V21_FWD_OUT_TRANSFORM) != 0;
  a.out_std = tout ? m->out_std : 1.0f;
  a.out_mean_scale = tout ? 1.0f : 0.0f;
  a.in_transform = (flags & V21_FWD_IN_TRANSFORM) ? 1 : 0;
  if (a.in_transform) a.tin = m->tin;
#ifdef V21_FUSED_STAMP  // diagnostic build only: where the cycle stamps go
  a.dbg = (unsigned long long*)(getenv("V21_FUSED_DBG_PTR") ? strtoull(getenv("V21_FUSED_DBG_PTR"), nullptr, 0) : 0ull);
#endif
  if (jk) {
    const hipError_t e = v21::jit_launch(jk, m->ctx->device, a, m->ctx->stream);
    if (e == hipSuccess) return V21_OK;
    (void)hipGetLastError();
    if (force_jit) {
      std::string why;
      v21::jit_state(jk, &why);
      return fail(V21_ERR_UNSUPPORTED, "V21_FWD_FORCE_JIT: %s (%s)", why.c_str(), hipGetErrorString(e));
    }
    // the code object could not be loaded or needs scratch memory (jit_launch marked it failed): this call and every
    // later one take the table-driven route
    m->route_count[route] -= 1;
    q.jit = 0;
    const int r2 = decide_forward(q);
    m->last_route = r2; m->route_count[r2] += 1;
    if (r2 == FWD_TABLE) return forward_chain(m, d_x, ldx, n, d_y, ldy, precision, flags);
    return forward_generic(m, d_x, ldx, n, d_y, ldy, precision, flags);
  }
  if (m->clk_stamps) {  // v21_debug_forward_clocked: the clock-stamped instantiation of the same kernel
    a.dbg = m->clk_stamps;
    HIPCHK((precision == V21_PREC_F16 ? launch_fused_S1_F16x2spClk : launch_fused_S1_BF16x2spClk)(a, m->ctx->stream));
    return V21_OK;
  }
  HIPCHK(g_fused[m->fused_id].fn[precision](a, m->ctx->stream));
  return V21_OK;
}
extern "C" int v21_debug_forward_clocked(v21_mlp* m, const float* d_x, int64_t ldx, int64_t n, float* d_y, int64_t ldy, int precision,
                                         int flags, unsigned long long* d_stamps) {
  if (!m || !d_stamps) return fail(V21_ERR_ARG, "null argument");
  if (m->fused_id < 0 || g_fused[m->fused_id].dims != ArchS1::dims || (precision != V21_PREC_F16 && precision != V21_PREC_BF16))
    return fail(V21_ERR_UNSUPPORTED, "clock-stamped kernels exist for the headline stack (archs.h S1) in f16 / bf16 only");
  if (flags & (V21_FWD_FORCE_GENERIC | V21_FWD_FORCE_CHAIN | V21_FWD_FORCE_JIT)) return fail(V21_ERR_ARG, "route flags do not apply");
  m->clk_stamps = d_stamps;
  const int r = v21_mlp_forward_dev(m, d_x, ldx, n, d_y, ldy, precision, flags | V21_FWD_NO_SMALL);
  m->clk_stamps = nullptr;
  return r;
}

// ---- run-time instantiation of the fused kernel (csrc/jit.h) through the C ABI
extern "C" int v21_mlp_jit(v21_mlp* m, int precision, int wait_ms, int* status) {
  if (!m || !status) return fail(V21_ERR_ARG, "null argument");
  if (precision < 0 || precision > 2) return fail(V21_ERR_ARG, "precision %d unknown", precision);
  *status = -1;
  if (m->fused_id >= 0) { *status = 1; return V21_OK; }  // compiled into the library (archs.h)
  std::string why;
  if (!v21::jit_eligible(m->L, m->dims.data(), m->act.data(), &why)) return fail(V21_ERR_UNSUPPORTED, "no fused kernel for this stack: %s", why.c_str());
  if (!m->jit_asked[precision]) {
    m->jit[precision] = v21::jit_request(m->L, m->dims.data(), m->act.data(), precision);
    m->jit_asked[precision] = true;
  }
  v21::JitKernel* k = m->jit[precision];
  if (!k) return fail(V21_ERR_UNSUPPORTED, "run-time compilation is switched off (V21_JIT=0) and no cached kernel exists");
  int s = v21::jit_state(k);
  if (s == v21::JIT_COMPILING && wait_ms != 0) s = v21::jit_wait(k, wait_ms);
  *status = s;
  if (s == v21::JIT_FAILED) {
    v21::jit_state(k, &why);
    return fail(V21_ERR_UNSUPPORTED, "fused kernel of this stack: %s", why.c_str());
  }
  return V21_OK;
}
extern "C" int v21_jit_prebuild(int n_layers, const int* dims, const int* act, int precision, const char* dir) {
  if (!dims || !act) return fail(V21_ERR_ARG, "null argument");
  std::string why;
  if (v21::jit_prebuild(n_layers, dims, act, precision, dir, &why) != 0) return fail(V21_ERR_UNSUPPORTED, "%s", why.c_str());
  return V21_OK;
}

extern "C" int v21_mlp_forward(v21_mlp* m, const void* x, int x_dtype, int64_t n, float* y, int precision, int flags) {
  if (!m || !x || !y) return fail(V21_ERR_ARG, "null argument");
  if (n < 0) return fail(V21_ERR_ARG, "negative row count");
  if (x_dtype != V21_DTYPE_F32 && x_dtype != V21_DTYPE_F64) return fail(V21_ERR_ARG, "x_dtype %d unknown", x_dtype);
  if (n == 0) return V21_OK;
  CHK(use(m->ctx));
  hipStream_t st = m->ctx->stream;
  const int din = m->dims[0], dout = m->dims[m->L];
  const long long chunk = 1 << 18;  // rows per host round trip
  const long long need = std::min<long long>(n, chunk);
  if (m->stage_rows < need) {
    if (m->d_xs) HIPCHK(hipFree(m->d_xs));
    if (m->d_ys) HIPCHK(hipFree(m->d_ys));
    if (m->d_xs64) HIPCHK(hipFree(m->d_xs64));
    HIPCHK(hipMalloc((void**)&m->d_xs, (size_t)need * din * sizeof(float)));
    HIPCHK(hipMalloc((void**)&m->d_xs64, (size_t)need * din * sizeof(double)));
    HIPCHK(hipMalloc((void**)&m->d_ys, (size_t)need * dout * sizeof(float)));
    m->stage_rows = need;
  }
  std::vector<float> tmp;
  flags &= 0xFF;
  if (precision < 0 || precision > 2) return fail(V21_ERR_ARG, "precision %d unknown", precision);
  if ((flags & V21_FWD_IN_TRANSFORM) && !m->has_tin) return fail(V21_ERR_STATE, "input transform requested but not set");
  if ((flags & V21_FWD_OUT_TRANSFORM) && !m->has_tout) return fail(V21_ERR_STATE, "output transform requested but not set");
  const bool tin = (flags & V21_FWD_IN_TRANSFORM) != 0;
  // par_transform of one value ON THE HOST, for the few-row route below: par_transform.h's two functions themselves
  // (__host__ __device__; r5: until r4 the host used libm's log10f and the device the float64 log10 rounded to float32,
  // so a float32 parameter vector could differ in its last bit depending on how many rows the call had -- ADVICE r4)
  auto host_value = [&](long long r, int j) -> float {
    const v21_affine_in& a = m->tin;
    if (x_dtype == V21_DTYPE_F64) {
      const double t = ((const double*)x)[r * din + j];
      if (!tin) return (float)t;  // Keras casts float64 inputs to float32 [K]
      return par_transform_f64(t, a.log_mask[j], a.zero_floor[j], a.lo[j], a.span[j]);
    }
    const float f = ((const float*)x)[r * din + j];
    if (!tin) return f;
    return par_transform_f32(f, a.log_mask[j], a.zero_floor[j], a.lo[j], a.span[j]);
  };
  const FwdQuery hq{m->L, m->dims.data(), m->act.data(), m->fused_id >= 0, 0, precision, (long long)n, flags & ~V21_FWD_IN_TRANSFORM, (long long)dout};
  if (decide_forward(hq) == FWD_SMALL && (!tin || din <= 8)) {
    // few rows: transform (if asked) and pad the rows on the host, so the first layer reads the staging buffer in
    // place (two launches fewer than transforming on the device)
    const long long ldp = p16(din);
    if (m->stage_pad_rows < n) {
      if (m->d_xpad) HIPCHK(hipFree(m->d_xpad));
      HIPCHK(hipMalloc((void**)&m->d_xpad, (size_t)(n + 2) * ldp * sizeof(float)));
      m->stage_pad_rows = n;
    }
    tmp.assign((size_t)n * ldp, 0.f);
    for (long long r = 0; r < n; ++r)
      for (int j = 0; j < din; ++j) tmp[(size_t)r * ldp + j] = host_value(r, j);
    HIPCHK(hipMemcpyAsync(m->d_xpad, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice, st));
    m->last_route = FWD_SMALL; m->route_count[FWD_SMALL] += 1;
    CHK(forward_small(m, m->d_xpad, ldp, n, m->d_ys, dout, precision, (flags & ~V21_FWD_IN_TRANSFORM) | V21_FWD_X_PADDED));
    HIPCHK(hipMemcpyAsync(y, m->d_ys, (size_t)n * dout * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return V21_OK;
  }
  // The route is decided ONCE per host call: a call that starts before the run-time kernel of its stack has arrived
  // keeps the table-driven route for every chunk and slice (the two routes round differently: ADVICE r4), the next call
  // takes the kernel.  v21_mlp_jit(mlp, precision, -1) waits for it: bit-stable results from the first call on.
  const bool rt_off = m->fused_id < 0 && !(flags & (V21_FWD_FORCE_JIT | V21_FWD_FORCE_GENERIC | V21_FWD_FORCE_CHAIN)) && !jit_ready_now(m, precision);
  for (long long r0 = 0; r0 < n; r0 += chunk) {
    const long long rows = std::min(chunk, n - r0);
    int fl = flags | (rt_off ? V21_FWD_RT_LATCH_OFF : 0);
    if (x_dtype == V21_DTYPE_F64 && tin) {
      // float64 parameters: staged as they are and transformed in float64 on the device (the reference's float64
      // branch, preprocess.py:74-108), the float32 cast after the map as Keras does it [K]
      const double* xd = (const double*)x + r0 * din;
      HIPCHK(hipMemcpyAsync(m->d_xs64, xd, (size_t)rows * din * sizeof(double), hipMemcpyHostToDevice, st));
      const long long tot = rows * din;
      hipLaunchKernelGGL(affine_in_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m->d_xs, (long long)din,
                         (const double*)m->d_xs64, (long long)din, rows, m->tin);
      HIPCHK(hipGetLastError());
      fl &= ~V21_FWD_IN_TRANSFORM;
    } else {
      const float* src;
      if (x_dtype == V21_DTYPE_F64) {  // Keras casts float64 inputs to float32 [K]
        tmp.resize((size_t)rows * din);
        const double* xd = (const double*)x + r0 * din;
        for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = (float)xd[i];
        src = tmp.data();
      } else {
        src = (const float*)x + r0 * din;
      }
      HIPCHK(hipMemcpyAsync(m->d_xs, src, (size_t)rows * din * sizeof(float), hipMemcpyHostToDevice, st));
    }
    // The results are 1,804 B per row against 28-56 B of input: the call is bound by their way back over PCIe
    // (65,536 rows: 118 MB, ~2.1 ms).  Slices of kSliceRows rows are computed on the context's stream and copied
    // out on a second one, so that only the FIRST slice's kernel is not hidden under a copy (f32, 65,536 rows:
    // 0.46 ms of kernel + 2.2 ms of copy one after the other -> 0.12 + 2.2 ms).
    constexpr long long kSliceRows = 16384;
    if (rows <= kSliceRows) {
      CHK(v21_mlp_forward_dev(m, m->d_xs, din, rows, m->d_ys, dout, precision, fl));
      HIPCHK(hipMemcpyAsync(y + r0 * dout, m->d_ys, (size_t)rows * dout * sizeof(float), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      continue;
    }
    v21_ctx* c = m->ctx;
    if (!c->copy_stream) {
      HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
      for (hipEvent_t& e : c->slice_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // equal slices of whole 256-row workgroup rounds; every slice takes the route the whole call would take (a short
    // last slice must not fall onto the few-row path: another summation order within one result)
    const long long nsl = (rows + kSliceRows - 1) / kSliceRows;
    const long long per = ((rows + nsl - 1) / nsl + 255) / 256 * 256;
    int k = 0;
    for (long long s0 = 0; s0 < rows; s0 += per, ++k) {
      const long long srows = std::min(per, rows - s0);
      CHK(v21_mlp_forward_dev(m, m->d_xs + s0 * din, din, srows, m->d_ys + s0 * dout, dout, precision, fl | V21_FWD_NO_SMALL));
      // (an event is reused every other slice: the copy that waited on its previous record was enqueued before this one)
      HIPCHK(hipEventRecord(c->slice_done[k & 1], st));
      HIPCHK(hipStreamWaitEvent(c->copy_stream, c->slice_done[k & 1], 0));
      HIPCHK(hipMemcpyAsync(y + (r0 + s0) * dout, m->d_ys + s0 * dout, (size_t)srows * dout * sizeof(float), hipMemcpyDeviceToHost,
                            c->copy_stream));
    }
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    HIPCHK(hipStreamSynchronize(st));
  }
  return V21_OK;
}


// Model.predict on a few rows (emulator.py:402 called from a sampler): one gemm_nt launch per layer
static int forward_small(v21_mlp* m, const float* d_x, long long ldx, long long n, float* d_y, long long ldy, int prec,
                         int flags) {
  hipStream_t st = m->ctx->stream;
  const int L = m->L, rows = (int)n;
  if (!m->d_wt) {
    long long ot = 0;
    for (int l = 0; l < L; ++l) { m->wt_off.push_back(ot); ot += (long long)(m->nw(l) + 32) * p16(m->dims[l]); }
    HIPCHK(hipMalloc((void**)&m->d_wt, (size_t)(ot + 64) * sizeof(float)));
    HIPCHK(hipMemsetAsync(m->d_wt, 0, (size_t)(ot + 64) * sizeof(float), st));
    for (int i = 0; i < 2; ++i) {
      const size_t nb = (size_t)(V21_SMALL_BATCH_ROWS + 32) * p16(m->maxdim) * sizeof(float);
      HIPCHK(hipMalloc((void**)&m->d_small[i], nb));
      HIPCHK(hipMemsetAsync(m->d_small[i], 0, nb, st));
    }
  }
  if (!m->wt_ok) {
    AdamArgs a{};
    a.w = m->d_w; a.wt = m->d_wt; a.n = (long long)m->nparams; a.L = L;
    for (int l = 0; l < L; ++l) {
      AdamLayer& al = a.lt[l];
      al.w_off = m->w_off[l]; al.wt_off = m->wt_off[l]; al.K = m->dims[l]; al.N = m->nw(l); al.ldwt = p16(al.K);
    }
    hipLaunchKernelGGL(wt_pack_kernel, dim3((unsigned)((m->nparams + 255) / 256)), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    m->wt_ok = true;
  }
  const long long ld0 = p16(m->dims[0]);
  const float* a0 = m->d_small[0];
  long long lda0 = ld0;
  if (flags & V21_FWD_X_PADDED) {
    a0 = d_x; lda0 = ldx;
  } else if ((flags & V21_FWD_IN_TRANSFORM) && m->has_tin) {
    const long long tot = (long long)rows * m->dims[0];
    hipLaunchKernelGGL(affine_in_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m->d_small[0], ld0, d_x,
                       ldx, (long long)rows, m->tin);
  } else {
    const long long tot = (long long)rows * ld0;
    hipLaunchKernelGGL(copy_pad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m->d_small[0], ld0, d_x,
                       ldx, (long long)rows, m->dims[0]);
  }
  HIPCHK(hipGetLastError());
  int cur = 0;
  bool unpre_done = false;
  for (int l = 0; l < L; ++l) {
    const bool last = l == L - 1;
    NtGroup grp{};
    grp.count = 1;
    NtArgs& g = grp.p[0];
    g.A = l == 0 ? a0 : m->d_small[cur]; g.lda = l == 0 ? lda0 : p16(m->dims[l]);
    g.B = m->d_wt + m->wt_off[l]; g.ldb = p16(m->dims[l]);
    g.C = last ? d_y : m->d_small[cur ^ 1]; g.ldc = last ? ldy : p16(m->dims[l + 1]);
    g.CT = nullptr;
    g.M = rows; g.N = m->dims[l + 1]; g.K = m->dims[l];  // V21_ACT_GAUSS: the z_mean rows of W^T only
    g.bias = m->d_w + m->b_off[l];
    g.ep = m->act[l] == V21_ACT_RELU ? NT_FWD_RELU : NT_FWD;
    if (last && (flags & V21_FWD_OUT_TRANSFORM) && m->has_tout && m->act[l] != V21_ACT_RELU) {
      g.ep = NT_FWD_UNPRE; g.aff_mean = m->d_mean; g.aff_std = m->out_std;  // unpreproc in the epilogue
      unpre_done = true;
    }
    g.nz = 1;
    CHK(launch_nt(prec, grp, st));
    cur ^= 1;
  }
  if ((flags & V21_FWD_OUT_TRANSFORM) && m->has_tout && !unpre_done) {
    const long long tot = (long long)rows * m->dims[L];
    hipLaunchKernelGGL(affine_out_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_y, ldy, (long long)rows,
                       m->dims[L], m->out_std, m->d_mean);
    HIPCHK(hipGetLastError());
  }
  return V21_OK;
}

// ---- FORWARD mode of the chain kernel for a stack without a trainer (declared above v21_mlp_forward_dev)
static int ensure_chain_stream(v21_mlp* m, int prec) {
  if (m->cfw_ok[prec]) return V21_OK;
  hipStream_t st = m->ctx->stream;
  const int L = m->L;
  const int pc = prec == V21_PREC_F32 ? 1 : 0;
  const int esz = pc ? 4 : 2;
  if (m->cfw_off[pc].empty()) {
    long long of = 0, ob = 0;  // elements
    for (int l = 0; l < L; ++l) {
      const int K = m->dims[l], N = m->nw(l);
      m->cfw_off[pc].push_back(of); of += (long long)((N + 31) / 32) * (pc ? chain32_frags(K) * 256 : chain_steps(K) * 512);
      m->cbw_off[pc].push_back(ob); ob += (long long)((K + 31) / 32) * (pc ? chain32_frags(N) * 256 : chain_steps(N) * 512);
    }
    m->cfw_bytes[pc] = of * esz; m->cbw_bytes[pc] = ob * esz;
  }
  if (!m->d_cfw[prec]) {
    HIPCHK(hipMalloc(&m->d_cfw[prec], (size_t)m->cfw_bytes[pc] + kChainStreamSlack)); HIPCHK(hipMemsetAsync(m->d_cfw[prec], 0, (size_t)m->cfw_bytes[pc] + kChainStreamSlack, st));
    HIPCHK(hipMalloc(&m->d_cbw[prec], (size_t)m->cbw_bytes[pc] + kChainStreamSlack)); HIPCHK(hipMemsetAsync(m->d_cbw[prec], 0, (size_t)m->cbw_bytes[pc] + kChainStreamSlack, st));
  }
  AdamArgs a{};  // the arena -> the packed streams (the trainer's repacking kernel without the Adam update)
  a.w = m->d_w; a.n = (long long)m->nparams; a.L = L; a.do_adam = 0; a.skip_nt = 1;
  a.fw = m->d_cfw[prec]; a.bw = m->d_cbw[prec]; a.cprec = prec == V21_PREC_F32 ? 3 : prec == V21_PREC_F16 ? 1 : 2;
  for (int l = 0; l < L; ++l) {
    AdamLayer& al = a.lt[l];
    al.w_off = m->w_off[l]; al.K = m->dims[l]; al.N = m->nw(l);
    al.fw_off = m->cfw_off[pc][l]; al.bw_off = m->cbw_off[pc][l];
    al.KS = pc ? chain32_frags(al.K) : chain_steps(al.K); al.NS = pc ? chain32_frags(al.N) : chain_steps(al.N);
  }
  hipLaunchKernelGGL(adam_repack_kernel, dim3((unsigned)((m->nparams + 255) / 256)), dim3(256), 0, st, a);
  HIPCHK(hipGetLastError());
  m->cfw_ok[prec] = true;
  return V21_OK;
}
static int forward_chain(v21_mlp* m, const float* d_x, long long ldx, long long n, float* d_y, long long ldy, int prec,
                         int flags) {
  hipStream_t st = m->ctx->stream;
  const int L = m->L;
  const int pc = prec == V21_PREC_F32 ? 1 : 0;
  CHK(ensure_chain_stream(m, prec));
  CHK(chain_attr(prec));
  ChainArgs a{};
  a.L = L;
  for (int l = 0; l < L; ++l) {
    ChainLayer& c = a.lt[l];
    c.K = m->dims[l]; c.N = m->nw(l);
    c.gauss = m->act[l] == V21_ACT_GAUSS;
    c.KS = pc ? chain32_frags(c.K) : chain_steps(c.K); c.NT = (c.N + 31) / 32;
    c.NS = pc ? chain32_frags(c.N) : chain_steps(c.N); c.KT = (c.K + 31) / 32;
    c.relu = m->act[l] == V21_ACT_RELU;
    c.mask_tile = -1;  // no backward pass: no ReLU masks kept
    c.fw_off = m->cfw_off[pc][l] / (pc ? 4 : 8); c.bw_off = m->cbw_off[pc][l] / (pc ? 4 : 8);  // units of one lane's 16 bytes
    c.b_off = m->b_off[l];
  }
  a.fw = m->d_cfw[prec]; a.bw = m->d_cbw[prec]; a.w = m->d_w;
  a.fw_bytes = m->cfw_bytes[pc]; a.bw_bytes = 0;  // (the prefetchers touch the forward stream only)
  a.zcap_layer = -1;
  a.sample = 0;  // a variational head evaluates z = z_mean (include/v21.h)
  a.x = d_x; a.ldx = ldx; a.rows = (int)n;
  a.fwd_only = 1;
  a.out = d_y; a.ldo = ldy;
  const bool tout = (flags & V21_FWD_OUT_TRANSFORM) && m->has_tout;
  a.out_std = tout ? m->out_std : 1.0f;
  a.out_mean = tout ? m->d_mean : nullptr;
  a.tin = ((flags & V21_FWD_IN_TRANSFORM) && m->has_tin) ? m->d_tin : nullptr;
  if (pc) return launch_chain32_args(a, st);
  a.ncons = (int)(((n + 31) / 32 + 7) / 8 * 8);
  a.npref = chain_prefetchers(a.ncons, 1);
  const dim3 grid(a.ncons + 8 * a.npref), block(64 * kChainWaves);
  launch_chain_forward_mode(prec, grid, block, st, a);
  HIPCHK(hipGetLastError());
  return V21_OK;
}


