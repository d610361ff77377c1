// v21_api.hip -- implementation of the C ABI declared in include/v21.h.
//
// Host-side orchestration of the HIP kernels in fused_fwd.h (K1), gemm.h (K3) and
// train_kernels.h (K2, K4, K5): contexts and streams, the flat fp32 parameter arena
// (Keras get_weights() order), the packed weight stream of the fused forward kernel,
// the per-layer fallback, the Keras-fit()-shaped epoch driver and the RCCL gradient
// all-reduce (K6; RCCL is dlopen'ed so single-GPU users never need it).
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

#include "../../include/v21.h"
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
#ifdef V21_CHAIN_FINE
constexpr int kStampSlots = 2048;  // (diagnostic build: per-wave stamps)
#else
constexpr int kStampSlots = 64;
#endif
#include "dw_adam.h"

using namespace v21;

// ---------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
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

extern "C" const char* v21_last_error(void) { return g_err.c_str(); }
static inline long long p16(int d) { return (d + 15) & ~15; }  // row pitch: whole 16-float groups
// Zeroed bytes behind every packed weight stream of the chain kernels: two 4-KiB chunks.  A stream is whole chunks, so
// its end is a page boundary; the rolling prefetch requests addresses AHEAD of what it uses, and a request must never
// leave the allocation (train_chain32s.h: the r3 abort).
constexpr size_t kChainStreamSlack = 8192;
// floats behind the P parameters of an arena: the loss slot, then room to round P + 1 up to whole shards of up to
// 64 ranks (sharded data-parallel Adam works on nranks * ceil((P + 1) / nranks) elements in place)
constexpr size_t kArenaPad = 4 + 64;
extern "C" int v21_version(void) { return 100; }
extern "C" int v21_device_count(int* n) {
  if (!n) return fail(V21_ERR_ARG, "null n");
  HIPCHK(hipGetDeviceCount(n));
  return V21_OK;
}

// ---------------------------------------------------------------------------------
// RCCL, loaded at run time
// ---------------------------------------------------------------------------------
struct nccl_uid { char internal[128]; };
typedef void* nccl_comm;
typedef int (*fn_GetUniqueId)(nccl_uid*);
typedef int (*fn_CommInitRank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_ReduceScatter)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
typedef int (*fn_CommDestroy)(nccl_comm);
typedef const char* (*fn_GetErrorString)(int);
typedef int (*fn_CommCount)(nccl_comm, int*);
struct RcclApi {
  void* lib = nullptr;
  fn_GetUniqueId GetUniqueId = nullptr;
  fn_CommInitRank CommInitRank = nullptr;
  fn_AllReduce AllReduce = nullptr;
  fn_ReduceScatter ReduceScatter = nullptr;
  fn_AllGather AllGather = nullptr;
  fn_CommDestroy CommDestroy = nullptr;
  fn_GetErrorString GetErrorString = nullptr;
  fn_CommCount CommCount = nullptr, CommUserRank = nullptr;
};
static RcclApi g_rccl;
static int load_rccl() {
  if (g_rccl.lib) return V21_OK;
  const char* cands[] = {getenv("V21_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* c : cands) {
    if (!c || !*c) continue;
    lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(V21_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
  g_rccl.GetUniqueId = (fn_GetUniqueId)dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (fn_CommInitRank)dlsym(lib, "ncclCommInitRank");
  g_rccl.AllReduce = (fn_AllReduce)dlsym(lib, "ncclAllReduce");
  g_rccl.ReduceScatter = (fn_ReduceScatter)dlsym(lib, "ncclReduceScatter");
  g_rccl.AllGather = (fn_AllGather)dlsym(lib, "ncclAllGather");
  g_rccl.CommDestroy = (fn_CommDestroy)dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (fn_GetErrorString)dlsym(lib, "ncclGetErrorString");
  g_rccl.CommCount = (fn_CommCount)dlsym(lib, "ncclCommCount");
  g_rccl.CommUserRank = (fn_CommCount)dlsym(lib, "ncclCommUserRank");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy || !g_rccl.ReduceScatter ||
      !g_rccl.AllGather)
    return fail(V21_ERR_COMM, "librccl lacks a required symbol");
  g_rccl.lib = lib;
  return V21_OK;
}
static const char* rccl_err(int r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"; }

// ---------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------
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
  // v21_mlp_forward on many rows: results leave over PCIe on a second stream, slice by slice, while the next slice
  // is being computed (created on first use)
  hipStream_t copy_stream = nullptr;
  hipEvent_t slice_done[2] = {nullptr, nullptr};
  // v21_debug_clock_probe_*: the sampling wave runs on its own stream beside the kernels under test
  hipStream_t probe_stream = nullptr;
  unsigned long long* d_probe = nullptr;
  int probe_cap = 0;
};
// ncclDataType_t / ncclRedOp_t values of rccl.h (the library is dlopen'ed, its header is not included)
constexpr int kNcclFloat32 = 7, kNcclSum = 0;
static int use(v21_ctx* c) {
  if (!c) return fail(V21_ERR_ARG, "null context");
  HIPCHK(hipSetDevice(c->device));
  return V21_OK;
}

extern "C" int v21_ctx_create(int device, v21_ctx** out) {
  if (!out) return fail(V21_ERR_ARG, "null out");
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(V21_ERR_ARG, "device %d out of range (%d visible)", device, n);
  HIPCHK(hipSetDevice(device));
  v21_ctx* c = new v21_ctx();
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return fail(V21_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
  c->stream = c->own;
  *out = c;
  return V21_OK;
}
extern "C" int v21_ctx_destroy(v21_ctx* c) {
  if (!c) return V21_OK;
  hipSetDevice(c->device);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->own) { hipStreamSynchronize(c->own); hipStreamDestroy(c->own); }
  if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
  for (hipEvent_t e : c->slice_done) if (e) hipEventDestroy(e);
  if (c->probe_stream) { hipStreamSynchronize(c->probe_stream); hipStreamDestroy(c->probe_stream); }
  if (c->d_probe) hipFree(c->d_probe);
  delete c;
  return V21_OK;
}
extern "C" int v21_ctx_sync(v21_ctx* c) { CHK(use(c)); HIPCHK(hipStreamSynchronize(c->stream)); return V21_OK; }
extern "C" int v21_ctx_set_stream(v21_ctx* c, void* s) { CHK(use(c)); c->stream = s ? (hipStream_t)s : c->own; return V21_OK; }
extern "C" int v21_ctx_get_stream(v21_ctx* c, void** s) { if (!c || !s) return fail(V21_ERR_ARG, "null"); *s = (void*)c->stream; return V21_OK; }

extern "C" int v21_malloc(v21_ctx* c, size_t bytes, void** p) {
  CHK(use(c));
  if (!p) return fail(V21_ERR_ARG, "null dptr");
  HIPCHK(hipMalloc(p, bytes ? bytes : 4));
  return V21_OK;
}
extern "C" int v21_host_alloc(v21_ctx* c, size_t bytes, void** p) {
  CHK(use(c));
  if (!p || bytes == 0) return fail(V21_ERR_ARG, "bad host allocation request");
  HIPCHK(hipHostMalloc(p, bytes, hipHostMallocDefault));
  return V21_OK;
}
extern "C" int v21_host_free(v21_ctx* c, void* p) { CHK(use(c)); if (p) HIPCHK(hipHostFree(p)); return V21_OK; }
extern "C" int v21_free(v21_ctx* c, void* p) { CHK(use(c)); if (p) HIPCHK(hipFree(p)); return V21_OK; }
extern "C" int v21_memcpy_h2d(v21_ctx* c, void* d, const void* s, size_t b) {
  CHK(use(c));
  HIPCHK(hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return V21_OK;
}
extern "C" int v21_memcpy_d2h(v21_ctx* c, void* d, const void* s, size_t b) {
  CHK(use(c));
  HIPCHK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return V21_OK;
}
extern "C" int v21_memset(v21_ctx* c, void* d, int v, size_t b) { CHK(use(c)); HIPCHK(hipMemsetAsync(d, v, b, c->stream)); return V21_OK; }
extern "C" int v21_event_create(v21_ctx* c, void** ev) {
  CHK(use(c));
  hipEvent_t e;
  HIPCHK(hipEventCreate(&e));
  *ev = (void*)e;
  return V21_OK;
}
extern "C" int v21_event_destroy(v21_ctx* c, void* ev) { CHK(use(c)); HIPCHK(hipEventDestroy((hipEvent_t)ev)); return V21_OK; }
extern "C" int v21_event_record(v21_ctx* c, void* ev) { CHK(use(c)); HIPCHK(hipEventRecord((hipEvent_t)ev, c->stream)); return V21_OK; }
extern "C" int v21_event_elapsed_ms(v21_ctx* c, void* a, void* b, float* ms) {
  CHK(use(c));
  HIPCHK(hipEventSynchronize((hipEvent_t)b));
  HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return V21_OK;
}

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
  // width of layer l's Dense output: dims[l+1], or 2*dims[l+1] = [z_mean | z_log_var] for V21_ACT_GAUSS
  int nw(int l) const { return act[l] == V21_ACT_GAUSS ? 2 * dims[l + 1] : dims[l + 1]; }
};

static int fpi_of(int prec) { return prec == V21_PREC_F32 ? 8 : 16; }
static void stream_geometry(const v21_mlp* m, int prec, int* total, int* padded) {
  int f = 0;
  for (int l = 0; l < m->L; ++l) f += ((m->dims[l + 1] + 31) / 32) * ((m->dims[l] + fpi_of(prec) - 1) / fpi_of(prec) + 1);
  *total = f;
  *padded = (f + 7) / 8 * 8;  // whole DMA rounds of the 4- and 8-wave kernels (fused_fwd.h: Geo::padded)
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
  for (size_t i = 0; i < sizeof(g_fused) / sizeof(g_fused[0]); ++i) {
    const FusedEntry& fe = g_fused[i];
    if (fe.L != n_layers) continue;
    bool same = true;
    for (int k = 0; k <= n_layers && same; ++k) same = fe.dims[k] == dims[k];
    for (int k = 0; k < n_layers && same; ++k) same = fe.act[k] == act[k];
    if (same) { m->fused_id = (int)i; break; }
  }
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
static void invalidate_streams(v21_mlp* m) {
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
static bool chain_fwd_eligible(const v21_mlp* m, int precision, int flags) {
  if (flags & V21_FWD_FORCE_GENERIC) return false;
  for (int l = 0; l <= m->L; ++l)
    if (m->dims[l] > kChainMaxDim) return false;
  for (int l = 0; l < m->L; ++l)
    if (m->act[l] == V21_ACT_GAUSS && (precision == V21_PREC_F32 || m->dims[l + 1] > kChainMaxLatent || l == m->L - 1)) return false;
  if ((flags & V21_FWD_IN_TRANSFORM) && m->dims[0] > 8) return false;
  return true;
}
static int forward_chain(v21_mlp* m, const float* d_x, long long ldx, long long n, float* d_y, long long ldy, int prec,
                         int flags);
// internal: d_x rows are already zero-padded to a multiple of 16 floats (ldx) in a buffer with slack --
// the small-batch path reads them in place (set by v21_mlp_forward, which pads on the host)
#define V21_FWD_X_PADDED 0x100
static bool takes_small_path(const v21_mlp* m, long long n, int precision, int flags) {
  const bool fused = m->fused_id >= 0 && !(flags & V21_FWD_FORCE_GENERIC) &&
                     (!(flags & V21_FWD_IN_TRANSFORM) || m->dims[0] <= 8);
  return n <= V21_SMALL_BATCH_ROWS && !(flags & (V21_FWD_NO_SMALL | V21_FWD_FORCE_GENERIC | V21_FWD_FORCE_CHAIN | V21_FWD_FORCE_JIT)) &&
         (precision == V21_PREC_F32 || !fused) && (!(flags & V21_FWD_IN_TRANSFORM) || m->dims[0] <= 8) &&
         m->maxdim <= kNtMaxKPerWg;
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
  const bool fused = m->fused_id >= 0 && !(flags & (V21_FWD_FORCE_GENERIC | V21_FWD_FORCE_CHAIN | V21_FWD_FORCE_JIT)) &&
                     (!(flags & V21_FWD_IN_TRANSFORM) || m->dims[0] <= 8) && ldy < (1ll << 21);
  // few rows: one latency-oriented launch per layer beats one wave walking the whole stack in f32
  // (and the K-loop GEMM of the generic path in any precision)
  if (takes_small_path(m, n, precision, flags) && ldy < (1ll << 21))
    return forward_small(m, d_x, ldx, n, d_y, ldy, precision, flags);
  // a stack outside archs.h: the same fused kernel, instantiated for it at run time (jit.h).  The first call asks for
  // it; the calls that arrive before its code object does take the table-driven routes below.
  v21::JitKernel* jk = nullptr;
  const bool force_jit = (flags & V21_FWD_FORCE_JIT) != 0;
  if ((m->fused_id < 0 || force_jit) && !(flags & (V21_FWD_FORCE_GENERIC | V21_FWD_FORCE_CHAIN)) && ldy < (1ll << 21) &&
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
  if (!fused && !jk && chain_fwd_eligible(m, precision, flags) && n < (1ll << 30)) return forward_chain(m, d_x, ldx, n, d_y, ldy, precision, flags);
  if (!fused && !jk) return forward_generic(m, d_x, ldx, n, d_y, ldy, precision, flags);
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
    if (chain_fwd_eligible(m, precision, flags) && n < (1ll << 30)) return forward_chain(m, d_x, ldx, n, d_y, ldy, precision, flags);
    return forward_generic(m, d_x, ldx, n, d_y, ldy, precision, flags);
  }
  HIPCHK(g_fused[m->fused_id].fn[precision](a, m->ctx->stream));
  return V21_OK;
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
  // par_transform of one value ON THE HOST, for the few-row route below: the same two branches as the device's
  // (par_transform.h) with libm's log10 / log10f -- what numpy calls for float64 / float32 arrays
  auto host_value = [&](long long r, int j) -> float {
    if (x_dtype == V21_DTYPE_F64) {
      double t = ((const double*)x)[r * din + j];
      if (!tin) return (float)t;  // Keras casts float64 inputs to float32 [K]
      const v21_affine_in& a = m->tin;
      if (a.zero_floor[j] > 0.0 && t == 0.0) t = a.zero_floor[j];
      if (a.log_mask[j]) t = std::log10(t);
      t -= a.lo[j]; t /= a.span[j]; t *= 2.0; t -= 1.0;  // preprocess.py:105-108, in this order
      return (float)t;
    }
    float f = ((const float*)x)[r * din + j];
    if (!tin) return f;
    const v21_affine_in& a = m->tin;
    if (a.zero_floor[j] > 0.0 && f == 0.f) f = (float)a.zero_floor[j];
    double t = a.log_mask[j] ? (double)std::log10(f) : (double)f;  // (log10f: np.log10 of a float32 array)
    t -= a.lo[j]; t /= a.span[j]; t *= 2.0; t -= 1.0;
    return (float)t;
  };
  if (takes_small_path(m, n, precision, flags & ~V21_FWD_IN_TRANSFORM) && (!tin || din <= 8)) {
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
    CHK(forward_small(m, m->d_xpad, ldp, n, m->d_ys, dout, precision, (flags & ~V21_FWD_IN_TRANSFORM) | V21_FWD_X_PADDED));
    HIPCHK(hipMemcpyAsync(y, m->d_ys, (size_t)n * dout * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return V21_OK;
  }
  for (long long r0 = 0; r0 < n; r0 += chunk) {
    const long long rows = std::min(chunk, n - r0);
    int fl = flags;
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

// ---------------------------------------------------------------------------------
// communicator
// ---------------------------------------------------------------------------------
extern "C" int v21_comm_get_unique_id(v21_ctx* c, void* id) {
  CHK(use(c));
  if (!id) return fail(V21_ERR_ARG, "null id");
  CHK(load_rccl());
  nccl_uid u;
  int r = g_rccl.GetUniqueId(&u);
  if (r != 0) return fail(V21_ERR_COMM, "ncclGetUniqueId: %s", rccl_err(r));
  memcpy(id, &u, sizeof u);
  return V21_OK;
}
extern "C" int v21_comm_init(v21_ctx* c, int nranks, int rank, const void* id) {
  CHK(use(c));
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(V21_ERR_ARG, "bad communicator arguments");
  if (c->comm || c->host_comm) return fail(V21_ERR_STATE, "communicator already initialised");
  CHK(load_rccl());
  nccl_uid u;
  memcpy(&u, id, sizeof u);
  int r = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
  if (r != 0) { c->comm = nullptr; return fail(V21_ERR_COMM, "ncclCommInitRank: %s", rccl_err(r)); }
  c->nranks = nranks;
  c->rank = rank;
  return V21_OK;
}
extern "C" int v21_comm_init_host(v21_ctx* c, int nranks, int rank, const v21_comm_host_ops* ops) {
  CHK(use(c));
  if (!ops || !ops->allreduce_sum_f32 || !ops->reduce_scatter_sum_f32 || !ops->allgather_f32 || nranks < 1 || rank < 0 ||
      rank >= nranks)
    return fail(V21_ERR_ARG, "bad communicator arguments");
  if (c->comm || c->host_comm) return fail(V21_ERR_STATE, "communicator already initialised");
  c->host = *ops;
  c->host_comm = true;
  c->nranks = nranks;
  c->rank = rank;
  return V21_OK;
}
extern "C" int v21_comm_destroy(v21_ctx* c) {
  CHK(use(c));
  if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
  c->host_comm = false;
  c->nranks = 1; c->rank = 0; c->sharded = 0;
  return V21_OK;
}
// what the attached communicator itself says (RCCL: ncclCommCount / ncclCommUserRank; host transport: what the host
// passed): transport 0 = none, 1 = RCCL inside the library, 2 = host-staged callbacks
extern "C" int v21_comm_info(v21_ctx* c, int* nranks, int* rank, int* transport) {
  if (!c || !nranks || !rank || !transport) return fail(V21_ERR_ARG, "null argument");
  *nranks = c->nranks; *rank = c->rank;
  *transport = c->comm ? 1 : (c->host_comm ? 2 : 0);
  if (c->comm && g_rccl.CommCount && g_rccl.CommUserRank) {
    int r = g_rccl.CommCount(c->comm, nranks);
    if (r == 0) r = g_rccl.CommUserRank(c->comm, rank);
    if (r != 0) return fail(V21_ERR_COMM, "ncclCommCount / ncclCommUserRank: %s", rccl_err(r));
  }
  return V21_OK;
}
extern "C" int v21_comm_set_sharded(v21_ctx* c, int on) {
  if (!c) return fail(V21_ERR_ARG, "null context");
  c->sharded = on ? 1 : 0;
  return V21_OK;
}
static int host_stage(v21_ctx* c, size_t n) {
  if (c->h_stage_n >= n) return V21_OK;
  if (c->h_stage) HIPCHK(hipHostFree(c->h_stage));
  HIPCHK(hipHostMalloc((void**)&c->h_stage, n * sizeof(float), hipHostMallocDefault));
  c->h_stage_n = n;
  return V21_OK;
}
// device buffer -> page-locked host copy -> callback -> back (the stream is drained on both sides: the callback
// blocks in the host's transport)
template <class F>
static int host_collective(v21_ctx* c, float* d_buf, size_t n, F&& call, const char* what) {
  CHK(host_stage(c, n));
  HIPCHK(hipMemcpyAsync(c->h_stage, d_buf, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int r = call(c->h_stage);
  if (r != 0) return fail(V21_ERR_COMM, "host %s callback returned %d", what, r);
  HIPCHK(hipMemcpyAsync(d_buf, c->h_stage, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return V21_OK;
}
extern "C" int v21_comm_allreduce_f32(v21_ctx* c, float* d_buf, size_t n) {
  CHK(use(c));
  if (c->nranks <= 1) return V21_OK;  // single rank: identity
  if (c->host_comm)
    return host_collective(c, d_buf, n, [&](float* h) { return c->host.allreduce_sum_f32(c->host.user, h, n); }, "all-reduce");
  int r = g_rccl.AllReduce(d_buf, d_buf, n, kNcclFloat32, kNcclSum, c->comm, c->stream);
  if (r != 0) return fail(V21_ERR_COMM, "ncclAllReduce: %s", rccl_err(r));
  return V21_OK;
}
// in place over nranks * n_per floats: rank r ends up with the sums of elements [r n_per, (r+1) n_per) there
extern "C" int v21_comm_reduce_scatter_f32(v21_ctx* c, float* d_buf, size_t n_per) {
  CHK(use(c));
  if (c->nranks <= 1) return V21_OK;
  if (c->host_comm)
    return host_collective(c, d_buf, n_per * c->nranks,
                           [&](float* h) { return c->host.reduce_scatter_sum_f32(c->host.user, h, n_per); }, "reduce-scatter");
  int r = g_rccl.ReduceScatter(d_buf, d_buf + (size_t)c->rank * n_per, n_per, kNcclFloat32, kNcclSum, c->comm, c->stream);
  if (r != 0) return fail(V21_ERR_COMM, "ncclReduceScatter: %s", rccl_err(r));
  return V21_OK;
}
// in place over nranks * n_per floats: every rank contributes elements [r n_per, (r+1) n_per) and receives all
extern "C" int v21_comm_allgather_f32(v21_ctx* c, float* d_buf, size_t n_per) {
  CHK(use(c));
  if (c->nranks <= 1) return V21_OK;
  if (c->host_comm)
    return host_collective(c, d_buf, n_per * c->nranks, [&](float* h) { return c->host.allgather_f32(c->host.user, h, n_per); },
                           "all-gather");
  int r = g_rccl.AllGather(d_buf + (size_t)c->rank * n_per, d_buf, n_per, kNcclFloat32, c->comm, c->stream);
  if (r != 0) return fail(V21_ERR_COMM, "ncclAllGather: %s", rccl_err(r));
  return V21_OK;
}

// ---------------------------------------------------------------------------------
// trainer (NT path: gemm_nt.h).  Every contraction of a step reads operands whose
// contraction index is contiguous; the producers write the transposed copies.
//   h[l]   (batch x p16(dims[l]))      activations, row-major          (forward A operand, ReLU mask)
//   ht[l]  ((dims[l]+1) x Bp)          activations transposed + a row of ones (weight-gradient A operand)
//   dz[l]  (batch x p16(dims[l]))      gradient w.r.t. pre-activation of layer l-1's output (backward A operand)
//   dzt[l] (dims[l] x Bp)              its transpose                    (weight-gradient B operand)
//   wt[l]  (N x p16(K)) = W^T          forward B operand;   wp[l] (K x p16(N)) = row-padded W: backward B operand
// ---------------------------------------------------------------------------------

struct v21_trainer {
  v21_mlp* mlp = nullptr;
  v21_ctx* ctx = nullptr;
  int prec = 0, max_batch = 0;
  v21_adam adam{1e-3f, 0.9f, 0.999f, 1e-7f};
  long long iter = 0;
  size_t P = 0;
  float *d_g = nullptr, *d_m = nullptr, *d_v = nullptr;  // P + 4 floats; d_g[P] = loss slot
  float* d_x[2] = {nullptr, nullptr};
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
  struct StepGraph { int rows, brows; const void *x, *y, *rw, *idx; long long row0; hipGraph_t graph; hipGraphExec_t exec; };
  std::vector<StepGraph> graphs;
  int graph_misses = 0;
};
constexpr long long kDescRing = 1024;
static StepCtx step_ctx(const v21_trainer* t) { return t->capturing ? StepCtx{t->d_desc, t->d_cur} : StepCtx{nullptr, nullptr}; }
static void destroy_graphs(v21_trainer* t) {
  for (auto& g : t->graphs) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); }
  t->graphs.clear();
  t->desc_count = 0; t->desc_next = 0; t->desc_iter0 = -1;
}

static int zalloc(float** p, size_t nfloat, hipStream_t st) {
  HIPCHK(hipMalloc((void**)p, nfloat * sizeof(float)));
  HIPCHK(hipMemsetAsync(*p, 0, nfloat * sizeof(float), st));
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
  {  // eligibility of the chain kernel
    const char* env = getenv("V21_TRAIN_CHAIN");
    bool ok = precision != V21_PREC_F32 && !(env && env[0] == '0') &&
              (t->gl < 0 || m->dims[t->gl + 1] <= kChainMaxLatent);
    int mask_tiles = 0;
    for (int l = 0; l <= L && ok; ++l) ok = m->dims[l] <= kChainMaxDim;
    for (int l = 0; l + 1 < L; ++l) mask_tiles += m->act[l] == V21_ACT_RELU ? (m->dims[l + 1] + 31) / 32 : 0;
    ok = ok && mask_tiles <= kChainMaskTiles;
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
      t->BS = ((long long)max_batch + 31) / 32 * 2 + 2;
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
  {  // eligibility of the fp32 chain kernel (train_chain32.h)
    const char* env = getenv("V21_TRAIN_CHAIN");
    bool ok = precision == V21_PREC_F32 && !(env && env[0] == '0');
    int mask_tiles = 0;
    for (int l = 0; l <= L && ok; ++l) ok = m->dims[l] <= kChainMaxDim;
    // a trainer of small batches (the reference's 256 rows) takes the 8-row kernel: twice the workgroups, half the
    // matrix work in each (train_chain32s.h); V21_CHAIN32S = 0 / 1 overrides the choice
    const char* es = getenv("V21_CHAIN32S");
    t->chain32s = es ? es[0] == '1' : max_batch <= kC32sMaxBatch;
    // a variational head: the small-batch kernel carries it (latent <= kChainMaxLatent), the 16-row kernel does not
    if (t->gl >= 0) ok = ok && t->chain32s && m->dims[t->gl + 1] <= kChainMaxLatent;
    for (int l = 0; l + 1 < L; ++l) mask_tiles += m->act[l] == V21_ACT_RELU ? t->c32_tiles(m->dims[l + 1]) : 0;
    ok = ok && mask_tiles <= (t->chain32s ? kC32sMaskTiles : kC32MaskTiles);
    if (!ok) t->chain32s = false;
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
  if (t->d_desc) hipFree(t->d_desc);
  if (t->h_desc) hipHostFree(t->h_desc);
  if (t->d_cur) hipFree(t->d_cur);
  if (t->d_steploss) hipFree(t->d_steploss);
  if (t->d_slab) hipFree(t->d_slab);
  if (t->d_zs) { hipFree(t->d_zs); hipFree(t->d_dzs); hipFree(t->d_dzst); hipFree(t->d_klrow); }
  if (t->d_dworder) hipFree(t->d_dworder);
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
static float grad_opscale(int brows, int dout) {
  const double s = (double)brows * (double)dout / 16.0;
  return (float)std::ldexp(1.0, std::max(0, std::min(24, (int)std::lround(std::log2(std::max(1.0, s))))));
}
static float adam_alpha(const v21_adam& a, long long t) {
  // [K] alpha_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in float32
  const float b1p = powf(a.beta1, (float)t), b2p = powf(a.beta2, (float)t);
  return a.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
}

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
    if (g.k_chunk > kNtMaxKPerWg) return fail(V21_ERR_UNSUPPORTED, "contraction range %d > %d per workgroup", g.k_chunk, kNtMaxKPerWg);
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

// Adam (do_adam) and/or refresh of the W^T / padded-W copies from the arena
static AdamArgs adam_args(v21_trainer* t, bool do_adam, float alpha, bool skip_nt = false);
static int adam_and_copies(v21_trainer* t, bool do_adam, float alpha, bool skip_nt = false, int nslab = 1) {
  AdamArgs a = adam_args(t, do_adam, alpha, skip_nt);
  if (nslab > 1) { a.gw = t->d_g; a.slab = t->d_slab; a.nslab = nslab; a.slab_stride = (long long)t->P + 4; }
  hipLaunchKernelGGL(adam_repack_kernel, dim3((unsigned)((t->P + 255) / 256)), dim3(256), 0, t->ctx->stream, a);
  HIPCHK(hipGetLastError());
  t->copies_ok = true;
  t->nt_ok = !skip_nt;
  return V21_OK;
}
static AdamArgs adam_args(v21_trainer* t, bool do_adam, float alpha, bool skip_nt) {
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
static int reduce_and_update(v21_trainer* t, bool chain_copies, int fold) {
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
    HIPCHK(hipMemcpyAsync(t->d_g + P, w + P, sizeof(float), hipMemcpyDeviceToDevice, st));  // the loss slot, on every rank
    CHK(adam_and_copies(t, false, 0.f, chain_copies));  // packed copies from the gathered arena
    return V21_OK;
  }
  CHK(v21_comm_allreduce_f32(c, t->d_g, P + 1));
  t->iter += 1;
  CHK(adam_and_copies(t, true, adam_alpha(t->adam, t->iter), chain_copies, fold));
  return V21_OK;
}

// need_nt: the caller reads the fp32 copies (per-layer forward/backward); chain steps do not
static int ensure_copies(v21_trainer* t, bool need_nt = true) {
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
  if (rows > 0) {
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
  CHK(reduce_and_update(t, false, 1));
  if (loss_out && !in_table) HIPCHK(hipMemcpyAsync(loss_out, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
  invalidate_streams(m);
  m->wpad_ok = true;  // ... but our own copies were just refreshed
  return V21_OK;
}

static int gather_batch(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy_src,
                        const float* rw, const int* d_idx, long long first, int rows);

// forward + loss + activation gradients of the chain path: ONE launch (train_chain.h)
static ChainModel chain_model(v21_trainer* t) {
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
static ChainStep chain_step(const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                            const int* d_idx, long long first, int rows, int brows, int dout,
                            const v21_trainer* vae = nullptr, long long row0 = 0) {
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
static void launch_joint32_kernel(int rpw, bool gauss, dim3 grid, dim3 block, hipStream_t st, const ChainModel* tab, const ChainStep& sa,
                                  const ChainStep& sb) {
  if (rpw == 4) {
    if (gauss) hipLaunchKernelGGL((train_chain32s_joint_kernel<4, true>), grid, block, kC32sLdsBytes, st, tab, sa, sb);
    else hipLaunchKernelGGL((train_chain32s_joint_kernel<4, false>), grid, block, kC32sLdsBytes, st, tab, sa, sb);
  } else {
    if (gauss) hipLaunchKernelGGL((train_chain32s_joint_kernel<8, true>), grid, block, kC32sLdsBytes, st, tab, sa, sb);
    else hipLaunchKernelGGL((train_chain32s_joint_kernel<8, false>), grid, block, kC32sLdsBytes, st, tab, sa, sb);
  }
}
static void launch_joint_kernel(int prec, bool gauss, dim3 grid, dim3 block, hipStream_t st, const ChainModel* tab, const ChainStep& sa,
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
static int chain_attr(int prec) {
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
static int chain_prefetchers(int ncons, int models) {
  if (models > 1) return 0;  // a sweep: measured slower with them (8 models, 24 prefetchers each: 106 k -> 95 k model-steps/s)
  const int idle = 256 - ncons;
  static const char* env = getenv("V21_CHAIN_PREF");  // (diagnosis: prefetcher workgroups per XCD, 0 = none)
  if (env) return std::max(0, std::min(atoi(env), idle / 8));
  // (per XCD: none 45.6 us per f16 step at 4,096 rows, 2-4 43.4-43.6, 8 43.8-43.9, 16 44.2; f32 at batch 256: 46.4 / 42.6-42.8 / 42.9 / 43.4)
  return idle >= 8 ? std::min(4, idle / 8) : 0;
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
static ChainModel chain_model32(v21_trainer* t) {
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
static int launch_chain32_args(ChainArgs& a, hipStream_t st, bool small = false) {
  CHK(chain_attr(V21_PREC_F32));
  if (small) {  // the 8-row kernel (train_chain32s.h), or its 4-row form
    const char* er = getenv("V21_C32S_ROWS");  // (tests force either form on every case)
    const int force_rows = er ? atoi(er) : 0;
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
  if (prec == V21_PREC_F16) launch_chain_kernel<PrecF16>(kChainFwd | kChainOut | kChainGauss, grid, block, st, a);
  else launch_chain_kernel<PrecBF16>(kChainFwd | kChainOut | kChainGauss, grid, block, st, a);
  HIPCHK(hipGetLastError());
  return V21_OK;
}

// weight gradients of chain-mode trainers: problems in groups of <= 16 per launch
static void dw16_problems(v21_trainer* t, int rows, int brows, int* nslice_out, std::vector<Dw16Args>& probs,
                          float* loss_out2 = nullptr) {
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
      HIPCHK(hipFuncSetAttribute((const void*)gemm_dw16_lds_kernel<PrecF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kDwLdsBytes));
    else
      HIPCHK(hipFuncSetAttribute((const void*)gemm_dw16_lds_kernel<PrecBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kDwLdsBytes));
    attr_done[prec] = true;
  }
  return V21_OK;
}
static int launch_dw16(int prec, const std::vector<Dw16Args>& probs, hipStream_t st) {
  // large batches: 128x128 tiles staged through LDS (half the bytes pulled into a CU per MFMA)
  const bool big = !probs.empty() && probs[0].steps >= 64;
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
    blocks *= grp.p[0].nz;
    if (blocks <= 0) continue;
    const dim3 grid((blocks + 7) / 8 * 8);  // whole rounds of the 8 XCDs (the kernels remap block ids XCD-wise)
    if (big) {
      if (prec == V21_PREC_F16) hipLaunchKernelGGL(gemm_dw16_lds_kernel<PrecF16>, grid, dim3(kDwThreads), kDwLdsBytes, st, grp);
      else hipLaunchKernelGGL(gemm_dw16_lds_kernel<PrecBF16>, grid, dim3(kDwThreads), kDwLdsBytes, st, grp);
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
static int refresh_dw_adam_table(const std::vector<v21_trainer*>& tr, DwAdamModel** d_tab, std::vector<DwAdamModel>& h_tab,
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
static int launch_dw_adam_group(const std::vector<v21_trainer*>& tr, const DwAdamModel* d_tab,
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
  if (tr[0]->prec == V21_PREC_F16) hipLaunchKernelGGL(dw16_adam_group_kernel<PrecF16>, grid, dim3(64 * kDwAdamWaves), 0, st, d_tab, stp);
  else hipLaunchKernelGGL(dw16_adam_group_kernel<PrecBF16>, grid, dim3(64 * kDwAdamWaves), 0, st, d_tab, stp);
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
static int launch_nt_many(int prec, std::vector<NtArgs>& probs, hipStream_t st);
// one f32 optimizer step in THREE launches (train_chain32.h): the chain over this rank's rows, every layer's weight
// gradient in one grouped NT launch on the fp32 operands the chain left, Adam (which also rebuilds the packed fp32
// streams and, on a single rank, publishes the batch loss)
static int train_on_rows_chain32(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy, const float* rw,
                                 const int* d_idx, long long first, int rows, int brows, float* loss_out, long long row0,
                                 bool chain_done = false /* the joint step: the chain of this model ran in the joint launch */) {
  v21_mlp* m = t->mlp;
  hipStream_t st = t->ctx->stream;
  const int L = m->L, dout = m->dims[L];
  if (rows > t->max_batch) return fail(V21_ERR_ARG, "batch of %d rows exceeds max_batch %d", rows, t->max_batch);
  const bool single = t->ctx->nranks == 1;
  const bool in_table = single && rows > 0 && loss_out && t->d_steploss && loss_out >= t->d_steploss &&
                        loss_out < t->d_steploss + t->steploss_cap;
  int fold = 1;
  if (rows > 0) {
    if (!chain_done) {
      CHK(ensure_copies(t, false));
      ChainArgs a{};
      static_cast<ChainModel&>(a) = chain_model32(t);
      static_cast<ChainStep&>(a) = chain_step(x, ldx, y, ldy, rw, d_idx, first, rows, brows, dout, t, row0);
      a.gs = 1.0f;  // fp32 operands: no scaling of the gradients
      CHK(launch_chain32_args(a, st, t->chain32s));
    }
    // one rank, a step of <= kDw32MaxRows rows, 32 x 32 tiles: gradients, Adam, packed streams and batch loss in ONE
    // launch whose workgroups walk the whole batch in slabs of 256 rows (dw_adam32.h)
    static const bool lds_rows = !(getenv("V21_DW32_LDS") && getenv("V21_DW32_LDS")[0] == '0');
    static const bool fused_ok = !(getenv("V21_DW32_ADAM") && getenv("V21_DW32_ADAM")[0] == '0');
    long long work = 0;
    for (int l = 0; l < L; ++l) work += (long long)((m->dims[l] + 1 + 63) / 64) * ((m->nw(l) + 63) / 64);
    const bool dw32 = single && fused_ok && lds_rows && L <= kNtMaxGroup && work < 192 && rows <= kDw32MaxRows;
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
    if (single && nslice == 1 && L <= kNtMaxGroup && fused_ok) {
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
      if (T == 2) hipLaunchKernelGGL(gemm_nt_dwadam_kernel<2>, dim3(blocks), dim3(256), 0, st, grp, ad);
      else if (rows <= kDw32MaxRows && lds_rows) hipLaunchKernelGGL(dwadam32_kernel, dim3(blocks), dim3(256), 0, st, grp, ad);  // operands through LDS in whole rows (dw_adam32.h)
      else hipLaunchKernelGGL(gemm_nt_dwadam_kernel<1>, dim3(blocks), dim3(256), 0, st, grp, ad);
      HIPCHK(hipGetLastError());
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
  } else {
    HIPCHK(hipMemsetAsync(t->d_g, 0, (t->P + 1) * sizeof(float), st));
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
  if (rows > 0) {
    CHK(ensure_copies(t, false));
    CHK(launch_chain(t, x, ldx, y, ldy, rw, d_idx, first, rows, brows, row0));
    // Single rank, nothing to exchange: gradients, Adam and the packed copies in one launch (dw_adam.h) -- up to the
    // batch where its 32 x 32 tiles, each pulling its operands over the WHOLE batch through one CU, lose to the
    // 128 x 128 LDS-staged split-K kernel + an Adam launch that sums the slabs (V21_DW_SPLIT_ROWS overrides the
    // threshold; measured r3, autoencoder stack, f16: see DESIGN.md section 3)
    static const int split_rows = getenv("V21_DW_SPLIT_ROWS") ? atoi(getenv("V21_DW_SPLIT_ROWS")) : 8192;
    if (t->ctx->nranks == 1 && (rows < split_rows || t->capturing)) {
      if (!t->capturing) t->iter += 1;
      // an epoch's per-step loss slot is written by the kernel itself (a device-to-device copy per step is a launch)
      const bool in_table = loss_out && t->d_steploss && loss_out >= t->d_steploss && loss_out < t->d_steploss + t->steploss_cap;
      CHK(launch_dw_adam(t, rows, brows, t->capturing ? 0.f : adam_alpha(t->adam, t->iter),
                         in_table ? (int)(loss_out - t->d_steploss) : -1));
      if (t->capturing) return V21_OK;
      if (loss_out && !in_table) HIPCHK(hipMemcpyAsync(loss_out, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
      invalidate_streams(m);
      m->wpad_ok = true;
      return V21_OK;
    }
    int nslice = 1;
    std::vector<Dw16Args> probs;
    dw16_problems(t, rows, brows, &nslice, probs);  // every weight gradient in one launch: [dW; db] = [H^T; 1^T] dZ
    CHK(launch_dw16(t->prec, probs, st));
    fold = nslice > 1 && t->ctx->nranks == 1 ? nslice : 1;  // single rank: Adam sums the slabs itself
    if (nslice > 1 && fold == 1) {
      const long long n4 = ((long long)t->P + 3) / 4;
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, t->d_g,
                         (const float*)t->d_slab, nslice, (long long)t->P + 4, (long long)t->P);
      HIPCHK(hipGetLastError());
    }
  } else {
    HIPCHK(hipMemsetAsync(t->d_g, 0, (t->P + 1) * sizeof(float), st));
  }
  if (t->capturing) {
    CHK(adam_and_copies(t, true, 0.f, true, fold));
    return V21_OK;
  }
  CHK(reduce_and_update(t, true, fold));
  if (loss_out) HIPCHK(hipMemcpyAsync(loss_out, t->d_g + t->P, sizeof(float), hipMemcpyDeviceToDevice, st));
  invalidate_streams(m);
  m->wpad_ok = true;
  return V21_OK;
}

static int gather_batch(v21_trainer* t, const float* x, long long ldx, const float* y, long long ldy_src,
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
  delete s;
  return V21_OK;
}

// launch `probs` in groups of <= kNtMaxGroup
static int launch_nt_many(int prec, std::vector<NtArgs>& probs, hipStream_t st) {
  for (size_t o = 0; o < probs.size(); o += kNtMaxGroup) {
    NtGroupBig grp{};
    grp.count = (int)std::min<size_t>(kNtMaxGroup, probs.size() - o);
    for (int i = 0; i < grp.count; ++i) grp.p[i] = probs[o + i];
    CHK(launch_nt(prec, grp, st));
  }
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
    csp.npref = chain_prefetchers(csp.ncons, G);
    const dim3 grid(csp.ncons + 8 * csp.npref, G), block(64 * kChainWaves);
    bool gauss = false;  // (train_chain.h: FEAT)
    for (v21_trainer* t : s->tr) gauss = gauss || t->gl >= 0;
    if (t0->prec == V21_PREC_F16) {
      if (gauss) hipLaunchKernelGGL((train_chain_group_kernel<PrecF16, true>), grid, block, kChainLdsBytes, st, (const ChainModel*)s->d_chain, csp);
      else hipLaunchKernelGGL((train_chain_group_kernel<PrecF16, false>), grid, block, kChainLdsBytes, st, (const ChainModel*)s->d_chain, csp);
    } else {
      if (gauss) hipLaunchKernelGGL((train_chain_group_kernel<PrecBF16, true>), grid, block, kChainLdsBytes, st, (const ChainModel*)s->d_chain, csp);
      else hipLaunchKernelGGL((train_chain_group_kernel<PrecBF16, false>), grid, block, kChainLdsBytes, st, (const ChainModel*)s->d_chain, csp);
    }
    HIPCHK(hipGetLastError());
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
static int launch_dw32_group(const std::vector<v21_trainer*>& trs, const Dw32Model* d_tab, int rows, long long step_index, int max_blocks,
                             hipStream_t st) {
  const int G = (int)trs.size();
  Dw32Step ds{};
  ds.rows = rows; ds.slot = (int)step_index;
  for (int k = 0; k < G; ++k) {
    v21_trainer* t = trs[k];
    t->iter += 1;
    ds.alpha[k] = adam_alpha(t->adam, t->iter);
  }
  hipLaunchKernelGGL(dwadam32_group_kernel, dim3(max_blocks, G), dim3(256), 0, st, d_tab, ds);
  HIPCHK(hipGetLastError());
  for (v21_trainer* t : trs) {
    t->copies_ok = true; t->nt_ok = false;
    invalidate_streams(t->mlp);
    t->mlp->wpad_ok = true;
  }
  return V21_OK;
}
// builds / refreshes the device table of launch_dw32_group; *ok = false: a member's gradient launch takes 64 x 64 tiles
static int refresh_dw32_table(const std::vector<v21_trainer*>& trs, Dw32Model* d_tab, std::vector<Dw32Model>& h_tab, int* max_blocks, bool* ok,
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
static int sweep_step_chain32(v21_sweep* s, const ChainStep& cs, long long step_index, int max_blocks) {
  hipStream_t st = s->ctx->stream;
  const int G = (int)s->tr.size(), rows = cs.rows;
  for (v21_trainer* t : s->tr) CHK(ensure_copies(t, false));
  CHK(chain_attr(V21_PREC_F32));
  // 4 rows per workgroup while every model's row blocks fit the chip in one round (train_chain32s.h)
  const char* er = getenv("V21_C32S_ROWS");
  const int force_rows = er ? atoi(er) : 0;
  const int rpw = force_rows == 4 || force_rows == 8 ? force_rows : ((long long)G * ((rows + 3) / 4) <= 256 ? 4 : 8);
  ChainStep csp = cs;
  csp.ncons = ((rows + rpw - 1) / rpw + 7) / 8 * 8;
  csp.npref = 0;
  const dim3 grid(csp.ncons * G), block(64 * kC32sWaves);
  bool gauss = false;
  for (v21_trainer* t : s->tr) gauss = gauss || t->gl >= 0;
  const ChainModel* tab = (const ChainModel*)s->d_chain;
  if (rpw == 4) {
    if (gauss) hipLaunchKernelGGL((train_chain32s_group_kernel<4, true>), grid, block, kC32sLdsBytes, st, tab, csp, G);
    else hipLaunchKernelGGL((train_chain32s_group_kernel<4, false>), grid, block, kC32sLdsBytes, st, tab, csp, G);
  } else {
    if (gauss) hipLaunchKernelGGL((train_chain32s_group_kernel<8, true>), grid, block, kC32sLdsBytes, st, tab, csp, G);
    else hipLaunchKernelGGL((train_chain32s_group_kernel<8, false>), grid, block, kC32sLdsBytes, st, tab, csp, G);
  }
  HIPCHK(hipGetLastError());
  return launch_dw32_group(s->tr, s->d_dw32, rows, step_index, max_blocks, st);
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
  j->ae = ae; j->em = em; j->latent_layer = latent_layer; j->f32 = f32;
  hipError_t e = hipMalloc((void**)&j->d_tab, 2 * sizeof(ChainModel));
  if (e != hipSuccess) { delete j; return fail(V21_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
  *out = j;
  return V21_OK;
}
extern "C" int v21_joint_destroy(v21_joint* j) {
  if (!j) return V21_OK;
  hipSetDevice(j->ae->ctx->device);
  hipStreamSynchronize(j->ae->ctx->stream);
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
// ---- diagnostics: every CU's LDS filled with a bit pattern (tests/test_poison_gpu.py).  A fresh process finds LDS
// benign; a kernel that relies on what LDS holds before it writes it (an uncleared padding column, a stale mask
// tile) only shows when the previous tenant left NaN / Inf patterns there.  Each workgroup takes the whole 160 KB of
// a CU (so at most one is resident per CU), writes the pattern, and idles for a while so that the dispatcher has to
// spread the grid over all CUs instead of recycling the first ones that finish.
__global__ void __launch_bounds__(256) lds_poison_kernel(unsigned pattern, int words, int spin, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned poison_smem[];
  for (int i = threadIdx.x; i < words; i += 256) poison_smem[i] = pattern;
  __syncthreads();
  for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(32);
  // (read back, so that the stores cannot be dropped as dead)
  if (poison_smem[(threadIdx.x * 37) % words] != pattern) atomicAdd(sink, 1u);
}
extern "C" int v21_debug_poison_lds(v21_ctx* c, uint32_t pattern) {
  CHK(use(c));
  constexpr int kBytes = 160 * 1024;
  static bool attr_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_dev[dev & 63]) {
    HIPCHK(hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes));
    attr_dev[dev & 63] = true;
  }
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, dev));
  unsigned* sink = nullptr;
  HIPCHK(hipMalloc((void**)&sink, 16));
  HIPCHK(hipMemsetAsync(sink, 0, 16, c->stream));
  // two workgroups per CU's worth of grid; each lingers ~2 us after its writes
  hipLaunchKernelGGL(lds_poison_kernel, dim3(2 * prop.multiProcessorCount), dim3(256), kBytes, c->stream, pattern,
                     kBytes / 4, 64, sink);
  HIPCHK(hipGetLastError());
  unsigned bad = 0;
  HIPCHK(hipMemcpyAsync(&bad, sink, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(sink));
  if (bad) return fail(V21_ERR_HIP, "LDS read-back mismatch in %u threads", bad);
  return V21_OK;
}

// diagnostics: the shader clock while other kernels run (train_kernels.h: clock_probe_kernel).  start: one sampling wave
// on a private stream for `duration_ms`, a sample every `period_us`; read: waits for it and reduces the samples.
extern "C" int v21_debug_clock_probe_start(v21_ctx* c, double duration_ms, double period_us) {
  CHK(use(c));
  if (!(duration_ms > 0.0) || duration_ms > 2000.0 || !(period_us >= 1.0)) return fail(V21_ERR_ARG, "clock probe: duration in (0, 2000] ms, period >= 1 us");
  if (!c->probe_stream) HIPCHK(hipStreamCreateWithFlags(&c->probe_stream, hipStreamNonBlocking));
  const int nmax = (int)std::min(65536.0, duration_ms * 1000.0 / period_us + 2.0);
  if (c->probe_cap < nmax) {
    if (c->d_probe) HIPCHK(hipFree(c->d_probe));
    HIPCHK(hipMalloc((void**)&c->d_probe, ((size_t)2 * nmax + 1) * sizeof(unsigned long long)));
    c->probe_cap = nmax;
  }
  HIPCHK(hipMemsetAsync(c->d_probe, 0, ((size_t)2 * c->probe_cap + 1) * sizeof(unsigned long long), c->probe_stream));
  // (s_memrealtime: 100 MHz -> 100 ticks per microsecond)
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, c->probe_stream, c->d_probe, c->probe_cap,
                     (unsigned long long)(period_us * 100.0), (unsigned long long)(duration_ms * 100000.0));
  HIPCHK(hipGetLastError());
  return V21_OK;
}
extern "C" int v21_debug_clock_probe_read(v21_ctx* c, double* ghz_mean, double* ghz_min, double* ghz_max, int* samples) {
  CHK(use(c));
  if (!ghz_mean || !ghz_min || !ghz_max || !samples) return fail(V21_ERR_ARG, "null argument");
  if (!c->probe_stream || !c->d_probe) return fail(V21_ERR_STATE, "no clock probe was started");
  std::vector<unsigned long long> h((size_t)2 * c->probe_cap + 1);
  HIPCHK(hipMemcpyAsync(h.data(), c->d_probe, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->probe_stream));
  HIPCHK(hipStreamSynchronize(c->probe_stream));
  const int n = (int)h[(size_t)2 * c->probe_cap];
  *samples = n;
  *ghz_mean = *ghz_min = *ghz_max = 0.0;
  if (n < 2) return fail(V21_ERR_STATE, "the clock probe took %d samples", n);
  double lo = 1e30, hi = 0.0;
  for (int i = 1; i < n; ++i) {
    const double g = (double)(h[2 * i] - h[2 * i - 2]) / (double)(h[2 * i + 1] - h[2 * i - 1]) * 0.1;  // cycles per 10 ns
    lo = std::min(lo, g); hi = std::max(hi, g);
  }
  *ghz_mean = (double)(h[2 * (n - 1)] - h[0]) / (double)(h[2 * (n - 1) + 1] - h[1]) * 0.1;
  *ghz_min = lo; *ghz_max = hi;
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
extern "C" int v21_trainer_use_graph(v21_trainer* t, int enable) {
  if (!t) return fail(V21_ERR_ARG, "null trainer");
  CHK(use(t->ctx));
  if (enable && (t->ctx->nranks > 1 || t->gl >= 0))
    return fail(V21_ERR_UNSUPPORTED, "captured steps need one rank and a stack without a variational layer");
  if (!enable) { HIPCHK(hipStreamSynchronize(t->ctx->stream)); destroy_graphs(t); }
  t->graph_mode = enable ? 1 : 0;
  return V21_OK;
}
