// api_base.hip -- errors, contexts and streams, device memory, events, the data-parallel communicator (RCCL is
// dlopen'ed: a single-GPU user never needs it) and the diagnostics entry points of include/v21.h.
#include "api_internal.h"

// ---------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------
static thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

extern "C" const char* v21_last_error(void) { return g_err.c_str(); }
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
// ncclDataType_t / ncclRedOp_t values of rccl.h (the library is dlopen'ed, its header is not included)
constexpr int kNcclFloat32 = 7, kNcclSum = 0;
int use(v21_ctx* c) {
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
  if (c->comm_stream) { hipStreamSynchronize(c->comm_stream); hipStreamDestroy(c->comm_stream); }
  for (hipEvent_t e : {c->ev_bucket[0], c->ev_bucket[1], c->ev_comm_done}) if (e) hipEventDestroy(e);
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
  if (c->comm || c->host_comm || c->null_comm) return fail(V21_ERR_STATE, "communicator already initialised");
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
  if (c->comm || c->host_comm || c->null_comm) return fail(V21_ERR_STATE, "communicator already initialised");
  c->host = *ops;
  c->host_comm = true;
  c->nranks = nranks;
  c->rank = rank;
  return V21_OK;
}
extern "C" int v21_comm_init_null(v21_ctx* c, int nranks, int rank) {
  CHK(use(c));
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(V21_ERR_ARG, "bad communicator arguments");
  if (c->comm || c->host_comm || c->null_comm) return fail(V21_ERR_STATE, "communicator already initialised");
  c->null_comm = true;
  c->nranks = nranks;
  c->rank = rank;
  return V21_OK;
}
extern "C" int v21_comm_set_buckets(v21_ctx* c, int buckets) {
  if (!c) return fail(V21_ERR_ARG, "null context");
  if (buckets != 1 && buckets != 2) return fail(V21_ERR_ARG, "buckets must be 1 or 2");
  CHK(use(c));
  if (buckets == 2 && !c->comm_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&c->ev_bucket[0], &c->ev_bucket[1], &c->ev_comm_done}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  c->buckets = buckets;
  return V21_OK;
}
extern "C" int v21_comm_destroy(v21_ctx* c) {
  CHK(use(c));
  if (c->comm_stream) HIPCHK(hipStreamSynchronize(c->comm_stream));
  if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
  c->host_comm = false; c->null_comm = false;
  c->nranks = 1; c->rank = 0; c->sharded = 0; c->buckets = 1;
  return V21_OK;
}
// what the attached communicator itself says (RCCL: ncclCommCount / ncclCommUserRank; host transport: what the host
// passed): transport 0 = none, 1 = RCCL inside the library, 2 = host-staged callbacks, 3 = none on purpose (v21_comm_init_null)
extern "C" int v21_comm_info(v21_ctx* c, int* nranks, int* rank, int* transport) {
  if (!c || !nranks || !rank || !transport) return fail(V21_ERR_ARG, "null argument");
  *nranks = c->nranks; *rank = c->rank;
  *transport = c->comm ? 1 : (c->host_comm ? 2 : (c->null_comm ? 3 : 0));
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
static int host_collective(v21_ctx* c, float* d_buf, size_t n, F&& call, const char* what, hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  CHK(host_stage(c, n));
  HIPCHK(hipMemcpyAsync(c->h_stage, d_buf, n * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int r = call(c->h_stage);
  if (r != 0) return fail(V21_ERR_COMM, "host %s callback returned %d", what, r);
  HIPCHK(hipMemcpyAsync(d_buf, c->h_stage, n * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return V21_OK;
}
int comm_allreduce_on(v21_ctx* c, float* d_buf, size_t n, hipStream_t st) {
  if (c->nranks <= 1 || c->null_comm || n == 0) return V21_OK;  // single rank / no transport: identity
  if (c->host_comm)
    return host_collective(c, d_buf, n, [&](float* h) { return c->host.allreduce_sum_f32(c->host.user, h, n); }, "all-reduce", st);
  int r = g_rccl.AllReduce(d_buf, d_buf, n, kNcclFloat32, kNcclSum, c->comm, st);
  if (r != 0) return fail(V21_ERR_COMM, "ncclAllReduce: %s", rccl_err(r));
  return V21_OK;
}
extern "C" int v21_comm_allreduce_f32(v21_ctx* c, float* d_buf, size_t n) {
  CHK(use(c));
  return comm_allreduce_on(c, d_buf, n, c->stream);
}
// in place over nranks * n_per floats: rank r ends up with the sums of elements [r n_per, (r+1) n_per) there
extern "C" int v21_comm_reduce_scatter_f32(v21_ctx* c, float* d_buf, size_t n_per) {
  CHK(use(c));
  if (c->nranks <= 1 || c->null_comm) return V21_OK;
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
  if (c->nranks <= 1 || c->null_comm) return V21_OK;
  if (c->host_comm)
    return host_collective(c, d_buf, n_per * c->nranks, [&](float* h) { return c->host.allgather_f32(c->host.user, h, n_per); },
                           "all-gather");
  int r = g_rccl.AllGather(d_buf + (size_t)c->rank * n_per, d_buf, n_per, kNcclFloat32, c->comm, c->stream);
  if (r != 0) return fail(V21_ERR_COMM, "ncclAllGather: %s", rccl_err(r));
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


