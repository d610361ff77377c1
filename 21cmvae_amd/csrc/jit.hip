// jit.hip -- run-time instantiation of fused_fwd<Arch, Prec> -- and, r5, of the fused training kernel
// fused_train16<Arch, Prec> -- through hiprtc (see jit.h).  A kernel's KIND rides in bit 4 of the precision argument of
// the internal interface (kJitTrain16 | V21_PREC_F16 ...), so that the registry, the cache and the compiler process
// (v21_jitc -> v21_jit_prebuild) serve both with the same code.
#include "jit.h"
#include "fused_train16.h"

#include <dlfcn.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace v21 {
namespace {

// include/v21.h + par_transform.h + fused_fwd.h, own #include lines and `#pragma once` removed (Makefile: fused_src.inc)
const unsigned char kFusedSrc[] = {
#include "fused_src.inc"
    0};
// ... + chain_types.h + fused_train.h + fused_train16.h (Makefile: fused_train_src.inc)
const unsigned char kTrainSrc[] = {
#include "fused_train_src.inc"
    0};
inline bool is_train(int prec) { return (prec & kJitTrain16) != 0; }
inline int base_prec(int prec) { return prec & 15; }

// ---- hiprtc, loaded at run time (a user who never predicts on a custom stack needs no libhiprtc)
typedef void* rtc_prog;
struct RtcApi {
  void* lib = nullptr;
  int (*CreateProgram)(rtc_prog*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*DestroyProgram)(rtc_prog*) = nullptr;
  int (*AddNameExpression)(rtc_prog, const char*) = nullptr;
  int (*CompileProgram)(rtc_prog, int, const char**) = nullptr;
  int (*GetProgramLogSize)(rtc_prog, size_t*) = nullptr;
  int (*GetProgramLog)(rtc_prog, char*) = nullptr;
  int (*GetLoweredName)(rtc_prog, const char*, const char**) = nullptr;
  int (*GetCodeSize)(rtc_prog, size_t*) = nullptr;
  int (*GetCode)(rtc_prog, char*) = nullptr;
  int (*Version)(int*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RtcApi g_rtc;
std::mutex g_rtc_mu;
bool load_rtc(std::string* why) {
  std::lock_guard<std::mutex> lk(g_rtc_mu);
  if (g_rtc.lib) return true;
  const char* cands[] = {getenv("V21_HIPRTC_LIB"), "libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
  void* lib = nullptr;
  for (const char* c : cands) {
    if (!c || !*c) continue;
    lib = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) { if (why) *why = std::string("cannot dlopen libhiprtc: ") + dlerror(); return false; }
  RtcApi a;
  a.lib = lib;
#define V21_RTC_SYM(field, name) *(void**)(&a.field) = dlsym(lib, name)
  V21_RTC_SYM(CreateProgram, "hiprtcCreateProgram");
  V21_RTC_SYM(DestroyProgram, "hiprtcDestroyProgram");
  V21_RTC_SYM(AddNameExpression, "hiprtcAddNameExpression");
  V21_RTC_SYM(CompileProgram, "hiprtcCompileProgram");
  V21_RTC_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
  V21_RTC_SYM(GetProgramLog, "hiprtcGetProgramLog");
  V21_RTC_SYM(GetLoweredName, "hiprtcGetLoweredName");
  V21_RTC_SYM(GetCodeSize, "hiprtcGetCodeSize");
  V21_RTC_SYM(GetCode, "hiprtcGetCode");
  V21_RTC_SYM(Version, "hiprtcVersion");
  V21_RTC_SYM(GetErrorString, "hiprtcGetErrorString");
#undef V21_RTC_SYM
  if (!a.CreateProgram || !a.DestroyProgram || !a.AddNameExpression || !a.CompileProgram || !a.GetProgramLogSize ||
      !a.GetProgramLog || !a.GetLoweredName || !a.GetCodeSize || !a.GetCode) {
    if (why) *why = "libhiprtc lacks a required symbol";
    return false;
  }
  g_rtc = a;
  return true;
}

// V21_JIT_WIDE=1 (diagnostics; read when a kernel is first requested): the 16-bit kernels in their ONE-workgroup-per-CU
// form -- one wave per SIMD with TWO column tiles (64 signals), every weight fragment read from LDS feeds two MFMAs,
// up to 512 registers -- instead of "x2sp" (two workgroups per CU, one column tile per wave, 256 registers).  Half the
// LDS bytes per MFMA; r1 measured it slower (62-65 against 52 us on the headline stack) and r4 measured it again
// through this switch.
// diagnostics (scripts/diag/k1_wide_probe.py): run-time 16-bit kernels in the one-workgroup-per-CU, two-column-tiles-per-wave
// form.  Measured slower in r1 and again in r4 (61.9 against 52 us); the switch exists only in a diagnostic build
// (make EXTRA=-DV21_DIAG_JIT_WIDE ...): the product library has one form.
#ifdef V21_DIAG_JIT_WIDE
bool jit_wide() { const char* e = getenv("V21_JIT_WIDE"); return e && e[0] == '1'; }
#else
bool jit_wide() { return false; }
#endif
const char* prec_type(int prec, bool wide) {
  if (is_train(prec)) return base_prec(prec) == 1 ? "PrecF16t16" : "PrecBF16t16";
  return prec == 0 ? "PrecF32" : (prec == 1 ? (wide ? "PrecF16" : "PrecF16x2sp") : (wide ? "PrecBF16" : "PrecBF16x2sp"));
}
// LDS read-ahead (fragments) of a run-time training kernel: the template's default (V21_TRAIN16_DEPTH = 4) unless the loss
// layer has several tiles and too few k-steps per tile for it (fused_train16.h: the target double buffer needs
// ks_of(L - 1) > depth) -- the sample notebook's 7 -> [64, 128] -> 451 has four: depth 3.  Passed to hiprtc as
// -DV21_TRAIN16_DEPTH and part of the kernel's name in the cache.
int train_depth(int L, const int* dims) {
  if (dims[L] <= 16) return V21_TRAIN16_DEPTH;
  const int ks = (dims[L - 1] + 31) / 32;
  return ks - 1 < V21_TRAIN16_DEPTH ? ks - 1 : V21_TRAIN16_DEPTH;
}
struct Launch { int rows_per_wg, threads, lds; };
template <class P> constexpr Launch launch_of() { return Launch{P::WAVES * P::CT * 32, 64 * P::WAVES, fused_lds_alloc<P>()}; }
Launch launch_geometry(int prec, bool wide) {
  if (is_train(prec))
    return base_prec(prec) == 1 ? Launch{kTrain16RowsPerWg, 64 * PrecF16t16::WAVES, fused_train16_lds<PrecF16t16>()}
                                : Launch{kTrain16RowsPerWg, 64 * PrecBF16t16::WAVES, fused_train16_lds<PrecBF16t16>()};
  if (prec == 0) return launch_of<PrecF32>();
  if (prec == 1) return wide ? launch_of<PrecF16>() : launch_of<PrecF16x2sp>();
  return wide ? launch_of<PrecBF16>() : launch_of<PrecBF16x2sp>();
}

unsigned long long fnv1a(const void* p, size_t n, unsigned long long h = 1469598103934665603ull) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

const char* kOptions[] = {"--offload-arch=gfx950", "-std=c++20", "-O3", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1"};
constexpr int kNumOptions = sizeof(kOptions) / sizeof(kOptions[0]);

std::string lib_dir() {
  Dl_info info{};
  if (dladdr((const void*)&fnv1a, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
  }
  return ".";
}
void mkdirs(const std::string& d, mode_t leaf = 0755) {
  for (size_t i = 1; i <= d.size(); ++i)
    if (i == d.size() || d[i] == '/') mkdir(d.substr(0, i).c_str(), i == d.size() ? leaf : 0755);
}
// A directory whose code objects this process will LOAD AND RUN must be the user's own and closed to others: a
// world-writable or foreign directory (somebody else created /tmp/21cmvae_amd_kernels_<uid> first) would let another
// local user plant GPU code.  lstat, not stat: a symbolic link in that place is refused as well.
bool private_dir(const std::string& d) {
  struct stat sb;
  if (lstat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) return false;
  return sb.st_uid == getuid() && (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0 && access(d.c_str(), W_OK | X_OK) == 0;
}
// A directory this process only READS code objects from (kernel_cache/ next to the library, written by the build): it and
// the file must belong to this user or to root and be closed to group and others -- the objects are loaded onto the GPU
// and run (only the per-user cache directory was checked until r4).
bool trusted_owner(const struct stat& sb) {
  return (sb.st_uid == getuid() || sb.st_uid == 0) && (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
bool trusted_dir(const std::string& d) {
  struct stat sb;
  return lstat(d.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode) && trusted_owner(sb);
}
// $V21_KERNEL_CACHE (it must pass the same test as the default: the user's own, closed to others), else ~/.cache/21cmvae_amd/kernels, else (no home directory, or one
// that cannot be written: a container running as another user) a per-user directory under /tmp created 0700; "" when
// neither is the user's own private directory -- kernels are then compiled per process and nothing is cached on disk
std::string user_cache_dir() {
  if (const char* e = getenv("V21_KERNEL_CACHE")) {
    mkdirs(e, 0700);
    return private_dir(e) ? std::string(e) : std::string();
  }
  const char* home = getenv("HOME");
  if (home && *home) {
    const std::string d = std::string(home) + "/.cache/21cmvae_amd/kernels";
    mkdirs(d, 0700);  // (the leaf holds code objects this process loads and runs: the user's alone)
    if (private_dir(d)) return d;
  }
  const std::string t = "/tmp/21cmvae_amd_kernels_" + std::to_string((long)getuid());
  mkdir(t.c_str(), 0700);
  return private_dir(t) ? t : std::string();
}

constexpr char kMagic[8] = {'V', '2', '1', 'K', 'O', 'B', 'J', '1'};
bool read_cache(const std::string& path, std::string& sym, std::vector<char>& code) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  struct stat sb;  // (the open file itself, not the name: nobody can swap it between the check and the read)
  if (fstat(fileno(f), &sb) != 0 || !S_ISREG(sb.st_mode) || !trusted_owner(sb)) { fclose(f); return false; }
  char magic[8];
  unsigned n = 0;
  bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kMagic, 8) == 0 && fread(&n, 4, 1, f) == 1 && n > 0 && n < 4096;
  if (ok) {
    sym.resize(n);
    ok = fread(sym.data(), 1, n, f) == n;
  }
  if (ok) {
    const long at = ftell(f);
    fseek(f, 0, SEEK_END);
    const long end = ftell(f);
    fseek(f, at, SEEK_SET);
    ok = end > at;
    if (ok) {
      code.resize((size_t)(end - at));
      ok = fread(code.data(), 1, code.size(), f) == code.size();
    }
  }
  fclose(f);
  return ok;
}
bool write_cache(const std::string& dir, const std::string& path, const std::string& sym, const std::vector<char>& code) {
  mkdirs(dir);
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const unsigned n = (unsigned)sym.size();
  bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&n, 4, 1, f) == 1 && fwrite(sym.data(), 1, n, f) == n &&
            fwrite(code.data(), 1, code.size(), f) == code.size();
  ok = fclose(f) == 0 && ok;
  if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;  // (atomic: another process may be writing the same kernel)
  if (!ok) unlink(tmp.c_str());
  return ok;
}

}  // namespace

struct JitKernel {
  int L = 0, prec = 0;
  bool wide = false;  // V21_JIT_WIDE at request time
  int dims[17] = {}, act[16] = {};
  std::string spec, file;  // "7x64x128x451_a110_PrecF16x2sp", "<spec>_<hash>.v21k"
  std::atomic<int> state{JIT_COMPILING};
  std::string why;         // (written before `state` leaves JIT_COMPILING)
  std::string sym;
  std::vector<char> code;
  std::mutex mu;
  std::condition_variable cv;
  struct Loaded { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; };
  std::map<int, Loaded> loaded;  // per device
  std::thread th;               // waits for the compiler process (v21_jitc)
  std::atomic<int> child{0};    // its pid while it runs (written under `mu`, together with the spawn)
  bool cancel = false;          // the process is ending (Registry::~Registry, under `mu`): do not start a compiler any more
};

namespace {
struct Registry {
  std::mutex mu;
  std::map<std::string, JitKernel*> all;
  ~Registry() {  // the process ends: a compiler process still running is of no use to anybody
    for (auto& kv : all) {
      // under the kernel's mutex: its thread spawns and publishes the pid under the same mutex, so either the child is
      // seen here and killed, or it is never started (a child spawned but not yet published made exit() wait
      // for the whole compilation)
      std::lock_guard<std::mutex> lk(kv.second->mu);
      kv.second->cancel = true;
      const int pid = kv.second->child.load();
      if (pid > 0) kill(pid, SIGKILL);
    }
    for (auto& kv : all)
      if (kv.second->th.joinable()) kv.second->th.join();
  }
};
Registry g_reg;

std::string make_spec(int L, const int* dims, const int* act, int prec, bool wide) {
  std::string s = is_train(prec) ? "train16_" : "";
  if (is_train(prec) && train_depth(L, dims) != V21_TRAIN16_DEPTH) s += "d" + std::to_string(train_depth(L, dims)) + "_";
  for (int l = 0; l <= L; ++l) s += (l ? "x" : "") + std::to_string(dims[l]);
  s += "_a";
  for (int l = 0; l < L; ++l) s += act[l] ? "1" : "0";
  s += "_";
  s += prec_type(prec, wide);
  return s;
}
std::string make_source(const JitKernel& k) {
  std::string src((const char*)(is_train(k.prec) ? kTrainSrc : kFusedSrc));
  src += "\nnamespace v21 {\nstruct ArchRT {\n  static constexpr int L = " + std::to_string(k.L) + ";\n  static constexpr int dims[" +
         std::to_string(k.L + 1) + "] = {";
  for (int l = 0; l <= k.L; ++l) src += (l ? ", " : "") + std::to_string(k.dims[l]);
  src += "};\n  static constexpr int act[" + std::to_string(k.L) + "] = {";
  for (int l = 0; l < k.L; ++l) src += (l ? ", " : "") + std::to_string(k.act[l] ? 1 : 0);
  src += "};\n};\n}  // namespace v21\n";
  return src;
}
// What a cached code object depends on besides the sources, the options and the stack: the format of the cache file and of
// the host-side launch contract (kCacheFormat: bump it when FusedArgs, the launch geometry or the file layout change), and
// the toolchain that compiled it -- the version of the HIP runtime this library runs with (hiprtc ships with it; the
// compiler library itself is only ever loaded in the child process: csrc/jitc_main.cpp says why).  A stale object from
// before a ROCm upgrade is then never loaded.
constexpr int kCacheFormat = 2;
int toolchain_version() {
  // (the compiler process is told by its parent -- V21_JIT_TOOLCHAIN in the environment it is spawned with: asking the HIP
  //  runtime initialises it, and a process that has the GPU open counts against the box's limit of six -- a test run with
  //  ten compilations in flight was killed for it, r5)
  if (const char* e = getenv("V21_JIT_TOOLCHAIN")) return atoi(e);
  static const int cached = [] {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); v = 0; }
    return v;
  }();
  return cached;
}
std::string file_name(const JitKernel& k) {
  unsigned long long h = is_train(k.prec) ? fnv1a(kTrainSrc, sizeof(kTrainSrc)) : fnv1a(kFusedSrc, sizeof(kFusedSrc));
  for (int i = 0; i < kNumOptions; ++i) h = fnv1a(kOptions[i], strlen(kOptions[i]), h);
  h = fnv1a(k.spec.data(), k.spec.size(), h);
  const int meta[3] = {kCacheFormat, toolchain_version(), HIP_VERSION};
  h = fnv1a(meta, sizeof meta, h);
  char buf[32];
  snprintf(buf, sizeof buf, "%016llx", h);
  return "fused_" + k.spec + "_" + buf + ".v21k";
}

int compile(const JitKernel& k, std::string& sym, std::vector<char>& code, std::string& why) {
  if (!load_rtc(&why)) return -1;
  const std::string src = make_source(k);
  const std::string expr = std::string(is_train(k.prec) ? "v21::fused_train16<v21::ArchRT, v21::" : "v21::fused_fwd<v21::ArchRT, v21::") +
                           prec_type(k.prec, k.wide) + ">";
  rtc_prog prog = nullptr;
  int r = g_rtc.CreateProgram(&prog, src.c_str(), "v21_fused_rt.hip", 0, nullptr, nullptr);
  if (r != 0) { why = "hiprtcCreateProgram failed"; return -1; }
  r = g_rtc.AddNameExpression(prog, expr.c_str());
  std::vector<const char*> opts(kOptions, kOptions + kNumOptions);
  std::string depth_opt;
  if (is_train(k.prec) && train_depth(k.L, k.dims) != V21_TRAIN16_DEPTH) {
    depth_opt = "-DV21_TRAIN16_DEPTH=" + std::to_string(train_depth(k.L, k.dims));
    opts.push_back(depth_opt.c_str());
  }
  if (r == 0) r = g_rtc.CompileProgram(prog, (int)opts.size(), opts.data());
  if (r != 0) {
    size_t n = 0;
    g_rtc.GetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) g_rtc.GetProgramLog(prog, log.data());
    if (log.size() > 1500) log.resize(1500);
    why = std::string("hiprtc: ") + (g_rtc.GetErrorString ? g_rtc.GetErrorString(r) : "?") + ": " + log;
    g_rtc.DestroyProgram(&prog);
    return -1;
  }
  const char* lowered = nullptr;
  size_t n = 0;
  if (g_rtc.GetLoweredName(prog, expr.c_str(), &lowered) != 0 || !lowered || g_rtc.GetCodeSize(prog, &n) != 0 || n == 0) {
    why = "hiprtc returned no code object";
    g_rtc.DestroyProgram(&prog);
    return -1;
  }
  sym = lowered;
  code.resize(n);
  r = g_rtc.GetCode(prog, code.data());
  g_rtc.DestroyProgram(&prog);
  if (r != 0) { why = "hiprtcGetCode failed"; return -1; }
  return 0;
}

void finish(JitKernel* k, int state, const std::string& why) {
  {
    std::lock_guard<std::mutex> lk(k->mu);
    k->why = why;
    k->state.store(state, std::memory_order_release);
  }
  k->cv.notify_all();
}
void fill(JitKernel* k, int L, const int* dims, const int* act, int prec) {
  k->L = L; k->prec = prec; k->wide = jit_wide();
  for (int l = 0; l <= L; ++l) k->dims[l] = dims[l];
  for (int l = 0; l < L; ++l) k->act[l] = act[l];
  k->spec = make_spec(L, dims, act, prec, k->wide);
  k->file = file_name(*k);
}
}  // namespace

bool jit_eligible(int L, const int* dims, const int* act, std::string* why) {
  auto no = [&](const char* w) { if (why) *why = w; return false; };
  if (L < 1 || L > 16) return no("layer count not in [1, 16]");
  if (act[L - 1] != 0) return no("the fused kernel's output layer is linear");
  for (int l = 0; l < L; ++l)
    if (act[l] != 0 && act[l] != 1) return no("the fused kernel has no variational head");
  for (int l = 0; l <= L; ++l)
    if (dims[l] < 1 || dims[l] > 1024) return no("layer width not in [1, 1024]");
  // fused_fwd.h, Geo::spread_limit: an output layer of several tiles needs more k-steps per tile than the read-ahead depth
  // (f32: 8 features per k-step, depth 2)
  if (dims[L] > 32 && dims[L - 1] <= 32) return no("the output layer's input is too narrow for the fused kernel's aux double buffer");
  return true;
}

// The fused training kernel on 16 rows per wave (fused_train16.h) for a stack outside archs.h.  What the template's own
// static_asserts and its register budget allow: a linear output layer, ReLU / linear hidden layers, no variational head,
// 2 .. 8 layers up to 512 wide (the chain kernel's limit: the trainer falls back to it), at most 44 ReLU mask tile pairs,
// a loss layer of one 16-wide tile or fed by more than 32 x (read-ahead) features.  A kernel that spills anyway is
// rejected when it is loaded (scratch memory), as the forward kernels are.
bool jit_train_eligible(int L, const int* dims, const int* act, std::string* why) {
  auto no = [&](const char* w) { if (why) *why = w; return false; };
  if (L < 2 || L > 8) return no("layer count not in [2, 8]");
  if (act[L - 1] != 0) return no("the loss is taken on a linear output layer");
  int pairs = 0;
  for (int l = 0; l < L; ++l) {
    if (act[l] != 0 && act[l] != 1) return no("the fused training kernel has no variational head");
    if (l + 1 < L && act[l]) pairs += ((dims[l + 1] + 15) / 16 + 1) / 2;
  }
  for (int l = 0; l <= L; ++l)
    if (dims[l] < 1 || dims[l] > 512) return no("layer width not in [1, 512]");
  if (pairs > kTrain16MaskPairs) return no("too many ReLU mask tiles for the kernel's LDS area");
  if (train_depth(L, dims) < 1) return no("the loss layer's input is too narrow for the target double buffer");
  return true;
}

JitKernel* jit_request(int L, const int* dims, const int* act, int prec) {
  if (is_train(prec)) {
    if ((base_prec(prec) != 1 && base_prec(prec) != 2) || !jit_train_eligible(L, dims, act, nullptr)) return nullptr;
  } else if (prec < 0 || prec > 2 || !jit_eligible(L, dims, act, nullptr)) return nullptr;
  const std::string spec = make_spec(L, dims, act, prec, jit_wide());
  std::lock_guard<std::mutex> lk(g_reg.mu);
  auto it = g_reg.all.find(spec);
  if (it != g_reg.all.end()) return it->second;
  JitKernel* k = new JitKernel();
  fill(k, L, dims, act, prec);
  const std::string built = lib_dir() + "/kernel_cache";
  const std::string dirs[2] = {trusted_dir(built) ? built : std::string(), user_cache_dir()};
  for (const std::string& d : dirs)
    if (!d.empty() && read_cache(d + "/" + k->file, k->sym, k->code)) {
      k->state.store(JIT_READY);
      g_reg.all[spec] = k;
      return k;
    }
  const char* e = getenv("V21_JIT");
  if (e && e[0] == '0') { delete k; return nullptr; }  // (cached kernels are still used; nothing is compiled)
  g_reg.all[spec] = k;
  // The compilation runs in a CHILD PROCESS (csrc/jitc_main.cpp says why): `v21_jitc` next to this library loads the
  // library, compiles into the user's cache directory and exits; the thread below only waits for it.
  k->th = std::thread([k] {
    const std::string dir = user_cache_dir(), helper = lib_dir() + "/v21_jitc";
    if (dir.empty()) {
      finish(k, JIT_FAILED, "no private directory for compiled kernels (neither ~/.cache/21cmvae_amd/kernels nor /tmp/21cmvae_amd_kernels_<uid> "
                            "is this user's own, closed to others): set V21_KERNEL_CACHE");
      return;
    }
    mkdirs(dir);
    const std::string errfile = dir + "/" + k->file + ".err";
    std::vector<std::string> av = {helper, dir, errfile, std::to_string(k->prec), std::to_string(k->L)};
    for (int l = 0; l <= k->L; ++l) av.push_back(std::to_string(k->dims[l]));
    for (int l = 0; l < k->L; ++l) av.push_back(std::to_string(k->act[l]));
    std::vector<char*> argv;
    for (auto& a : av) argv.push_back(a.data());
    argv.push_back(nullptr);
    // The child gets a SCRUBBED environment: a parent under rocprofv3 or any LD_PRELOAD tool would otherwise hand the pure
    // compile job its tool libraries (the profiler then initialises the GPU in the compiler process and writes its own
    // output into the profile directory: ADVICE r4).  Everything else -- PATH, HOME, ROCM_PATH, V21_* -- is kept.
    std::vector<std::string> envs;
    for (char** e = environ; e && *e; ++e) {
      const char* drop[] = {"V21_JIT_TOOLCHAIN=", "LD_PRELOAD=", "HSA_TOOLS_LIB=", "HSA_TOOLS_REPORT_LOAD_FAILURE=", "ROCP_", "ROCPROFILER_", "ROCPROF_", "ROCTRACER_",
                            "ROCTX_", "HIP_TOOLS_LIB=", "OMPT_TOOL_LIBRARIES="};
      bool keep = true;
      for (const char* d : drop) keep = keep && strncmp(*e, d, strlen(d)) != 0;
      if (keep) envs.push_back(*e);
    }
    envs.push_back("V21_JIT_TOOLCHAIN=" + std::to_string(toolchain_version()));
    std::vector<char*> envp;
    for (auto& e : envs) envp.push_back(e.data());
    envp.push_back(nullptr);
    pid_t pid = 0;
    int rc = 0;
    {
      std::lock_guard<std::mutex> lk(k->mu);
      if (k->cancel) rc = ECANCELED;
      else rc = posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), envp.data());
      if (rc == 0) k->child.store((int)pid);
    }
    if (rc != 0) { finish(k, JIT_FAILED, "cannot start the compiler process " + helper + ": " + strerror(rc)); return; }
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
    k->child.store(0);
    std::string sym;
    std::vector<char> code;
    if (WIFEXITED(status) && WEXITSTATUS(status) == 0 && read_cache(dir + "/" + k->file, sym, code)) {
      k->sym = sym;
      k->code = std::move(code);
      finish(k, JIT_READY, "");
      return;
    }
    std::string why = "the compiler process failed";
    if (FILE* f = fopen(errfile.c_str(), "r")) {
      char buf[1600];
      const size_t n = fread(buf, 1, sizeof buf - 1, f);
      buf[n] = 0;
      fclose(f);
      unlink(errfile.c_str());
      if (n) why = buf;
    } else if (WIFSIGNALED(status)) {
      why += " (signal " + std::to_string(WTERMSIG(status)) + ")";
    }
    finish(k, JIT_FAILED, why);
  });
  return k;
}

int jit_state(JitKernel* k, std::string* why) {
  if (!k) { if (why) *why = "no kernel"; return JIT_FAILED; }
  const int s = k->state.load(std::memory_order_acquire);
  if (why && s == JIT_FAILED) { std::lock_guard<std::mutex> lk(k->mu); *why = k->why; }
  return s;
}

int jit_wait(JitKernel* k, int timeout_ms) {
  if (!k) return JIT_FAILED;
  std::unique_lock<std::mutex> lk(k->mu);
  auto done = [&] { return k->state.load(std::memory_order_acquire) != JIT_COMPILING; };
  if (timeout_ms < 0) k->cv.wait(lk, done);
  else k->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), done);
  return k->state.load(std::memory_order_acquire);
}

static hipError_t jit_launch_any(JitKernel* k, int device, void* arg_block, long long rows, hipStream_t st);
hipError_t jit_launch(JitKernel* k, int device, const FusedArgs& a, hipStream_t st) {
  FusedArgs copy = a;
  return jit_launch_any(k, device, &copy, a.n_rows, st);
}
hipError_t jit_launch_train(JitKernel* k, int device, const ChainArgs& a, hipStream_t st) {
  ChainArgs copy = a;
  return jit_launch_any(k, device, &copy, a.rows, st);
}
static hipError_t jit_launch_any(JitKernel* k, int device, void* arg_block, long long rows, hipStream_t st) {
  if (!k) return hipErrorInvalidValue;
  const int s = k->state.load(std::memory_order_acquire);
  if (s == JIT_COMPILING) return hipErrorNotReady;
  if (s != JIT_READY) return hipErrorInvalidValue;
  const Launch g = launch_geometry(k->prec, k->wide);
  JitKernel::Loaded ld;
  {
    std::lock_guard<std::mutex> lk(k->mu);
    auto it = k->loaded.find(device);
    if (it == k->loaded.end()) {
      JitKernel::Loaded n;
      hipError_t e = hipModuleLoadData(&n.mod, k->code.data());
      if (e == hipSuccess) e = hipModuleGetFunction(&n.fn, n.mod, k->sym.c_str());
      int scratch = 0;
      if (e == hipSuccess) e = hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, n.fn);
      if (e == hipSuccess && scratch > 64) {  // (a handful of spilled words outside the inner loop is tolerated)
        // registers spilled to scratch memory: the activations of this stack do not fit a wave's registers -- the
        // table-driven chain kernel is the better route for it
        k->why = "the stack is too wide for the fused kernel's register budget (" + std::to_string(scratch) + " bytes of scratch per lane)";
        k->state.store(JIT_FAILED, std::memory_order_release);
        (void)hipModuleUnload(n.mod);
        return hipErrorInvalidValue;
      }
      if (e != hipSuccess) {
        k->why = std::string("loading the code object: ") + hipGetErrorString(e);
        k->state.store(JIT_FAILED, std::memory_order_release);
        (void)hipGetLastError();
        return e;
      }
      // more than 64 KB of dynamic LDS needs the attribute (as the compiled-in kernels set it in fused_inst.hip)
      (void)hipFuncSetAttribute((const void*)n.fn, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds);
      (void)hipGetLastError();
      it = k->loaded.emplace(device, n).first;
    }
    ld = it->second;
  }
  long long nwg = (rows + g.rows_per_wg - 1) / g.rows_per_wg;
  if (nwg <= 0) return hipSuccess;
  if (is_train(k->prec)) nwg = (nwg + 7) / 8 * 8;  // (whole rounds of the 8 XCDs: the kernel maps block numbers XCD-major and the surplus leaves)
  void* args[] = {arg_block};
  return hipModuleLaunchKernel(ld.fn, (unsigned)nwg, 1, 1, (unsigned)g.threads, 1, 1, (unsigned)g.lds, st, args, nullptr);
}

int jit_prebuild(int L, const int* dims, const int* act, int prec, const char* dir, std::string* why) {
  std::string w;
  const bool ok = is_train(prec) ? ((base_prec(prec) == 1 || base_prec(prec) == 2) && jit_train_eligible(L, dims, act, &w))
                                 : (prec >= 0 && prec <= 2 && jit_eligible(L, dims, act, &w));
  if (!ok) { if (why) *why = w.empty() ? "bad precision" : w; return -1; }
  JitKernel k;
  fill(&k, L, dims, act, prec);
  const std::string d = dir && *dir ? dir : lib_dir() + "/kernel_cache";
  std::string sym;
  std::vector<char> code;
  if (read_cache(d + "/" + k.file, sym, code)) return 0;  // already there, same sources and options
  if (compile(k, sym, code, w) != 0) { if (why) *why = w; return -1; }
  if (!write_cache(d, d + "/" + k.file, sym, code)) { if (why) *why = "cannot write " + d + "/" + k.file; return -1; }
  return 0;
}

}  // namespace v21
