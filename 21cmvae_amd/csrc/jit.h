// jit.h -- run-time instantiation of the fused forward kernel (fused_fwd.h) for stacks the library was not built for.
//
// `_gen_model` accepts any `hidden_dims` (/root/reference/VeryAccurateEmulator/emulator.py:12-48; the notebooks build
// [64, 128] and [32, 128, 256] models), and fused_fwd<Arch, Prec> is straight-line code per (stack, precision): four
// stacks are compiled into libv21.so (archs.h), every other one used to take the table-driven chain kernel in FORWARD
// mode -- 2.8-2.9x slower on the headline stack, 21 % of the HBM write rate on the sample notebook's stack (r3).  Here
// the SAME kernel template is instantiated for the stack at hand by hiprtc (libhiprtc.so, dlopen'ed; the sources are
// embedded in the library at build time), in a background thread; until the code object is there the chain kernel
// serves the calls.  Code objects are cached on disk ($V21_KERNEL_CACHE, default ~/.cache/21cmvae_amd/kernels; the
// directory `kernel_cache/` next to libv21.so is searched first: `__graft_entry__.build()` prebuilds the notebook
// stacks there -- hiprtc needs no GPU to compile).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "fused_fwd.h"
#include "chain_types.h"

namespace v21 {
struct JitKernel;  // one (stack, precision); lives until the library is unloaded
enum { JIT_COMPILING = 0, JIT_READY = 1, JIT_FAILED = -1 };
// r5: `prec | kJitTrain16` asks for the fused TRAINING kernel of the stack (fused_train16<Arch, Prec>, f16 / bf16) instead
// of the forward kernel: same registry, same caches, same compiler process
constexpr int kJitTrain16 = 16;
bool jit_train_eligible(int L, const int* dims, const int* act, std::string* why);
hipError_t jit_launch_train(JitKernel* k, int device, const ChainArgs& a, hipStream_t st);

// false (with a reason) for stacks fused_fwd cannot express: a non-linear output layer, a variational head, > 16 layers
bool jit_eligible(int L, const int* dims, const int* act, std::string* why);
// The kernel of (dims, act, precision); the first request starts the compilation (or finds it in a cache directory).
// nullptr: not eligible, or run-time compilation is switched off (V21_JIT=0) and nothing is cached.
JitKernel* jit_request(int L, const int* dims, const int* act, int prec);
int jit_state(JitKernel* k, std::string* why = nullptr);
int jit_wait(JitKernel* k, int timeout_ms);  // < 0: until the compilation has ended
// Launch on the current device (the code object is loaded per device on first use).  hipErrorNotReady while compiling;
// a kernel that needs scratch memory (a stack too wide for the register budget) is marked failed at load time.
hipError_t jit_launch(JitKernel* k, int device, const FusedArgs& a, hipStream_t st);
// compile into `dir` without running anything (no GPU needed): returns 0, or -1 with a reason
int jit_prebuild(int L, const int* dims, const int* act, int prec, const char* dir, std::string* why);
}  // namespace v21
