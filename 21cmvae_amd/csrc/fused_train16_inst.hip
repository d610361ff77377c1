// fused_train16_inst.hip -- one translation unit per (stack, precision) of the 16-rows-per-wave fused training kernel (fused_train16.h);
// compiled with -DV21_ARCH=T1 -DV21_PREC=F16t16 etc. (see Makefile).
#include "fused_train16.h"
#include "archs.h"

#define V21_CAT3(a, b, c) a##b##c
#define V21_SYMNAME(a, p) V21_CAT3(launch_fused_train16_, a, _##p)
#define V21_XCAT(a, b) a##b
#define V21_ARCH_T(a) V21_XCAT(Arch, a)
#define V21_PREC_T(p) V21_XCAT(Prec, p)
#define V21_EXPAND_SYM(a, p) V21_SYMNAME(a, p)

namespace v21 {

hipError_t V21_EXPAND_SYM(V21_ARCH, V21_PREC)(const ChainArgs& a, hipStream_t st) {
  using A = V21_ARCH_T(V21_ARCH);
  using P = V21_PREC_T(V21_PREC);
  auto kern = fused_train16<A, P>;
  static bool attr_done_dev[64] = {};  // the attribute belongs to (function, device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool& attr_done = attr_done_dev[dev & 63];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, fused_train16_lds<P>());
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int nrb = (a.rows + kTrain16RowsPerWg - 1) / kTrain16RowsPerWg;
  if (nrb <= 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3((unsigned)((nrb + 7) / 8 * 8)), dim3(64 * P::WAVES), fused_train16_lds<P>(), st, a);
  return hipGetLastError();
}

}  // namespace v21
