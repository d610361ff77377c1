// par_transform.h -- the reference's preprocess.par_transform for one value, as the prologue of every forward kernel
// computes it (fused_fwd.h, train_chain.h, train_chain32.h, train_kernels.h:affine_in_kernel).
#pragma once
#ifndef __HIPCC_RTC__  // (hiprtc brings its own runtime header: csrc/jit.hip compiles this file at run time)
#include <hip/hip_runtime.h>
#endif

#include "../../include/v21_types.h"

namespace v21 {
// preprocess.par_transform of ONE value (include/v21.h: v21_affine_in).  The map is float64 in the reference's own order
// of operations (preprocess.py:105-108: subtract the minimum, divide by the span, times two, minus one -- r5: until r4 the
// device multiplied by 2 / span, one float64 rounding apart from the host's few-row route, which could move the last bit
// of the float32 result: ADVICE r4); what differs between the two forms is the dtype of floor and log10:
//   par_transform_f32: the reference's branch for float32 arrays -- floor (float)1e-6, np.log10 of a float32 array is
//     a float32 (here AND on the host's few-row route: the float64 log10 rounded to float32, i.e. the correctly rounded
//     log10f -- one definition on both sides, so a row's transformed parameters do not depend on how many rows the call has);
//   par_transform_f64: float64 arrays -- everything in float64.
// (Until r3 this was `__log10f` and an f32 affine map: ~1e-6 of error on inputs of order one, which the five layers
//  amplify to ~1e-4 x std at the output -- ten times the stated f32 tolerance.  The float64 log10 and division cost ~130
//  VALU instructions per logged column and row, issued while the first weight block is still on its way.)
__host__ __device__ __forceinline__ float par_transform_f32(float x, int lm, double zf, double lo, double span) {
  if (zf > 0.0 && x == 0.f) x = (float)zf;
  double t = (double)x;
  if (lm) t = (double)(float)log10(t);
  t -= lo; t /= span; t *= 2.0; t -= 1.0;
  return (float)t;
}
__host__ __device__ __forceinline__ float par_transform_f64(double t, int lm, double zf, double lo, double span) {
  if (zf > 0.0 && t == 0.0) t = zf;
  if (lm) t = log10(t);
  t -= lo; t /= span; t *= 2.0; t -= 1.0;
  return (float)t;
}
}  // namespace v21
