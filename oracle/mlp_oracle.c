/* mlp_oracle.c -- plain-C restatement of the dense forward pass and of one Keras-Adam
 * step.  TEST INFRASTRUCTURE ONLY (a second, independent statement of oracle/ref_numpy.py;
 * never linked into the product).  Parity status: as ref_numpy.py (TensorFlow bit-level
 * outputs unpinned).
 *
 *   forward : Keras Dense, act(x W + b), ReLU hidden / linear last  (emulator.py:43,45)
 *   adam    : Keras 2.7 ResourceApplyAdam [K]: m += (g-m)(1-b1); v += (g*g-v)(1-b2);
 *             w -= alpha m / (sqrt(v) + eps), alpha = lr sqrt(1-b2^t)/(1-b1^t)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* y (n, dims[L]) = stack(x (n, dims[0])); w = flat arena W0|b0|W1|...; act[l] 1 = ReLU.
 * Accumulates each dot product in double when acc64 != 0, else in float (k ascending). */
int oracle_mlp_forward(const float* w, const int* dims, const int* act, int L, const float* x, long n,
                       float* y, int acc64) {
  int maxd = 0;
  for (int i = 0; i <= L; ++i) if (dims[i] > maxd) maxd = dims[i];
  float* a = (float*)malloc(sizeof(float) * (size_t)maxd * 2);
  if (!a) return -1;
  float* b = a + maxd;
  for (long r = 0; r < n; ++r) {
    memcpy(a, x + r * dims[0], sizeof(float) * (size_t)dims[0]);
    const float* p = w;
    for (int l = 0; l < L; ++l) {
      const int K = dims[l], N = dims[l + 1];
      const float* W = p; const float* bias = p + (size_t)K * N;
      for (int j = 0; j < N; ++j) {
        float v;
        if (acc64) {
          double s = 0.0;
          for (int k = 0; k < K; ++k) s += (double)a[k] * (double)W[(size_t)k * N + j];
          v = (float)(s + (double)bias[j]);
        } else {
          float s = 0.f;
          for (int k = 0; k < K; ++k) s = fmaf(a[k], W[(size_t)k * N + j], s);
          v = s + bias[j];
        }
        b[j] = (act[l] && v < 0.f) ? 0.f : v;
      }
      p += (size_t)K * N + N;
      float* t = a; a = b; b = t;
    }
    memcpy(y + r * dims[L], a, sizeof(float) * (size_t)dims[L]);
  }
  free(a < b ? a : b);
  return 0;
}

void oracle_adam_step(float* w, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                      float eps, long t) {
  const float b1p = powf(b1, (float)t), b2p = powf(b2, (float)t);
  const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
  for (long i = 0; i < n; ++i) {
    m[i] = m[i] + (g[i] - m[i]) * (1.0f - b1);
    v[i] = v[i] + (g[i] * g[i] - v[i]) * (1.0f - b2);
    w[i] = w[i] - (m[i] * alpha) / (sqrtf(v[i]) + eps);
  }
}
