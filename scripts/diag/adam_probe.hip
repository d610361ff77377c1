// Diagnostic: where do the 8 us of adam_repack_kernel (AE stack, 333 k parameters, 8 split-K slabs) go?
// Times variants of the SAME kernel back to back (data stays in the Infinity Cache, as in a real step):
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -o scripts/diag/adam_probe scripts/diag/adam_probe.hip
#include "../../21cmvae_amd/csrc/train_kernels.h"

#include <cstdio>
#include <vector>
using namespace v21;

__global__ void empty_kernel(const AdamArgs a) { if (a.n < 0) a.w[0] = 0.f; }
// the fp32 side only, four elements per thread as 16-byte accesses (what a vectorised update would cost)
__global__ void adam_vec4_kernel(const AdamArgs a) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 >= a.n) return;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 g = *(const f4*)(a.slab + i);
  for (int k = 1; k < 8; ++k) g += *(const f4*)(a.slab + k * a.slab_stride + i);
  f4 m = *(f4*)(a.m + i), v = *(f4*)(a.v + i), w = *(f4*)(a.w + i);
  m += (g - m) * a.omb1; v += (g * g - v) * a.omb2;
  for (int j = 0; j < 4; ++j) w[j] -= m[j] * a.alpha / (sqrtf(v[j]) + a.eps);
  *(f4*)(a.m + i) = m; *(f4*)(a.v + i) = v; *(f4*)(a.w + i) = w; *(f4*)(a.gw + i) = g;
}

int main() {
  const int dims[6] = {451, 352, 9, 32, 352, 451};
  AdamArgs a{};
  long long P = 0, fo = 0, bo = 0;
  a.L = 5;
  for (int l = 0; l < 5; ++l) {
    AdamLayer& al = a.lt[l];
    al.w_off = P; al.K = dims[l]; al.N = dims[l + 1];
    P += (long long)al.K * al.N + al.N;
    al.KS = chain_steps(al.K); al.NS = chain_steps(al.N);
    al.fw_off = fo; fo += (long long)((al.N + 31) / 32) * al.KS * 512;
    al.bw_off = bo; bo += (long long)((al.K + 31) / 32) * al.NS * 512;
  }
  const long long stride = ((P + 4 + 3) / 4) * 4;
  float *w, *m, *v, *g, *slab; void *fw, *bw;
  hipMalloc(&w, (P + 8) * 4); hipMalloc(&m, (P + 8) * 4); hipMalloc(&v, (P + 8) * 4); hipMalloc(&g, (P + 8) * 4);
  hipMalloc(&slab, 8 * stride * 4); hipMalloc(&fw, fo * 2); hipMalloc(&bw, bo * 2);
  hipMemset(w, 0, P * 4); hipMemset(m, 0, P * 4); hipMemset(v, 0, P * 4); hipMemset(slab, 0, 8 * stride * 4);
  a.w = w; a.m = m; a.v = v; a.g = g; a.n = P; a.alpha = 1e-3f; a.omb1 = 0.1f; a.omb2 = 1e-3f; a.eps = 1e-7f;
  a.fw = fw; a.bw = bw; a.skip_nt = 1; a.gw = g; a.slab = slab; a.slab_stride = stride;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %6.2f us/launch\n", name, ms * 5.f);
  };
  const unsigned nb = (unsigned)((P + 255) / 256);
  auto run = [&](const char* name, int do_adam, int nslab, int cprec) {
    AdamArgs b = a; b.do_adam = do_adam; b.nslab = nslab; b.cprec = cprec;
    timeit(name, [&] { hipLaunchKernelGGL(adam_repack_kernel, dim3(nb), dim3(256), 0, 0, b); });
  };
  timeit("empty kernel, same grid", [&] { hipLaunchKernelGGL(empty_kernel, dim3(nb), dim3(256), 0, 0, a); });
  run("full: 8 slabs + adam + f16 packs", 1, 8, 1);
  run("1 slab + adam + f16 packs", 1, 1, 1);
  run("8 slabs + adam, no packs", 1, 8, 0);
  run("1 slab + adam, no packs", 1, 1, 0);
  run("packs only", 0, 1, 1);
  timeit("vec4: 8 slabs + adam, no packs", [&] {
    AdamArgs b = a; b.nslab = 8;
    hipLaunchKernelGGL(adam_vec4_kernel, dim3((unsigned)((P / 4 + 255) / 256)), dim3(256), 0, 0, b); });
  run("full again", 1, 8, 1);
  return 0;
}
