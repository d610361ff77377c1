// train_kernels.h -- the non-GEMM kernels of a training step and of the generic
// forward path (gfx950).  All are HBM/L2 streaming kernels: coalesced 4-byte or 16-byte
// accesses, one wave per row where a row reduction is needed.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/v21.h"
#include "par_transform.h"
#include "chain_types.h"

namespace v21 {

// (StepDesc / StepCtx: chain_types.h)
static __global__ void step_tick_kernel(int* cur) { *cur += 1; }

// K2: loss_i = w_i sum_j (p - y)^2 (relative_mse_loss, emulator.py:68-81, with
// w_i = 1/(D amp_i^2); plain MSE w_i = 1/D) and dL/dp = scale * w_i * (p - y),
// scale = 2 / B_global ([K]: batch loss = mean of per-sample losses).
// One wave per row; WRITE_GRAD = false is the validation pass.
template <bool WRITE_GRAD>
__global__ void loss_grad_kernel(const float* __restrict__ p, long long ldp,
                                 const float* __restrict__ y, long long ldy,
                                 const float* __restrict__ w, float* __restrict__ dz,
                                 long long lddz, float* __restrict__ rowloss, int n, int d,
                                 float scale, const float* __restrict__ extra) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float wi = w[row];
  const float gs = scale * wi;
  const float* pr = p + (long long)row * ldp;
  const float* yr = y + (long long)row * ldy;
  float s = 0.f;
  for (int j = lane; j < d; j += 64) {
    const float df = pr[j] - yr[j];
    s += df * df;
    if (WRITE_GRAD) dz[(long long)row * lddz + j] = gs * df;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) rowloss[row] = wi * s + (extra ? extra[row] : 0.f);  // extra: kl_weight * KL_i (A13)
}

// deterministic sum of n floats by ONE workgroup -> out[0] (n <= a few 10^5)
static __global__ void sum_kernel(const float* __restrict__ v, int n, float* __restrict__ out, int accumulate,
                           float* __restrict__ slots = nullptr, StepCtx sc = StepCtx{nullptr, nullptr}, int slot = -1) {
  __shared__ double part[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)v[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
    out[0] = accumulate ? out[0] + (float)t : (float)t;
    // the step's loss, where run_epoch collects it: the replayed step's descriptor names the slot, an eager epoch passes it
    if (slots && (sc.desc || slot >= 0)) slots[sc.desc ? sc.desc[*sc.cur].slot : slot] = (float)t;
  }
}

// deterministic split-K reduction: g[i] = slab_0[i] + slab_1[i] + ... (fixed order)
static __global__ void reduce_slabs_kernel(float* __restrict__ g, const float* __restrict__ slabs, int nslab,
                                    long long stride, long long n) {
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    float4 s = *(const float4*)(slabs + i4);
    for (int k = 1; k < nslab; ++k) {
      const float4 t = *(const float4*)(slabs + k * stride + i4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *(float4*)(g + i4) = s;
  } else {
    for (long long i = i4; i < n; ++i) {
      float s = slabs[i];
      for (int k = 1; k < nslab; ++k) s += slabs[k * stride + i];
      g[i] = s;
    }
  }
}

// ... over the arena elements [lo, hi) only (r5: a gradient bucket of a data-parallel step; element by element the same
// sums in the same order as the whole-arena kernel)
static __global__ void reduce_slabs_range_kernel(float* __restrict__ g, const float* __restrict__ slabs, int nslab,
                                          long long stride, long long lo, long long hi) {
  const long long i4 = (lo & ~3ll) + ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= lo && i4 + 3 < hi) {
    float4 s = *(const float4*)(slabs + i4);
    for (int k = 1; k < nslab; ++k) {
      const float4 t = *(const float4*)(slabs + k * stride + i4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *(float4*)(g + i4) = s;
  } else {
    for (long long i = i4 < lo ? lo : i4; i < i4 + 4 && i < hi; ++i) {
      float s = slabs[i];
      for (int k = 1; k < nslab; ++k) s += slabs[k * stride + i];
      g[i] = s;
    }
  }
}

// ---- kernels of the NT training path (gemm_nt.h) ------------------------------------
// K5: the Keras data adapter's shuffled batch (emulator.py:369-378 [K]); one wave per batch row: x[idx] -> H0 row and H0^T column, y[idx] -> Y row, w[idx]
static __global__ void gather_batch_kernel(const float* __restrict__ x, int din, float* __restrict__ h0, long long ldh,
                                    float* __restrict__ h0t, long long ldt, const float* __restrict__ y, int dout,
                                    float* __restrict__ yb, long long ldy, const float* __restrict__ w,
                                    float* __restrict__ wb, const int* __restrict__ idx, long long first, int n,
                                    long long ldx_src, long long ldy_src, StepCtx sc = StepCtx{nullptr, nullptr}) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  if (sc.desc) first = sc.desc[*sc.cur].first;
  const long long s = idx ? (long long)idx[first + row] : first + row;
  const float* xs = x + s * ldx_src;
  for (int j = lane; j < din; j += 64) {
    const float v = xs[j];
    h0[(long long)row * ldh + j] = v;
    h0t[(long long)j * ldt + row] = v;
  }
  if (y) {
    const float* ys = y + s * ldy_src;
    for (int j = lane; j < dout; j += 64) yb[(long long)row * ldy + j] = ys[j];
  }
  if (lane == 0 && w) wb[row] = w[s];
}

// K2 (training form): like loss_grad_kernel, also writes the transposed gradient dZ^T (dout x batch)
static __global__ void loss_grad_t_kernel(const float* __restrict__ p, long long ldp, const float* __restrict__ y,
                                   long long ldy, const float* __restrict__ w, float* __restrict__ dz,
                                   long long lddz, float* __restrict__ dzt, long long ldt,
                                   float* __restrict__ rowloss, int n, int d, float scale,
                                   const float* __restrict__ extra) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float wi = w[row];
  const float gs = scale * wi;
  const float* pr = p + (long long)row * ldp;
  const float* yr = y + (long long)row * ldy;
  float s = 0.f;
  for (int j = lane; j < d; j += 64) {
    const float df = pr[j] - yr[j];
    s += df * df;
    const float gr = gs * df;
    dz[(long long)row * lddz + j] = gr;
    dzt[(long long)j * ldt + row] = gr;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) rowloss[row] = wi * s + (extra ? extra[row] : 0.f);
}

// ---- A13: the variational latent layer (V21_ACT_GAUSS, include/v21.h) ------------------
// counter-based standard normal: splitmix64 finaliser of (seed, step, row, d), two 24-bit
// uniforms, Box-Muller cosine branch (the test checker restates it: gauss_eps).
__device__ __forceinline__ float gauss_eps(unsigned long long seed, unsigned long long step, unsigned long long row,
                                           unsigned long long d) {
  unsigned long long x = seed + step * 0x9E3779B97F4A7C15ull + row * 0xD1B54A32D192ED03ull + d * 0x8CB92BA72F3D8DD7ull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  const float u1 = (float)((x >> 40) + 1ull) * (1.0f / 16777216.0f);         // (0, 1]
  const float u2 = (float)((x >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);  // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
struct GaussArgs {
  const float* zs; long long ldz;   // (rows x 2L): [z_mean | z_log_var], output of the Dense before
  int L, n;
  float* h; long long ldh;          // forward: z (rows x L) ...
  float* ht; long long ldt;         // ... and z^T (nullable)
  float* klrow;                     // forward: kl_weight * KL_i
  const float* dz; long long lddz;  // backward: dL/dz (rows x L)
  float* dzs; long long lddzs;      // backward: dL/d[z_mean | z_log_var] ...
  float* dzst;                      // ... and its transpose (pitch ldt)
  float beta;                       // forward: kl_weight; backward: kl_weight / B_global
  int sample;
  unsigned long long seed, step, row0;
};
// one wave per row: z = mu + exp(lv/2) eps;  KL_i = -1/2 sum_d (1 + lv - mu^2 - exp lv)
static __global__ void gauss_sample_kernel(const GaussArgs a) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.n) return;
  const float* zr = a.zs + (long long)row * a.ldz;
  float kl = 0.f;
  for (int d = lane; d < a.L; d += 64) {
    const float mu = zr[d], lv = zr[a.L + d];
    const float sd = expf(0.5f * lv);
    const float e = a.sample ? gauss_eps(a.seed, a.step, a.row0 + row, d) : 0.f;
    const float z = mu + sd * e;
    a.h[(long long)row * a.ldh + d] = z;
    if (a.ht) a.ht[(long long)d * a.ldt + row] = z;
    kl += -0.5f * (1.0f + lv - mu * mu - sd * sd);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kl += __shfl_xor(kl, o, 64);
  if (lane == 0) a.klrow[row] = a.beta * kl;
}
// d mu = dz + beta mu;  d lv = dz eps exp(lv/2)/2 + beta (exp lv - 1)/2     (beta = kl_weight / B)
static __global__ void gauss_sample_bwd_kernel(const GaussArgs a) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.n) return;
  const float* zr = a.zs + (long long)row * a.ldz;
  for (int d = lane; d < a.L; d += 64) {
    const float mu = zr[d], lv = zr[a.L + d];
    const float sd = expf(0.5f * lv);
    const float e = a.sample ? gauss_eps(a.seed, a.step, a.row0 + row, d) : 0.f;
    const float g = a.dz[(long long)row * a.lddz + d];
    const float dmu = g + a.beta * mu;
    const float dlv = g * e * 0.5f * sd + a.beta * 0.5f * (sd * sd - 1.0f);
    a.dzs[(long long)row * a.lddzs + d] = dmu;
    a.dzs[(long long)row * a.lddzs + a.L + d] = dlv;
    a.dzst[(long long)d * a.ldt + row] = dmu;
    a.dzst[(long long)(a.L + d) * a.ldt + row] = dlv;
  }
}

// ---- sweep forms (BASELINE configs[4]: many models, one batch): blockIdx.y = model ----------
constexpr int kSweepMax = 64;  // (r5: BASELINE configs[4] names 64 concurrent configs; 16 until r4.  The by-value argument blocks below stay under the 4-KiB kernel-argument segment: static_asserts)
// train_chain.h walks the packed weight streams in chunks of kChainUnit 1-KiB fragments (k-steps of 16
// features); a tile's k-steps are padded to whole chunks
constexpr int kChainUnit = 4;
__host__ __device__ constexpr int chain_steps(int d) { return ((d + 15) / 16 + kChainUnit - 1) / kChainUnit * kChainUnit; }
struct LossGroup {
  const float* p[kSweepMax]; long long ldp[kSweepMax];
  float* dz[kSweepMax]; long long lddz[kSweepMax];
  float* dzt[kSweepMax];
  float* rowloss[kSweepMax];
  const float* y; long long ldy;  // the batch's targets and row weights are shared by all models
  const float* w;
  long long ldt;
  int n, d;
  float scale;
};
static_assert(sizeof(LossGroup) <= 4096, "kernel arguments are limited to 4 KiB");
static __global__ void loss_grad_t_group_kernel(const LossGroup a) {
  const int k = blockIdx.y;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.n) return;
  const float wi = a.w[row];
  const float gs = a.scale * wi;
  const float* pr = a.p[k] + (long long)row * a.ldp[k];
  const float* yr = a.y + (long long)row * a.ldy;
  float* dz = a.dz[k] + (long long)row * a.lddz[k];
  float* dzt = a.dzt[k];
  float s = 0.f;
  for (int j = lane; j < a.d; j += 64) {
    const float df = pr[j] - yr[j];
    s += df * df;
    const float gr = gs * df;
    dz[j] = gr;
    dzt[(long long)j * a.ldt + row] = gr;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) a.rowloss[k][row] = wi * s;
}
struct SumGroup { const float* v[kSweepMax]; float* out[kSweepMax]; float* out2[kSweepMax]; int n; };
static __global__ void sum_group_kernel(const SumGroup a) {  // one workgroup per model, same order as sum_kernel
  __shared__ double part[16];
  const float* v = a.v[blockIdx.x];
  double s = 0.0;
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) s += (double)v[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
    a.out[blockIdx.x][0] = (float)t;
    if (a.out2[blockIdx.x]) a.out2[blockIdx.x][0] = (float)t;
  }
}

// K4: Keras-2.7 Adam (tf.raw_ops.ResourceApplyAdam [K]):
//   m += (g - m)(1 - b1);  v += (g^2 - v)(1 - b2);  w -= alpha m / (sqrt(v) + eps)
// (alpha = lr sqrt(1-b2^t)/(1-b1^t) from the host, f32) over the flat arena, plus the
// refresh of the two GEMM-friendly weight copies the next step reads:
// W^T (rows = outputs, padded) for the forward, row-padded W for the backward.
struct AdamLayer {
  long long w_off, wt_off, wp_off; int K, N; long long ldwt, ldwp;
  long long fw_off, bw_off; int KS, NS;  // train_chain.h streams (element offsets; k-/n-steps padded to whole units: chain_steps())
  // fused_train.h stream (AdamArgs::ts): first fragment of this layer as a forward virtual layer and as an
  // activation-gradient virtual layer (-1: layer 0 has none), k-steps of 16 per tile of either (a tile = 1 aux + k-steps)
  int tsf, tsb, tkf, tkb;
};
struct AdamArgs {
  float *w, *m, *v;
  const float* g;
  float *wt, *wp;
  long long n;
  float alpha, omb1, omb2, eps;
  int do_adam, L;
  void *fw, *bw;  // packed compute-type weight streams of train_chain.h (cprec != 0)
  int cprec;      // 0: none, 1: f16, 2: bf16
  int skip_nt;    // leave the fp32 W^T / padded-W copies alone (chain steps do not read them)
  // split-K slabs of the weight gradient still to be summed (single rank: the separate
  // reduce_slabs launch is folded in here; g is rewritten with the sum, same fixed order)
  float* gw; const float* slab; int nslab; long long slab_stride;
  StepCtx sc;  // replayed step: alpha comes from the descriptor
  long long i0;  // first element this launch works on (sharded data-parallel Adam: a rank's slice of the arena)
  int no_pack;   // update w, m, v only: the packed copies are rebuilt in a second pass once every rank's slice is back
  // f32 chain steps on a single rank (train_chain32.h): thread 0 turns the fixed-point batch loss into the float
  // slot(s) and clears the accumulator (the 16-bit chain leaves that to its weight-gradient kernel)
  unsigned long long* loss_acc; float* loss_out; float* loss_out2; int loss_slot;
  // the packed stream of the fused training kernel (fused_train.h), rewritten in this pass when the step that ends here took
  // that kernel (the next one probably will: it then needs no pack launch); nullptr otherwise.  Same element placement as
  // pack_stream_kernel below; padding and the zero aux fragments of the activation-gradient layers are never written.
  unsigned char* ts; int ts_bf16;
  int ts_fmt16;  // the stream is fused_train16.h's (16-feature tiles, 32-feature k-steps: tkf / tkb count those)
  AdamLayer lt[16];
};
// The layer an arena element belongs to: every block of 256 consecutive elements lies in ONE layer except the few
// that straddle a boundary, so the block looks its layer up once with scalar compares (uniform index -> the layer
// record comes through scalar loads) and only a straddling block falls back to the per-lane search (a chain of
// dependent vector loads from the argument block).
__device__ __forceinline__ int adam_layer_of(const AdamArgs& a, long long i) {
  int l = 0;
  for (int j = 1; j < a.L; ++j) l += i >= a.lt[j].w_off ? 1 : 0;
  return l;
}
// one arena element: summed gradient -> (m, v, w); returns the new weight
__device__ __forceinline__ float adam_update_element(const AdamArgs& a, long long i, float alpha) {
  float wi = a.w[i];
  if (a.do_adam) {
    float gi;
    if (a.nslab == 8) {  // the usual split: eight independent loads, summed in the same fixed order as the loop below
      float sl[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) sl[k] = a.slab[k * a.slab_stride + i];
      gi = sl[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) gi += sl[k];
      a.gw[i] = gi;
    } else if (a.nslab > 1) {
      gi = a.slab[i];
      for (int k = 1; k < a.nslab; ++k) gi += a.slab[k * a.slab_stride + i];
      a.gw[i] = gi;
    } else {
      gi = a.g[i];
    }
    const float mi = a.m[i] + (gi - a.m[i]) * a.omb1;
    const float vi = a.v[i] + (gi * gi - a.v[i]) * a.omb2;
    a.m[i] = mi; a.v[i] = vi;
    wi = wi - (mi * alpha) / (sqrtf(vi) + a.eps);
    a.w[i] = wi;
  }
  return wi;
}
// 16-bit element e of lane `lane` of fragment `frag` of the fused_train.h stream
__device__ __forceinline__ void ts_store(const AdamArgs& a, long long frag, int lane, int e, float v) {
  unsigned char* p = a.ts + ((frag * 64 + lane) << 4) + 2 * e;
  if (a.ts_bf16) *reinterpret_cast<__bf16*>(p) = (__bf16)v;
  else *reinterpret_cast<_Float16*>(p) = (_Float16)v;
}
__device__ __forceinline__ void adam_repack_element(const AdamArgs& a, long long i, float alpha, const AdamLayer& L) {
  const float wi = adam_update_element(a, i, alpha);
  if (a.no_pack) return;
  const long long r = i - L.w_off;
  if (a.ts && r >= (long long)L.K * L.N) {
    const int n = (int)(r - (long long)L.K * L.N);
    if (a.ts_fmt16) {  // float n % 16 of the aux fragment of tile n / 16
      const long long frag = L.tsf + (long long)(n >> 4) * (L.tkf + 1);
      reinterpret_cast<float*>(a.ts + (frag << 10))[n & 15] = wi;
    } else {
      // a bias: element [h][reg] of the aux fragment of its forward tile (fp32: the accumulator's initial value),
      // n = 32 nt + (reg & 3) + 8 (reg >> 2) + 4 h
      const int m = n & 31;
      const long long frag = L.tsf + (long long)(n >> 5) * (L.tkf + 1);
      reinterpret_cast<float*>(a.ts + (frag << 10))[((m >> 2) & 1) * 16 + (m >> 3) * 4 + (m & 3)] = wi;
    }
  }
  if (r < (long long)L.K * L.N) {  // a kernel element (biases have no copies)
    int k, n;
    if ((long long)L.K * L.N < (1ll << 32)) {  // (uniform) 32-bit division: a 64-bit one is ~150 instructions
      const unsigned ru = (unsigned)r, Nu = (unsigned)L.N, q = ru / Nu;
      k = (int)q; n = (int)(ru - q * Nu);
    } else {
      k = (int)(r / L.N); n = (int)(r % L.N);
    }
    if (!a.skip_nt) {
      a.wt[L.wt_off + (long long)n * L.ldwt + k] = wi;
      a.wp[L.wp_off + (long long)k * L.ldwp + n] = wi;
    }
    if (a.ts && a.ts_fmt16) {
      // fused_train16.h: forward tile n / 16, k-step k / 32: lane = 16 g + n % 16, half e with k % 32 = 16 (e / 4) + 4 g + e % 4;
      // activation-gradient tile k / 16, step n / 32: the same with k and n exchanged
      ts_store(a, L.tsf + (long long)(n >> 4) * (L.tkf + 1) + 1 + (k >> 5), 16 * ((k >> 2) & 3) + (n & 15), 4 * ((k >> 4) & 1) + (k & 3), wi);
      if (L.tsb >= 0)
        ts_store(a, L.tsb + (long long)(k >> 4) * (L.tkb + 1) + 1 + (n >> 5), 16 * ((n >> 2) & 3) + (k & 15), 4 * ((n >> 4) & 1) + (n & 3), wi);
    } else if (a.ts) {
      // fused_train.h (pack_stream_kernel's placement): forward tile n/32, k-step k/16: lane = 32 h + n%32 with
      // k%16 = 8 (e >> 2) + 4 h + (e & 3); activation-gradient tile k/32, step n/16: the same with k and n exchanged
      ts_store(a, L.tsf + (long long)(n >> 5) * (L.tkf + 1) + 1 + (k >> 4), 32 * ((k >> 2) & 1) + (n & 31), 4 * ((k >> 3) & 1) + (k & 3), wi);
      if (L.tsb >= 0)
        ts_store(a, L.tsb + (long long)(k >> 5) * (L.tkb + 1) + 1 + (n >> 4), 32 * ((n >> 2) & 1) + (k & 31), 4 * ((n >> 3) & 1) + (n & 3), wi);
    }
    if (a.cprec) {
      // forward fragment (tile n/32, k-step k/16): lane = 32*((k%16)/8) + n%32, element k%8
      const long long pf = L.fw_off + ((((long long)(n >> 5) * L.KS + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) << 3) + (k & 7);
      // backward fragment (tile k/32, n-step n/16): lane = 32*((n%16)/8) + k%32, element n%8
      const long long pb = L.bw_off + ((((long long)(k >> 5) * L.NS + (n >> 4)) * 64 + ((n >> 3) & 1) * 32 + (k & 31)) << 3) + (n & 7);
      if (a.cprec == 4) {
        // fp32 streams of train_chain32s.h (KS / NS = fragments of four k / n per 64-wide tile):
        // forward fragment (tile n/64, k/4): lane = n%64, element k%4; backward (tile k/64, n/4): lane = k%64, element n%4
        const long long qf = L.fw_off + ((((long long)(n >> 6) * L.KS + (k >> 2)) * 64 + (n & 63)) << 2) + (k & 3);
        const long long qb = L.bw_off + ((((long long)(k >> 6) * L.NS + (n >> 2)) * 64 + (k & 63)) << 2) + (n & 3);
        reinterpret_cast<float*>(a.fw)[qf] = wi;
        reinterpret_cast<float*>(a.bw)[qb] = wi;
      } else if (a.cprec == 3) {
        // fp32 streams of train_chain32.h (KS / NS = fragments per 32-wide tile, two per k16 / n16 step):
        // forward fragment (tile n/32, step k/16, half (n/16)%2): lane = n%16 + 16*((k%16)/4), element k%4
        const long long qf = L.fw_off + ((((long long)(n >> 5) * L.KS + 2 * (k >> 4) + ((n >> 4) & 1)) * 64 + (n & 15) + 16 * ((k >> 2) & 3)) << 2) + (k & 3);
        // backward fragment (tile k/32, step n/16, half (k/16)%2): lane = k%16 + 16*((n%16)/4), element n%4
        const long long qb = L.bw_off + ((((long long)(k >> 5) * L.NS + 2 * (n >> 4) + ((k >> 4) & 1)) * 64 + (k & 15) + 16 * ((n >> 2) & 3)) << 2) + (n & 3);
        reinterpret_cast<float*>(a.fw)[qf] = wi;
        reinterpret_cast<float*>(a.bw)[qb] = wi;
      } else if (a.cprec == 1) {
        reinterpret_cast<_Float16*>(a.fw)[pf] = (_Float16)wi;
        reinterpret_cast<_Float16*>(a.bw)[pb] = (_Float16)wi;
      } else {
        reinterpret_cast<__bf16*>(a.fw)[pf] = (__bf16)wi;
        reinterpret_cast<__bf16*>(a.bw)[pb] = (__bf16)wi;
      }
    }
  }
}
__device__ __forceinline__ void adam_repack_block(const AdamArgs& a, long long b0, long long end, float alpha) {
  const long long i = b0 + threadIdx.x;
  const long long last = b0 + blockDim.x - 1 < end - 1 ? b0 + blockDim.x - 1 : end - 1;
  const int l0 = adam_layer_of(a, b0), l1 = adam_layer_of(a, last);  // scalar
  if (i >= end) return;
  if (l0 == l1) adam_repack_element(a, i, alpha, a.lt[l0]);
  else adam_repack_element(a, i, alpha, a.lt[adam_layer_of(a, i)]);
}
static __global__ void adam_repack_kernel(const AdamArgs a) {
  if (a.loss_acc && blockIdx.x == 0 && threadIdx.x == 0) {
    const float f = (float)((double)(long long)*a.loss_acc * (1.0 / 4294967296.0));
    *a.loss_out = f;
    if (a.loss_out2 && (a.sc.desc || a.loss_slot >= 0)) a.loss_out2[a.sc.desc ? a.sc.desc[*a.sc.cur].slot : a.loss_slot] = f;
    *a.loss_acc = 0ull;
  }
  adam_repack_block(a, a.i0 + (long long)blockIdx.x * blockDim.x, a.i0 + a.n, a.sc.desc ? a.sc.desc[*a.sc.cur].alpha : a.alpha);
}
// sweep form: blockIdx.y = model, argument blocks in device memory (one per model)
struct AlphaGroup { float a[kSweepMax]; };
static __global__ void adam_repack_group_kernel(const AdamArgs* __restrict__ tab, const AlphaGroup alpha) {
  const AdamArgs& a = tab[blockIdx.y];
  if ((long long)blockIdx.x * blockDim.x >= a.n) return;
  adam_repack_block(a, (long long)blockIdx.x * blockDim.x, a.n, alpha.a[blockIdx.y]);
}

// fp32 W^T copies (rows = outputs, pitch p16(K)) of the small-batch forward path: one thread per arena element
static __global__ void wt_pack_kernel(const AdamArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  int l = 0;
  while (l + 1 < a.L && i >= a.lt[l + 1].w_off) ++l;
  const AdamLayer L = a.lt[l];
  const long long r = i - L.w_off;
  if (r < (long long)L.K * L.N) {
    const int k = (int)(r / L.N), n = (int)(r % L.N);
    a.wt[L.wt_off + (long long)n * L.ldwt + k] = a.w[i];
  }
}
// rows of x -> zero-padded rows of pitch ldd (the latency GEMM reads whole 16-float groups)
static __global__ void copy_pad_kernel(float* __restrict__ dst, long long ldd, const float* __restrict__ src, long long lds_,
                                long long n, int d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ldd) return;
  const long long row = i / ldd;
  const int j = (int)(i % ldd);
  dst[i] = j < d ? src[row * lds_ + j] : 0.f;
}

// diagnostics: the shader clock UNDER LOAD.  One wave samples (s_memtime, s_memrealtime) pairs -- the shader-clock cycle
// counter and the constant 100 MHz reference counter -- every `period` reference ticks for `duration` reference ticks
// (or until `max_samples` pairs are written: both bounds are reached whatever else happens), while the kernels under
// test run beside it (it needs one wave slot and a handful of registers).  d(memtime) / d(memrealtime) x 100 MHz is the
// clock those kernels ran at (bench.py: roofline.clock_ghz).
static __global__ void clock_probe_kernel(unsigned long long* out, int max_samples, unsigned long long period, unsigned long long duration) {
  if (threadIdx.x != 0) return;
  unsigned long long t0, r0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  out[0] = t0; out[1] = r0;
  int n = 1;
  unsigned long long next = r0 + period;
  while (n < max_samples) {
    unsigned long long t, r;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r)::"memory");
    if (r >= next) {
      out[2 * n] = t; out[2 * n + 1] = r;
      ++n;
      next += period;
      if (r - r0 >= duration) break;
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  out[2 * max_samples] = (unsigned long long)n;
}

// per-layer-path prologue (the fused and the chain kernels do this while they gather their rows); SRC = float: rows in
// device memory / float32 host rows; SRC = double: float64 host rows staged as they are (v21_mlp_forward)
template <class SRC>
__global__ void affine_in_kernel(float* __restrict__ dst, long long ldd, const SRC* __restrict__ src,
                                 long long lds_, long long n, const v21_affine_in t) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * t.n) return;
  const long long row = i / t.n;
  const int j = (int)(i % t.n);
  const SRC x = src[row * lds_ + j];
  if constexpr (sizeof(SRC) == 8) dst[row * ldd + j] = par_transform_f64(x, t.log_mask[j], t.zero_floor[j], t.lo[j], t.span[j]);
  else dst[row * ldd + j] = par_transform_f32(x, t.log_mask[j], t.zero_floor[j], t.lo[j], t.span[j]);
}
static __global__ void affine_out_kernel(float* __restrict__ y, long long ldy, long long n, int d,
                                  float stdv, const float* __restrict__ mean) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * d) return;
  const long long row = i / d;
  const int j = (int)(i % d);
  y[row * ldy + j] = y[row * ldy + j] * stdv + mean[j];
}

// Weight stream of the fused forward kernel (fused_fwd.h): one thread per (fragment,
// lane).  Layer table in `lt`: per layer {K, N, ks, nt, w_off, b_off, first_frag}.
// flags: 1 = the layer's weights are read TRANSPOSED (element (f, n) = w[w_off + n * K + f]: an activation-gradient layer
// of fused_train.h, whose K is the real layer's N), 2 = no bias (zero aux fragment)
// fp32 rows -> 16-bit operand elements (f16 / bf16), row pitch ld16 halves, zero beyond the row's din features
static __global__ void rows_to_half_kernel(const float* x, int din, long long n, unsigned short* out, long long ld16, int is_bf16) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ld16) return;
  const long long r = i / ld16; const int c = (int)(i - r * ld16);
  const float v = c < din ? x[r * din + c] : 0.f;
  if (is_bf16) { const __bf16 h = (__bf16)v; out[i] = __builtin_bit_cast(unsigned short, h); }
  else { const _Float16 h = (_Float16)v; out[i] = __builtin_bit_cast(unsigned short, h); }
}
struct PackLayer { int K, N, ks, nt; long long w_off, b_off; int first; int flags; };
struct PackArgs {
  const float* w;        // flat arena
  const float* mean;     // out_dim floats or nullptr
  unsigned char* stream;
  int L, total, padded, fpi, epi, esize;  // esize: 2 (f16/bf16) or 4 (f32)
  int is_bf16;
  int all_hidden;        // fused_train.h: every layer's aux fragment is the accumulator's initial value (no output-orientation layer)
  int fmt16;             // fused_train16.h: 16-feature tiles, 32-feature k-steps (lt[].ks / nt counted that way), fragment order kmap
  PackLayer lt[32];
};
static __global__ void pack_stream_kernel(const PackArgs a) {
  const int F = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (F >= a.padded) return;
  unsigned char* dst = a.stream + (size_t)F * 1024;
  if (F >= a.total) {  // tail padding
    ((float4*)dst)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  int l = 0;
  while (l + 1 < a.L && F >= a.lt[l + 1].first) ++l;
  const PackLayer L = a.lt[l];
  const int rel = F - L.first, tl = L.ks + 1;
  const int nt = rel / tl, ks = rel % tl - 1;
  if (a.fmt16) {
    // fused_train16.h: lane = (out feature m = lane % 16 of the tile, group g = lane / 16); half e of the lane is input feature
    // kmap(ks, g, e) = 32 ks + 16 (e / 4) + 4 g + e % 4; aux fragment: bias[16 nt + 0..15] as 16 floats, zeros after
    const int m = lane & 15, g = lane >> 4;
    if (ks < 0) {
      float out[4] = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < 4; ++q) {
        const int idx = lane * 4 + q, n = 16 * nt + idx;
        if (idx < 16 && !(L.flags & 2) && n < L.N) out[q] = a.w[L.b_off + n];
      }
      ((float4*)dst)[lane] = make_float4(out[0], out[1], out[2], out[3]);
      return;
    }
    const int n = 16 * nt + m;
    for (int e = 0; e < 8; ++e) {
      const int f = 32 * ks + 16 * (e >> 2) + 4 * g + (e & 3);
      float v = 0.f;
      if (f < L.K && n < L.N) v = (L.flags & 1) ? a.w[L.w_off + (long long)n * L.K + f] : a.w[L.w_off + (long long)f * L.N + n];
      if (a.is_bf16) ((__bf16*)dst)[lane * 8 + e] = (__bf16)v;
      else ((_Float16*)dst)[lane * 8 + e] = (_Float16)v;
    }
    return;
  }
  const int r = lane & 31, h = lane >> 5;
  if (ks < 0) {  // aux fragment: 256 floats, 4 per lane
    float out[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 4; ++q) {
      const int idx = lane * 4 + q;
      float v = 0.f;
      if (l < a.L - 1 || a.all_hidden) {
        if (idx < 32 && !(L.flags & 2)) {  // [h][reg]: bias[32nt + rho(reg) + 4h]
          const int hh = idx >> 4, reg = idx & 15;
          const int n = 32 * nt + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
          if (n < L.N) v = a.w[L.b_off + n];
        }
      } else {
        if (idx < 32) { const int n = 32 * nt + idx; if (n < L.N) v = a.w[L.b_off + n]; }
        else if (idx < 64) { const int n = 32 * nt + idx - 32; if (n < L.N && a.mean) v = a.mean[n]; }
      }
      out[q] = v;
    }
    ((float4*)dst)[lane] = make_float4(out[0], out[1], out[2], out[3]);
    return;
  }
  const int n = 32 * nt + r;
  for (int e = 0; e < a.epi; ++e) {
    const int f = a.fpi * ks + 8 * (e >> 2) + 4 * h + (e & 3);
    float v = 0.f;
    if (f < L.K && n < L.N) v = (L.flags & 1) ? a.w[L.w_off + (long long)n * L.K + f] : a.w[L.w_off + (long long)f * L.N + n];
    if (a.esize == 4) ((float*)dst)[lane * 4 + e] = v;
    else if (a.is_bf16) ((__bf16*)dst)[lane * 8 + e] = (__bf16)v;
    else ((_Float16*)dst)[lane * 8 + e] = (_Float16)v;
  }
}

}  // namespace v21
