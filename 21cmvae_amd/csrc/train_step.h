// train_step.h -- a whole single-rank optimizer step in ONE launch (gfx950, f16 / bf16 operands).
//
// A chain step used to be two dependent launches: train_chain_kernel (forward + activation gradients of every
// row block; at the per-GPU batch of 4,096 rows 128 workgroups on 128 of the 256 CUs, ~30 us) and then
// dw16_adam_kernel (weight gradients over the whole batch + Adam + packed weight copies, ~17 us, of which ~3 us is
// the dependent launch itself).  The operands of a layer's weight gradient are complete long before the chain
// ends -- H_l^T after the forward pass, dZ_l^T when the backward pass has gone through layer l, top layer first --
// and half of the chip idles meanwhile.  Here ONE grid of at most one workgroup per CU runs both:
//
//   workgroups [0, ncons)           row blocks: train_chain_body, which now counts the fragments of every layer's
//                                   operand it has written into ready[layer] (release at agent scope);
//   workgroups [ncons, ncons + 8p)  touch the packed weight streams once (chain_prefetch);
//   every workgroup, afterwards     WORKER: takes 32 x 32 tiles of [dW; db] from eight queues (one per XCD: a
//                                   workgroup drains the queue of the XCD it runs on first -- its tiles share their
//                                   operand rows in that XCD's L2 -- then helps the others), top layer first; waits
//                                   until ready[layer] has reached the layer's fragment count (acquire at agent
//                                   scope: the fragments were written on other XCDs), contracts over the batch with
//                                   16 waves, applies Adam, and rewrites the tile's packed fragments once no row block
//                                   can still read them (ready[layer - 1] complete: every row block is past this
//                                   layer's backward contraction).
//
// Nothing waits on a worker, and a row block never waits on anything: progress does not depend on which workgroups
// are resident, provided the row blocks get their CUs -- they have the lowest block ids, and the grid never exceeds
// one workgroup per CU.  A bounded spin (seconds) turns an impossible wait into an error flag instead of a hang.
// The last workgroup to leave turns the fixed-point batch loss into the float slots and clears the counters.
#pragma once
#include "dw_adam.h"

namespace v21 {

constexpr int kStepQueues = 8;
struct StepSync {
  unsigned ready[16];            // fragments of layer l's two operands written so far (this step)
  unsigned next[kStepQueues];    // next entry of queue x
  unsigned done;                 // workgroups that have left
  unsigned error;                // 1: a wait ran into its limit (never cleared by the device)
};
struct StepPlan {
  StepSync* sync;
  const int* order;              // order[qfirst[x] + i] = i-th logical tile of queue x (top layer first)
  int qfirst[kStepQueues + 1];
  unsigned expect[16];           // fragments per step of layer l: row blocks x (2 ceil(K/32) + 2 ceil(N/32))
  const DwAdamModel* model;      // device copy (the block does not fit the kernel arguments beside ChainArgs)
};

__device__ __forceinline__ int step_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return (int)(v & 7u);
}

// wave-uniform wait until *p >= expect; returns false when the limit was hit (error flagged)
__device__ __forceinline__ void step_wait(const unsigned* p, unsigned expect, StepSync* sy) {
  unsigned spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
    __builtin_amdgcn_s_sleep(16);
    ++spins;
    // a second or so (the row blocks of a step take tens of microseconds); once one wait has given up, all do
    if (spins > (1u << 20) || ((spins & 1023u) == 0 && __hip_atomic_load(&sy->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      __hip_atomic_store(&sy->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <class P>
__global__ void __launch_bounds__(64 * kChainWaves) train_step_kernel(const ChainArgs a, const DwAdamStep ds, const StepPlan pl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  if ((int)blockIdx.x < a.ncons) train_chain_body<P>(a, a);
  else if ((int)blockIdx.x < a.ncons + 8 * a.npref) chain_prefetch(a, a);
  // ---- worker
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the row block's last LDS reads and stores are done
  const DwAdamModel& md = *pl.model;
  StepSync* sy = pl.sync;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int* slot = reinterpret_cast<int*>(chain_smem + dw_adam_lds_bytes<kChainWaves>());  // [0]: tile of this round
  const int home = step_xcc_id();
  unsigned seen = 0;  // layers this workgroup has already found complete (bit l)
  const float alpha = ds.sc.desc ? ds.sc.desc[*ds.sc.cur].alpha : ds.alpha[0];
  for (int hop = 0; hop < kStepQueues;) {
    const int q = (home + hop) & (kStepQueues - 1);
    const int qn = pl.qfirst[q + 1] - pl.qfirst[q];
    if (tid == 0) {
      const unsigned i = __hip_atomic_fetch_add(&sy->next[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      slot[0] = i < (unsigned)qn ? pl.order[pl.qfirst[q] + (int)i] : -1;
    }
    __syncthreads();
    const int lb = slot[0];
    __syncthreads();  // (slot[0] is rewritten in the next round)
    if (lb < 0) { ++hop; continue; }
    int l = 0;
    while (l + 1 < md.L && lb >= md.lt[l + 1].first) ++l;
    if (!((seen >> l) & 1u)) {
      if (wave == 0) step_wait(sy->ready + l, pl.expect[l], sy);
      __syncthreads();
      seen |= 1u << l;
    }
    dw16_adam_tile<P, kChainWaves>(md, lb, alpha, ds.out_scale[0], ds.steps, ds.slot, ds.sc, chain_smem, false, [&](int layer) {
      // the packed fragments of this layer are still being read by row blocks that have not gone through its
      // backward contraction: wait until every row block has flushed the gradient of the layer BELOW
      if (layer > 0 && !((seen >> (layer - 1)) & 1u)) {
        if (wave == 0) step_wait(sy->ready + layer - 1, pl.expect[layer - 1], sy);
        seen |= 1u << (layer - 1);  // (the barrier that follows in dw16_adam_tile orders the other waves)
      }
    });
    __syncthreads();  // the tile's LDS is reused by the next one
  }
  // ---- leave: the last workgroup publishes the loss and clears the counters for the next step
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned before = __hip_atomic_fetch_add(&sy->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1 == gridDim.x) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const DwAdamLayer& g = md.lt[0];
      const unsigned long long acc = __hip_atomic_load(g.loss_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float f = (float)((double)(long long)acc * (1.0 / 4294967296.0));
      *g.loss_out = f;
      if (g.loss_out2 && (ds.sc.desc || ds.slot >= 0)) g.loss_out2[ds.sc.desc ? ds.sc.desc[*ds.sc.cur].slot : ds.slot] = f;
      __hip_atomic_store(g.loss_acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = 0; i < 16; ++i) __hip_atomic_store(&sy->ready[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = 0; i < kStepQueues; ++i) __hip_atomic_store(&sy->next[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sy->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace v21
