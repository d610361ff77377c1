// chain_types.h -- the argument blocks of the chain and fused training kernels (train_chain.h, train_chain32*.h,
// fused_train.h, fused_train16.h) and the two device helpers the fused training kernels share with the chain kernel, in a
// header of their own: r5 instantiates fused_train16<Arch, Prec> at RUN TIME (csrc/jit.hip hands v21_types.h +
// par_transform.h + fused_fwd.h + this file + fused_train.h + fused_train16.h to hiprtc), and hiprtc must not be handed
// train_chain.h / train_kernels.h / gemm_nt.h with their dozens of kernels and host-side launch code.
#pragma once
#ifndef __HIPCC_RTC__  // (hiprtc brings its own runtime header)
#include <hip/hip_runtime.h>
#endif

#include "../../include/v21_types.h"

namespace v21 {

constexpr int kTrainRowsPerWg = 128;  // batch rows per workgroup of the fused training kernel (fused_train.h; the host sizes the operand buffers for whole blocks)

// What differs between two optimizer steps of one epoch when a captured step is replayed (hipGraph):
// where the batch starts, Adam's bias-corrected step size, and the slot of the step's loss.  The host
// writes one descriptor per step of the epoch; the kernels read descriptor number *cur; the last
// node of the captured step increments *cur.  desc == nullptr: the values in the kernel arguments.
struct StepDesc { long long first; float alpha; int slot; };
struct StepCtx { const StepDesc* desc; const int* cur; };

struct ChainLayer {
  int K, N;            // Dense input / output width
  int KS, NT;          // forward: k-steps of 16 (padded to whole chunks: chain_steps), 32-wide output tiles
  int NS, KT;          // backward: n-steps of 16 (padded likewise), 32-wide input tiles
  int relu;            // ReLU on this layer's output
  int gauss;           // variational head (V21_ACT_GAUSS): N = 2*latent Dense outputs [z_mean | z_log_var]; the
                       // next layer sees z = z_mean + exp(z_log_var/2) eps (latent wide)
  int mask_tile;       // first tile of this layer's output mask in LDS (-1: none)
  long long fw_off, bw_off;  // fragment offsets (units of 8 elements) into the packed streams
  long long b_off;     // bias offset in the arena
  // operands of the weight gradient (gemm_dw16_kernel below), written here in MFMA-fragment order:
  // element (feature f, batch row b) at ((f/32 * BS + b/16) * 64 + 32*((b%16)/8) + f%32) * 8 + b%8
  void* ht16;          // input of this layer (K features; feature K is a constant row of ones)
  void* dzt16;         // gs * gradient w.r.t. this layer's output (N features)
};
// what stays the same from step to step: one per model (a sweep keeps a table of them in HBM)
struct ChainModel {
  int L;
  ChainLayer lt[16];
  const void* fw; const void* bw;  // packed weight streams
  long long fw_bytes, bw_bytes;    // their sizes (the prefetcher workgroups touch every line once)
  const float* w;                  // arena (biases)
  long long BS;                    // batch steps of 16 per feature tile of the transposed buffers
  // batch loss: every workgroup adds its rows' losses as 2^-32 fixed point (an integer sum does not
  // depend on the order of arrival); gemm_dw16_kernel turns it into the float slot and clears it
  unsigned long long* loss_acc;
  unsigned long long* stamps;      // diagnostics: s_memtime of workgroup 0 at every phase boundary
  // variational head (per model, so the members of a sweep may differ): loss_i += kl_weight * KL_i;
  // eps keyed on (seed, step, row0 + row, d) as in train_kernels.h
  float kl_weight;
  int sample;
  unsigned long long seed, step;
  // joint step (train_chain_joint_kernel): the fp32 outputs of this (linear) layer stay in LDS as the NEXT
  // model's targets (-1: none) -- the autoencoder's latent layer, emulator.py:753-754 without the host round trip
  int zcap_layer;
  // train_chain32s.h: what every wave does in every layer, worked out by the host (C32sJob rows, read with scalar loads)
  const int* jobs;
};
// the batch of this step (shared by every model of a sweep)
struct ChainStep {
  const float* x; long long ldx;   // source rows
  // the same rows as 16-bit operand elements (f16 / bf16 as the trainer's precision), row pitch ldx16 halves, zero-padded to
  // a multiple of 32 features: what the fused training kernels gather instead of x when the step reads the trainer's
  // resident training set (api_trainer.hip: v21_trainer_set_data) -- large steps are bound by HBM traffic, and the input
  // rows are rounded to 16 bits on their way into the first MFMA either way; nullptr: none
  const unsigned short* x16; long long ldx16;
  const float* y; long long ldy;   // targets (nullptr: y == x, the autoencoder)
  const float* rw;                 // row weights w_i
  const int* idx; long long first; // row m of the batch = source row idx[first + m] (or first + m)
  int rows;                        // rows of this rank's batch
  float scale;                     // 2 / B_global
  float gs;                        // gradient operand scale (power of two)
  float inv_b;                     // 1 / B_global (the KL term's own gradient)
  unsigned long long row0;         // position of this rank's first row in the global batch (noise key)
  unsigned long long step_off;     // steps since the per-model `step` was stored (a sweep stores it once per epoch)
  StepCtx sc;                      // replayed step (hipGraph): `first` comes from the step descriptor
  int y_from_lds;                  // joint step: the targets are the rows the previous model left in LDS (zcap_layer)
  // workgroups [0, ncons) carry row blocks; workgroups ncons + 8 p + x (p < npref) are PREFETCHERS of XCD x: they
  // touch every 128-byte line of the model's weight streams once and leave (see chain_prefetch)
  int ncons, npref;
  // validation pass: gather, forward and loss only -- nothing is written but the loss accumulator, so the launch may
  // cover ANY number of rows (the weight-gradient operands, sized for max_batch, are not touched)
  int fwd_only;
  // joint step: the emulator's workgroups first run the ENCODER alone (forward layers [0, nfwd) of the autoencoder,
  // nothing written but the captured latents) -- nfwd = 0: the whole stack
  int nfwd;
  int blk0;  // physical block blk0 is logical block 0 of this body (the joint kernel runs two families of row blocks)
  // FORWARD mode (Model.predict of any stack up to 512 wide, emulator.py:402 / :789-790; with fwd_only): the last layer's
  // outputs leave as fp32 rows out[row * ldo + n] = z * out_std + out_mean[n] (preprocess.unpreproc, preprocess.py:27-46;
  // out_mean == nullptr: plain outputs) instead of entering a loss; tin != nullptr: preprocess.par_transform
  // (preprocess.py:49-110, statistics cached in *tin, device memory) is applied to the <= 8 input columns as they
  // are gathered.  out == nullptr: training / validation.
  float* out; long long ldo;
  float out_std; const float* out_mean;
  const v21_affine_in* tin;
};
struct ChainArgs : ChainModel, ChainStep {};

typedef short chain_s4 __attribute__((ext_vector_type(4)));
typedef short chain_s8 __attribute__((ext_vector_type(8)));
// hardware transpose read (ds_read_b64_tr_b16): per 16-lane group a 4-row x 16-column block of 16-bit
// elements; lane 4q+p supplies the address of row q, columns 4p..4p+3 and lane i receives column i of
// the four rows.  EXEC must be all ones.
__device__ __forceinline__ chain_s4 chain_tr_read(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((chain_s4 __attribute__((address_space(3)))*)p);
}

}  // namespace v21
