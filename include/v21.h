/* v21.h -- C ABI of libv21.so: the MI355X (gfx950) engine behind the 21cmVAE
 * predict()/train() hot path.
 *
 * The reference (christianhbye/21cmVAE) has no FFI of its own: its numeric engine is
 * TensorFlow/Keras, reached through eight call sites in
 * VeryAccurateEmulator/emulator.py (:369, :402, :739, :753-754, :756, :789, :790, :827).
 * Every entry point below names the reference call site it replaces.  The Python
 * shim (21cmvae_amd/engine.py) binds these with ctypes; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, opaque handles, no exceptions cross the boundary;
 *   - every function returns 0 on success or a negative V21_ERR_* code;
 *     v21_last_error() returns a thread-local human-readable message;
 *   - host buffers are caller-owned, C-contiguous, alive for the call only;
 *     device memory is owned by the library (or, for *_dev entry points, by the
 *     caller, who passes raw device pointers on the context's stream);
 *   - weights travel in Keras get_weights() order: W0 (in,out) row-major, b0, W1, ...
 *     (layout confirmed by the reference's shipped .h5 files);
 *   - one context per device; calls on one context must be serialised by the caller.
 */
#ifndef V21_H
#define V21_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct v21_ctx v21_ctx;
typedef struct v21_mlp v21_mlp;
typedef struct v21_trainer v21_trainer;

enum {
  V21_OK = 0,
  V21_ERR_ARG = -1,         /* bad argument (null, shape, range)              */
  V21_ERR_HIP = -2,         /* a HIP runtime call failed                      */
  V21_ERR_UNSUPPORTED = -3, /* valid request this build cannot serve          */
  V21_ERR_STATE = -4,       /* call order violated (e.g. no data set)         */
  V21_ERR_COMM = -5         /* RCCL failure                                   */
};

/* arithmetic of the dense contractions (accumulation is always fp32) */
enum { V21_PREC_F32 = 0, V21_PREC_F16 = 1, V21_PREC_BF16 = 2 };
/* V21_ACT_GAUSS: the variational latent layer (the KL/ELBO mode the
 * north star names; the reference snapshot has only its `z_mean` layer name,
 * models/autoencoder_based_emulator/encoder.h5).  A layer with this activation is
 * Dense(dims[l] -> 2*dims[l+1]) whose output is [z_mean | z_log_var]; what the next
 * layer sees is z = z_mean + exp(z_log_var / 2) * eps (dims[l+1] wide).  forward()
 * is deterministic (z = z_mean); the trainer samples eps and adds
 * kl_weight * KL(N(z_mean, exp z_log_var) || N(0, 1)) to every row's loss. */
enum { V21_ACT_LINEAR = 0, V21_ACT_RELU = 1, V21_ACT_GAUSS = 2 };
enum { V21_DTYPE_F32 = 0, V21_DTYPE_F64 = 1 };

/* Fused prologue = preprocess.par_transform (preprocess.py:49-110) with the training-set statistics cached:
 *   t = x_j;  if (zero_floor_j > 0 && t == 0) t = zero_floor_j      (fx == 0 -> 1e-6, preprocess.py:76)
 *   if (log_mask_j) t = log10(t)                                     (preprocess.py:77-78)
 *   y_j = (t - lo_j) / span_j * 2 - 1,  span_j = hi_j - lo_j         (preprocess.py:105-108, float64)
 * and the float32 cast Keras applies to the float64 result [K].  lo / hi are the column minima / maxima of the
 * log-transformed TRAINING parameters (float64, computed by the host as the reference computes them, :89-101).
 * The reference floors and takes log10 IN THE DTYPE OF THE ARRAY IT IS HANDED (:74-78) and does the affine map in
 * float64 (:81-85, :105-108); the library keeps both branches:
 *   float64 rows (v21_mlp_forward with V21_DTYPE_F64): floor, log10 and map in float64 on the device;
 *   float32 rows (V21_DTYPE_F32, and every device-resident input): floor = (float)zero_floor_j, log10 rounded to
 *     float32 (the correctly rounded log10f), then the float64 map -- numpy's float32 branch up to log10f's last bit.
 * n <= 8. */
#include "v21_types.h" /* v21_affine_in */

/* Fused epilogue = preprocess.unpreproc (preprocess.py:27-46): y*std + mean_j. */
typedef struct {
  float std;
  const float* mean; /* host pointer, out_dim floats */
  int32_t n;
} v21_affine_out;

const char* v21_last_error(void);
int v21_version(void);
int v21_device_count(int* n);

/* ---- context: one per device; owns a HIP stream ------------------------------ */
int v21_ctx_create(int device, v21_ctx** out);
int v21_ctx_destroy(v21_ctx* ctx);
int v21_ctx_sync(v21_ctx* ctx);
/* adopt an external hipStream_t (e.g. torch's current stream); NULL restores own */
int v21_ctx_set_stream(v21_ctx* ctx, void* hip_stream);
int v21_ctx_get_stream(v21_ctx* ctx, void** hip_stream);

/* device memory + copies + event timing, so a host program needs nothing else */
int v21_malloc(v21_ctx* ctx, size_t bytes, void** dptr);
int v21_free(v21_ctx* ctx, void* dptr);
/* page-locked host memory: a result buffer of this kind is filled by the device over PCIe
 * directly (no staging copy) -- the Python shim hands out numpy arrays backed by a pool of
 * them for large predict() results */
int v21_host_alloc(v21_ctx* ctx, size_t bytes, void** hptr);
int v21_host_free(v21_ctx* ctx, void* hptr);
int v21_memcpy_h2d(v21_ctx* ctx, void* dst, const void* src, size_t bytes);
int v21_memcpy_d2h(v21_ctx* ctx, void* dst, const void* src, size_t bytes);
int v21_memset(v21_ctx* ctx, void* dst, int value, size_t bytes);
int v21_event_create(v21_ctx* ctx, void** ev);
int v21_event_destroy(v21_ctx* ctx, void* ev);
int v21_event_record(v21_ctx* ctx, void* ev);               /* on the ctx stream */
int v21_event_elapsed_ms(v21_ctx* ctx, void* start, void* stop, float* ms); /* syncs stop */

/* ---- dense stack: replaces emulator._gen_model (emulator.py:12-48) + the Keras
 * Model object it returns.  A chain of models (emulator -> decoder,
 * emulator.py:789-790; encoder -> decoder, emulator.py:517) is ONE stack whose
 * `act` array has a linear layer in the middle. ----------------------------------- */
int v21_mlp_create(v21_ctx* ctx, int n_layers, const int* dims /* n_layers+1 */,
                   const int* act /* n_layers */, v21_mlp** out);
int v21_mlp_destroy(v21_mlp* mlp);
int v21_mlp_num_params(const v21_mlp* mlp, size_t* n);
/* Keras Model.set_weights / get_weights (flat fp32 arena, Keras order) */
int v21_mlp_set_weights(v21_mlp* mlp, const float* flat, size_t n);
int v21_mlp_get_weights(v21_mlp* mlp, float* flat, size_t n);
/* NULL clears the transform.  Both are copied. */
int v21_mlp_set_input_transform(v21_mlp* mlp, const v21_affine_in* t);
int v21_mlp_set_output_transform(v21_mlp* mlp, const v21_affine_out* t);
/* 1 if (dims, act) has a fully fused register-resident kernel, else 0 */
int v21_mlp_has_fused(const v21_mlp* mlp, int precision, int* yes);

/* Model.predict (emulator.py:402, :753-754, :789-790) and Model.__call__ (:517, :827):
 * host (n, in_dim) f32/f64 -> host (n, out_dim) f32.  flags: bit0 = apply input
 * transform, bit1 = apply output transform, bit2 = force the generic per-layer path,
 * bit3 = never take the small-batch latency path.
 * Path selection (results agree within fp32 rounding of the summation order): the fused
 * one-launch kernel for the shipped stack shapes; for up to V21_SMALL_BATCH_ROWS rows in
 * f32 (and for any stack without a fused kernel) one latency-oriented launch per layer
 * that spreads a layer's output tiles over the chip -- a sampler that calls
 * predict() row by row (the reference's use case: README "40 ms" per call) waits for one
 * memory round trip per layer instead of for one wave walking the whole stack. */
int v21_mlp_forward(v21_mlp* mlp, const void* x, int x_dtype, int64_t n, float* y,
                    int precision, int flags);
/* Same, device-resident, asynchronous on the context stream.  ldx/ldy = row pitch
 * in floats (>= in_dim / out_dim). */
int v21_mlp_forward_dev(v21_mlp* mlp, const float* d_x, int64_t ldx, int64_t n,
                        float* d_y, int64_t ldy, int precision, int flags);
/* The fully fused register-resident kernel (csrc/fused_fwd.h) is straight-line code per (stack, precision): the stacks of
 * csrc/archs.h are compiled into the library, every other `_gen_model` output (emulator.py:12-48: any `hidden_dims`)
 * gets it instantiated AT RUN TIME by hiprtc (csrc/jit.h), in a background thread started by the first
 * v21_mlp_forward[_dev] call above V21_SMALL_BATCH_ROWS rows; calls that arrive earlier take the table-driven one-launch
 * kernel (same results to operand rounding).  Code objects are cached under $V21_KERNEL_CACHE (default
 * ~/.cache/21cmvae_amd/kernels) and looked up in `kernel_cache/` next to libv21.so first; V21_JIT=0 switches the
 * compilation off (cached kernels are still used).
 * v21_mlp_jit: ask for the kernel now and wait up to wait_ms milliseconds (< 0: until the compilation has ended, 0: do
 *   not wait); *status = 1 ready (also for the compiled-in stacks), 0 compiling; V21_ERR_UNSUPPORTED (status -1) when
 *   the stack cannot have one (non-linear output layer, variational head, too wide for the register budget, no
 *   libhiprtc) -- v21_last_error says which.
 * v21_jit_prebuild: compile (stack, precision) into `dir` (NULL: kernel_cache/ next to the library) without touching a
 *   GPU -- a build step for deployments that know their stacks. */
int v21_mlp_jit(v21_mlp* mlp, int precision, int wait_ms, int* status);
/* r5: the same for the fused TRAINING kernel (csrc/fused_train16.h: forward + loss + activation gradients of steps of
 * >= 8,193 rows in one launch, 64-row workgroups).  `_gen_model` takes any hidden_dims (emulator.py:12-48); four stacks
 * have the kernel compiled in (archs.h T1-T4), every other f16 / bf16 trainer of >= 8,193 rows per step whose stack the
 * template can express (linear output, no variational head, 2-8 layers up to 512 wide) has it instantiated at run time:
 * asked for when the trainer is created, taken once the code object is there (until then, and if it never comes, the
 * chain kernel serves).  v21_trainer_jit waits up to wait_ms (< 0: until compiled): *status 1 ready, 0 compiling;
 * V21_ERR_UNSUPPORTED (status -1) when this trainer cannot have one.  v21_jit_prebuild(..., precision | 16, dir) compiles
 * it ahead of time. */
int v21_trainer_jit(v21_trainer* tr, int wait_ms, int* status);
int v21_jit_prebuild(int n_layers, const int* dims, const int* act, int precision, const char* dir);
#define V21_FWD_IN_TRANSFORM 1
#define V21_FWD_OUT_TRANSFORM 2
#define V21_FWD_FORCE_GENERIC 4
#define V21_FWD_NO_SMALL 8
/* diagnostics / benchmarks: the table-driven one-launch forward (csrc/train_chain.h, FORWARD mode: any stack up to 512
 * wide, f16 / bf16) even where a compiled fused kernel exists -- what every other stack gets by default */
#define V21_FWD_FORCE_CHAIN 16
/* diagnostics / benchmarks: the run-time-instantiated fused kernel (v21_mlp_jit) even for a stack that has a
 * compiled-in one; waits for the compilation, V21_ERR_UNSUPPORTED if the stack cannot have one */
#define V21_FWD_FORCE_JIT 32
#define V21_SMALL_BATCH_ROWS 4096

/* ---- trainer: replaces Model.compile + Model.fit (emulator.py:369-378, :739-747,
 * :756-764; optimizer/loss from notebooks/Training.ipynb cells 4 and 10). ------- */
typedef struct {
  float lr, beta1, beta2, eps; /* Keras Adam: 1e-3, 0.9, 0.999, 1e-7 */
} v21_adam;

/* The stack's output layer must be linear (V21_ERR_UNSUPPORTED otherwise): the reference's output Dense has no
 * activation (emulator.py:44), and the loss gradient is taken with respect to the Dense output. */
int v21_trainer_create(v21_mlp* mlp, int precision, int max_batch, v21_trainer** out);
int v21_trainer_destroy(v21_trainer* tr);
int v21_trainer_set_adam(v21_trainer* tr, const v21_adam* cfg);
int v21_trainer_set_lr(v21_trainer* tr, float lr);
int v21_trainer_get_lr(v21_trainer* tr, float* lr);
/* Resident training / validation matrices.  y == NULL means y = x (autoencoder fit,
 * emulator.py:739-741).  row_weight w_i defines the loss: loss_i = w_i sum_j (y-p)^2
 * (relative_mse_loss, emulator.py:68-81 -> w_i = 1/(D amp_i^2); plain MSE -> 1/D). */
int v21_trainer_set_data(v21_trainer* tr, int which /*0 train, 1 val*/, const float* x,
                         const float* y, const float* row_weight, int64_t n);
/* One Keras epoch: rows visited in `perm` order (n_train int32; NULL = in order),
 * batches of `batch` with the partial last batch kept, epoch loss =
 * sum(batch_loss * n_b) / N.  With a communicator attached each rank takes its
 * slice of every global batch and gradients are all-reduced (sum) before Adam. */
int v21_trainer_run_epoch(v21_trainer* tr, const int32_t* perm, int batch, double* loss);
/* validation pass (which = 1) or loss over the training set (which = 0).  f16 / bf16 trainers on the chain kernel
 * evaluate the whole split in ONE forward-only launch (`batch` is then only validated); f32 trainers walk it in
 * batches of min(batch, max_batch). */
int v21_trainer_eval(v21_trainer* tr, int which, int batch, double* loss);
/* device pointers of the resident split `which` (v21_trainer_set_data): x (n, in_dim), y (n, out_dim; the same pointer as x
 * when the split was set with y == NULL), row weights (n) -- for custom loops that step on slices of the resident training
 * set with v21_trainer_step_dev (a step that reads rows of the resident training inputs lets the fused training kernels gather
 * their 16-bit copy: half the bytes of the step's largest read).  Valid until the next set_data of that split. */
int v21_trainer_get_data_dev(v21_trainer* tr, int which, const float** x, const float** y, const float** row_weight, int64_t* n);
/* single optimizer step on caller-provided device batch (bench / custom loops) */
int v21_trainer_step_dev(v21_trainer* tr, const float* d_x, const float* d_y,
                         const float* d_row_weight, int n_rows, int global_rows);
int v21_trainer_last_step_loss(v21_trainer* tr, double* loss); /* syncs */
/* optimizer state: iter + m + v (Keras optimizer_weights order = arena order) */
int v21_trainer_get_state(v21_trainer* tr, int64_t* iter, float* m, float* v, size_t n);
int v21_trainer_set_state(v21_trainer* tr, int64_t iter, const float* m, const float* v,
                          size_t n);
/* gradient of the last step (after all-reduce), for tests */
int v21_trainer_get_grad(v21_trainer* tr, float* g, size_t n);
/* ---- joint step (BASELINE configs[2]).  The reference trains the autoencoder, THEN encodes the
 * training signals with the finished encoder and trains the latent emulator on those latents
 * (emulator.py:739-764).  A joint epoch takes one optimizer step of EACH model on the same rows of every
 * batch; the emulator's targets are the latents the encoder produces for those rows in that step (no
 * gradient flows back into the encoder).  With the autoencoder's learning rate at 0 it is exactly the
 * reference's second phase.  `ae` holds the signals (set_data(0, signals, NULL, w)), `em` the parameters of the
 * same rows (its y argument is ignored); latent_layer = index of the encoder's linear output layer in `ae`'s
 * stack (linear, or the V21_ACT_GAUSS head: the emulator then learns z_mean).  Two f16 / bf16 trainers (chain kernel), or
 * two f32 trainers of max_batch <= 2,048 (the reference's arithmetic); with a
 * communicator on the context every rank carries its share of every batch and each model's gradients are exchanged as
 * in a plain step.  losses[0] = autoencoder, losses[1] = emulator. */
typedef struct v21_joint v21_joint;
int v21_joint_create(v21_trainer* ae, v21_trainer* em, int latent_layer, v21_joint** out);
int v21_joint_destroy(v21_joint* j);
int v21_joint_run_epoch(v21_joint* j, const int32_t* perm /* nullable */, int batch, double* losses /* [2] */);
/* Validation of both models in one launch on the trainers' validation sets (set_data(1, ...): signals for `ae`, the
 * parameters of the same rows for `em`): losses[0] = the autoencoder's validation loss, losses[1] = the emulator's loss
 * against the latents the CURRENT encoder produces for the validation signals (emulator.py:754 without the host round
 * trip). */
int v21_joint_eval(v21_joint* j, double* losses /* [2] */);
/* Captured-step replay (hipGraph): run_epoch / step_dev capture one optimizer step per
 * batch geometry and replay it; first row, Adam step size and loss slot of each step come from a device
 * table.  Bit-identical to eager launches.  Off by default (the steps are GPU-bound on MI355X: replay frees
 * the host thread, it does not shorten a step); V21_ERR_UNSUPPORTED with a communicator or a
 * V21_ACT_GAUSS layer.  Replaces nothing in the reference (Keras fit, emulator.py:369-378, has no analogue). */
int v21_trainer_use_graph(v21_trainer* tr, int enable);
/* diagnostics ("race detection / sanitizers": GPU AddressSanitizer is not available on this
 * pool): fills the whole LDS of every CU with `pattern` (e.g. 0xFFFFFFFF = NaN as fp32, f16 and bf16) and returns
 * when that is done.  The poison tests launch it immediately before each kernel that keeps activations in LDS, so
 * that a column or mask tile read before it is written meets NaN instead of a fresh process's zeros. */
int v21_debug_poison_lds(v21_ctx* ctx, uint32_t pattern);
/* diagnostics: the shader clock UNDER LOAD.  start: one wave on a private stream samples the shader-clock cycle counter
 * against the constant 100 MHz reference counter every period_us for duration_ms, beside whatever runs on the context's
 * stream meanwhile; read: waits for it; mean / min / max of cycles per nanosecond over the sampling intervals.
 * bench.py reports it as roofline.clock_ghz (DVFS holds ~1.6-1.9 GHz under the fused kernel, 2.4 GHz is nominal). */
int v21_debug_clock_probe_start(v21_ctx* ctx, double duration_ms, double period_us);
/* diagnostics (r5): the shader clock FROM THE TIMED KERNEL ITSELF.  Launches the headline stack's fused kernel (archs.h
 * S1; f16 / bf16) in a separate instantiation whose wave 0 of every workgroup reads s_memtime (shader-clock cycles),
 * s_memrealtime (constant 100 MHz) and HW_REG_XCC_ID when the workgroup starts and when it ends.  d_stamps (device
 * memory): five 64-bit words per workgroup = ceil(n / 128) workgroups: [cycles at start, 100-MHz ticks at start, cycles at
 * end, ticks at end, XCD 0-7 in bits 0-3 | HW_REG_HW_ID of the workgroup's wave 0 in bits 8-39].  (end - start) cycles / ticks * 0.1 = GHz as that workgroup's CU saw it; bench.py reports
 * mean / min / max per XCD over a repeat of its K timed launches.  Otherwise the arguments of v21_mlp_forward_dev. */
int v21_debug_forward_clocked(v21_mlp* mlp, const float* d_x, int64_t ldx, int64_t n, float* d_y, int64_t ldy, int precision,
                              int flags, unsigned long long* d_stamps);
int v21_debug_clock_probe_read(v21_ctx* ctx, double* ghz_mean, double* ghz_min, double* ghz_max, int* samples);
/* diagnostics: the small-batch f32 chain kernel (csrc/train_chain32s.h) follows a host-built job table into the packed
 * weight streams without range checks; v21_trainer_create validates every address a row names against the allocated
 * streams (V21_ERR_STATE instead of a GPU memory fault).  This repeats that check against stream sizes the caller names
 * (bytes; negative = the real ones): the tests hand in a truncated stream.  V21_ERR_UNSUPPORTED for trainers on other paths. */
int v21_debug_check_chain_jobs(v21_trainer* tr, long long fw_bytes, long long bw_bytes);
/* ---- routes: WHICH kernels a call takes (csrc/routes.h holds the one decision the dispatch sites and these queries
 * share; INTEGRATION.md section 6 is the table, tests/test_routes.py asserts it).  The reference has one path (Keras:
 * emulator.py:369-378, 402); the library has several kernels for the same arithmetic, chosen from precision, row count,
 * stack and rank count.  The two v21_route_* queries are pure host logic (no GPU, no handles):
 *   v21_route_forward  the route v21_mlp_forward_dev takes for n rows of this stack (rt_ready: assume the run-time
 *                      instantiated kernel of a stack outside archs.h has arrived); route: 1 small (one NT launch per
 *                      layer), 2 fused_fwd compiled in, 3 fused_fwd instantiated at run time, 4 table-driven chain
 *                      kernel in FORWARD mode, 5 generic per-layer GEMM.
 *   v21_route_train    one optimizer step of `rows` rows of a trainer created with max_batch, on `nranks` ranks
 *                      (rt_ready: assume the run-time instantiated fused TRAINING kernel of a stack outside archs.h has arrived):
 *                      fwd: 1 per-layer NT, 2 train_chain_kernel (16-bit), 3 fused_train (128-row workgroups),
 *                      4 fused_train16 (64-row workgroups), 5 train_chain32_kernel, 6 / 7 train_chain32s_kernel<8> / <4>;
 *                      upd: 1 per-layer NT + adam_repack, 2 dw16_adam_kernel, 3 gemm_dw16[_lds] split-K + adam_repack,
 *                      4 dwadam32_kernel, 5 gemm_nt_dwadam_kernel, 6 sliced NT + adam_repack.
 * v21_mlp_last_route / v21_trainer_last_route report what the LAUNCH SITES recorded for the last call / eager step and
 * how many calls / steps took each route since creation (counts[route]; nullable).  v21_route_name: kind 0 forward,
 * 1 training forward, 2 update. */
int v21_route_forward(int n_layers, const int* dims, const int* act, int precision, int64_t n, int flags, int rt_ready, int* route);
int v21_route_train(int n_layers, const int* dims, const int* act, int precision, int max_batch, int rows, int nranks, int rt_ready, int* fwd, int* upd);
int v21_mlp_last_route(v21_mlp* mlp, int* route, long long counts[8]);
int v21_trainer_last_route(v21_trainer* tr, int* fwd, int* upd, long long fwd_counts[8], long long upd_counts[8]);
const char* v21_route_name(int kind, int route);
/* r5: where an eager optimizer step's time goes, by HIP events on the context's stream.  v21_trainer_phase_timing(tr, n,
 * cut) stamps the next n steps (n = 0: off) with TWO events each -- the step's start and one cut point:
 *   cut 1 after forward + loss + activation gradients (the chain / fused training launch, its stream pack included),
 *   cut 2 after the weight gradients (+ slab sums),  cut 3 after the gradient exchange has been joined,  cut 4 after Adam +
 *   packed copies (the whole step);
 * v21_trainer_phase_times returns the MEDIAN milliseconds from start to cut over the steps stamped since the last call.
 * A HIP event is a packet of its own (an empty interval between two reads ~5 us): the phases are DIFFERENCES of the
 * cumulative times of separate runs, in which the marker's cost cancels (21cmvae_amd/_native.py: Trainer.phase_profile).
 * Single-rank steps whose gradients and Adam are ONE launch have cut 2 = cut 3 = cut 1; the per-layer path has cut 1 = cut 2
 * = everything before the exchange.  Steps of a joint object or a sweep are not stamped. */
int v21_trainer_phase_timing(v21_trainer* tr, int steps, int cut);
int v21_trainer_phase_times(v21_trainer* tr, double* ms, int* steps);
/* diagnostics: which route the eager 16-bit steps of this trainer took since it was created:
 * out[0] steps through the 32-row chain kernel (csrc/train_chain.h), out[1] steps through the fused training kernel
 * (csrc/fused_train.h), out[2] of those that had to launch pack_stream_kernel first (the others found the kernel's
 * weight stream already written by the previous step's Adam pass), out[3] Adam passes that wrote that stream.
 * The tests assert on these so that "fused route == chain route" cannot pass with both taking the same kernel. */
int v21_debug_trainer_counters(v21_trainer* tr, long long out[4]);
/* diagnostics: s_memtime stamps of workgroup 0 of the last chain-kernel launch (train_chain.h):
 * [0] start, [1] batch gathered, [2..L+1] after forward layer l, [L+2] loss reduced,
 * [L+3..] after each backward layer (top down).  Off by default (a stamp holds its wave for ~600 cycles and the
 * other waves meet it at the next barrier: eleven stamps were 2-3 us of every step until r3); enable_stamps turns
 * them on for the launches that follow (steps already recorded into graphs keep what they were recorded with).
 * V21_ERR_STATE when the trainer runs the per-layer path (variational f32 or >512-wide stacks) or stamps are off. */
int v21_trainer_enable_stamps(v21_trainer* tr, int enable);
int v21_trainer_chain_stamps(v21_trainer* tr, uint64_t* out, int n);
/* Variational mode of a stack with a V21_ACT_GAUSS layer (A13; build-side extension, no
 * reference arithmetic exists for it): loss_i = recon_i + kl_weight * KL_i,
 * KL_i = -1/2 sum_d (1 + lv - mu^2 - exp lv).  sample = 0 -> eps = 0 (with kl_weight = 0
 * this is exactly the deterministic autoencoder of emulator.py:517); otherwise eps is a
 * counter-based standard normal keyed on (seed, step, row of the global batch, d) --
 * restated in oracle/ref_numpy.py:gauss_eps.  Evaluation passes use eps = 0. */
int v21_trainer_set_vae(v21_trainer* tr, float kl_weight, int sample, uint64_t seed);

/* ---- sweep: several models trained in lock step on ONE shared batch stream (BASELINE
 * configs[4]: "64 concurrent latent-dim/hidden-width configs packed as batched GEMM", 8 per
 * GPU).  New: the reference trains one model per fit() call (emulator.py:739-747); a sweep
 * runs the same fit arithmetic for every model, with phase k of the step (layer k forward,
 * loss, layer k backward, Adam) issued as one grouped launch for all models.  The trainers
 * must share context, precision, max_batch, depth, activations and in/out width (hidden and
 * latent widths differ); trainer 0 holds the training set (v21_trainer_set_data).  Each
 * trainer keeps its own Adam settings, state and weights and can be used on its own
 * (eval, get_state, ...) between sweep epochs.  count <= 64 (r5; 16 until r4: configs[4] names 64
 * concurrent configs, and a group of 8 models x 8 row blocks leaves three quarters of the chip idle). */
typedef struct v21_sweep v21_sweep;
int v21_sweep_create(v21_trainer** trainers, int count, v21_sweep** out);
int v21_sweep_destroy(v21_sweep* sw); /* the trainers stay alive */
/* one Keras-style epoch of every model (same shuffle, same batches); losses[count] */
int v21_sweep_run_epoch(v21_sweep* sw, const int32_t* perm, int batch, double* losses);

/* ---- data-parallel communicator (new: the reference is single-process, emulator.py:369,739,756).  One
 * process per GPU; every rank trains on its contiguous share of each global batch and the gradients are summed
 * before Adam.  Two transports behind the same step logic:
 *   v21_comm_init       RCCL inside the library (librccl.so.1 is dlopen'ed, so a single-GPU user needs none):
 *                       ncclAllReduce, or ncclReduceScatter + ncclAllGather in sharded mode, on the context stream.
 *   v21_comm_init_host  collectives supplied by the host as callbacks on page-locked HOST buffers (the library
 *                       stages device -> host -> device around them): any transport the host has -- the test
 *                       suite runs two ranks on one GPU over torch.distributed/gloo through it.
 * v21_comm_set_sharded(1): reduce-scatter of the gradient arena -> each rank applies Adam to its 1/R slice of
 * (w, m, v) only -> all-gather of the updated weights; 0 (default): one all-reduce, identical
 * Adam on every rank.  In sharded mode v21_trainer_get_state is a collective call (it gathers m and v).  Switching
 * the mode of a context does not touch trainers that already exist on it: after sharded steps a rank holds current
 * moments for ITS slice only, so call v21_trainer_get_state + v21_trainer_set_state (or create a new trainer) before
 * such a trainer goes on in all-reduce mode. -------- */
#define V21_COMM_ID_BYTES 128
typedef struct v21_comm_host_ops {
  void* user;
  /* all blocking, 0 = ok; buffers are host memory of nranks * n_per (or n) floats, results in place */
  int (*allreduce_sum_f32)(void* user, float* buf, size_t n);
  int (*reduce_scatter_sum_f32)(void* user, float* buf, size_t n_per); /* rank r: sums of [r n_per, (r+1) n_per) there */
  int (*allgather_f32)(void* user, float* buf, size_t n_per);          /* rank r contributes [r n_per, (r+1) n_per) */
} v21_comm_host_ops;
int v21_comm_get_unique_id(v21_ctx* ctx, void* id /* V21_COMM_ID_BYTES */);
int v21_comm_init(v21_ctx* ctx, int nranks, int rank, const void* id);
int v21_comm_init_host(v21_ctx* ctx, int nranks, int rank, const v21_comm_host_ops* ops);
/* r5: a communicator WITHOUT a transport: the context reports `nranks` ranks, a trainer on it computes this rank's share
 * of every global batch and takes the N > 1 step structure (operands -> weight gradients -> [exchange: nothing] -> Adam),
 * so the compute side of a data-parallel step can be timed on ONE GPU (bench.py: dp_compute_only).  The gradients are
 * NOT summed over anything: results are those of a rank whose peers contributed zeros. */
int v21_comm_init_null(v21_ctx* ctx, int nranks, int rank);
/* r5: the all-reduce form's gradient exchange in 1 message (default: the whole arena + loss slot after all weight
 * gradients) or 2: the weight gradients are formed in two launches, the output-side layers first; their half of the arena
 * (and the loss slot) is all-reduced on a second stream while the input-side layers' gradients are still being formed.
 * Same sums per element; with two ranks bit-identical to the one-message form.  Applies to 16-bit chain trainers in
 * all-reduce mode; every rank of a communicator must set the same value. */
int v21_comm_set_buckets(v21_ctx* ctx, int buckets);
int v21_comm_destroy(v21_ctx* ctx);
int v21_comm_set_sharded(v21_ctx* ctx, int on);
/* what the attached communicator reports about itself (RCCL: ncclCommCount / ncclCommUserRank); transport: 0 none,
 * 1 RCCL inside the library, 2 host-staged callbacks.  bench.py prints it beside the data-parallel legs. */
int v21_comm_info(v21_ctx* ctx, int* nranks, int* rank, int* transport);
int v21_comm_allreduce_f32(v21_ctx* ctx, float* d_buf, size_t n);           /* sum, in place */
int v21_comm_reduce_scatter_f32(v21_ctx* ctx, float* d_buf, size_t n_per);  /* in place over nranks * n_per floats */
int v21_comm_allgather_f32(v21_ctx* ctx, float* d_buf, size_t n_per);       /* in place over nranks * n_per floats */

#ifdef __cplusplus
}
#endif
#endif /* V21_H */
