/*
 * ctc_amd.h -- C ABI of the MI355X-native CTC loss (loss, analytic gradient, analytic Hessian).
 *
 * This is the drop-in boundary for the hot path of alexeytochin/tf_seq2seq_losses.  The reference has
 * no FFI of its own (it is pure Python on TensorFlow); the entry points below are what a binding for
 * its two public functions and its loss-data properties would call.  Each one cites the reference
 * interface it replaces (paths relative to the reference repo, v0.3.0).
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HIP, gfx950) unless a parameter says "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Every entry point is
 *     asynchronous on that stream, performs no allocation, no host<->device copy and no device
 *     synchronisation, and is therefore capturable into a hipGraph.
 *   - The caller owns every buffer, including the workspace (size from ctc_amd_workspace_bytes).
 *     Outputs are fully overwritten.  The library keeps no global state besides a thread-local
 *     error string and the test-only tier override of ctc_amd_debug_override (never set by the product path);
 *     calls are re-entrant for distinct (stream, workspace) pairs.  The override is a plain process-wide
 *     variable: ctc_amd_debug_override is NOT thread-safe against calls running in other threads -- set it
 *     between calls, from the one thread that makes them (tests and benchmarks only).
 *   - Layouts are dense row-major: logits[B][T][V] float32, labels[B][label_stride] int32,
 *     label_length[B], logit_length[B] int32, loss[B], grad[B][T][V], hess[B][T][V][T][V] float32.
 *   - `U` is a static upper bound on label_length (the reference uses the dynamic max(label_length),
 *     base_loss.py:482-486; any U >= max(label_length) gives identical loss/gradient/Hessian because
 *     the extra lattice states stay at log 0).  A sample with label_length[b] > U gets loss = +inf.
 *   - Return value: 0 on success, negative CTC_AMD_E* code otherwise; ctc_amd_last_error() has text.
 *     Not errors (reference semantics, classic_ctc_loss.py:50-52, base_loss.py:240-245,283-288):
 *     infeasible alignment => loss = +inf, gradient = 0, Hessian = 0; B == 0; T == 0.
 */
#ifndef CTC_AMD_H
#define CTC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTC_AMD_ABI_VERSION 5

/* lattice variant */
#define CTC_AMD_CLASSIC 0    /* classic_ctc_loss.py:33-70   (collapse repeats, then drop blanks)   */
#define CTC_AMD_SIMPLIFIED 1 /* simplified_ctc_loss.py:32-67 (drop blanks only)                     */

/* what the float input is / what the derivatives are taken with respect to */
#define CTC_AMD_WRT_LOGITS 0   /* input = logits; log_softmax fused (base_loss.py:59, tools.py:27-40);
                                  derivatives w.r.t. logits (what tf.GradientTape returns to the user) */
#define CTC_AMD_WRT_LOGPROBS 1 /* input = log-probabilities treated as independent variables
                                  (base_loss.py:71-99); derivatives w.r.t. them (loss_data.gradient /
                                  loss_data.hessian, base_loss.py:186-268)                            */

/* error codes */
#define CTC_AMD_OK 0
#define CTC_AMD_EINVAL (-1)     /* bad argument (null pointer, negative size, unsupported shape)      */
#define CTC_AMD_EWORKSPACE (-2) /* workspace too small                                                */
#define CTC_AMD_EHIP (-3)       /* HIP runtime error (launch failure)                                 */
#define CTC_AMD_ELABEL (-4)     /* ctc_amd_check_labels: a label outside [0, V) or equal to the blank */

/* Limits (CTC_AMD_EINVAL beyond them; the reference has none, its cost just grows):
 *   U <= CTC_AMD_MAX_U          label positions (16 per lane of one wavefront)
 *   V <= CTC_AMD_MAX_V          tokens for loss / gradient / alpha-beta (one 64 KB LDS token row per wavefront)
 *   V <= CTC_AMD_MAX_V_HESSIAN  tokens for ctc_amd_hessian / ctc_amd_hvp (V + 4 floats of LDS per wavefront)
 * ctc_amd_loss_grad* / ctc_amd_grad_resume / ctc_amd_alpha_beta: vector (16-byte / 8-byte) row accesses are used when V, the
 * strides AND the base pointers are aligned; any other alignment runs element-wise paths with identical results.
 * ctc_amd_hessian / ctc_amd_hvp require 16-byte aligned tensor pointers (CTC_AMD_EINVAL otherwise). */
#define CTC_AMD_MAX_U 1024
#define CTC_AMD_MAX_V 16384
#define CTC_AMD_MAX_V_HESSIAN 16380

/* selector for ctc_amd_workspace_bytes */
#define CTC_AMD_WS_LOSS_GRAD 0
#define CTC_AMD_WS_ALPHA_BETA 1
#define CTC_AMD_WS_HESSIAN 2
#define CTC_AMD_WS_HVP 3
/* ctc_amd_loss_grad* / ctc_amd_grad_resume with wrt == CTC_AMD_WRT_LOGITS and float32 (or 8-byte aligned bfloat16) tensors:
 * the workspace of the pipeline such a call selects -- checkpoint rows only when that is a fused tier (64 MB instead of
 * 680 MB at B=256 T=1000 U=128).  CTC_AMD_WS_LOSS_GRAD stays valid for every call (log-probability input, any format). */
#define CTC_AMD_WS_LOSS_GRAD_LOGITS 4

/* element types of the producer formats (ctc_amd_loss_grad_ex) */
#define CTC_AMD_F32 0
#define CTC_AMD_BF16 1
#define CTC_AMD_F16 2 /* IEEE half: read / written by the three-kernel pipeline (the fused tiers take float32 and bfloat16) */

/* ABI version of the loaded library (== CTC_AMD_ABI_VERSION of the header it was built from). */
int ctc_amd_abi_version(void);

/* Thread-local text of the last error returned on this thread ("" if none). */
const char *ctc_amd_last_error(void);

/* Name of the kernel pipeline ctc_amd_loss_grad would run for contiguous float32 tensors of these shapes ("fused6",
 * "fused5" or "v1"); diagnostic only (benchmarks and tests report it), never needed for correctness. */
const char *ctc_amd_pipeline_name(int kind, int wrt, int B, int T, int V, int U, int want_grad);

/*
 * Diagnostic override, for parity tests and benchmarks only (process-wide; set it between calls, not during one):
 *   key "pipeline": "" (best eligible tier, default), "v1", "fused5" -- forces a lower tier of ctc_amd_loss_grad
 *   key "hessian":  "" (default) or "slab" -- the general Hessian kernel also for labels of <= 32 positions
 *   key "hvp":      "" (default) or "v1"   -- the log-domain Hessian-vector pipeline also where the fused kernel applies
 * The library never reads the environment.  Returns CTC_AMD_EINVAL for an unknown key or value.
 */
int ctc_amd_debug_override(const char *key /*host*/, const char *value /*host*/);

/*
 * Diagnostic (tests/tools/soak.py, the parity tests): byte offset, inside the workspace of a float32 logits call of these
 * shapes (CTC_AMD_WS_LOSS_GRAD_LOGITS layout), of the int32[B] flag words the linear-domain fused kernel leaves -- 0 = the
 * utterance was computed in the linear domain, != 0 = it was redone by the log-domain roles (bits D1..D7).
 * Returns CTC_AMD_EINVAL when that call would not run the "fused6" pipeline.
 */
int ctc_amd_debug_flags_offset(int kind, int B, int T, int V, int U, size_t *out_offset /*host*/);
/* The same for the fused Hessian-vector kernel: offset of its int32[B] flag words inside a CTC_AMD_WS_HVP workspace
 * (CTC_AMD_EINVAL for shapes ctc_amd_hvp does not run that kernel for). */
int ctc_amd_debug_hvp_flags_offset(int kind, int B, int T, int V, int U, size_t *out_offset /*host*/);

/*
 * out2[0] = sum of the finite entries of loss[B], out2[1] = their number (as float): the two scalars a data-parallel
 * training loop all-reduces (tf.reduce_sum / reduce_mean of the loss: README.md:62, tests/benchmark.py:199), in one launch.
 * Asynchronous on `stream` like the compute entry points.
 */
int ctc_amd_reduce_loss(const float *loss, int B, float *out2, void *stream);

/*
 * Opt-in validation of the labels (the ONLY entry point that synchronises the stream and allocates -- a few bytes from the
 * stream-ordered pool; keep it off the hot path).  Returns CTC_AMD_ELABEL if any label inside label_length (and inside
 * U / label_stride) lies outside [0, V) or equals blank_index.  Without this check such a label is not an error in the
 * compute entry points: it is an impossible emission, the sample comes out infeasible (loss +inf, zero gradient).
 * Replaces: the InvalidArgumentError of tf.gather on TF-CPU for out-of-range labels (base_loss.py:328-344).
 */
int ctc_amd_check_labels(const int32_t *labels, int label_stride, const int32_t *label_length, int blank_index,
                         int B, int V, int U, void *stream);

/*
 * Measurement aid (bench.py's box probe): copies `bytes` (a multiple of 16) from src to dst with the access shape of the
 * kernels' row traffic (16 bytes per lane, non-temporal stores), asynchronously on `stream`.  Not part of the hot path.
 */
int ctc_amd_probe_copy(void *dst, const void *src, size_t bytes, void *stream);

/*
 * Measurement aid (bench.py --emulate-collective): ONE workgroup of `threads` threads holding `lds_bytes` of LDS that polls
 * the device clock for `microseconds` and exits -- the footprint of a latency-bound RCCL all-reduce kernel, to measure on
 * one GPU whether such a kernel runs beside the loss kernel or waits for its tail.  Asynchronous on `stream`.
 */
int ctc_amd_probe_spin(int threads, int lds_bytes, float microseconds, void *stream);

/* Bytes of device workspace the call selected by `what` needs for these shapes. */
int ctc_amd_workspace_bytes(int what, int kind, int B, int T, int V, int U, size_t *out_bytes /*host*/);

/*
 * Loss and (optionally) its gradient.
 * Replaces: classic_ctc_loss / simplified_ctc_loss forward (classic_ctc_loss.py:33-70,
 * simplified_ctc_loss.py:32-67 -> base_loss.py:38-99 -> loss_data.loss) and the first-order backward
 * forward_fn.backprop = d_loss[:,None,None] * gradient (base_loss.py:140-155, 262-298) composed with TF's
 * autodiff of log_softmax (tools.py:37-39) when wrt == CTC_AMD_WRT_LOGITS.
 *   grad   may be NULL (loss only).
 *   d_loss may be NULL (== ones); otherwise [B] upstream gradient that scales each sample's gradient.
 */
int ctc_amd_loss_grad(int kind, int wrt,
                      const float *logits, const int32_t *labels, int label_stride,
                      const int32_t *label_length, const int32_t *logit_length, int blank_index,
                      int B, int T, int V, int U,
                      float *loss, float *grad, const float *d_loss,
                      void *workspace, size_t workspace_bytes, void *stream);

/*
 * Forward/backward lattice variables in the reference's own layout and units (natural log, -inf for
 * impossible states): classic alpha/beta[B][T+1][U+1][2] (s=0 closed, s=1 open), simplified [B][T+1][U+1].
 * Replaces: ClassicCtcLossData.alpha/.beta (classic_ctc_loss.py:310-462), SimplifiedCtcLossData.alpha/.beta
 * (simplified_ctc_loss.py:291-438).  Parity/debug entry point; not on the fast path.
 * Here U must equal max(label_length) for the shapes to match the reference's.
 */
int ctc_amd_alpha_beta(int kind, int wrt,
                       const float *logits, const int32_t *labels, int label_stride,
                       const int32_t *label_length, const int32_t *logit_length, int blank_index,
                       int B, int T, int V, int U,
                       float *loss, float *alpha, float *beta,
                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * lg[B][T][V] = natural log of the posterior P(frame t emits token k | label) -- minus the gradient w.r.t. log-probabilities,
 * in log space: finite where the float32 gradient has underflowed (a posterior of e^-150 is returned as -150), -inf for
 * tokens no lattice state emits, for frames beyond logit_length and for infeasible samples.  Also writes loss[B].
 * Replaces: loss_data.logarithmic_logproba_gradient (base_loss.py:270-298): the segment log-sum-exp of
 * _combine_transition_probabilities(alpha[:, :-1], beta[:, 1:]) by token (base_loss.py:420-468, tools.py:74-119).
 * Debug / analysis entry point like ctc_amd_alpha_beta (three-kernel pipeline, full lattice rows).  V <= 8192.
 * Workspace: CTC_AMD_WS_ALPHA_BETA.
 */
int ctc_amd_log_posterior(int kind, int wrt,
                          const float *logits, const int32_t *labels, int label_stride,
                          const int32_t *label_length, const int32_t *logit_length, int blank_index,
                          int B, int T, int V, int U,
                          float *loss, float *lg,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * Dense Hessian hess[B][T][V][T][V] (the O(l^4) path), plus loss and gradient (grad may be NULL).
 * Replaces: loss_data.hessian (base_loss.py:186-260) for wrt == CTC_AMD_WRT_LOGPROBS, and
 * tape.batch_jacobian(tape.gradient(sum(loss), logits), logits) (README.md:58-71) for
 * wrt == CTC_AMD_WRT_LOGITS.  The reference's gamma tensor (classic_ctc_loss.py:167-308,
 * simplified_ctc_loss.py:85-191) is never materialised.
 */
int ctc_amd_hessian(int kind, int wrt,
                    const float *logits, const int32_t *labels, int label_stride,
                    const int32_t *label_length, const int32_t *logit_length, int blank_index,
                    int B, int T, int V, int U,
                    float *loss, float *grad, float *hess,
                    void *workspace, size_t workspace_bytes, void *stream);

/*
 * ctc_amd_loss_grad for the formats a producer kernel hands over: logits (and the gradient written back) as float32,
 * bfloat16 or float16 (CTC_AMD_F32 / CTC_AMD_BF16 / CTC_AMD_F16), with arbitrary element strides of the batch and time axes -- time-major [T,B,V] activations are
 * logits_stride_b = V, logits_stride_t = B*V; the token axis is contiguous.  Arithmetic is float32 either way.
 * Replaces: the `logit_to_logproba` entry of ctc_loss (base_loss.py:59, tools.py:27-40), which the reference can only
 * feed with a contiguous float32 [B,T,V] tensor (a transposed or bfloat16 producer pays one more 262 MB pass there).
 * SURVEY.md section 8(f) rank 3.  Workspace: CTC_AMD_WS_LOSS_GRAD.  Strides are in elements and must be >= V.
 */
int ctc_amd_loss_grad_ex(int kind, int wrt,
                         const void *logits, int logits_dtype, int64_t logits_stride_b, int64_t logits_stride_t,
                         const int32_t *labels, int label_stride,
                         const int32_t *label_length, const int32_t *logit_length, int blank_index,
                         int B, int T, int V, int U,
                         float *loss, void *grad, int grad_dtype, int64_t grad_stride_b, int64_t grad_stride_t,
                         const float *d_loss,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * ctc_amd_loss_grad_ex for PACKED (ragged) batches: no padding frames in memory.  Utterance b owns the logit_length[b] rows
 * row_offsets[b] .. row_offsets[b] + logit_length[b] - 1 of a [total_rows][row_stride] tensor (row_stride >= V elements, the
 * token axis contiguous); the gradient has the same packing (and its own element type).  T is max(logit_length) (a bound is
 * fine: it sizes the workspace, CTC_AMD_WS_LOSS_GRAD).  row_offsets is a DEVICE array of B int64; the ranges must not overlap.
 * Rows beyond logit_length[b] do not exist and are neither read nor written.  Runs the three-kernel pipeline.
 * Replaces: the [batch, max_length, num_tokens] padding the reference requires of its caller (base_loss.py:105-138) together
 * with the masks that undo it (base_loss.py:283-298).  SURVEY.md section 8(f) rank 3 (producer formats).
 */
int ctc_amd_loss_grad_packed(int kind, int wrt,
                             const void *logits, int logits_dtype, const int64_t *row_offsets, int64_t row_stride,
                             const int32_t *labels, int label_stride,
                             const int32_t *label_length, const int32_t *logit_length, int blank_index,
                             int B, int T, int V, int U,
                             float *loss, void *grad, int grad_dtype, int64_t grad_row_stride,
                             const float *d_loss,
                             void *workspace, size_t workspace_bytes, void *stream);

/*
 * ctc_amd_loss_grad_ex that also accumulates what a training loop takes from the losses, without a launch of its own.
 * Replaces: tf.reduce_sum / reduce_mean of the loss over the finite samples (README.md:62, tests/benchmark.py:199).
 *   sum2[0] += sum over the finite loss[b] of round(loss[b] * 2^20)   (int64, fixed point: integer adds give the same
 *              bits whatever order the workgroups finish in -- and whatever order ranks are all-reduced in; a single
 *              loss beyond +-2^42 ~ 4.4e12 enters clamped to that, so that it cannot wrap the sum)
 *   sum2[1] += number of finite loss[b]
 *   sum2 must hold zeros on entry (or a running total the caller wants to extend); zero_next, if not NULL, points at the
 *   two int64 of the NEXT step and is cleared by this call -- alternate two buffers and nothing ever needs a memset.
 * The linear-domain fused kernel adds inside its one launch; the other pipelines append one small launch.
 * Everything else as ctc_amd_loss_grad_ex.
 */
int ctc_amd_loss_grad_sum(int kind, int wrt,
                          const void *logits, int logits_dtype, int64_t logits_stride_b, int64_t logits_stride_t,
                          const int32_t *labels, int label_stride,
                          const int32_t *label_length, const int32_t *logit_length, int blank_index,
                          int B, int T, int V, int U,
                          float *loss, void *grad, int grad_dtype, int64_t grad_stride_b, int64_t grad_stride_t,
                          const float *d_loss, long long *sum2, long long *zero_next,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * First half of a forward -> backward pair (ABI v5): the losses alone, for a caller that WILL ask for the gradient of the same
 * batch with ctc_amd_grad_resume (torch.autograd: forward now, backward once d_loss is known).
 * Replaces: forward_fn (base_loss.py:140-149) when a backward pass follows.
 * Same work and same workspace contents as ctc_amd_loss_grad_ex with grad == NULL.  The difference is which utterances the
 * linear-domain kernel hands to its log-domain roles.  A loss-only call has no posterior mass to check its sweeps against; a
 * stand-alone one (ctc_amd_loss_grad* with grad == NULL: inference, scoring) therefore sends every utterance that shows one of
 * the kernel's conservative signs there -- which includes every utterance with logits as sharp as a trained model's (D7).  This
 * call trusts the linear sweeps' loss only where the sound detector (the posterior mass check of calls with a gradient) finds
 * nothing to redo -- at least 64 frames to spare over what the labels need, at most 12 frames per label position, P decaying by at
 * most 10 bits per frame (11.75 on the simplified lattice: logits up to about N(0, 3.25^2) over 256 tokens), one or two label
 * positions per lane (U <= 128) -- and keeps every sign outside those bounds.  Every loss-only call, stand-alone or first half, also
 * takes the log-domain roles for an utterance with more than 40 frames per label position when lanes hold two or more label
 * positions (U > 64; flag 2048).  The resume
 * call checks every utterance's posterior mass and redoes what fails, so the gradient is always verified; the loss of a
 * non-binding utterance is taken from the linear sweeps as it stands (measured: tests/tools/flag_stats.py).
 * Shapes that do not run the linear-domain fused kernel behave exactly like ctc_amd_loss_grad_ex with grad == NULL.
 * Workspace: CTC_AMD_WS_LOSS_GRAD (or CTC_AMD_WS_LOSS_GRAD_LOGITS), to be handed to ctc_amd_grad_resume untouched.
 */
int ctc_amd_loss_forward(int kind, int wrt,
                         const void *logits, int logits_dtype, int64_t logits_stride_b, int64_t logits_stride_t,
                         const int32_t *labels, int label_stride,
                         const int32_t *label_length, const int32_t *logit_length, int blank_index,
                         int B, int T, int V, int U,
                         float *loss,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * Second half of a forward -> backward pair: the gradient for a loss that ctc_amd_loss_forward (or ctc_amd_loss_grad /
 * ctc_amd_loss_grad_ex with grad == NULL) has just computed, weighted by d_loss (which a training loop only knows once the backward pass runs).
 * Replaces: forward_fn.backprop (base_loss.py:150-153) when the forward pass has already run.
 * The loss-only call stops where the alpha and beta chains meet and leaves its checkpoints, the softmax statistics and
 * log P in the workspace; this call runs the remaining half of the same kernel from there (together: one loss+gradient
 * call's work, split over two launches).  Contract: same arguments as that loss-only call, same workspace, nothing else
 * written to the workspace in between.  `loss` is rewritten with the same values for the utterances that the log-domain
 * kernel redoes (normally none).  Shapes that do not run the linear-domain fused kernel ("fused6") compute loss and
 * gradient anew, so the call is always valid after ANY loss-only call with the same arguments.
 * Workspace: CTC_AMD_WS_LOSS_GRAD.
 */
int ctc_amd_grad_resume(int kind, int wrt,
                        const void *logits, int logits_dtype, int64_t logits_stride_b, int64_t logits_stride_t,
                        const int32_t *labels, int label_stride,
                        const int32_t *label_length, const int32_t *logit_length, int blank_index,
                        int B, int T, int V, int U,
                        float *loss, void *grad, int grad_dtype, int64_t grad_stride_b, int64_t grad_stride_t,
                        const float *d_loss,
                        void *workspace, size_t workspace_bytes, void *stream);

/*
 * Hessian-vector product  out[b,t,k] = sum_{t2,k2} H[b,t,k,t2,k2] * vec[b,t2,k2]  without materialising H
 * (H as ctc_amd_hessian would fill it for the same `wrt`).  O(T*U) memory and work per utterance (tangent-mode
 * alpha/beta recursion), so it works at sizes where the [B,T,V,T,V] tensor does not fit.
 * Replaces: gradient_fn.backprop (base_loss.py:157-175), i.e. the contraction the reference performs with a
 * materialised Hessian when the gradient is differentiated once more (README.md:58-71).
 *   vec, out  [B,T,V] float   (H is symmetric, so this is also vec^T H)
 *   loss      [B] (out), grad [B,T,V] (out, NULL to skip): as ctc_amd_hessian
 * Workspace: CTC_AMD_WS_HVP.
 */
int ctc_amd_hvp(int kind, int wrt,
                const float *logits, const int32_t *labels, int label_stride,
                const int32_t *label_length, const int32_t *logit_length, int blank_index,
                int B, int T, int V, int U,
                const float *vec, float *loss, float *grad, float *out,
                void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CTC_AMD_H */
