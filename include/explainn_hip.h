/*
 * explainn_hip.h -- C ABI of libexplainn_hip.so: the MI355X (gfx950) implementation of the
 * ExplaiNN batched forward/backward over one-hot DNA.
 *
 * The reference (oriolfornes/ExplaiNN) has no FFI: its boundary for this path is the Python
 * nn.Module surface.  Each entry point below names the reference interface it replaces
 * (paths relative to /root/reference/explainn/).  All pointers except `explainn_ctx*`,
 * `explainn_params*`, `explainn_grads*` and the out-parameters documented as host are DEVICE
 * pointers (fp32, contiguous, row-major, the reference's state_dict shapes).  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  No call synchronises the device unless
 * it says so; nothing here allocates in the launch path (scratch is owned by the context).
 *
 * Every function returns 0 on success, a negative EXPLAINN_E_* code otherwise;
 * explainn_last_error() gives the message for the calling thread.
 */
#ifndef EXPLAINN_HIP_H
#define EXPLAINN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXPLAINN_OK 0
#define EXPLAINN_E_ARG (-1)        /* bad shape / argument                                   */
#define EXPLAINN_E_HIP (-2)        /* a HIP runtime call failed                               */
#define EXPLAINN_E_BATCH1 (-3)     /* train mode with B == 1 (torch BatchNorm raises there)   */
#define EXPLAINN_E_STATE (-4)      /* backward without a matching train-mode forward          */
#define EXPLAINN_E_UNSUPPORTED (-5)

#define EXPLAINN_LOSS_BCE_WITH_LOGITS 0 /* architectures/__init__.py:453-455 nn.BCEWithLogitsLoss() */
#define EXPLAINN_LOSS_MSE 1             /* architectures/__init__.py:456     nn.MSELoss()           */

typedef struct explainn_ctx explainn_ctx;

/* The 14 parameters + 6 running statistics of the reference module, by state_dict key
 * (architectures/__init__.py:72-104).  U = cnn_units, k = kernel_size, n = floor((L-k+1)/7),
 * T = n_features.  The running_* arrays are updated in place by a train-mode forward, exactly
 * as torch.nn.BatchNorm1d does (momentum 0.1, unbiased variance into running_var); the
 * num_batches_tracked pointers may be NULL. */
typedef struct explainn_params {
    const float* conv_w;   /* linears.0.weight   (U,4,k)                                  */
    const float* conv_b;   /* linears.0.bias     (U)                                      */
    const float* bn1_w;    /* linears.1.weight   (U)                                      */
    const float* bn1_b;    /* linears.1.bias     (U)                                      */
    float* bn1_rm;         /* linears.1.running_mean (U)                                  */
    float* bn1_rv;         /* linears.1.running_var  (U)                                  */
    const float* fc1_w;    /* linears.6.weight   (100U,n,1)                               */
    const float* fc1_b;    /* linears.6.bias     (100U)                                   */
    const float* bn2_w;    /* linears.7.weight   (100U)                                   */
    const float* bn2_b;    /* linears.7.bias     (100U)                                   */
    float* bn2_rm;         /* linears.7.running_mean (100U)                               */
    float* bn2_rv;         /* linears.7.running_var  (100U)                               */
    const float* fc2_w;    /* linears.10.weight  (U,100,1)                                */
    const float* fc2_b;    /* linears.10.bias    (U)                                      */
    const float* bn3_w;    /* linears.11.weight  (U)                                      */
    const float* bn3_b;    /* linears.11.bias    (U)                                      */
    float* bn3_rm;         /* linears.11.running_mean (U)                                 */
    float* bn3_rv;         /* linears.11.running_var  (U)                                 */
    const float* final_w;  /* final.weight       (T,U)                                    */
    const float* final_b;  /* final.bias         (T)                                      */
    int64_t* bn1_nbt;      /* linears.1.num_batches_tracked  () int64, may be NULL        */
    int64_t* bn2_nbt;      /* linears.7.num_batches_tracked                               */
    int64_t* bn3_nbt;      /* linears.11.num_batches_tracked                              */
    /* Version of the VALUES behind the pointers above, maintained by the caller: bump it whenever
     * a parameter or running statistic changes (the Python binding derives it from torch's tensor
     * version counters).  Eval-mode entry points keep the folded BatchNorm / filter tables of the
     * last version they saw and rebuild them only when it differs.  0 = unknown: rebuild on
     * every call (what a zero-initialised struct gets). */
    uint64_t version;
} explainn_params;

/* Gradient outputs, same shapes as the parameters; every array is OVERWRITTEN (not
 * accumulated).  What `loss.backward()` leaves in `.grad` at selene/__init__.py:291. */
typedef struct explainn_grads {
    float* conv_w; float* conv_b; float* bn1_w; float* bn1_b;
    float* fc1_w;  float* fc1_b;  float* bn2_w; float* bn2_b;
    float* fc2_w;  float* fc2_b;  float* bn3_w; float* bn3_b;
    float* final_w; float* final_b;
} explainn_grads;

/* Replaces ExplaiNN.__init__ (architectures/__init__.py:44-107) as far as device state goes:
 * fixes (cnn_units, kernel_size, sequence_length, n_features) and allocates scratch for
 * batches of up to max_batch sequences on HIP device `device`.  Synchronous. */
int explainn_create(explainn_ctx** out, int cnn_units, int kernel_size, int sequence_length,
                    int n_features, int max_batch, int device);
void explainn_destroy(explainn_ctx* ctx);
const char* explainn_last_error(void);
/* bytes of device scratch the context holds */
int64_t explainn_scratch_bytes(const explainn_ctx* ctx);

/* ExplaiNN.forward in eval mode (architectures/__init__.py:109-114; callers predict.py:81-82,
 * selene/__init__.py:334).  x: (B,4,L) fp32 one-hot rows ACGT, N = all-zero column
 * (sequence/__init__.py:19-26).  logits: (B,T). */
int explainn_forward_eval(explainn_ctx* ctx, const float* x, int B, const explainn_params* p,
                          float* logits, void* stream);

/* ExplaiNN.forward in train mode (caller selene/__init__.py:288): batch statistics in the three
 * BatchNorms, running statistics updated in place, Dropout(0.3) after the first FC
 * (architectures/__init__.py:92).  keep_mask: optional (B,100U) uint8 keep-mask (1 = keep)
 * that replaces the built-in generator -- for parity tests; NULL uses the counter-based
 * generator keyed by (seed, b, unit, channel).  dropout_p == 0 disables dropout.
 * Keeps what explainn_backward needs inside ctx (one step in flight per context). */
int explainn_forward_train(explainn_ctx* ctx, const float* x, int B, const explainn_params* p,
                           const uint8_t* keep_mask, float dropout_p, uint64_t seed,
                           float* logits, void* stream);

/* Autograd backward of the train-mode forward above (selene/__init__.py:291 loss.backward()):
 * dlogits (B,T) -> all 14 parameter gradients.  freeze_top_n_filters zeroes rows [0,n) of the
 * filter gradient (the hook selene/__init__.py:509-515 / :254-257). */
int explainn_backward(explainn_ctx* ctx, const float* dlogits, int B, const explainn_params* p,
                      const explainn_grads* g, int freeze_top_n_filters, void* stream);

/* get_loss (architectures/__init__.py:446-456), mean reduction, fused with its gradient:
 * loss_out (1 float, device) and dlogits (B,T, device). */
int explainn_loss_grad(explainn_ctx* ctx, int loss_kind, const float* logits, const float* targets,
                       int B, float* loss_out, float* dlogits, void* stream);

/* One whole training step of the hot loop selene/__init__.py:288-291 without the optimiser:
 * train forward + loss + backward, enqueued back to back on `stream`. */
int explainn_train_step(explainn_ctx* ctx, const float* x, const float* targets, int B,
                        const explainn_params* p, const explainn_grads* g, int loss_kind,
                        float dropout_p, uint64_t seed, int freeze_top_n_filters,
                        float* logits, float* loss_out, void* stream);

/* The same step in the two halves a data-parallel run overlaps its gradient all-reduce with
 * (selene/__init__.py:288-291 has no such split; the reference is single-device):
 * explainn_train_step_fc runs forward, loss and the backward down to the per-unit FC stage -- on
 * return (in stream order) every gradient from fc1_w to final_b, the contiguous tail of a flat
 * buffer laid out in explainn_grads order and 97-99 % of its bytes, is final and may be handed to
 * RCCL on another stream; explainn_train_step_conv finishes the step (conv_w, conv_b, bn1_w,
 * bn1_b).  fc + conv == explainn_train_step, bit for bit. */
int explainn_train_step_fc(explainn_ctx* ctx, const float* x, const float* targets, int B,
                           const explainn_params* p, const explainn_grads* g, int loss_kind,
                           float dropout_p, uint64_t seed, float* logits, float* loss_out,
                           void* stream);
int explainn_train_step_conv(explainn_ctx* ctx, int B, const explainn_params* p,
                             const explainn_grads* g, int freeze_top_n_filters, void* stream);

/* model.linears(x_rep) in eval mode (test.py:151): per-unit outputs (B,U). */
int explainn_unit_outputs(explainn_ctx* ctx, const float* x, int B, const explainn_params* p,
                          float* outs, void* stream);
/* model.linears[:3](x_rep) in eval mode (test.py:159-160): exp(BN(conv)) per position,
 * (B,U,L-k+1). */
int explainn_unit_activations(explainn_ctx* ctx, const float* x, int B, const explainn_params* p,
                              float* acts, void* stream);

/* Filter -> PWM export, replacing the dense float16 (N,U,Lo) host array of
 * test.py:128-166 and the Python loops of interpret.py:363-459.  Both calls stream batches; the
 * accumulators live in caller memory (device pointers) and persist across calls.
 *
 * select: [B] bytes or NULL -- 1 = the sequence is one of the "well predicted" ones
 * (interpret.py:310-361, host logic); unselected sequences contribute nothing.
 *
 * explainn_filter_act_max: unit_max[u] = max(unit_max[u], max over selected sequences and positions
 * of the eval-mode activation rounded to float16 as test.py:137 stores it).  Zero unit_max before
 * the first batch; interpret.py:373's threshold is 0.5 * unit_max (in float16). */
int explainn_filter_act_max(explainn_ctx* ctx, const float* x, int B, const explainn_params* p,
                            const uint8_t* select, float* unit_max, void* stream);
/* explainn_filter_sites: every start position j of a selected sequence whose float16 activation
 * exceeds thresholds[u] is a site x[j : j+k] (interpret.py:401-421); pfm[u][t][a] (int32, (U,k,4),
 * A/C/G/T; an N inside a site counts for no letter) += its letters (interpret.py:431-459).  Sites
 * are ranked in (call, sequence, position) order -- feed the forward strand's batches first, then the
 * reverse strand's, as interpret.py:385-429 iterates -- and only the first site_cap per unit count
 * (interpret.py:423-425, 1e6 there); site_total[u] (int32 [U], zero before the first batch) carries
 * the rank across calls and ends as min(#sites, site_cap).  hit: optional (B,U) bytes out, 1 = the
 * sequence has at least one position above the unit's threshold (interpret.py:485-490). */
int explainn_filter_sites(explainn_ctx* ctx, const float* x, int B, const explainn_params* p,
                          const uint8_t* select, const float* thresholds, int site_cap,
                          int32_t* site_total, int32_t* pfm, uint8_t* hit, void* stream);

/* Base-code input: instead of the fp32 one-hot (16*L bytes per sequence) hand the
 * sequences over as a (B,L) byte matrix of base codes -- 0,1,2,3 = A,C,G,T, 4 = N, the integers
 * sequence.one_hot_encode (sequence/__init__.py:8-28) turns into one-hot columns -- and let the
 * kernel reverse-complement them on the fly when reverse_complement != 0 (the augmentation of
 * train.py:275-278 / predict.py:78-79 without a second copy of the data).  The codes are packed
 * into the context; every entry point above that takes `x` then accepts x == NULL, meaning "the
 * staged batch" (B must match, else EXPLAINN_E_STATE; passing a real x discards the staged batch).
 * Other byte values are treated as N and raise bit 0 of explainn_input_flags. */
int explainn_stage_codes(explainn_ctx* ctx, const uint8_t* codes, int B, int reverse_complement,
                         void* stream);

/* The fp32 one-hot packed into the context ahead of the forward: like explainn_stage_codes, the
 * entry points then take x == NULL.  Lets the caller read explainn_input_flags BEFORE anything
 * depends on the batch -- and route a batch that is not one-hot to the dense kernels (next entry)
 * instead of running it as if its soft columns were N. */
int explainn_stage_onehot(explainn_ctx* ctx, const float* x, int B, void* stream);

/* Soft inputs.  The reference's forward takes ANY float (B,4,L) tensor (architectures/__init__.py:111
 * feeds x to a grouped Conv1d); the fast path here needs one-hot columns.  enable != 0 makes every
 * following entry point read x as a general dense tensor (csrc/dense.hip: dense input moments,
 * dense filter bank + pooling, dense filter-gradient scatter, dense activations; everything behind
 * the pooled activations is the regular pipeline) -- same results, to rounding, as the fast path
 * gives on one-hot x.  x must stay valid until the backward of a train forward has been enqueued.
 * The filter -> PWM export (explainn_filter_*) has no dense form: EXPLAINN_E_UNSUPPORTED. */
int explainn_dense_input(explainn_ctx* ctx, int enable);

/* PWM scan: the reference's `PWM` module forward (architectures/__init__.py:157-168).
 * x: fp32 (B,4,L) rows A,C,G,T; pwms: fp32 (G,4,k); scores: fp32 (B,G) = max (EXPLAINN_PWM_MAX) or
 * sum (EXPLAINN_PWM_SUM) of the window scores over both strands.  No context needed. */
#define EXPLAINN_PWM_SUM 0
#define EXPLAINN_PWM_MAX 1
int explainn_pwm_scan(const float* x, int B, int L, const float* pwms, int G, int k, int scoring,
                      float* scores, void* stream);

/* One Adam step over n_tensors parameter tensors in a single launch -- torch.optim.Adam(params, lr)
 * with its defaults, the optimiser the reference builds (architectures/__init__.py:463-464) and
 * steps at selene/__init__.py:291.  params/grads/exp_avg/exp_avg_sq: HOST arrays of n_tensors
 * device pointers (fp32, contiguous), sizes: HOST array of element counts; step: 1 for the first
 * update; the hyper-parameters are doubles because torch derives 1-beta and the bias corrections
 * from Python floats (1.f - 0.999f is off by 5e-5).  No context needed. */
int explainn_adam_step(int n_tensors, float* const* params, const float* const* grads,
                       float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
                       int64_t step, double lr, double beta1, double beta2, double eps, void* stream);

/* Input validation result of every pack since the last call: bit 0 set = some column of x was
 * neither one-hot nor all-zero (such columns were treated as N).  Synchronises `stream`,
 * writes the flags to *flags_host and clears them. */
int explainn_input_flags(explainn_ctx* ctx, int* flags_host, void* stream);

/* Measurement aid (bench.py's per-kernel roofline; the reference only logs steps per second,
 * selene/__init__.py:296-304): with timing enabled every stage of the training step is bracketed
 * by HIP events on the launch stream; explainn_stage_times synchronises the device and returns
 * the microseconds of each stage of the LAST step (-1 for stages that did not run), in the order
 * of explainn_stage_name(0 .. explainn_stage_count()-1).  Off by default: the events cost a few
 * microseconds per stage. */
int explainn_stage_timing(explainn_ctx* ctx, int enable);
int explainn_stage_count(void);
const char* explainn_stage_name(int i);
int explainn_stage_times(explainn_ctx* ctx, float* us, int cap);

/* Test aid for the built-in dropout generator (architectures/__init__.py:92 nn.Dropout(0.3); the
 * reference exposes no mask either -- torch draws it inside the op).  After a train-mode forward
 * of B sequences, copies the per-(unit, sequence) 100-bit words "pre-activation > 0 AND kept by
 * dropout" the backward will use into out: device, uint32 (U, B, 4), channel r = bit (r & 31) of
 * word r >> 5.  With BatchNorm2's weight = 0 and bias > 0 every pre-activation is positive and the
 * words ARE the keep mask (tests/test_gpu_parity.py measures its rate, scaling and independence
 * that way).  EXPLAINN_E_STATE without a train forward of that batch size in flight. */
int explainn_debug_keep_bits(explainn_ctx* ctx, int B, uint32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EXPLAINN_HIP_H */
