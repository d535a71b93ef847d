// extern "C" entry points of libexplainn_hip.so (declared in include/explainn_hip.h) and the
// context that owns the device scratch.  Each entry point enqueues its pipeline stages on the
// caller's stream and returns; see DESIGN.md section 3 for the stage list.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <cstdlib>
#include <cstring>

#include "common.h"

static thread_local char g_err[512] = "";

void explainn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* explainn_last_error(void) { return g_err; }

namespace {
struct Carver {
    int64_t off = 0;
    char* base = nullptr;
    template <typename T>
    void take(T** p, int64_t count) {
        off = (off + 255) & ~int64_t(255);
        if (base) *p = reinterpret_cast<T*>(base + off);
        off += count * (int64_t)sizeof(T);
    }
};

void carve(explainn_ctx* c, Carver& cv) {
    const int64_t U = c->U, U4 = c->U4, n = c->n, Bs = c->Bs, NS = c->NS, K4 = c->K4;
    cv.take(&c->codesT, (int64_t)c->L * Bs);
    cv.take(&c->pk2, (int64_t)c->PW * Bs);
    cv.take(&c->nmask, (int64_t)c->NW * Bs);
    cv.take(&c->bm, (int64_t)4 * (Bs / 64) * c->Lp);
    cv.take(&c->G, K4 * K4);
    cv.take(&c->m, K4);
    cv.take(&c->alpha, U4);
    cv.take(&c->shift, U4);
    cv.take(&c->mug, U4);
    cv.take(&c->sig1, U4);
    cv.take(&c->Gw, U4 * K4);
    cv.take(&c->Wt, (int64_t)c->Uq * c->k * 20);
    cv.take(&c->Wf, (int64_t)conv_tiles_padded(c->U, c->k) * conv_ksteps(c->k) * 3 * 512);
    cv.take(&c->Wsg, (int64_t)conv_tiles_padded(c->U, c->k) * 32);
    // (rows up to whole unit groups of the filter-bank GEMM: its waves store their padding rows too)
    const int64_t Upad = (int64_t)32 * conv_tiles_padded(c->U, c->k);
    // (+ 64: the dump words behind the last row, see cpm_position)
    cv.take(&c->ext, Upad * n * Bs + 64);
    cv.take(&c->idx, Upad * n * Bs + 64);
    cv.take(&c->qs0, U * NS);
    cv.take(&c->qS1p, U * c->QCH * NS);
    cv.take(&c->qS2p, U * c->QCH * NS * NS);
    cv.take(&c->qbar, U * NS);
    cv.take(&c->VC, U * FC_H * NS);
    cv.take(&c->A2, U * FC_H * NS);
    cv.take(&c->A2f, c->NQ <= FC_BF_MAXN ? 0 : U * FC_MT * fc_nk4q(c->NQ) * 256);
    cv.take(&c->A2h, c->NQ <= FC_BF_MAXN ? U * FC_MT * fc_ks32(c->NQ) * 3 * 256 : 0);
    cv.take(&c->sh2, U * FC_H);
    cv.take(&c->sig2, U * FC_H);
    cv.take(&c->z, U * Bs);
    cv.take(&c->zhat, U * Bs);
    cv.take(&c->o, U * Bs);
    cv.take(&c->sig3, U);
    c->ZBLK = fc_fwd_blocks((int)Bs, c->NQ);
    cv.take(&c->z12p, U * c->ZBLK * 2);
    cv.take(&c->bits, U * Bs + 4);
    cv.take(&c->dz, U * Bs + 4);
    cv.take(&c->EQp, U * c->ACH * FC_H * NS);
    cv.take(&c->Sep, U * c->ACH * FC_H);
    cv.take(&c->EQs, U * FC_H * NS);
    cv.take(&c->Ttf, U * fc_nw16(c->NQ) * 3 * 4 * 256);          // bf16 x 3 pieces (fc.hip passB)
    cv.take(&c->Mff, U * fc_nw16(c->NQ) * 4 * fc_nw16(c->NQ) * 64);
    cv.take(&c->k0p, U * NS);
    cv.take(&c->dy, U4 * n * Bs);
    cv.take(&c->S12p, U * fc_ng(c->NQ) * (Bs / 16) * 2);
    cv.take(&c->Dspp, U * (Bs / 4) * K4);
    cv.take(&c->dlogits, (int64_t)c->maxB * c->T);
    cv.take(&c->flags, 64);
    cv.take(&c->dlT, (int64_t)c->T * Bs);
    cv.take(&c->gWp, c->T > HEAD_GEMM_MIN_T ? (int64_t)head_gw_chunks(c->maxB) * c->T * (c->U + 1) : 0);
    cv.take(&c->lossp, 256);
    cv.take(&c->site_cnt, U4 * Bs);
    cv.take(&c->site_off, U4 * Bs);
    cv.off = (cv.off + 255) & ~int64_t(255);
}

int check_batch(const explainn_ctx* c, int B) {
    if (!c) { explainn_set_error("null context"); return EXPLAINN_E_ARG; }
    if (B < 1 || B > c->maxB) {
        explainn_set_error("batch %d outside [1, max_batch=%d]", B, c->maxB);
        return EXPLAINN_E_ARG;
    }
    return EXPLAINN_OK;
}

#define TRY(call)                    \
    do {                             \
        int rc_ = (call);            \
        if (rc_ != EXPLAINN_OK) return rc_; \
    } while (0)

// Per-stage device times (explainn_stage_timing / explainn_stage_times): with timing on, every
// stage of the training step is bracketed by two HIP events on the launch stream.
const char* const kStageNames[ST_COUNT] = {
    "pack_tables", "moments", "prep1_stats", "conv_pool", "qmom", "prep2", "fc_fwd", "head_fwd",
    "loss", "head_bwd", "passA", "mid", "passB", "conv_bwd", "fin_bwd"};
#define STAGE(id, call)                                                                 \
    do {                                                                                \
        if (c->timing) HIP_TRY(hipEventRecord(c->ev0[id], s));                          \
        TRY(call);                                                                      \
        if (c->timing) { HIP_TRY(hipEventRecord(c->ev1[id], s)); c->timed |= 1u << (id); } \
    } while (0)

int eval_front(explainn_ctx* c, const float* x, int B, const explainn_params* p, hipStream_t s) {
    // every eval-mode entry point overwrites scratch a pending backward would read (codes, ext,
    // idx, z, bits ...): whatever train forward was in flight is gone, and its backward must fail
    // with E_STATE instead of returning the eval batch's gradients
    c->fwd_B = 0; c->tail_B = 0;
    if (c->dense) {
        if (!x) { explainn_set_error("dense input mode needs x"); return EXPLAINN_E_ARG; }
        c->staged_B = 0;
    } else {
        TRY(launch_pack(c, x, B, false, s));
    }
    // The folded tables depend on the parameters only: rebuilt when the caller's parameter version
    // moved (or is unknown), not per batch -- predict.py's loop and a validation pass run pack +
    // filter bank + FC + head per batch and nothing else.
    if (!(c->eval_valid && p->version != 0 && p->version == c->eval_version)) {
        TRY(launch_prep1_tables(c, p, s));
        TRY(launch_prep1(c, p, B, false, s));
        TRY(launch_prep2(c, p, B, false, s));
        c->eval_valid = true;
        c->eval_version = p->version;
    }
    return EXPLAINN_OK;
}
}  // namespace

extern "C" int explainn_create(explainn_ctx** out, int cnn_units, int kernel_size,
                               int sequence_length, int n_features, int max_batch, int device) {
    if (!out) { explainn_set_error("out is null"); return EXPLAINN_E_ARG; }
    *out = nullptr;
    if (cnn_units < 1 || n_features < 1 || max_batch < 1) {
        explainn_set_error("cnn_units, n_features and max_batch must be positive");
        return EXPLAINN_E_ARG;
    }
    if (kernel_size < 2 || kernel_size > MAX_K) {
        explainn_set_error("kernel_size %d unsupported (2..%d)", kernel_size, MAX_K);
        return EXPLAINN_E_UNSUPPORTED;
    }
    const int Lo = sequence_length - kernel_size + 1;
    const int n = Lo / POOLW;
    if (n < 1) {
        explainn_set_error("sequence_length %d too short for kernel_size %d and MaxPool1d(7,7)",
                           sequence_length, kernel_size);
        return EXPLAINN_E_ARG;
    }
    const int NQ = nq_bucket(n);
    if (NQ == 0) {
        explainn_set_error("pooled length n=%d exceeds the largest instantiated kernel (%d)", n, MAX_NQ);
        return EXPLAINN_E_UNSUPPORTED;
    }
    HIP_TRY(hipSetDevice(device));                 // (nothing allocated yet)
    explainn_ctx* c = new explainn_ctx();
    memset(c, 0, sizeof(*c));
    c->U = cnn_units; c->k = kernel_size; c->L = sequence_length; c->T = n_features;
    c->maxB = max_batch; c->device = device;
    c->Lo = Lo; c->n = n; c->U4 = (cnn_units + 3) & ~3; c->Uq = c->U4 / 4;
    c->NQ = NQ; c->NS = ns_stride(NQ); c->K4 = 4 * kernel_size;
    // Batch stride of every [..][b] array: a multiple of 64 lanes, but an ODD multiple, so that the
    // row stride (4*Bs bytes) is never a multiple of 512 B: with Bs = 1024 every row of ext/dy/...
    // was exactly 4096 B apart and all rows of a wavefront (and of every unit) landed on the same
    // memory channel (channel camping: ~8 us per dependent load, profiles/r01_c).
    c->Bs = (max_batch + 63) & ~63;
    if (((c->Bs / 64) & 1) == 0) c->Bs += 64;
    c->NW = (sequence_length + 31) / 32 + 2; c->PW = 2 * c->NW;
    c->Lp = ((c->NW * 32 + 63) / 64) * 64;
    {
        int q = (max_batch + 127) / 128;
        const int64_t per = (int64_t)c->U * c->NS * c->NS * 4;
        const int cap = (int)((int64_t)(64 << 20) / (per > 0 ? per : 1));
        if (q > 8) q = 8;
        if (q > cap) q = cap;
        if (q < 1) q = 1;
        c->QCH = q;
        int a = (max_batch + 255) / 256;                 // passA: two 128-sequence waves per chunk (fc.hip)
        if (a > 16) a = 16;
        if (a < 1) a = 1;
        c->ACH = a;
        // tuning overrides (experiments only; defaults above are what is tested and benchmarked)
        if (const char* e = getenv("EXPLAINN_ACH")) { const int v = atoi(e); if (v >= 1 && v <= 16) c->ACH = v; }
        if (const char* e = getenv("EXPLAINN_QCH")) { const int v = atoi(e); if (v >= 1 && v <= cap && v <= 8) c->QCH = v; }
    }
    Carver dry;
    carve(c, dry);
    c->bytes = dry.off;
    hipError_t e = hipMalloc(&c->base, c->bytes);
    if (e != hipSuccess) {
        explainn_set_error("hipMalloc(%lld bytes) failed: %s", (long long)c->bytes, hipGetErrorString(e));
        delete c;
        return EXPLAINN_E_HIP;
    }
    Carver real;
    real.base = c->base;
    carve(c, real);
    e = hipMemset(c->base, 0, c->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        explainn_set_error("scratch memset failed: %s", hipGetErrorString(e));
        (void)hipFree(c->base);
        delete c;
        return EXPLAINN_E_HIP;
    }
    // from here on every failure releases what was acquired (explainn_destroy copes with the
    // members that are still null)
    int rc = [&]() -> int {
        TRY(prep_configure(c));
        TRY(bwd_configure(c));
        TRY(fc_configure(c));
        return EXPLAINN_OK;
    }();
    if (rc != EXPLAINN_OK) { explainn_destroy(c); return rc; }
    *out = c;
    return EXPLAINN_OK;
}

extern "C" void explainn_destroy(explainn_ctx* c) {
    if (!c) return;
    for (int i = 0; i < ST_COUNT; ++i) {
        if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]);
        if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
    }
    if (c->base) (void)hipFree(c->base);
    delete c;
}

extern "C" int64_t explainn_scratch_bytes(const explainn_ctx* c) { return c ? c->bytes : 0; }

extern "C" int explainn_forward_eval(explainn_ctx* c, const float* x, int B,
                                     const explainn_params* p, float* logits, void* stream) {
    TRY(check_batch(c, B));
    hipStream_t s = static_cast<hipStream_t>(stream);
    TRY(eval_front(c, x, B, p, s));
    if (c->dense) {
        TRY(launch_dense_conv_pool(c, x, p, B, s));
        TRY(launch_fc_fwd(c, p, B, false, nullptr, 0.f, 0, s));
        TRY(launch_head_fwd(c, p, B, false, logits, nullptr, s));
        return EXPLAINN_OK;
    }
    TRY(launch_conv_pool(c, p, B, false, s));
    TRY(launch_fc_fwd(c, p, B, false, nullptr, 0.f, 0, s));
    TRY(launch_head_fwd(c, p, B, false, logits, nullptr, s));
    return EXPLAINN_OK;
}

extern "C" int explainn_unit_outputs(explainn_ctx* c, const float* x, int B,
                                     const explainn_params* p, float* outs, void* stream) {
    TRY(check_batch(c, B));
    hipStream_t s = static_cast<hipStream_t>(stream);
    TRY(eval_front(c, x, B, p, s));
    if (c->dense) {
        TRY(launch_dense_conv_pool(c, x, p, B, s));
        TRY(launch_fc_fwd(c, p, B, false, nullptr, 0.f, 0, s));
    } else {
        TRY(launch_conv_pool(c, p, B, false, s));
        TRY(launch_fc_fwd(c, p, B, false, nullptr, 0.f, 0, s));
    }
    TRY(launch_head_fwd(c, p, B, false, nullptr, outs, s));
    return EXPLAINN_OK;
}

extern "C" int explainn_unit_activations(explainn_ctx* c, const float* x, int B,
                                         const explainn_params* p, float* acts, void* stream) {
    TRY(check_batch(c, B));
    hipStream_t s = static_cast<hipStream_t>(stream);
    TRY(eval_front(c, x, B, p, s));
    if (c->dense) return launch_dense_conv_act(c, x, B, acts, s);
    TRY(launch_conv_act(c, B, acts, s));
    return EXPLAINN_OK;
}

extern "C" int explainn_stage_codes(explainn_ctx* c, const uint8_t* codes, int B,
                                    int reverse_complement, void* stream) {
    TRY(check_batch(c, B));
    if (!codes) { explainn_set_error("codes is null"); return EXPLAINN_E_ARG; }
    c->fwd_B = 0; c->tail_B = 0;       // the packed codes of a pending backward are overwritten
    return launch_pack_codes(c, codes, B, reverse_complement ? 1 : 0, static_cast<hipStream_t>(stream));
}

extern "C" int explainn_filter_act_max(explainn_ctx* c, const float* x, int B,
                                       const explainn_params* p, const uint8_t* select,
                                       float* unit_max, void* stream) {
    TRY(check_batch(c, B));
    if (!unit_max) { explainn_set_error("unit_max is null"); return EXPLAINN_E_ARG; }
    if (c->dense) { explainn_set_error("the filter export works on base codes: one-hot input only"); return EXPLAINN_E_UNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    TRY(eval_front(c, x, B, p, s));
    return launch_filter_act_max(c, B, select, unit_max, s);
}

extern "C" int explainn_filter_sites(explainn_ctx* c, const float* x, int B, const explainn_params* p,
                                     const uint8_t* select, const float* thresholds, int site_cap,
                                     int32_t* site_total, int32_t* pfm, uint8_t* hit, void* stream) {
    TRY(check_batch(c, B));
    if (!thresholds || !site_total || !pfm) {
        explainn_set_error("thresholds, site_total and pfm are required");
        return EXPLAINN_E_ARG;
    }
    if (site_cap <= 0) { explainn_set_error("site_cap must be positive"); return EXPLAINN_E_ARG; }
    if (c->dense) { explainn_set_error("the filter export works on base codes: one-hot input only"); return EXPLAINN_E_UNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    TRY(eval_front(c, x, B, p, s));
    return launch_filter_sites(c, B, select, thresholds, site_cap, site_total, pfm, hit, s);
}

extern "C" int explainn_forward_train(explainn_ctx* c, const float* x, int B,
                                      const explainn_params* p, const uint8_t* keep_mask,
                                      float dropout_p, uint64_t seed, float* logits, void* stream) {
    TRY(check_batch(c, B));
    if (B == 1) {
        // torch raises here (BatchNorm over one value); the reason for train.py:297-302
        explainn_set_error("Expected more than 1 value per channel when training, got input size "
                           "[1, %d, 1]", FC_H * c->U);
        return EXPLAINN_E_BATCH1;
    }
    if (dropout_p < 0.f || dropout_p >= 1.f) {
        explainn_set_error("dropout_p must be in [0,1)");
        return EXPLAINN_E_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    c->fwd_B = 0;
    c->eval_valid = false;             // the train-mode folds overwrite the eval-mode tables
    // the one-hot batch is packed and the filter tables are built by one launch; a staged batch of
    // base codes (x == NULL) is already packed and only needs the tables
    c->dense_x = nullptr;
    if (c->dense) {
        // soft input: no codes to pack; the stages that touch x take the dense kernels (dense.hip)
        if (!x) { explainn_set_error("dense input mode needs x"); return EXPLAINN_E_ARG; }
        c->staged_B = 0;
        STAGE(ST_PACK, launch_prep1_tables(c, p, s));
        STAGE(ST_MOMENTS, launch_dense_moments(c, x, B, s));
        STAGE(ST_PREP1, launch_prep1(c, p, B, true, s));
        STAGE(ST_CONV_POOL, launch_dense_conv_pool(c, x, p, B, s));
        c->dense_x = x;
    } else {
    if (x) STAGE(ST_PACK, launch_pack_tables(c, x, p, B, s));
    else {
        TRY(launch_pack(c, x, B, false, s));
        STAGE(ST_PACK, launch_prep1_tables(c, p, s));
    }
    // input moments (bit masks -> Gram -> BatchNorm1 fold), then the filter bank.  (Round 1 forked the
    // moment chain onto a side stream beside the filter bank; the filter bank fills every wave slot
    // of the chip, the two stretched each other, and in series -- now that the chain takes 5 + 4 us
    // -- the step is 4 us shorter.)
    STAGE(ST_MOMENTS, launch_moments(c, B, s));
    STAGE(ST_PREP1, launch_prep1(c, p, B, true, s));
    STAGE(ST_CONV_POOL, launch_conv_pool(c, p, B, true, s));
    }
    STAGE(ST_QMOM, launch_qmoments(c, B, s));
    STAGE(ST_PREP2, launch_prep2(c, p, B, true, s));
    STAGE(ST_FC_FWD, launch_fc_fwd(c, p, B, true, keep_mask, dropout_p, seed, s));
    STAGE(ST_HEAD_FWD, launch_head_fwd(c, p, B, true, logits, nullptr, s));
    c->fwd_B = B;
    return EXPLAINN_OK;
}

namespace {
// the backward after the head, in the two halves a data-parallel run overlaps its all-reduce with:
// after backward_fc every gradient from fc1_w to final_b (the tail of explainn_grads) is final;
// backward_conv then produces conv_w, conv_b, bn1_w, bn1_b
int backward_fc(explainn_ctx* c, int B, const explainn_params* p, const explainn_grads* g,
                const pa_head_args* head, hipStream_t s) {
    STAGE(ST_PASSA, launch_passA(c, B, head, s));
    STAGE(ST_MID, launch_mid_bwd(c, p, g, B, s));
    return EXPLAINN_OK;
}

int backward_conv(explainn_ctx* c, int B, const explainn_params* p, const explainn_grads* g,
                  int freeze_top_n_filters, hipStream_t s) {
    STAGE(ST_PASSB, launch_passB(c, B, s));
    if (c->dense_x) STAGE(ST_CONV_BWD, launch_dense_conv_bwd(c, c->dense_x, B, s));
    else STAGE(ST_CONV_BWD, launch_conv_bwd(c, B, s));
    STAGE(ST_FIN, launch_fin_bwd(c, p, g, B, freeze_top_n_filters, s));
    return EXPLAINN_OK;
}

// Few tasks and a small batch: the head backward runs inside passA (fc.hip; mode 1 = dlogits given,
// 2 = from the loss).  Every passA wave then makes the full-batch pass of BatchNorm3's backward
// itself: worth a launch (~5 us) up to a few hundred sequences -- 0.120 -> 0.115 ms per step at 100
// units x 64 sequences -- and a wash at 1024 (passA +9.6 us, head_bwd -9.6 us; MI355X), so larger
// batches keep the per-unit kernel.
bool head_rides_in_passA(const explainn_ctx* c, int B) { return c->T <= PA_HEAD_MAX_T && B <= 512; }

pa_head_args head_in_passA(explainn_ctx* c, const explainn_params* p, const explainn_grads* g, int mode,
                           const float* dl, int kind, const float* logits, const float* y,
                           float* loss_out) {
    pa_head_args h = {mode, c->T, kind, dl, logits, y, loss_out, p->final_w, p->bn3_w, c->o, c->zhat,
                      c->sig3, c->dz, g->final_w, g->final_b, g->bn3_w, g->bn3_b, g->fc2_b};
    return h;
}

int backward_tail(explainn_ctx* c, int B, const explainn_params* p, const explainn_grads* g,
                  int freeze_top_n_filters, const pa_head_args* head, hipStream_t s) {
    TRY(backward_fc(c, B, p, g, head, s));
    return backward_conv(c, B, p, g, freeze_top_n_filters, s);
}
}  // namespace

extern "C" int explainn_backward(explainn_ctx* c, const float* dlogits, int B,
                                 const explainn_params* p, const explainn_grads* g,
                                 int freeze_top_n_filters, void* stream) {
    TRY(check_batch(c, B));
    if (c->fwd_B != B) {
        explainn_set_error("backward(B=%d) without a matching train-mode forward (last B=%d)", B,
                           c->fwd_B);
        return EXPLAINN_E_STATE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (head_rides_in_passA(c, B)) {
        const pa_head_args h = head_in_passA(c, p, g, 1, dlogits, 0, nullptr, nullptr, nullptr);
        return backward_tail(c, B, p, g, freeze_top_n_filters, &h, s);
    }
    STAGE(ST_HEAD_BWD, launch_head_bwd(c, p, g, dlogits, B, s));
    return backward_tail(c, B, p, g, freeze_top_n_filters, nullptr, s);
}

extern "C" int explainn_loss_grad(explainn_ctx* c, int loss_kind, const float* logits,
                                  const float* targets, int B, float* loss_out, float* dlogits,
                                  void* stream) {
    TRY(check_batch(c, B));
    if (loss_kind != EXPLAINN_LOSS_BCE_WITH_LOGITS && loss_kind != EXPLAINN_LOSS_MSE) {
        explainn_set_error("unknown loss kind %d", loss_kind);
        return EXPLAINN_E_ARG;
    }
    return launch_loss(c, loss_kind, logits, targets, B, loss_out, dlogits,
                       static_cast<hipStream_t>(stream));
}

namespace {
int train_step_front(explainn_ctx* c, const float* x, const float* targets, int B,
                     const explainn_params* p, const explainn_grads* g, int loss_kind,
                     float dropout_p, uint64_t seed, float* logits, float* loss_out, void* stream) {
    if (loss_kind != EXPLAINN_LOSS_BCE_WITH_LOGITS && loss_kind != EXPLAINN_LOSS_MSE) {
        explainn_set_error("unknown loss kind %d", loss_kind);
        return EXPLAINN_E_ARG;
    }
    if (c) c->tail_B = 0;
    TRY(explainn_forward_train(c, x, B, p, nullptr, dropout_p, seed, logits, stream));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (head_rides_in_passA(c, B)) {
        // few tasks, small batch: loss, loss gradient and head backward all ride in passA's prologue
        const pa_head_args h = head_in_passA(c, p, g, 2, nullptr, loss_kind, logits, targets, loss_out);
        return backward_fc(c, B, p, g, &h, s);
    }
    if (c->T <= 4) {
        // few tasks: the loss gradient is recomputed inside the head backward (one launch less)
        STAGE(ST_HEAD_BWD, launch_head_bwd_fused_loss(c, p, g, loss_kind, logits, targets, loss_out, B, s));
    } else {
        STAGE(ST_LOSS, launch_loss_deferred(c, loss_kind, logits, targets, B, loss_out, c->dlogits, s));
        STAGE(ST_HEAD_BWD, launch_head_bwd(c, p, g, c->dlogits, B, s));
    }
    return backward_fc(c, B, p, g, nullptr, s);
}
}  // namespace

extern "C" int explainn_train_step(explainn_ctx* c, const float* x, const float* targets, int B,
                                   const explainn_params* p, const explainn_grads* g, int loss_kind,
                                   float dropout_p, uint64_t seed, int freeze_top_n_filters,
                                   float* logits, float* loss_out, void* stream) {
    TRY(train_step_front(c, x, targets, B, p, g, loss_kind, dropout_p, seed, logits, loss_out, stream));
    return backward_conv(c, B, p, g, freeze_top_n_filters, static_cast<hipStream_t>(stream));
}

extern "C" int explainn_train_step_fc(explainn_ctx* c, const float* x, const float* targets, int B,
                                      const explainn_params* p, const explainn_grads* g,
                                      int loss_kind, float dropout_p, uint64_t seed, float* logits,
                                      float* loss_out, void* stream) {
    TRY(train_step_front(c, x, targets, B, p, g, loss_kind, dropout_p, seed, logits, loss_out, stream));
    c->tail_B = B;
    return EXPLAINN_OK;
}

extern "C" int explainn_train_step_conv(explainn_ctx* c, int B, const explainn_params* p,
                                        const explainn_grads* g, int freeze_top_n_filters,
                                        void* stream) {
    TRY(check_batch(c, B));
    if (c->tail_B != B) {
        explainn_set_error("train_step_conv(B=%d) without a matching train_step_fc (pending B=%d)",
                           B, c->tail_B);
        return EXPLAINN_E_STATE;
    }
    c->tail_B = 0;
    return backward_conv(c, B, p, g, freeze_top_n_filters, static_cast<hipStream_t>(stream));
}

extern "C" int explainn_dense_input(explainn_ctx* c, int enable) {
    if (!c) { explainn_set_error("null context"); return EXPLAINN_E_ARG; }
    c->dense = enable != 0;
    c->fwd_B = 0; c->tail_B = 0;
    return EXPLAINN_OK;
}

extern "C" int explainn_stage_onehot(explainn_ctx* c, const float* x, int B, void* stream) {
    TRY(check_batch(c, B));
    if (!x) { explainn_set_error("x is null"); return EXPLAINN_E_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    c->fwd_B = 0; c->tail_B = 0;        // the packed codes of a pending backward are overwritten
    TRY(launch_pack(c, x, B, true, s));  // with the bit masks: the batch may feed a train forward
    c->staged_B = B;
    return EXPLAINN_OK;
}

extern "C" int explainn_stage_timing(explainn_ctx* c, int enable) {
    if (!c) { explainn_set_error("null context"); return EXPLAINN_E_ARG; }
    if (enable && !c->ev0[0]) {
        for (int i = 0; i < ST_COUNT; ++i) {
            HIP_TRY(hipEventCreate(&c->ev0[i]));
            HIP_TRY(hipEventCreate(&c->ev1[i]));
        }
    }
    c->timing = enable != 0;
    c->timed = 0;
    return EXPLAINN_OK;
}

extern "C" int explainn_stage_count(void) { return ST_COUNT; }
extern "C" const char* explainn_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

extern "C" int explainn_stage_times(explainn_ctx* c, float* us, int cap) {
    if (!c || !us || cap < ST_COUNT) { explainn_set_error("stage_times: need room for %d floats", ST_COUNT); return EXPLAINN_E_ARG; }
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < ST_COUNT; ++i) {
        us[i] = -1.f;                                   // stage did not run in the last step
        if (c->timed & (1u << i)) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
            us[i] = ms * 1e3f;
        }
    }
    c->timed = 0;
    return EXPLAINN_OK;
}

extern "C" int explainn_debug_keep_bits(explainn_ctx* c, int B, uint32_t* out, void* stream) {
    TRY(check_batch(c, B));
    if (!out) { explainn_set_error("out is null"); return EXPLAINN_E_ARG; }
    if (c->fwd_B != B) {
        explainn_set_error("keep_bits(B=%d) without a train-mode forward of that batch in flight (last B=%d)",
                           B, c->fwd_B);
        return EXPLAINN_E_STATE;
    }
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)B * sizeof(uint4), c->bits, (size_t)c->Bs * sizeof(uint4),
                             (size_t)B * sizeof(uint4), (size_t)c->U, hipMemcpyDeviceToDevice,
                             static_cast<hipStream_t>(stream)));
    return EXPLAINN_OK;
}

extern "C" int explainn_input_flags(explainn_ctx* c, int* flags_host, void* stream) {
    if (!c || !flags_host) { explainn_set_error("null argument"); return EXPLAINN_E_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(flags_host, c->flags, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(c->flags, 0, sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));
    return EXPLAINN_OK;
}
