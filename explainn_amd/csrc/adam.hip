// Adam over all parameter tensors in ONE launch (SURVEY.md 8a row a9 / 8b "optional adam_step"):
// the reference's optimiser is torch.optim.Adam(params, lr) (architectures/__init__.py:463-464),
// whose step on 14 small tensors is a string of foreach launches.  Same update rule, same order of
// operations as torch's (lerp for exp_avg, mul+addcmul for exp_avg_sq, sqrt/bias-correction/eps
// for the denominator, addcdiv into the parameter); no amsgrad, no weight decay (the defaults the
// reference uses).  Stand-alone: needs no explainn_ctx.
#include "common.h"

namespace {

constexpr int ADAM_MAX_TENSORS = 32;
constexpr int ADAM_T = 256;
constexpr int ADAM_CHUNK = ADAM_T * 8;          // elements per block

struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];      // prefix of per-tensor block counts
    int count;
};

__global__ __launch_bounds__(ADAM_T) void adam_kernel(AdamTable t, float omb1, float beta2,
                                                      float omb2, float eps, float step_size,
                                                      float bc2_sqrt) {
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;   // block-uniform
    const long long base = (long long)((int)blockIdx.x - t.first_block[ti]) * ADAM_CHUNK;
    float* __restrict__ p = t.p[ti];
    const float* __restrict__ g = t.g[ti];
    float* __restrict__ m = t.m[ti];
    float* __restrict__ v = t.v[ti];
    const long long n = t.n[ti];
#pragma unroll
    for (int i = 0; i < ADAM_CHUNK / ADAM_T; ++i) {
        const long long e = base + (long long)i * ADAM_T + threadIdx.x;
        if (e < n) {
            const float ge = g[e];
            const float me = m[e] + (ge - m[e]) * omb1;
            const float ve = v[e] * beta2 + omb2 * ge * ge;
            const float denom = sqrtf(ve) / bc2_sqrt + eps;
            m[e] = me;
            v[e] = ve;
            p[e] = p[e] - step_size * (me / denom);
        }
    }
}

}  // namespace

extern "C" int explainn_adam_step(int n_tensors, float* const* params, const float* const* grads,
                                  float* const* exp_avg, float* const* exp_avg_sq,
                                  const int64_t* sizes, int64_t step, double lr, double beta1,
                                  double beta2, double eps, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !sizes))) {
        explainn_set_error("adam_step: null argument");
        return EXPLAINN_E_ARG;
    }
    if (step < 1) { explainn_set_error("adam_step: step counts from 1"); return EXPLAINN_E_ARG; }
    // bias corrections in double on the host, as torch's single-tensor Adam computes them
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    // 1 - beta in double, then rounded: torch passes these weights as doubles to lerp/addcmul
    // (1.f - 0.999f is off by 5e-5 relative)
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    for (int i = 0; i < n_tensors;) {
        // one launch per table of up to ADAM_MAX_TENSORS non-empty tensors; `i` carries on from the
        // last tensor the previous table consumed (empty tensors are skipped without a slot)
        AdamTable tab;
        tab.count = 0;
        tab.first_block[0] = 0;
        for (; i < n_tensors && tab.count < ADAM_MAX_TENSORS; ++i) {
            if (sizes[i] < 0 || (sizes[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]))) {
                explainn_set_error("adam_step: tensor %d has a null pointer or negative size", i);
                return EXPLAINN_E_ARG;
            }
            if (sizes[i] == 0) continue;
            const int c = tab.count++;
            tab.p[c] = params[i]; tab.g[c] = grads[i]; tab.m[c] = exp_avg[i]; tab.v[c] = exp_avg_sq[i];
            tab.n[c] = sizes[i];
            tab.first_block[c + 1] = tab.first_block[c] + (int)((sizes[i] + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        if (tab.count == 0) continue;
        hipLaunchKernelGGL(adam_kernel, dim3(tab.first_block[tab.count]), dim3(ADAM_T), 0,
                           static_cast<hipStream_t>(stream), tab, omb1, (float)beta2, omb2, (float)eps, step_size,
                           bc2_sqrt);
        LAUNCH_CHECK();
    }
    return EXPLAINN_OK;
}
