// PWM scan: the reference's `PWM` module (architectures/__init__.py:116-170, used
// by pwm-scoring.py:81-105) -- a frozen bank of G position weight matrices slid over both strands of
// every sequence, reduced to one score per (sequence, PWM): the maximum, or the sum, over all
// 2*(L-k+1) window scores.  Stand-alone: needs no explainn_ctx, only device pointers.
// Block = (sequence, four PWMs); the sequence's 4 x L one-hot tile and the four matrices sit in LDS,
// threads walk the window starts.  The tile is kept as fp32 (not base codes): the reference module
// is a plain convolution and also accepts soft inputs.
#include "common.h"

namespace {

constexpr int PWM_T = 256;

__global__ __launch_bounds__(PWM_T) void pwm_scan_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ pwms,
                                                         float* __restrict__ scores, int L, int G,
                                                         int k, int use_max) {
    extern __shared__ float psm[];             // xs[4][L] | W[4 pwms][4][k]
    float* xs = psm;
    float* W = psm + 4 * L;
    __shared__ float red[4][PWM_T / 64];
    const int b = blockIdx.x, quad = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 4 * L; i += PWM_T) xs[i] = x[(size_t)b * 4 * L + i];
    for (int i = tid; i < 4 * 4 * k; i += PWM_T) {
        const int g = quad * 4 + i / (4 * k);
        W[i] = g < G ? pwms[(size_t)g * 4 * k + (i % (4 * k))] : 0.f;
    }
    __syncthreads();
    const int Lo = L - k + 1;
    float acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = use_max ? -INFINITY : 0.f;
    // window starts 0..Lo-1 on the forward strand, Lo..2Lo-1 on the reverse complement
    // (x_rev[a][p] = x[3-a][L-1-p], architectures/__init__.py:159)
    for (int w = tid; w < 2 * Lo; w += PWM_T) {
        const bool rev = w >= Lo;
        const int p = rev ? w - Lo : w;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < k; ++j) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float xv = rev ? xs[(3 - a) * L + (L - 1 - p - j)] : xs[a * L + p + j];
#pragma unroll
                for (int g = 0; g < 4; ++g) s[g] = fmaf(W[(g * 4 + a) * k + j], xv, s[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = use_max ? fmaxf(acc[g], s[g]) : acc[g] + s[g];
    }
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v = acc[g];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float t = __shfl_xor(v, o, 64);
            v = use_max ? fmaxf(v, t) : v + t;
        }
        if (lane == 0) red[g][wave] = v;
    }
    __syncthreads();
    if (tid < 4 && quad * 4 + tid < G) {
        float v = red[tid][0];
        for (int w = 1; w < PWM_T / 64; ++w) v = use_max ? fmaxf(v, red[tid][w]) : v + red[tid][w];
        scores[(size_t)b * G + quad * 4 + tid] = v;
    }
}

}  // namespace

extern "C" int explainn_pwm_scan(const float* x, int B, int L, const float* pwms, int G, int k,
                                 int scoring, float* scores, void* stream) {
    if (!x || !pwms || !scores || B < 1 || G < 1 || k < 1 || L < k) {
        explainn_set_error("pwm_scan: need B,G,k >= 1 and sequence_length >= kernel_size "
                           "(B=%d G=%d k=%d L=%d)", B, G, k, L);
        return EXPLAINN_E_ARG;
    }
    if (scoring != EXPLAINN_PWM_SUM && scoring != EXPLAINN_PWM_MAX) {
        explainn_set_error("pwm_scan: scoring must be EXPLAINN_PWM_SUM or EXPLAINN_PWM_MAX");
        return EXPLAINN_E_ARG;
    }
    const size_t sm = ((size_t)4 * L + 16 * k) * sizeof(float);
    if (sm > 48 * 1024) {
        explainn_set_error("pwm_scan: sequence_length %d / kernel_size %d exceed the LDS tile", L, k);
        return EXPLAINN_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(pwm_scan_kernel, dim3(B, (G + 3) / 4), dim3(PWM_T), sm,
                       static_cast<hipStream_t>(stream), x, pwms, scores, L, G, k,
                       scoring == EXPLAINN_PWM_MAX ? 1 : 0);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}
