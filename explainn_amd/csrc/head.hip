// The narrow end of the network: BatchNorm3 + ReLU over the batch (architectures/__init__.py:99-100),
// the final nn.Linear(U,T) combiner (:104), the loss (:446-456) and their backward.  All arrays
// here are (U x B) or (B x T) -- a few MB at most -- so these are plain reduction kernels.
#include "common.h"

__device__ __forceinline__ double block_sum_256(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one block per unit: batch statistics of z, then zhat / o
__global__ __launch_bounds__(256) void head_fwd_train_kernel(
    const float* __restrict__ z, const float* __restrict__ c2, const float* __restrict__ g3,
    const float* __restrict__ b3, float* __restrict__ rm3, float* __restrict__ rv3, int64_t* nbt,
    float* __restrict__ zhat, float* __restrict__ o, float* __restrict__ sig3, int Bs, int B) {
    __shared__ double red[4];
    const int u = blockIdx.x, tid = threadIdx.x;
    const float* zu = z + (size_t)u * Bs;
    double s = 0;
    for (int b = tid; b < B; b += 256) s += (double)zu[b];
    const double mean = block_sum_256(s, red) / (double)B;
    double v = 0;
    for (int b = tid; b < B; b += 256) { const double d = (double)zu[b] - mean; v = fma(d, d, v); }
    double var = block_sum_256(v, red) / (double)B;
    const double sg = sqrt(var + BN_EPS_D);
    const float meanf = (float)mean, isg = (float)(1.0 / sg), gam = g3[u], bet = b3[u];
    for (int b = tid; b < B; b += 256) {
        const float zh = (zu[b] - meanf) * isg;
        zhat[(size_t)u * Bs + b] = zh;
        o[(size_t)u * Bs + b] = fmaxf(fmaf(gam, zh, bet), 0.f);
    }
    if (tid == 0) {
        sig3[u] = (float)sg;
        rm3[u] = (float)((1 - BN_MOM_D) * (double)rm3[u] + BN_MOM_D * (mean + (double)c2[u]));
        rv3[u] = (float)((1 - BN_MOM_D) * (double)rv3[u] + BN_MOM_D * var * (double)B / (double)(B - 1));
        if (u == 0 && nbt) *nbt += 1;
    }
}

// logits[b][t] = bf[t] + sum_u Wf[t][u] * o[u][b]; one block per (64 sequences, task): its 16
// waves each sum a slice of the units, then the slices are added in fixed order
__global__ __launch_bounds__(1024) void logits_kernel(const float* __restrict__ o,
                                                      const float* __restrict__ Wf,
                                                      const float* __restrict__ bf,
                                                      float* __restrict__ logits, int U, int T,
                                                      int Bs, int B) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane, t = blockIdx.y;
    const float* wr = Wf + (size_t)t * U;
    float acc = 0.f;
    for (int u0 = wv; u0 < U; u0 += 64) {              // four units (eight loads) in flight
        float wq[4], oq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = min(u0 + 16 * q, U - 1);
            wq[q] = wr[u];
            oq[q] = o[(size_t)u * Bs + b];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { KEEP(wq[q]); KEEP(oq[q]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = fmaf(u0 + 16 * q < U ? wq[q] : 0.f, oq[q], acc);
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && b < B) {
        float s = bf[t];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += part[i][lane];
        logits[(size_t)b * T + t] = s;
    }
}

// outs[b][u] = o[u][b]   (model.linears(x) output layout, test.py:151)
__global__ __launch_bounds__(256) void outs_kernel(const float* __restrict__ o,
                                                   float* __restrict__ outs, int U, int Bs, int B) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= B * U) return;
    const int u = gid / B, b = gid % B;
    outs[(size_t)b * U + u] = o[(size_t)u * Bs + b];
}

int launch_head_fwd(explainn_ctx* c, const explainn_params* p, int B, bool train, float* logits,
                    float* outs, hipStream_t s) {
    if (train) {
        hipLaunchKernelGGL(head_fwd_train_kernel, dim3(c->U), dim3(256), 0, s, c->z, p->fc2_b,
                           p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv, p->bn3_nbt, c->zhat, c->o,
                           c->sig3, c->Bs, B);
        LAUNCH_CHECK();
    }
    if (logits) {
        hipLaunchKernelGGL(logits_kernel, dim3((B + 63) / 64, c->T), dim3(1024), 0, s, c->o,
                           p->final_w, p->final_b, logits, c->U, c->T, c->Bs, B);
        LAUNCH_CHECK();
    }
    if (outs) {
        hipLaunchKernelGGL(outs_kernel, dim3((B * c->U + 255) / 256), dim3(256), 0, s, c->o, outs,
                           c->U, c->Bs, B);
        LAUNCH_CHECK();
    }
    return EXPLAINN_OK;
}

// loss + dlogits, one block (B*T is at most ~1e6); fixed reduction order -> deterministic
__global__ __launch_bounds__(1024) void loss_kernel(const float* __restrict__ logits,
                                                    const float* __restrict__ y, int kind, int N,
                                                    float* __restrict__ loss,
                                                    float* __restrict__ dlogits) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const float invN = 1.0f / (float)N;
    double acc = 0;
    for (int i = tid; i < N; i += 1024) {
        const float x = logits[i], t = y[i];
        float l, d;
        if (kind == EXPLAINN_LOSS_BCE_WITH_LOGITS) {
            l = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            d = (1.0f / (1.0f + expf(-x)) - t) * invN;
        } else {
            const float e = x - t;
            l = e * e;
            d = 2.0f * e * invN;
        }
        acc += (double)l;
        dlogits[i] = d;
    }
    acc = wave_sum_d(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0;
        for (int i = 0; i < 16; ++i) s += red[i];
        *loss = (float)(s / (double)N);
    }
}

int launch_loss(explainn_ctx* c, int kind, const float* logits, const float* y, int B, float* loss,
                float* dlogits, hipStream_t s) {
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, s, logits, y, kind, B * c->T, loss,
                       dlogits);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// d loss / d logit for one (sequence, task): either read from `dl`, or -- FUSED, used by
// explainn_train_step when T is small -- recomputed from logits and targets so that the separate
// loss launch (and its ~4 us floor) disappears; unit 0's block then also reduces the loss value.
template <bool FUSED>
__device__ __forceinline__ float dl_at(const float* __restrict__ dl, const float* __restrict__ logits,
                                       const float* __restrict__ y, int kind, float invN, int i) {
    if (!FUSED) return dl[i];
    const float x = logits[i], t = y[i];
    if (kind == EXPLAINN_LOSS_BCE_WITH_LOGITS) return (1.0f / (1.0f + expf(-x)) - t) * invN;
    return 2.0f * (x - t) * invN;
}

// one block per unit: final-layer gradients, BN3 backward -> dz[u][b]
template <bool FUSED>
__global__ __launch_bounds__(256) void head_bwd_kernel(
    const float* __restrict__ dl, const float* __restrict__ logits, const float* __restrict__ y,
    int kind, float* __restrict__ loss_out, const float* __restrict__ Wf,
    const float* __restrict__ g3, const float* __restrict__ o, const float* __restrict__ zhat,
    const float* __restrict__ sig3, float* __restrict__ dz, float* __restrict__ gWf,
    float* __restrict__ gbf, float* __restrict__ gg3, float* __restrict__ gb3,
    float* __restrict__ gc2, int U, int T, int Bs, int B) {
    __shared__ double red[4];
    const int u = blockIdx.x, tid = threadIdx.x;
    const float* ou = o + (size_t)u * Bs;
    const float* zh = zhat + (size_t)u * Bs;
    float* dzu = dz + (size_t)u * Bs;
    const float invN = 1.0f / (float)(B * T);
    double s1 = 0, s2 = 0;
    for (int b = tid; b < B; b += 256) {
        float dob = 0.f;
        for (int t = 0; t < T; ++t)
            dob = fmaf(dl_at<FUSED>(dl, logits, y, kind, invN, b * T + t), Wf[(size_t)t * U + u], dob);
        const float d3 = ou[b] > 0.f ? dob : 0.f;
        dzu[b] = d3;
        s1 += (double)d3;
        s2 = fma((double)d3, (double)zh[b], s2);
    }
    const double S1 = block_sum_256(s1, red);
    const double S2 = block_sum_256(s2, red);
    const float m1 = (float)(S1 / (double)B), m2 = (float)(S2 / (double)B);
    const float sc = g3[u] / sig3[u];
    for (int b = tid; b < B; b += 256) dzu[b] = sc * (dzu[b] - m1 - zh[b] * m2);
    if (tid == 0) { gg3[u] = (float)S2; gb3[u] = (float)S1; gc2[u] = 0.f; }
    for (int t = 0; t < T; ++t) {
        double a = 0;
        for (int b = tid; b < B; b += 256)
            a = fma((double)dl_at<FUSED>(dl, logits, y, kind, invN, b * T + t), (double)ou[b], a);
        const double tot = block_sum_256(a, red);
        if (tid == 0) gWf[(size_t)t * U + u] = (float)tot;
        if (u == 0) {
            double c = 0;
            for (int b = tid; b < B; b += 256) c += (double)dl_at<FUSED>(dl, logits, y, kind, invN, b * T + t);
            const double ct = block_sum_256(c, red);
            if (tid == 0) gbf[t] = (float)ct;
        }
    }
    if (FUSED && u == 0) {
        double acc = 0;
        for (int i = tid; i < B * T; i += 256) {
            const float x = logits[i], t = y[i];
            float l;
            if (kind == EXPLAINN_LOSS_BCE_WITH_LOGITS) l = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            else { const float e = x - t; l = e * e; }
            acc += (double)l;
        }
        const double tot = block_sum_256(acc, red);
        if (tid == 0) *loss_out = (float)(tot / (double)(B * T));
    }
}

int launch_head_bwd(explainn_ctx* c, const explainn_params* p, const explainn_grads* g,
                    const float* dlogits, int B, hipStream_t s) {
    hipLaunchKernelGGL(head_bwd_kernel<false>, dim3(c->U), dim3(256), 0, s, dlogits,
                       (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr, p->final_w,
                       p->bn3_w, c->o, c->zhat, c->sig3, c->dz, g->final_w, g->final_b, g->bn3_w,
                       g->bn3_b, g->fc2_b, c->U, c->T, c->Bs, B);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// loss + its gradient folded into the head backward (explainn_train_step, small T)
int launch_head_bwd_fused_loss(explainn_ctx* c, const explainn_params* p, const explainn_grads* g,
                               int kind, const float* logits, const float* y, float* loss_out, int B,
                               hipStream_t s) {
    hipLaunchKernelGGL(head_bwd_kernel<true>, dim3(c->U), dim3(256), 0, s, (const float*)nullptr,
                       logits, y, kind, loss_out, p->final_w, p->bn3_w, c->o, c->zhat, c->sig3, c->dz,
                       g->final_w, g->final_b, g->bn3_w, g->bn3_b, g->fc2_b, c->U, c->T, c->Bs, B);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}
