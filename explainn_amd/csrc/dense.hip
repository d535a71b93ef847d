// Fallback for inputs that are NOT one-hot.  The reference's forward accepts any float tensor
// (architectures/__init__.py:111 hands x straight to a grouped Conv1d); everything in this library
// that makes the filter bank fast -- base codes, one-hot bit operands, integer pair counts -- needs
// one-hot columns.  A "soft" input (position-probability matrices, a blend of sequences) therefore
// takes these four plain kernels for the stages that touch x, and the regular pipeline for
// everything behind the pooled activations:
//
//   dense_moments    G[(a,j),(a',j')] = mean_{b,p} x[b,a,p+j] x[b,a',p+j'],  m = mean_{b,p} x[b,a,p+j]
//                    (for one-hot x these are the pair counts of pack.hip; here real-valued, fp64)
//   dense_conv_pool  conv[b,u,p] = sum_{a,j} W[u,a,j] x[b,a,p+j]; sign-aware MaxPool1d(7,7) -> ext, idx
//   dense_conv_bwd   Dspp[u][a,j] += dy[b,u,w] x[b,a,7w+idx+j]      (the sparse filter-gradient term)
//   dense_conv_act   model.linears[:3]: exp(alpha*conv + shift) per position
//
// BatchNorm1's closed forms (prep1_stats, fin_bwd) are algebra on the window vectors f[b,p] and hold
// for any x.  Correct, not fast: this path is an escape hatch, the benchmarks never take it.
#include "common.h"

// one block per (gap d, row pair (a, a')): thread <-> position q, fp64 sums over the batch
__global__ __launch_bounds__(256) void dense_moments_kernel(const float* __restrict__ x,
                                                            double* __restrict__ G,
                                                            double* __restrict__ m, int B, int L,
                                                            int k) {
    extern __shared__ double dq[];                     // [L] products summed over b, then 4 partial sums
    const int d = blockIdx.x >> 4, a = (blockIdx.x >> 2) & 3, a2 = blockIdx.x & 3;
    const int tid = threadIdx.x, Lo = L - k + 1, K4 = 4 * k;
    double part = 0;
    for (int q = tid; q < L; q += 256) {
        double s = 0;
        if (q + d < L) {
            const float* r0 = x + (size_t)a * L + q;
            const float* r1 = x + (size_t)a2 * L + q + d;
            for (int b0 = 0; b0 < B; b0 += 8) {
                float u0[8], u1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const size_t o = (size_t)min(b0 + i, B - 1) * 4 * L;
                    u0[i] = r0[o]; u1[i] = r1[o];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) s += (b0 + i < B) ? (double)u0[i] * (double)u1[i] : 0.0;
            }
        }
        dq[q] = s;
        if (q < Lo) part += s;
    }
    double* red = dq + L;
    part = wave_sum_d(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const double T0 = red[0] + red[1] + red[2] + red[3];
    if (tid < k - d) {
        const int j = tid;
        double sj = T0;
        for (int q = 0; q < j; ++q) sj += dq[Lo + q] - dq[q];
        const double v = sj / ((double)B * (double)Lo);
        const int row = a * k + j, col = a2 * k + j + d;
        G[(size_t)row * K4 + col] = v;
        G[(size_t)col * K4 + row] = v;
    }
    // m[(a,j)] = mean window indicator = mean_{b} sum_{q=j}^{j+Lo-1} x[b,a,q] / Lo: blocks (0,a,a)
    if (d == 0 && a == a2) {
        __syncthreads();
        double s = 0;
        for (int q = tid; q < L; q += 256) {
            double sq = 0;
            for (int b0 = 0; b0 < B; b0 += 8) {
                float u0[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) u0[i] = x[((size_t)min(b0 + i, B - 1) * 4 + a) * L + q];
#pragma unroll
                for (int i = 0; i < 8; ++i) sq += (b0 + i < B) ? (double)u0[i] : 0.0;
            }
            dq[q] = sq;
            if (q < Lo) s += sq;
        }
        s = wave_sum_d(s);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        const double M0 = red[0] + red[1] + red[2] + red[3];
        if (tid < k) {
            double sj = M0;
            for (int q = 0; q < tid; ++q) sj += dq[Lo + q] - dq[q];
            m[a * k + tid] = sj / ((double)B * (double)Lo);
        }
    }
}

// one block per (sequence, unit quad): threads over positions, conv sums through LDS to the pooling
__global__ __launch_bounds__(256) void dense_conv_pool_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ Wt,
                                                              const float* __restrict__ gamma1,
                                                              float* __restrict__ ext,
                                                              uint8_t* __restrict__ idx, int U, int k,
                                                              int L, int Lo, int n, int Bs) {
    extern __shared__ float4 dsm[];            // Wsm [k][5] float4 | xs [4][L] | conv [4][Lo]
    float4* Wsm = dsm;
    float* xs = reinterpret_cast<float*>(Wsm + k * 5);
    float* cv = xs + 4 * L;
    const int b = blockIdx.x, quad = blockIdx.y;
    const float4* src = reinterpret_cast<const float4*>(Wt) + (size_t)quad * k * 5;
    for (int i = threadIdx.x; i < k * 5; i += 256) Wsm[i] = src[i];
    for (int i = threadIdx.x; i < 4 * L; i += 256) xs[i] = x[(size_t)b * 4 * L + i];
    __syncthreads();
    for (int p = threadIdx.x; p < Lo; p += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float xv = xs[a * L + p + j];
                const float4 w = Wsm[j * 5 + a];
                acc.x = fmaf(w.x, xv, acc.x); acc.y = fmaf(w.y, xv, acc.y);
                acc.z = fmaf(w.z, xv, acc.z); acc.w = fmaf(w.w, xv, acc.w);
            }
        cv[p] = acc.x; cv[Lo + p] = acc.y; cv[2 * Lo + p] = acc.z; cv[3 * Lo + p] = acc.w;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * n; e += 256) {
        const int uu = e / n, w = e % n, u = quad * 4 + uu;
        const float sg = (u < U && gamma1[u] < 0.f) ? -1.f : 1.f;
        const float* c = cv + uu * Lo + POOLW * w;
        float best = sg * c[0];
        int bi = 0;
#pragma unroll
        for (int i = 1; i < POOLW; ++i) {
            const float v = sg * c[i];
            if (v > best) { best = v; bi = i; }            // strict: first index wins ties
        }
        const size_t o = ((size_t)u * n + w) * Bs + b;
        ext[o] = sg * best; idx[o] = (uint8_t)bi;
    }
}

// one block per (unit, CB2 x 64 sequences); thread <-> filter entry (a, j); the dy / argmax rows of
// a pooling window are staged through LDS, the x values come straight from global memory
#define DENSE_BWD_SEQS 128                     // sequences per dense_conv_bwd block (one partial each)
__global__ __launch_bounds__(128) void dense_conv_bwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ dy,
                                                             const uint8_t* __restrict__ idx,
                                                             float* __restrict__ Dspp, int k, int L,
                                                             int n, int Bs, int B) {
    __shared__ float sdy[DENSE_BWD_SEQS];
    __shared__ int sps[DENSE_BWD_SEQS];
    const int u = blockIdx.y, b0 = blockIdx.x * DENSE_BWD_SEQS, tid = threadIdx.x;
    const int K4 = 4 * k, a = tid / k, j = tid % k;
    const bool mine = tid < K4;
    float acc = 0.f;
    for (int w = 0; w < n; ++w) {
        __syncthreads();
        for (int i = tid; i < DENSE_BWD_SEQS; i += 128) {
            const int b = b0 + i;
            const size_t o = ((size_t)u * n + w) * Bs + min(b, Bs - 1);
            sdy[i] = (b < B) ? dy[o] : 0.f;
            sps[i] = POOLW * w + (int)idx[o];
        }
        __syncthreads();
        if (mine) {
            for (int i0 = 0; i0 < DENSE_BWD_SEQS; i0 += 8) {
                float xv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    xv[q] = x[((size_t)min(b0 + i0 + q, B - 1) * 4 + a) * L + sps[i0 + q] + j];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = fmaf(sdy[i0 + q], xv[q], acc);
            }
        }
    }
    if (mine) Dspp[((size_t)u * (Bs / 4) + blockIdx.x) * K4 + tid] = acc;
}

__global__ __launch_bounds__(256) void dense_conv_act_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ Wt,
                                                             const float* __restrict__ alpha,
                                                             const float* __restrict__ shift,
                                                             float* __restrict__ acts, int U, int k,
                                                             int L, int Lo) {
    extern __shared__ float4 asm_[];           // Wsm [k][5] float4 | xs [4][L]
    float4* Wsm = asm_;
    float* xs = reinterpret_cast<float*>(Wsm + k * 5);
    const int b = blockIdx.x, quad = blockIdx.y;
    const float4* src = reinterpret_cast<const float4*>(Wt) + (size_t)quad * k * 5;
    for (int i = threadIdx.x; i < k * 5; i += 256) Wsm[i] = src[i];
    for (int i = threadIdx.x; i < 4 * L; i += 256) xs[i] = x[(size_t)b * 4 * L + i];
    __syncthreads();
    for (int p = threadIdx.x; p < Lo; p += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float xv = xs[a * L + p + j];
                const float4 w = Wsm[j * 5 + a];
                acc.x = fmaf(w.x, xv, acc.x); acc.y = fmaf(w.y, xv, acc.y);
                acc.z = fmaf(w.z, xv, acc.z); acc.w = fmaf(w.w, xv, acc.w);
            }
        const float av[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int u = quad * 4 + uu;
            if (u < U) acts[((size_t)b * U + u) * Lo + p] = qval(alpha[u], av[uu], shift[u]);
        }
    }
}

int launch_dense_moments(explainn_ctx* c, const float* x, int B, hipStream_t s) {
    const size_t sm = (size_t)(c->L + 4) * sizeof(double);
    hipLaunchKernelGGL(dense_moments_kernel, dim3(c->k * 16), dim3(256), sm, s, x, c->G, c->m, B, c->L,
                       c->k);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int launch_dense_conv_pool(explainn_ctx* c, const float* x, const explainn_params* p, int B,
                           hipStream_t s) {
    const size_t sm = (size_t)c->k * 5 * sizeof(float4) + (size_t)(4 * c->L + 4 * c->Lo) * sizeof(float);
    hipLaunchKernelGGL(dense_conv_pool_kernel, dim3(B, c->Uq), dim3(256), sm, s, x, c->Wt, p->bn1_w,
                       c->ext, c->idx, c->U, c->k, c->L, c->Lo, c->n, c->Bs);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int launch_dense_conv_bwd(explainn_ctx* c, const float* x, int B, hipStream_t s) {
    // one partial per 128 sequences, like conv_bwd's (two 64-sequence tiles per wave): fin_bwd sums them
    hipLaunchKernelGGL(dense_conv_bwd_kernel, dim3((B + DENSE_BWD_SEQS - 1) / DENSE_BWD_SEQS, c->U),
                       dim3(128), 0, s, x, c->dy, c->idx, c->Dspp, c->k, c->L, c->n, c->Bs, B);
    LAUNCH_CHECK();
    c->dsp_stride = c->Bs / 4; c->dsp_count = (B + DENSE_BWD_SEQS - 1) / DENSE_BWD_SEQS;
    return EXPLAINN_OK;
}

int launch_dense_conv_act(explainn_ctx* c, const float* x, int B, float* acts, hipStream_t s) {
    const size_t sm = (size_t)c->k * 5 * sizeof(float4) + (size_t)4 * c->L * sizeof(float);
    hipLaunchKernelGGL(dense_conv_act_kernel, dim3(B, c->Uq), dim3(256), sm, s, x, c->Wt, c->alpha,
                       c->shift, acts, c->U, c->k, c->L, c->Lo);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}
