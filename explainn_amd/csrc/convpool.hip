// The filter bank: Conv1d(4U->U, k, groups=U) + BatchNorm1 + exp + MaxPool1d(7,7)
// (architectures/__init__.py:73-81) as ONE gather kernel that never materialises the conv output.
//
// One-hot input makes the convolution a gather: conv[b,u,p] = sum_j W[u, s[b,p+j], j].  A lane owns
// one sequence and four units: the taps of a unit quad sit in LDS as W[j][code] -> float4 (code 4 =
// N = zeros), so one ds_read_b128 at byte offset (j*80 + code*16) feeds four accumulators; lanes
// that share a code hit the same address (broadcast) and the five codes of a tap occupy twenty
// consecutive banks, so the reads are conflict-free by construction.
// BatchNorm+exp are monotone per unit, so the 7-wide max-pool runs on the raw gather sums with the
// sign of alpha = gamma1/sigma1 choosing max or min; only the pooled extreme (and its offset, for
// the backward routing) leaves the kernel: ext[u][w][b], idx[u][w][b].
#include "common.h"

template <int K>
__global__ __launch_bounds__(64) void conv_pool_kernel(const uint8_t* __restrict__ codesT,
                                                       const float* __restrict__ Wt,
                                                       const float* __restrict__ alpha,
                                                       float* __restrict__ ext,
                                                       uint8_t* __restrict__ idx, int n, int Bs) {
    __shared__ float4 W[K * 5];
    const int quad = blockIdx.y, lane = threadIdx.x;
    const int b = blockIdx.x * 64 + lane;
    const float4* src = reinterpret_cast<const float4*>(Wt) + (size_t)quad * K * 5;
    for (int i = lane; i < K * 5; i += 64) W[i] = src[i];
    float sgn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sgn[i] = alpha[quad * 4 + i] >= 0.f ? 1.f : -1.f;
    __syncthreads();
    constexpr int WIN = POOLW + K - 1;
    int coff[WIN];
    const uint8_t* cp = codesT + b;
#pragma unroll
    for (int i = 0; i < K - 1; ++i) coff[i] = (int)cp[(size_t)i * Bs] * 16;
    const char* Wb = reinterpret_cast<const char*>(W);
    for (int w = 0; w < n; ++w) {
#pragma unroll
        for (int i = 0; i < POOLW; ++i)
            coff[K - 1 + i] = (int)cp[(size_t)(POOLW * w + K - 1 + i) * Bs] * 16;
        float4 acc[POOLW];
#pragma unroll
        for (int i = 0; i < POOLW; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < K; ++j) {
#pragma unroll
            for (int i = 0; i < POOLW; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(Wb + j * 80 + coff[i + j]);
                acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
            }
        }
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const float sg = sgn[uu];
            float best = sg * (uu == 0 ? acc[0].x : uu == 1 ? acc[0].y : uu == 2 ? acc[0].z : acc[0].w);
            int bi = 0;
#pragma unroll
            for (int i = 1; i < POOLW; ++i) {
                const float v = sg * (uu == 0 ? acc[i].x : uu == 1 ? acc[i].y : uu == 2 ? acc[i].z : acc[i].w);
                if (v > best) { best = v; bi = i; }          // strict: first index wins ties
            }
            const size_t o = ((size_t)(quad * 4 + uu) * n + w) * Bs + b;
            ext[o] = sg * best;
            idx[o] = (uint8_t)bi;
        }
#pragma unroll
        for (int i = 0; i < K - 1; ++i) coff[i] = coff[i + POOLW];
    }
}

#define K_DISPATCH(Kv, CALL)                                                                   \
    switch (Kv) {                                                                              \
        case 2: { CALL(2); } break;   case 3: { CALL(3); } break;   case 4: { CALL(4); } break;   \
        case 5: { CALL(5); } break;   case 6: { CALL(6); } break;   case 7: { CALL(7); } break;   \
        case 8: { CALL(8); } break;   case 9: { CALL(9); } break;   case 10: { CALL(10); } break; \
        case 11: { CALL(11); } break; case 12: { CALL(12); } break; case 13: { CALL(13); } break; \
        case 14: { CALL(14); } break; case 15: { CALL(15); } break; case 16: { CALL(16); } break; \
        case 17: { CALL(17); } break; case 18: { CALL(18); } break; case 19: { CALL(19); } break; \
        case 20: { CALL(20); } break; case 21: { CALL(21); } break; case 22: { CALL(22); } break; \
        case 23: { CALL(23); } break; case 24: { CALL(24); } break; case 25: { CALL(25); } break; \
        case 26: { CALL(26); } break; case 27: { CALL(27); } break; case 28: { CALL(28); } break; \
        case 29: { CALL(29); } break; case 30: { CALL(30); } break; case 31: { CALL(31); } break; \
        case 32: { CALL(32); } break;                                                          \
        default: explainn_set_error("kernel_size %d not instantiated (2..32)", Kv);            \
                 return EXPLAINN_E_UNSUPPORTED;                                                \
    }

int launch_conv_pool(explainn_ctx* c, int B, hipStream_t s) {
    const dim3 grid((B + 63) / 64, c->Uq);
#define CALL(KK)                                                                             \
    hipLaunchKernelGGL(conv_pool_kernel<KK>, grid, dim3(64), 0, s, c->codesT, c->Wt, c->alpha, \
                       c->ext, c->idx, c->n, c->Bs)
    K_DISPATCH(c->k, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// model.linears[:3] (test.py:159-160): per-position activations exp(alpha*conv+shift), (B,U,Lo).
// Auxiliary export path: one block per (sequence, unit quad), threads over positions.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_act_kernel(const uint8_t* __restrict__ codesT,
                                                       const float* __restrict__ Wt,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ shift,
                                                       float* __restrict__ acts, int U, int k,
                                                       int L, int Lo, int Bs) {
    extern __shared__ float4 Wsm[];            // [k][5], then codes [L] as bytes
    uint8_t* cs = reinterpret_cast<uint8_t*>(Wsm + k * 5);
    const int b = blockIdx.x, quad = blockIdx.y;
    const float4* src = reinterpret_cast<const float4*>(Wt) + (size_t)quad * k * 5;
    for (int i = threadIdx.x; i < k * 5; i += 256) Wsm[i] = src[i];
    for (int p = threadIdx.x; p < L; p += 256) cs[p] = codesT[(size_t)p * Bs + b];
    __syncthreads();
    for (int p = threadIdx.x; p < Lo; p += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; ++j) {
            const float4 v = Wsm[j * 5 + cs[p + j]];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const float av[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int u = quad * 4 + uu;
            if (u < U) acts[((size_t)b * U + u) * Lo + p] = qval(alpha[u], av[uu], shift[u]);
        }
    }
}

int launch_conv_act(explainn_ctx* c, int B, float* acts, hipStream_t s) {
    const size_t sm = (size_t)c->k * 5 * sizeof(float4) + ((c->L + 15) & ~15);
    hipLaunchKernelGGL(conv_act_kernel, dim3(B, c->Uq), dim3(256), sm, s, c->codesT, c->Wt, c->alpha,
                       c->shift, acts, c->U, c->k, c->L, c->Lo, c->Bs);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}
