// The per-unit two-layer FC (architectures/__init__.py:84-100) and its backward, as three
// streaming passes whose inner loop is one v_fmac with a SCALAR (SGPR) weight operand:
//
//   fc_fwd   lane = sequence.  q[0..n) of the lane's sequence sits in VGPRs; the BN2-folded FC1
//            row A2[u][r][:] is wave-uniform and arrives through scalar loads, so
//            y2 = sh2[r] + sum_w A2[r][w]*q[w] costs n VALU ops and no LDS/vector-memory traffic.
//            ReLU, dropout, FC2 (z += V2[r]*a) and the 100 "relu'>0 and kept" bits per
//            (unit, sequence) are produced in the same loop; nothing of size (B,100U) is stored.
//   passA    lane = hidden channel r.  EQ[r][w] = sum_b e[b,r] q[b,w] with e = dz[b]*bit[b,r];
//            q[b][:], dz[b] and the bit words are scalars.  Per-chunk partials, fixed-order sum.
//   passB    lane = sequence.  dq[w] = sum_r e[r] T[r][w] - k0'[w] - sum_v q[v] M[v][w], then
//            dy = dq*q routed to the pooling argmax, plus the two BN1-backward sums.
//
// Template parameter NQ >= n is the register-array length (bucketed; weights are zero padded).
#include "common.h"

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// dmode: 0 no dropout, 1 keep-mask array, 2 counter-based generator
template <int NQ, bool TRAIN>
__global__ __launch_bounds__(64) void fc_fwd_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, const float* __restrict__ A2, const float* __restrict__ sh2,
    const float* __restrict__ V2, uint4* __restrict__ bits, float* __restrict__ zout,
    const uint8_t* __restrict__ keep_mask, int dmode, uint32_t thresh16, float scale,
    uint32_t seed_lo, uint32_t seed_hi, const float* __restrict__ c2,
    const float* __restrict__ g3, const float* __restrict__ b3, const float* __restrict__ rm3,
    const float* __restrict__ rv3, float* __restrict__ oout, int n, int Bs, int B, int U) {
    constexpr int NS = (NQ + 3) & ~3;
    const int u = blockIdx.y, lane = threadIdx.x;
    const int b = blockIdx.x * 64 + lane;
    const float a1 = alpha[u], s1 = shift[u];
    float q[NQ];
#pragma unroll
    for (int w = 0; w < NQ; ++w)
        q[w] = (w < n) ? qval(a1, ext[((size_t)u * n + w) * Bs + b], s1) : 0.f;
    const float* Au = A2 + (size_t)u * FC_H * NS;
    const float* shu = sh2 + (size_t)u * FC_H;
    const float* V2u = V2 + (size_t)u * FC_H;
    uint32_t rs = 0;
    if (TRAIN && dmode == 2)
        rs = mix32(mix32(seed_lo ^ (uint32_t)b * 0x9E3779B9U) ^ mix32(seed_hi + (uint32_t)u)) | 1u;
    const uint8_t* km = (TRAIN && dmode == 1) ? keep_mask + (size_t)min(b, B - 1) * FC_H * U + (size_t)u * FC_H
                                              : nullptr;
    float zacc = 0.f;
    uint32_t words[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
        uint32_t cur = 0u;
        const int cntr = (wd == 3) ? (FC_H - 96) : 32;
#pragma unroll 4
        for (int rr = 0; rr < cntr; ++rr) {
            const int r = wd * 32 + rr;
            const float* Ar = Au + (size_t)r * NS;
            float y = shu[r];
#pragma unroll
            for (int w = 0; w < NQ; ++w) y = fmaf(Ar[w], q[w], y);
            bool pos = y > 0.f;
            if (TRAIN) {
                if (dmode == 2) {
                    uint32_t rnd;
                    if ((rr & 1) == 0) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; rnd = rs & 0xffffU; }
                    else rnd = rs >> 16;
                    pos = pos && (rnd >= thresh16);
                } else if (dmode == 1) {
                    pos = pos && (km[r] != 0);
                }
            }
            const float av = pos ? y * scale : 0.f;
            zacc = fmaf(V2u[r], av, zacc);
            cur |= (pos ? 1u : 0u) << rr;
        }
        words[wd] = cur;
    }
    if (TRAIN) {
        zout[(size_t)u * Bs + b] = zacc;
        bits[(size_t)u * Bs + b] = make_uint4(words[0], words[1], words[2], words[3]);
    } else {
        const float inv = g3[u] / sqrtf(rv3[u] + (float)BN_EPS_D);
        const float y3 = fmaf(inv, zacc + c2[u] - rm3[u], b3[u]);
        oout[(size_t)u * Bs + b] = fmaxf(y3, 0.f);
    }
}

int launch_fc_fwd(explainn_ctx* c, const explainn_params* p, int B, bool train,
                  const uint8_t* keep_mask, float drop_p, uint64_t seed, hipStream_t s) {
    const dim3 grid((B + 63) / 64, c->U);
    int dmode = 0;
    float scale = 1.f;
    uint32_t thresh = 0;
    if (train && drop_p > 0.f) {
        dmode = keep_mask ? 1 : 2;
        scale = 1.0f / (1.0f - drop_p);
        thresh = (uint32_t)(drop_p * 65536.0 + 0.5);
    }
    if (train) { c->fwd_drop = dmode != 0; c->fwd_scale = scale; }
#define CALL(N)                                                                                   \
    if (train)                                                                                    \
        hipLaunchKernelGGL((fc_fwd_kernel<N, true>), grid, dim3(64), 0, s, c->ext, c->alpha,      \
                           c->shift, c->A2, c->sh2, p->fc2_w, c->bits, c->z, keep_mask, dmode,    \
                           thresh, scale, (uint32_t)seed, (uint32_t)(seed >> 32), p->fc2_b,       \
                           p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv, c->o, c->n, c->Bs, B, c->U);  \
    else                                                                                          \
        hipLaunchKernelGGL((fc_fwd_kernel<N, false>), grid, dim3(64), 0, s, c->ext, c->alpha,     \
                           c->shift, c->A2, c->sh2, p->fc2_w, c->bits, c->z, keep_mask, dmode,    \
                           thresh, scale, (uint32_t)seed, (uint32_t)(seed >> 32), p->fc2_b,       \
                           p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv, c->o, c->n, c->Bs, B, c->U)
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(128) void passA_kernel(const float* __restrict__ qbw,
                                                    const float* __restrict__ dz,
                                                    const uint4* __restrict__ bits,
                                                    float* __restrict__ EQp,
                                                    float* __restrict__ Sep, int Bs, int B,
                                                    int ACH) {
    constexpr int NS = (NQ + 3) & ~3;
    const int u = blockIdx.y, ch = blockIdx.x, r = threadIdx.x;
    const int per = (B + ACH - 1) / ACH;
    const int bbeg = ch * per, bend = min(B, bbeg + per);
    const int wsel = r >> 5, bsh = r & 31;
    float acc[NQ];
#pragma unroll
    for (int w = 0; w < NQ; ++w) acc[w] = 0.f;
    float se = 0.f;
    const float* dzu = dz + (size_t)u * Bs;
    const uint4* bu = bits + (size_t)u * Bs;
#pragma unroll 2
    for (int b = bbeg; b < bend; ++b) {
        const float dzb = dzu[b];                       // scalar
        const uint4 wv = bu[b];                         // scalar, 16 B
        const uint32_t myw = wsel == 0 ? wv.x : (wsel == 1 ? wv.y : (wsel == 2 ? wv.z : wv.w));
        const float e = ((myw >> bsh) & 1u) ? dzb : 0.f;
        se += e;
        const float* row = qbw + ((size_t)u * Bs + b) * NS;   // scalar row
#pragma unroll
        for (int w = 0; w < NQ; ++w) acc[w] = fmaf(e, row[w], acc[w]);
    }
    if (r < FC_H) {
        float* dst = EQp + (((size_t)u * ACH + ch) * FC_H + r) * NS;
#pragma unroll
        for (int w = 0; w < NS; ++w) dst[w] = (w < NQ) ? acc[w < NQ ? w : 0] : 0.f;
        Sep[((size_t)u * ACH + ch) * FC_H + r] = se;
    }
}

int launch_passA(explainn_ctx* c, int B, hipStream_t s) {
    const dim3 grid(c->ACH, c->U);
#define CALL(N)                                                                            \
    hipLaunchKernelGGL(passA_kernel<N>, grid, dim3(128), 0, s, c->qbw, c->dz, c->bits, c->EQp, \
                       c->Sep, c->Bs, B, c->ACH)
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(64) void passB_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, const float* __restrict__ dz, const uint4* __restrict__ bits,
    const float* __restrict__ Tt, const float* __restrict__ M, const float* __restrict__ k0p,
    const double* __restrict__ mug, const double* __restrict__ sig1, float* __restrict__ dy,
    float* __restrict__ S12p, int n, int Bs, int B) {
    constexpr int NS = (NQ + 3) & ~3;
    const int u = blockIdx.y, lane = threadIdx.x;
    const int b = blockIdx.x * 64 + lane;
    const float a1 = alpha[u], s1 = shift[u];
    float q[NQ], acc[NQ];
    const float* k0u = k0p + (size_t)u * NS;
#pragma unroll
    for (int w = 0; w < NQ; ++w) {
        q[w] = (w < n) ? qval(a1, ext[((size_t)u * n + w) * Bs + b], s1) : 0.f;
        acc[w] = -k0u[w];
    }
    const uint4 wv = bits[(size_t)u * Bs + b];
    const float dzb = dz[(size_t)u * Bs + b];
    const uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
    const float* Tu = Tt + (size_t)u * FC_H * NS;
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
        const int cntr = (wd == 3) ? (FC_H - 96) : 32;
#pragma unroll 4
        for (int rr = 0; rr < cntr; ++rr) {
            const float e = ((words[wd] >> rr) & 1u) ? dzb : 0.f;
            const float* Tr = Tu + (size_t)(wd * 32 + rr) * NS;
#pragma unroll
            for (int w = 0; w < NQ; ++w) acc[w] = fmaf(e, Tr[w], acc[w]);
        }
    }
    const float* Mu = M + (size_t)u * NS * NS;
#pragma unroll 2
    for (int v = 0; v < n; ++v) {
        // q[v] with a runtime (uniform) v: select through a static unrolled chain is costly, so
        // recompute it from ext (L2-resident) instead
        const float nq = -qval(a1, ext[((size_t)u * n + v) * Bs + b], s1);
        const float* Mr = Mu + (size_t)v * NS;
#pragma unroll
        for (int w = 0; w < NQ; ++w) acc[w] = fmaf(nq, Mr[w], acc[w]);
    }
    const float mu = (float)mug[u];
    const float isg = (float)(1.0 / sig1[u]);
    float sA = 0.f, sB = 0.f;
    const bool live = b < B;
#pragma unroll
    for (int w = 0; w < NQ; ++w) {
        if (w < n) {
            const float dyv = live ? acc[w] * q[w] : 0.f;
            const float ch = (ext[((size_t)u * n + w) * Bs + b] - mu) * isg;
            sA += dyv;
            sB = fmaf(dyv, ch, sB);
            dy[((size_t)u * n + w) * Bs + b] = dyv;
        }
    }
    sA = wave_sum(sA);
    sB = wave_sum(sB);
    if (lane == 0) {
        float* d = S12p + ((size_t)u * (Bs / 64) + blockIdx.x) * 2;
        d[0] = sA; d[1] = sB;
    }
}

int launch_passB(explainn_ctx* c, int B, hipStream_t s) {
    const dim3 grid((B + 63) / 64, c->U);
#define CALL(N)                                                                                 \
    hipLaunchKernelGGL(passB_kernel<N>, grid, dim3(64), 0, s, c->ext, c->alpha, c->shift, c->dz, \
                       c->bits, c->Tt, c->M, c->k0p, c->mug, c->sig1, c->dy, c->S12p, c->n,     \
                       c->Bs, B)
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}
