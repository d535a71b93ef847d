// Backward "small algebra" and the filter-gradient scatter.
//
//   mid_fused / mid_big   per unit: sum passA's chunk partials; gradients of the FC2 weights and
//             the BN2 affine parameters; the gradient of the FC1 weights (BN2 backward folded in
//             through V1.C, which prep2 left in VC); the tables passB consumes: T[r][w], the n x n
//             matrix M and the vector k0' that carry the BN2-backward mean terms into dq.
//             n <= 72: everything in LDS, fp64.  n > 72: M on the fp32 MFMA, EQ through global.
//   conv_bwd  sparse term of the filter gradient: every pooling window sends dy to the one position
//             that won the max, so dW gets dy added at (base at p*+j, tap j) for the k taps.
//             Lane = sequence, register accumulators per tap, bases from the 2-bit packed codes
//             through a 64-bit funnel window.
//   fin_bwd   per unit: BN1 backward closed form -> dW, d gamma1, d beta1.
#include "common.h"

typedef float f32x16b __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4b __attribute__((ext_vector_type(4)));
#define CB_PF 4              // pooling windows whose dy / idx a conv_bwd lane fetches per batch

// Register budget: this kernel must stay at <= 64 VGPRs so that TWO 1024-thread blocks share a CU --
// 300 units on 256 CUs otherwise take two rounds.  (Batching the per-channel partial-sum loads below
// made the block 11 % faster in isolation, took 127 VGPRs, and cost 5 us per step in the pipeline;
// check -Rpass-analysis=kernel-resource-usage after touching it.)
// n <= 72: one 1024-thread block per unit with V1, A2, EQ and M
// staged in LDS, so every inner loop reads LDS instead of chasing dependent global loads.
__global__ __launch_bounds__(1024) void mid_fused_kernel(
    const float* __restrict__ EQp, const float* __restrict__ Sep, const float* __restrict__ A2,
    const float* __restrict__ sh2, const float* __restrict__ sig2, const float* __restrict__ fc1_w,
    const float* __restrict__ fc2_w, const float* __restrict__ g2, const double* __restrict__ qbar,
    const float* __restrict__ VC, float* __restrict__ Ttf,
    float* __restrict__ Mff, float* __restrict__ k0p, float* __restrict__ g_fc2_w, float* __restrict__ g_bn2_w,
    float* __restrict__ g_bn2_b, float* __restrict__ g_fc1_b, float* __restrict__ g_fc1_w, int n,
    int NS, int NW16, int B, int ACH, float scale) {
    extern __shared__ float fsm[];
    const int ld = n + 1;
    float* V1s = fsm;                        // [100][ld]
    float* A2s = V1s + FC_H * ld;            // [100][ld]
    float* EQl = A2s + FC_H * ld;            // [100][ld]
    float* Ms = EQl + FC_H * ld;             // [n][n]
    float* qb = Ms + n * n;                  // [n]
    float* se = qb + n;                      // [100]
    float* md2s = se + FC_H;                 // [100]
    float* cfs = md2s + FC_H;                // [100]  md2h / sig2
    float* md2hs = cfs + FC_H;               // [100]
    float* VCl = md2hs + FC_H;               // [100][ld]  (V1.C), read once here: the dV1 loop below
    float* svl = VCl + FC_H * ld;            // [100]      then runs on LDS alone (its five global
    float* gsl = svl + FC_H;                 // [100]      loads per element made it a latency chain:
    float* sgl = gsl + FC_H;                 // [100]      12 K of the block's 31 K cycles)
    const int u = blockIdx.x, tid = threadIdx.x;
    STAMP(0);
    for (int e = tid; e < FC_H * n; e += 1024) {
        const int r = e / n, w = e % n;
        const size_t ch = (size_t)u * FC_H + r;
        V1s[r * ld + w] = fc1_w[ch * n + w];
        A2s[r * ld + w] = A2[ch * NS + w];
        VCl[r * ld + w] = VC[ch * NS + w];
    }
    // passA's partial sums are stored w-major (EQp[..][w][r]): r is the fast index here
    for (int e = tid; e < FC_H * n; e += 1024) {
        const int w = e / FC_H, r = e % FC_H;
        double eq = 0;
        for (int c0 = 0; c0 < ACH; c0 += 8) {         // eight partials in flight, fixed-order sum
            float pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                pv[i] = EQp[(((size_t)u * ACH + min(c0 + i, ACH - 1)) * NS + w) * FC_H + r];
#pragma unroll
            for (int i = 0; i < 8; ++i) KEEP(pv[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) eq += (c0 + i < ACH) ? (double)pv[i] : 0.0;
        }
        EQl[r * ld + w] = (float)eq;
    }
    double* qbd = reinterpret_cast<double*>((reinterpret_cast<size_t>(sgl + FC_H) + 7) & ~size_t(7));   // [n] doubles
    for (int w = tid; w < n; w += 1024) { const double v = qbar[(size_t)u * NS + w]; qb[w] = (float)v; qbd[w] = v; }
    for (int r = tid; r < FC_H; r += 1024) {
        double s = 0;
        for (int c = 0; c < ACH; ++c) s += (double)Sep[((size_t)u * ACH + c) * FC_H + r];
        se[r] = (float)s;
    }
    __syncthreads();
    STAMP(1);
    const double sc = (double)scale;
    if (tid < FC_H) {
        const int r = tid;
        const size_t ch = (size_t)u * FC_H + r;
        double sAE = 0, sVE = 0;
        const double ser = (double)se[r];
        for (int w = 0; w < n; ++w) {
            const double eq = (double)EQl[r * ld + w];
            sAE = fma((double)A2s[r * ld + w], eq, sAE);
            sVE = fma((double)V1s[r * ld + w], eq - ser * qbd[w], sVE);
        }
        const double v2 = (double)fc2_w[ch], sg = (double)sig2[ch];
        g_fc2_w[ch] = (float)(sc * (sAE + (double)sh2[ch] * ser));
        const double db2 = sc * v2 * ser, dg2 = sc * v2 / sg * sVE;
        g_bn2_b[ch] = (float)db2;
        g_bn2_w[ch] = (float)dg2;
        g_fc1_b[ch] = 0.f;
        md2s[r] = (float)(db2 / (double)B);
        md2hs[r] = (float)(dg2 / (double)B);
        cfs[r] = (float)((dg2 / (double)B) / sg);
        svl[r] = (float)v2; gsl[r] = (float)((double)g2[ch] / sg); sgl[r] = (float)((double)B / sg);
    }
    __syncthreads();
    STAMP(2);
    // M[v][w] = sum_r cf[r] V1[r][v] A2[r][w] on the fp32 MFMA (its consumer passB is fp32), in 16x16
    // tiles of v_mfma_f32_16x16x4_f32, one tile per wave (4 tiles at n <= 32), K = 100 channels = 25
    // steps with all operands read ahead.  (One 32x32x2 tile on ONE wave -- 50 dependent 64-cycle
    // steps plus a 16-value scattered epilogue -- was the block's critical path: every other wave sat
    // at the barrier below for 12 K of the block's 34 K cycles.  As an fp64 VALU loop: worse still.)
    const int NT2 = NW16;
    const int ntile_m = NT2 * NT2;
    {
        const int wave = tid >> 6, lane = tid & 63, c = lane & 15, gq = lane >> 4;
        for (int tile = wave; tile < ntile_m; tile += 16) {
            const int vt = tile / NT2, wt = tile % NT2;
            const int va = 16 * vt + c, wb = 16 * wt + c;
            const bool alive = va < n, blive = wb < n;
            const float* acol = V1s + min(va, n - 1);
            const float* bcol = A2s + min(wb, n - 1);
            f32x4b acc = f32x4b{0.f, 0.f, 0.f, 0.f};
            // five steps' operands at a time: the block must stay at <= 64 VGPRs (two blocks per CU)
#pragma unroll
            for (int q0 = 0; q0 < FC_H / 4; q0 += 5) {
                float av[5], bv[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int r = 4 * (q0 + q) + gq;
                    av[q] = cfs[r] * acol[r * ld];
                    bv[q] = bcol[r * ld];
                }
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(alive ? av[q] : 0.f, blive ? bv[q] : 0.f, acc, 0, 0, 0);
            }
            // D[v][w]: rows v = 16vt + 4gq + i, column wb; Mff in passB's k order (j', i'): v = 16j' + 4g + i'
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 16 * vt + 4 * gq + i;
                Mff[(((size_t)u * NW16 + wt) * (4 * NW16) + 4 * vt + i) * 64 + 16 * gq + c] = acc[i];
                if (v < n && wb < n) Ms[v * n + wb] = acc[i];
            }
        }
    }
    STAMP(3);
    // dV1 and T do not need M: the waves that had no M tile above do this loop while the tile waves
    // are still in their MFMA chains (with everybody taking an equal share the block waited for
    // wave 0 to finish its tile AND its share)
    const int busy = min(ntile_m, 8) * 64;             // threads of the tile waves
    for (int e = tid - busy; e < FC_H * NS; e += 1024 - busy) {
        if (e < 0) break;                              // tile waves skip
        const int r = e / NS, w = e % NS;
        const size_t ch = (size_t)u * FC_H + r;
        const double sv = sc * (double)svl[r];
        float tv = 0.f;
        if (w < n) {
            tv = (float)(sv * (double)A2s[r * ld + w]);
            // (V1.C)[r][w], computed once by prep2, times B / sigma2 (per-channel, from LDS)
            const double hq = (double)VCl[r * ld + w] * (double)sgl[r];
            const double val = (double)gsl[r] *
                               (sv * (double)EQl[r * ld + w] -
                                (double)md2s[r] * (double)B * qbd[w] -
                                (double)md2hs[r] * hq);
            g_fc1_w[ch * n + w] = (float)val;
        }
        {
            // T[r][w] as three bf16 pieces (hi + mid + lo = tv exactly) in the A-fragment order of
            // v_mfma_f32_16x16x32_bf16: lane 16((r>>3)&3) + (w&15), element r&7 of k-step r>>5
            const uint32_t hb = __float_as_uint(tv) & 0xffff0000u;
            const float r1 = tv - __uint_as_float(hb);
            const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mb);
            uint16_t* tb = reinterpret_cast<uint16_t*>(Ttf) + ((size_t)u * NW16 + (w >> 4)) * (3 * 4 * 512) +
                           ((size_t)(r >> 5) * 64 + 16 * ((r >> 3) & 3) + (w & 15)) * 8 + (r & 7);
            tb[0] = (uint16_t)(hb >> 16);
            tb[4 * 512] = (uint16_t)(mb >> 16);
            tb[8 * 512] = (uint16_t)(__float_as_uint(r2) >> 16);
        }
    }
    __syncthreads();
    STAMP(4);
    for (int w = tid; w < NS; w += 1024) {
        double k0 = 0;
        if (w < n) {
            for (int r = 0; r < FC_H; ++r) k0 = fma((double)A2s[r * ld + w], (double)md2s[r], k0);
            for (int v = 0; v < n; ++v) k0 -= qbd[v] * (double)Ms[v * n + w];
        }
        k0p[(size_t)u * NS + w] = (float)k0;
    }
    STAMP(5);
}

// Same algebra for large pooled lengths (72 < n <= 160, configs C4/C5): V1 and A2 stay in LDS
// (2 x 100 x (n+1) floats), the EQ sums go through global memory (EQs), C is streamed through LDS in
// 32-column chunks, and the k0' correction uses  sum_v qbar[v] M[v][w] = sum_r cf[r] (V1[r].qbar) A2[r][w]
// so M itself never has to be resident.
__global__ __launch_bounds__(1024) void mid_big_kernel(
    const float* __restrict__ EQp, const float* __restrict__ Sep, const float* __restrict__ A2,
    const float* __restrict__ sh2, const float* __restrict__ sig2, const float* __restrict__ fc1_w,
    const float* __restrict__ fc2_w, const float* __restrict__ g2, const double* __restrict__ qbar,
    const float* __restrict__ VC, float* __restrict__ EQs,
    float* __restrict__ Ttf, float* __restrict__ Mff,
    float* __restrict__ k0p, float* __restrict__ g_fc2_w, float* __restrict__ g_bn2_w,
    float* __restrict__ g_bn2_b, float* __restrict__ g_fc1_b, float* __restrict__ g_fc1_w, int n,
    int NS, int NW16, int B, int ACH, float scale) {
    extern __shared__ float bsm[];
    const int ld = n + 1;
    float* V1s = bsm;                        // [100][ld]
    float* A2s = V1s + FC_H * ld;            // [100][ld]
    float* qb = A2s + FC_H * ld;             // [n]
    float* se = qb + n;                      // [100]
    float* md2s = se + FC_H;                 // [100]
    float* cfs = md2s + FC_H;                // [100]  md2h / sig2
    float* md2hs = cfs + FC_H;               // [100]
    float* kco = md2hs + FC_H;               // [100]  md2 - cf * (V1[r].qbar)
    const int u = blockIdx.x, tid = threadIdx.x;
    constexpr int NT = 1024;
    for (int e = tid; e < FC_H * n; e += NT) {
        const int r = e / n, w = e % n;
        const size_t ch = (size_t)u * FC_H + r;
        V1s[r * ld + w] = fc1_w[ch * n + w];
        A2s[r * ld + w] = A2[ch * NS + w];
    }
    // passA's partial sums are stored w-major (EQp[..][w][r]): r is the fast index here
    for (int e = tid; e < FC_H * n; e += NT) {
        const int w = e / FC_H, r = e % FC_H;
        const size_t ch = (size_t)u * FC_H + r;
        double eq = 0;
        for (int c0 = 0; c0 < ACH; c0 += 8) {         // eight partials in flight, fixed-order sum
            float pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                pv[i] = EQp[(((size_t)u * ACH + min(c0 + i, ACH - 1)) * NS + w) * FC_H + r];
#pragma unroll
            for (int i = 0; i < 8; ++i) KEEP(pv[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) eq += (c0 + i < ACH) ? (double)pv[i] : 0.0;
        }
        EQs[ch * NS + w] = (float)eq;
    }
    for (int w = tid; w < n; w += NT) qb[w] = (float)qbar[(size_t)u * NS + w];
    for (int r = tid; r < FC_H; r += NT) {
        double s = 0;
        for (int c = 0; c < ACH; ++c) s += (double)Sep[((size_t)u * ACH + c) * FC_H + r];
        se[r] = (float)s;
    }
    __syncthreads();                         // (also orders this block's EQs writes before its reads)
    const double sc = (double)scale;
    // per-channel sums: 8 threads per channel, shuffle-reduced
    {
        const int r = tid >> 3, part = tid & 7;
        const int rr = r < FC_H ? r : FC_H - 1;
        const size_t ch = (size_t)u * FC_H + rr;
        double sAE = 0, sVE = 0, qv = 0;
        const double ser = (double)se[rr];
        for (int w = part; w < n; w += 8) {
            const double eq = (double)EQs[ch * NS + w];
            const double qw = qbar[(size_t)u * NS + w];
            sAE = fma((double)A2s[rr * ld + w], eq, sAE);
            sVE = fma((double)V1s[rr * ld + w], eq - ser * qw, sVE);
            qv = fma((double)V1s[rr * ld + w], qw, qv);
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            sAE += __shfl_xor(sAE, off, 64); sVE += __shfl_xor(sVE, off, 64); qv += __shfl_xor(qv, off, 64);
        }
        if (r < FC_H && part == 0) {
            const double v2 = (double)fc2_w[ch], sg = (double)sig2[ch];
            g_fc2_w[ch] = (float)(sc * (sAE + (double)sh2[ch] * ser));
            const double db2 = sc * v2 * ser, dg2 = sc * v2 / sg * sVE;
            g_bn2_b[ch] = (float)db2;
            g_bn2_w[ch] = (float)dg2;
            g_fc1_b[ch] = 0.f;
            md2s[r] = (float)(db2 / (double)B);
            md2hs[r] = (float)(dg2 / (double)B);
            const double cf = (dg2 / (double)B) / sg;
            cfs[r] = (float)cf;
            kco[r] = (float)(db2 / (double)B - cf * qv);
        }
    }
    __syncthreads();
    // M[v][w] = sum_r cf[r] V1[r][v] A2[r][w] on the matrix cores (fp32 MFMA; its consumer passB is
    // fp32 too): NWT x NWT tiles of 32x32, K = 100 hidden channels in steps of 2, a tile per wave
    {
        const int wave = tid >> 6, lane = tid & 63, rc = lane & 31, kk = lane >> 5;
        const int NT2 = (NS + 31) >> 5;
        for (int tile = wave; tile < NT2 * NT2; tile += NT / 64) {
            const int vt = tile / NT2, wt = tile % NT2;
            const int va = 32 * vt + rc, wb = 32 * wt + rc;
            const bool alive = va < n, blive = wb < n;
            const float* acol = V1s + min(va, n - 1);
            const float* bcol = A2s + min(wb, n - 1);
            f32x16b acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.f;
#pragma unroll 5
            for (int s2 = 0; s2 < FC_H / 2; ++s2) {
                const int r = 2 * s2 + kk;
                const float a = alive ? cfs[r] * acol[r * ld] : 0.f;
                const float bb = blive ? bcol[r * ld] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
            }
            if (wb < NS) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int v = 32 * vt + (g & 3) + 8 * (g >> 2) + 4 * kk;
                    if (v < NS) {
                        if (v < 16 * NW16 && wb < 16 * NW16)       // k order (j', i'): v = 16j' + 4g + i'
                            Mff[(((size_t)u * NW16 + (wb >> 4)) * (4 * NW16) + 4 * (v >> 4) + (v & 3)) * 64 +
                                16 * ((v >> 2) & 3) + (wb & 15)] = acc[g];
                    }
                }
            }
        }
    }
    // dV1 and T; (V1.C)[r][w] was computed once by prep2
    for (int e = tid; e < FC_H * NS; e += NT) {
        const int r = e / NS, w = e % NS;
        const size_t ch = (size_t)u * FC_H + r;
        const double sv = sc * (double)fc2_w[ch];
        float tv = 0.f;
        if (w < n) {
            tv = (float)(sv * (double)A2s[r * ld + w]);
            const double sg = (double)sig2[ch];
            const double hq = (double)VC[ch * NS + w] * (double)B / sg;
            const double val = ((double)g2[ch] / sg) *
                               (sv * (double)EQs[ch * NS + w] -
                                (double)md2s[r] * (double)B * qbar[(size_t)u * NS + w] -
                                (double)md2hs[r] * hq);
            g_fc1_w[ch * n + w] = (float)val;
        }
        {
            // T[r][w] as three bf16 pieces (hi + mid + lo = tv exactly) in the A-fragment order of
            // v_mfma_f32_16x16x32_bf16: lane 16((r>>3)&3) + (w&15), element r&7 of k-step r>>5
            const uint32_t hb = __float_as_uint(tv) & 0xffff0000u;
            const float r1 = tv - __uint_as_float(hb);
            const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mb);
            uint16_t* tb = reinterpret_cast<uint16_t*>(Ttf) + ((size_t)u * NW16 + (w >> 4)) * (3 * 4 * 512) +
                           ((size_t)(r >> 5) * 64 + 16 * ((r >> 3) & 3) + (w & 15)) * 8 + (r & 7);
            tb[0] = (uint16_t)(hb >> 16);
            tb[4 * 512] = (uint16_t)(mb >> 16);
            tb[8 * 512] = (uint16_t)(__float_as_uint(r2) >> 16);
        }
    }
    for (int w = tid; w < NS; w += NT) {
        double k0 = 0;
        if (w < n)
            for (int r = 0; r < FC_H; ++r) k0 = fma((double)A2s[r * ld + w], (double)kco[r], k0);
        k0p[(size_t)u * NS + w] = (float)k0;
    }
}

static size_t mid_big_lds(int n) {
    return ((size_t)2 * FC_H * (n + 1) + n + 5 * FC_H) * sizeof(float);
}

static size_t mid_fused_lds(int n) {
    return ((size_t)4 * FC_H * (n + 1) + (size_t)n * n + n + 7 * FC_H + 2) * sizeof(float) + (size_t)n * sizeof(double);
}

int launch_mid_bwd(explainn_ctx* c, const explainn_params* p, const explainn_grads* g, int B,
                   hipStream_t s) {
    if (c->n <= 72) {
        hipLaunchKernelGGL(mid_fused_kernel, dim3(c->U), dim3(1024), mid_fused_lds(c->n), s, c->EQp,
                           c->Sep, c->A2, c->sh2, c->sig2, p->fc1_w, p->fc2_w, p->bn2_w, c->qbar,
                           c->VC, c->Ttf, c->Mff, c->k0p, g->fc2_w, g->bn2_w, g->bn2_b,
                           g->fc1_b, g->fc1_w, c->n, c->NS, fc_nw16(c->NQ), B, c->ACH,
                           c->fwd_scale);
        LAUNCH_CHECK();
        return EXPLAINN_OK;
    }
    hipLaunchKernelGGL(mid_big_kernel, dim3(c->U), dim3(1024), mid_big_lds(c->n), s, c->EQp, c->Sep,
                       c->A2, c->sh2, c->sig2, p->fc1_w, p->fc2_w, p->bn2_w, c->qbar, c->VC, c->EQs,
                       c->Ttf, c->Mff, c->k0p, g->fc2_w, g->bn2_w, g->bn2_b, g->fc1_b,
                       g->fc1_w, c->n, c->NS, fc_nw16(c->NQ), B, c->ACH,
                       c->fwd_scale);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm1 backward closed form for one unit, by `nthr` threads of
// one block: sums the per-tile partials in fixed order -> dW, d gamma1, d beta1.
struct fin_args {
    const float* S12p; const double* m; const double* Gw; const double* mug; const double* sig1;
    const float* g1; float* g_conv_w; float* g_conv_b; float* g_bn1_w; float* g_bn1_b;
    int K4; int freeze_n; int NG;
    int dsp_stride, dsp_count;        // Dspp[u][dsp_stride][4k]: the first dsp_count partials are live
};

__device__ __forceinline__ void fin_unit(const fin_args& f, const float* __restrict__ Dspp, int u,
                                         int tid, int nthr, int Bs, int B) {
    const int K4 = f.K4;
    const int NT = f.dsp_stride, nt = f.dsp_count;     // the filter-gradient partials of this step
    const int NT16 = Bs / 16, nt16 = (B + 15) / 16;
    // S1, S2: the per-tile partials are spread over the threads (one load each, then a fixed-order
    // tree) instead of every thread walking all of them in batches
    __shared__ double fin_red[2][4];
    double S1 = 0, S2 = 0;
    for (int t = tid; t < f.NG * nt16; t += nthr) {      // (w-tile group, 16-sequence tile) partials
        const int grp = t / nt16, tl = t - grp * nt16;
        const float2 pv = *reinterpret_cast<const float2*>(&f.S12p[(((size_t)u * f.NG + grp) * NT16 + tl) * 2]);
        S1 += (double)pv.x; S2 += (double)pv.y;
    }
    S1 = wave_sum_d(S1); S2 = wave_sum_d(S2);
    if ((tid & 63) == 0) { fin_red[0][tid >> 6] = S1; fin_red[1][tid >> 6] = S2; }
    __syncthreads();
    S1 = 0; S2 = 0;
    for (int w = 0; w < (nthr + 63) / 64; ++w) { S1 += fin_red[0][w]; S2 += fin_red[1][w]; }
    const double sg = f.sig1[u], a = (double)f.g1[u] / sg, mu = f.mug[u];
    for (int i = tid; i < K4; i += nthr) {
        double D = 0;
        for (int t0 = 0; t0 < nt; t0 += 16) {          // sixteen partials in flight, fixed-order sum
            float pv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) pv[q] = Dspp[((size_t)u * NT + min(t0 + q, nt - 1)) * K4 + i];
#pragma unroll
            for (int q = 0; q < 16; ++q) KEEP(pv[q]);
#pragma unroll
            for (int q = 0; q < 16; ++q) D += (t0 + q < nt) ? (double)pv[q] : 0.0;
        }
        const double val = a * (D - S1 * f.m[i] - (S2 / sg) * (f.Gw[(size_t)u * K4 + i] - mu * f.m[i]));
        f.g_conv_w[(size_t)u * K4 + i] = (u < f.freeze_n) ? 0.f : (float)val;
    }
    if (tid == 0) {
        f.g_bn1_b[u] = (float)S1;
        f.g_bn1_w[u] = (float)S2;
        f.g_conv_b[u] = 0.f;
    }
}

template <int K>
__global__ __launch_bounds__(64, 3) void conv_bwd_kernel(const float* __restrict__ dy,
                                                         const uint8_t* __restrict__ idx,
                                                         const uint32_t* __restrict__ pk2,
                                                         const uint32_t* __restrict__ nmask,
                                                         float* __restrict__ Dspp, int U, int n,
                                                         int Bs, int PW, int NW, int nlds_off,
                                                         int B) {
    // One wavefront = CB_TILES x 64 sequences x one unit; lane = sequence.  The partial sums of a lane
    // live in REGISTERS: per tap the sums for bases C,G,T; base A is recovered at the end as
    // (sum of all dy) - C - G - T.  N positions are packed as 'C'; the lanes that have one also add
    // that dy to a per-tap LDS cell, which is taken out of C at the end.  (LDS float atomics
    // serialise per lane on gfx950 and LDS read-modify-write chains were latency-bound,
    // profiles/r01_c.)  Two sequence tiles per wave: 2400 waves at C2, all resident at 3 waves per SIMD,
    // which leaves the registers for the row prefetch below.
    // The packed codes are staged per chunk of CBW pooling windows (the positions a chunk touches
    // span CBW*7 + K - 1 bases): a fixed ~7 KB of LDS per wave whatever the sequence length, so the
    // 5 waves/SIMD hold for L = 1000 too (staging the whole sequence cost 26 KB there and left
    // 1.5 waves/SIMD).  Columns are lane-private: no barrier between chunks.
    constexpr int CBW = 32;
    constexpr int PWC = ((POOLW * CBW + K + 15) >> 4) + 2, NWC = ((POOLW * CBW + K + 31) >> 5) + 2;
    extern __shared__ uint32_t smem[];        // one-hot rows [4] float4, pk2 chunk [PWC][64], nmask chunk [NWC][64]
    uint32_t* pks = smem + 16;
    uint32_t* nms = pks + (size_t)PWC * 64;
    const int lane = threadIdx.x, u = blockIdx.y;
    STAMP(0);
    constexpr uint32_t KMASK = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
    // One-hot rows {c==A, c==C, c==G, c==T} per 2-bit code as floats: acc[j] += dy * row[code_j] is
    // one ds_read_b128 (4 distinct addresses per wave: broadcast, no conflicts), one v_pk_fma_f32
    // (C, G) and one v_fma_f32 (T) per tap; A = total - C - G - T at the end.  (Three compare-select-add triples per tap were 10 VALU instructions
    // per tap and made this kernel the VALU-issue-bound maximum of the step, profiles/r01_final.)
    f32x2 a12[K];
    float a3[K];
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) { a12[j] = f32x2{0.f, 0.f}; a3[j] = 0.f; }
    // dy that landed in the C bucket because an N base is packed as 'C', per tap: one LDS float per
    // tap, fed by the few lanes that have an N (one wavefront per block, so the order of the adds
    // -- lane order, instruction order -- is fixed and the sum reproducible)
    float* nlds = reinterpret_cast<float*>(smem) + nlds_off;
    // the one-hot rows sit at the very start of the dynamic LDS, so that a row's address is the
    // code field itself (no base add): row c = {c==C, c==G, c==T, 0}
    float4* oh = reinterpret_cast<float4*>(smem);
    if (lane < K) nlds[lane] = 0.f;
    if (lane < 4) oh[lane] = make_float4(lane == 1 ? 1.f : 0.f, lane == 2 ? 1.f : 0.f, lane == 3 ? 1.f : 0.f, 0.f);
    __syncthreads();
    typedef __attribute__((address_space(3))) f32x4b lds_f32x4;
    typedef __attribute__((address_space(3))) char lds_char;
    // 32-bit LDS address of the rows (64-byte aligned: the code field is OR-ed in, one v_and_or_b32)
    const uint32_t ohbase = (uint32_t)(size_t)(const lds_char*)(reinterpret_cast<const char*>(oh));
    const float* __restrict__ dyu = dy + (size_t)u * n * Bs;
    const uint8_t* __restrict__ idxu = idx + (size_t)u * n * Bs;
    for (int tl = 0; tl < CB_TILES; ++tl) {
    const int tile = blockIdx.x * CB_TILES + tl;
    if (tile * 64 >= B) break;                         // wave-uniform
    const int b = tile * 64 + lane;
    for (int wc = 0; wc < n; wc += CBW) {
        // chunk origin in words: POOLW*CBW = 224 positions = 14 code words = 7 mask words
        const int w_lo = (POOLW * wc) >> 4, n_lo = (POOLW * wc) >> 5;
        stage_columns2<PWC, NWC>(pks + lane, pk2 + (size_t)w_lo * Bs + b, min(PWC, PW - w_lo),
                                 nms + lane, nmask + (size_t)n_lo * Bs + b, min(NWC, NW - n_lo), Bs);
        if (wc == 0) STAMP(1);
        const int wend = min(wc + CBW, n);
        // dy / idx are fetched CB_PF windows at a time, one batch ahead of the taps that consume them:
        // with a single window in flight every event waited out a full L2 round trip (49 us of
        // latency for 25 us of issue, profiles/r02)
        float dy_n[CB_PF];
        int ps_n[CB_PF];
#pragma unroll
        for (int q = 0; q < CB_PF; ++q) {
            const int off = min(wc + q, wend - 1) * Bs + b;       // 32-bit lane offset, uniform base
            dy_n[q] = dyu[off];
            ps_n[q] = (int)idxu[off];
        }
        for (int wb0 = wc; wb0 < wend; wb0 += CB_PF) {
        float dy_c[CB_PF];
        int ps_c[CB_PF];
#pragma unroll
        for (int q = 0; q < CB_PF; ++q) { KEEP(dy_n[q]); KEEP(ps_n[q]); dy_c[q] = dy_n[q]; ps_c[q] = ps_n[q]; }
#pragma unroll
        for (int q = 0; q < CB_PF; ++q) {
            const int off = min(wb0 + CB_PF + q, wend - 1) * Bs + b;
            dy_n[q] = dyu[off];
            ps_n[q] = (int)idxu[off];
        }
#pragma unroll
        for (int q = 0; q < CB_PF; ++q) {
            const int wb = wb0 + q;
            float dyv = (wb < wend) ? dy_c[q] : 0.f;
            const int ps = ps_c[q] + POOLW * min(wb, wend - 1);
            // lanes past the batch (the last, partly filled tile) must not contribute: their dy is
            // whatever an earlier, larger batch left there
            dyv = (b < B) ? dyv : 0.f;
            const int w0 = (ps >> 4) - w_lo, sh = (ps & 15) * 2;
            const uint32_t c0 = pks[w0 * 64 + lane], c1 = pks[(w0 + 1) * 64 + lane],
                           c2 = pks[(w0 + 2) * 64 + lane];
            const uint32_t lo = __funnelshift_r(c0, c1, sh), hi = __funnelshift_r(c1, c2, sh);
            const int n0 = (ps >> 5) - n_lo, nsh = ps & 31;
            uint32_t nm = __funnelshift_r(nms[n0 * 64 + lane], nms[(n0 + 1) * 64 + lane], nsh) & KMASK;
            const f32x2 dy2 = f32x2{dyv, dyv};
            // rows are fetched five at a time, one group ahead of the sums that consume them (all K
            // in flight would need 4K registers; one group at a time left the LDS latency exposed
            // four times per window)
            auto row_addr = [&](int j) -> uint32_t {
                // byte offset of the code's one-hot row: code * 16
                const uint32_t off16 = j < 16 ? ((j >= 2 ? (lo >> (2 * j - 4)) : (lo << (4 - 2 * j))) & 0x30u)
                                              : ((j >= 18 ? (hi >> (2 * (j - 16) - 4)) : (hi << (4 - 2 * (j - 16)))) & 0x30u);
                return off16 | ohbase;
            };
            constexpr int NGRP = (K + 4) / 5;
            f32x4b r[2][5];
#pragma unroll
            for (int jj = 0; jj < 5; ++jj)
                if (jj < K) r[0][jj] = *(const volatile lds_f32x4*)(size_t)row_addr(jj);
#pragma unroll
            for (int gq = 0; gq < NGRP; ++gq) {
                const int j0 = 5 * gq;
                if (gq + 1 < NGRP) {
#pragma unroll
                    for (int jj = 0; jj < 5; ++jj)
                        if (j0 + 5 + jj < K)
                            // (volatile: the fourth component is unused and a narrowed ds_read_b96
                            // costs twice the LDS cycles of the b128)
                            r[(gq + 1) & 1][jj] = *(const volatile lds_f32x4*)(size_t)row_addr(j0 + 5 + jj);
                }
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) {
                    const int j = j0 + jj;
                    if (j < K) {
                        a12[j] = __builtin_elementwise_fma(dy2, f32x2{r[gq & 1][jj][0], r[gq & 1][jj][1]}, a12[j]);
                        a3[j] = fmaf(dyv, r[gq & 1][jj][2], a3[j]);
                        // pinned here: otherwise the sums sink below the N loop into the loop latch
                        // and all K rows (4K registers) stay live across it
                        KEEP(a12[j]); KEEP(a3[j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            tot += dyv;
            // N bases (packed as 'C'): the few lanes that have one add that dy to the tap's LDS cell
            while (__any(nm != 0u)) {
                if (nm != 0u) {
                    const int j = __ffs(nm) - 1;
                    nm &= nm - 1u;
                    atomicAdd(&nlds[j], dyv);
                }
            }
        }
        }
    }
    }
    STAMP(2);
    // sums over the 64 lanes through LDS, one base at a time: every lane parks K values as a column
    // of a [K][65] tile (the code tiles are dead by now), then lane j adds up row j
    float* red = reinterpret_cast<float*>(smem) + 16;
    float* out = Dspp + ((size_t)u * (Bs / 32) + blockIdx.x) * 4 * K;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float v = a == 0 ? (tot - a12[j][0] - a12[j][1] - a3[j]) : (a == 1 ? a12[j][0] : (a == 2 ? a12[j][1] : a3[j]));
            red[j * 65 + lane] = v;
        }
        __syncthreads();
        if (lane < K) {
            float sacc = 0.f;
#pragma unroll 16
            for (int l = 0; l < 64; ++l) sacc += red[lane * 65 + l];
            if (a == 1) sacc -= nlds[lane];            // lane j holds tap j: take its N sum out
            out[a * K + lane] = sacc;
        }
    }
    STAMP(3);
}

#define KB_DISPATCH(Kv, CALL)                                                                  \
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

// ---------------------------------------------------------------------------------------------
// Filter gradient on the bf16 matrix core (round 3).  The pooled gradient of (unit u, window w,
// sequence b) lands on ONE position p = 7w + idx[u][w][b]; with D[u][(b,p)] = dy there and 0
// elsewhere, and the window indicator F[(b,p)][(a,j)] = [s[b][p+j] == a],
//     dW[u][a][j] = sum_(b,p) D[u][(b,p)] F[(b,p)][(a,j)]
// is a GEMM (M = units, N = 4k filter taps, K = sequences x positions) whose B operand is a BIT matrix
// and whose A operand is 1/7 dense.  F is exact in bf16; dy is split exactly into three bf16 pieces
// (hi / mid / lo, 8+8+8 mantissa bits, as passA does): every product is exact and the matrix core
// accumulates in fp32, so this is an fp32 sum of the same terms the register formulation adds, in
// another order.  (The register formulation -- one LDS row fetch and two packed FMAs per (window,
// tap, 64 sequences) -- is LDS-bandwidth and issue bound at 42 us for C2: tools/issue_rate.hip
// measures 3.8 clk per wave-instruction for its mix.  Here the same sum costs 1.66 M MFMAs = 11 us
// of matrix-core time.)  N bases need no correction: an N sets no bit in any base's mask.
//
// k-step of v_mfma_f32_16x16x32_bf16 = (one position p, 32 sequences).  Workgroup = (block of 32
// sequences, tile of 16 units); its four waves take the windows of a chunk in turn.
//   B operand  the batch as bit masks per (base, position) already exists (pack.hip: bm, the input
//              of the moment kernel).  The workgroup expands the 32 bits of its sequence block to
//              bf16 once per (base, position): X[q][a][g] = the 8 bf16 of sequences 8g..8g+7 -- the
//              B fragment of column (a, j) at position p is the 16 bytes X[p + j][a][g]: one
//              ds_read_b128 per (16-column tile, position), no arithmetic.
//   A operand  lane (unit row c, g) loads dy and idx of its 8 sequences for window w once (32 + 8
//              contiguous bytes), splits dy into the three pieces, and for each of the 7 positions of
//              the window keeps the pieces where idx == r (packed 16-bit compare -> mask).
// 15 MFMAs (5 column tiles x 3 pieces) per (unit tile, window, position, 32 sequences).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 cb_bf16x8;
typedef uint32_t cb_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short cb_u16x2 __attribute__((ext_vector_type(2)));
#define CBM_CHUNK 32                  // pooling windows per LDS image of the expanded masks
__host__ __device__ constexpr int cbm_tiles(int K) { return (4 * K + 15) / 16; }

template <int K>
__global__ __launch_bounds__(256, 3) void conv_bwd_mm_kernel(
    const float* __restrict__ dy, const uint8_t* __restrict__ idx,
    const unsigned long long* __restrict__ bm, float* __restrict__ Dspp, int U, int n, int Bs, int B,
    int NT64, int Lp, int LQ, int NP) {
    constexpr int K4 = 4 * K, NTL = cbm_tiles(K);
    extern __shared__ __attribute__((aligned(16))) unsigned char cbsm[];
    cb_u32x4* X = reinterpret_cast<cb_u32x4*>(cbsm);           // [LQ positions][4 bases][4 slots] x 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * 32, u0 = blockIdx.y * 16;
    const int tile64 = b0 >> 6, half = (b0 >> 5) & 1;
    const int ua = min(u0 + c, U - 1);                          // the unit this lane feeds as an A row
    // sequences past the batch: their idx bytes are forced to 7, which no position matches
    uint32_t dead0 = 0u, dead1 = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (b0 + 8 * g + j >= B) { if (j < 4) dead0 |= 7u << (8 * j); else dead1 |= 7u << (8 * (j - 4)); }
    // Column order inside the matrix-core tiles: n = 4 j + a (tap-major), so the 16 columns of tile t
    // are taps 4t .. 4t+3 x the four bases.  X holds one 256-byte row per position q, [base][slot]
    // with slot = g ^ 2 (q & 1): with that swizzle the 16 lanes of every ds_read_b128 lane group hit
    // 16 distinct 16-byte bank slots whatever q is (searched exhaustively; the plain [a][q][g]
    // image was 2.6-way conflicted).  Byte offset of this lane's fragment for column tile t at an
    // EVEN relative position; an odd one flips bit 5 (slot ^ 2).
    const int ca = c & 3;
    uint32_t boff[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = min(4 * t + (c >> 2), K - 1);
        boff[t] = (uint32_t)(j * 256 + ca * 64 + ((g ^ ((j & 1) << 1)) * 16));
    }
    f32x4b acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[t] = f32x4b{0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ dyu = dy + (size_t)ua * n * Bs + b0 + 8 * g;
    const uint8_t* __restrict__ ixu = idx + (size_t)ua * n * Bs + b0 + 8 * g;
    STAMP(0);
    for (int wc = 0; wc < n; wc += CBM_CHUNK) {
        const int nwin = min(CBM_CHUNK, n - wc), q0 = POOLW * wc, nq = POOLW * nwin + K - 1;
        // this wave's first window of the chunk: its loads fly while the masks are expanded
        int w = wc + wave;
        float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0;
        uint2 iw = make_uint2(0u, 0u);
        if (w < wc + nwin) {
            d0 = *reinterpret_cast<const float4*>(dyu + (size_t)w * Bs);
            d1 = *reinterpret_cast<const float4*>(dyu + (size_t)w * Bs + 4);
            iw = *reinterpret_cast<const uint2*>(ixu + (size_t)w * Bs);
        }
        __syncthreads();                                        // the previous chunk's X is dead
        for (int e = tid; e < 4 * nq; e += 256) {
            const int a = e / nq, q = e - a * nq;
            const unsigned long long m = bm[((size_t)a * NT64 + tile64) * Lp + q0 + q];
            const uint32_t bits32 = half ? (uint32_t)(m >> 32) : (uint32_t)m;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const uint32_t by = (bits32 >> (8 * gg)) & 0xffu;
                cb_u32x4 v;
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    v[d] = ((by >> (2 * d)) & 1u) * 0x00003f80u + ((by >> (2 * d + 1)) & 1u) * 0x3f800000u;
                X[(q * 4 + a) * 4 + (gg ^ ((q & 1) << 1))] = v;
            }
        }
        __syncthreads();
        if (wc == 0) STAMP(1);
        for (; w < wc + nwin; w += 4) {
            // pieces of the 8 gradients: x = hi + mid + lo exactly, each piece a bf16
            const float xs[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            uint32_t ph[4], pm[4], pl[4];
            {
                uint32_t hb[8], mb[8], lb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t xb = __float_as_uint(xs[j]);
                    const float r1 = xs[j] - __uint_as_float(xb & 0xffff0000u);
                    const uint32_t r1b = __float_as_uint(r1);
                    const float r2 = r1 - __uint_as_float(r1b & 0xffff0000u);
                    hb[j] = xb; mb[j] = r1b; lb[j] = __float_as_uint(r2);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {       // (upper halves of sequences 2d+1, 2d) -> one dword
                    ph[d] = __builtin_amdgcn_perm(hb[2 * d + 1], hb[2 * d], 0x07060302u);
                    pm[d] = __builtin_amdgcn_perm(mb[2 * d + 1], mb[2 * d], 0x07060302u);
                    pl[d] = __builtin_amdgcn_perm(lb[2 * d + 1], lb[2 * d], 0x07060302u);
                }
            }
            // one-hot of the argmax offset per sequence, as 16-bit pairs in the order of the packed
            // pieces: bit r of a half says "this sequence's gradient sits at position 7w + r"
            const uint32_t i0 = iw.x | dead0, i1 = iw.y | dead1;
            uint32_t ip[4];
            ip[0] = (1u << (i0 & 0xffu)) | (0x10000u << ((i0 >> 8) & 0xffu));
            ip[1] = (1u << ((i0 >> 16) & 0xffu)) | (0x10000u << (i0 >> 24));
            ip[2] = (1u << (i1 & 0xffu)) | (0x10000u << ((i1 >> 8) & 0xffu));
            ip[3] = (1u << ((i1 >> 16) & 0xffu)) | (0x10000u << (i1 >> 24));
            const int wn = w + 4;                               // next window of this wave: in flight below
            if (wn < wc + nwin) {
                d0 = *reinterpret_cast<const float4*>(dyu + (size_t)wn * Bs);
                d1 = *reinterpret_cast<const float4*>(dyu + (size_t)wn * Bs + 4);
                iw = *reinterpret_cast<const uint2*>(ixu + (size_t)wn * Bs);
            }
            const uint32_t wbase = (uint32_t)(POOLW * (w - wc) * 256);
            const uint32_t wpar = (uint32_t)((w - wc) & 1) << 5;   // parity of this window's first position
            typedef __attribute__((address_space(3))) cb_u32x4 lds_u32x4;
            typedef __attribute__((address_space(3))) unsigned char lds_uchar;
            const uint32_t xbase = (uint32_t)(size_t)(const lds_uchar*)cbsm + wbase;
#pragma unroll
            for (int r = 0; r < POOLW; ++r) {
                cb_u32x4 ah, am, al;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // 0xffff per half whose bit r is set
                    const uint32_t mk = __umul24((ip[d] >> r) & 0x00010001u, 0xffffu);
                    ah[d] = ph[d] & mk; am[d] = pm[d] & mk; al[d] = pl[d] & mk;
                }
                const cb_bf16x8 fh = __builtin_bit_cast(cb_bf16x8, ah), fm = __builtin_bit_cast(cb_bf16x8, am),
                                fl = __builtin_bit_cast(cb_bf16x8, al);
                cb_bf16x8 fb[NTL];
#pragma unroll
                for (int t = 0; t < NTL; ++t)
                    fb[t] = __builtin_bit_cast(cb_bf16x8, *(const lds_u32x4*)(size_t)(
                        xbase + ((boff[t] ^ wpar ^ (uint32_t)((r & 1) << 5)) + (uint32_t)(r * 256))));
                // piece outermost: consecutive MFMAs go to different accumulators
#pragma unroll
                for (int t = 0; t < NTL; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, fb[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NTL; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm, fb[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NTL; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, fb[t], acc[t], 0, 0, 0);
            }
        }
    }
    STAMP(2);
    // the four waves' tiles are added in a fixed order through LDS (the mask image is dead), then
    // stored as ONE partial per (sequence block, unit): D[row = unit 4g + i][col c] of column tile t
    __syncthreads();
    float* red = reinterpret_cast<float*>(cbsm);                // [4 waves][NTL][4][64]
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((wave * NTL + t) * 4 + i) * 64 + lane] = acc[t][i];
    __syncthreads();
    for (int e = tid; e < NTL * 4 * 64; e += 256) {
        const float v = ((red[e] + red[NTL * 256 + e]) + red[2 * NTL * 256 + e]) + red[3 * NTL * 256 + e];
        const int l = e & 63, i = (e >> 6) & 3, t = e >> 8;
        const int u = u0 + 4 * (l >> 4) + i, j = 4 * t + ((l & 15) >> 2), a = l & 3;
        if (u < U && j < K) Dspp[((size_t)u * NP + blockIdx.x) * K4 + a * K + j] = v;
    }
    STAMP(3);
}

static size_t conv_bwd_mm_lds(const explainn_ctx* c, int* lq_out) {
    const int nwin = c->n < CBM_CHUNK ? c->n : CBM_CHUNK;
    const int LQ = POOLW * nwin + c->k - 1;
    if (lq_out) *lq_out = LQ;
    size_t sm = (size_t)4 * LQ * 64;
    const size_t red = (size_t)4 * cbm_tiles(c->k) * 256 * sizeof(float);
    return sm > red ? sm : red;
}

int launch_conv_bwd_mm(explainn_ctx* c, int B, hipStream_t s) {
    int LQ = 0;
    const size_t sm = conv_bwd_mm_lds(c, &LQ);
    const int NP = (B + 31) / 32;
    const dim3 grid(NP, (c->U + 15) / 16);
#define CALL(KK)                                                                                   \
    hipLaunchKernelGGL(conv_bwd_mm_kernel<KK>, grid, dim3(256), sm, s, c->dy, c->idx, c->bm, c->Dspp, \
                       c->U, c->n, c->Bs, B, (B + 63) / 64, c->Lp, LQ, c->Bs / 32)
    KB_DISPATCH(c->k, CALL);
#undef CALL
    LAUNCH_CHECK();
    c->dsp_stride = c->Bs / 32; c->dsp_count = NP;
    return EXPLAINN_OK;
}

int launch_conv_bwd(explainn_ctx* c, int B, hipStream_t s) {
    {   // A/B switch while both formulations exist (EXPLAINN_CONV_BWD=reg selects the register form)
        const char* e = getenv("EXPLAINN_CONV_BWD");
        if (!(e && e[0] == 'r')) return launch_conv_bwd_mm(c, B, s);
    }
    const dim3 grid(((B + 63) / 64 + CB_TILES - 1) / CB_TILES, c->U);
    // chunk tiles (see the kernel: [PWC + NWC][64] words) or the [k][65] reduction tile
    const int pwc = ((POOLW * 32 + c->k + 15) >> 4) + 2, nwc = ((POOLW * 32 + c->k + 31) >> 5) + 2;
    size_t sm = (size_t)(pwc + nwc) * 64 * sizeof(uint32_t);
    const size_t red_bytes = (size_t)c->k * 65 * sizeof(float);
    if (sm < red_bytes) sm = red_bytes;
    sm += 4 * sizeof(float4);                          // the one-hot rows in front of the tiles
    const int nlds_off = (int)(sm / sizeof(float));    // k floats behind the tiles: the N corrections
    sm += (size_t)((c->k + 15) & ~15) * sizeof(float);
#define CALL(KK)                                                                               \
    hipLaunchKernelGGL(conv_bwd_kernel<KK>, grid, dim3(64), sm, s, c->dy, c->idx, c->pk2, c->nmask, \
                       c->Dspp, c->U, c->n, c->Bs, c->PW, c->NW, nlds_off, B)
    KB_DISPATCH(c->k, CALL);
#undef CALL
    LAUNCH_CHECK();
    c->dsp_stride = c->Bs / 32; c->dsp_count = (int)grid.x;
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// One block per unit.  (Folding this into conv_bwd's last-arriving tile was tried: the device-scope
// release every tile then needs -- an L2 write-back on this multi-XCD part -- cost 130 us per step.)
__global__ __launch_bounds__(128) void fin_bwd_kernel(const fin_args fin,
                                                      const float* __restrict__ Dspp, int Bs, int B) {
    fin_unit(fin, Dspp, blockIdx.x, threadIdx.x, 128, Bs, B);
}

int launch_fin_bwd(explainn_ctx* c, const explainn_params* p, const explainn_grads* g, int B,
                   int freeze_n, hipStream_t s) {
    const fin_args fin = {c->S12p, c->m, c->Gw, c->mug, c->sig1, p->bn1_w, g->conv_w, g->conv_b,
                          g->bn1_w, g->bn1_b, c->K4, freeze_n, fc_ng(c->NQ), c->dsp_stride, c->dsp_count};
    hipLaunchKernelGGL(fin_bwd_kernel, dim3(c->U), dim3(128), 0, s, fin, c->Dspp, c->Bs, B);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int bwd_configure(explainn_ctx* c) {
    {
        const size_t sm = conv_bwd_mm_lds(c, nullptr);
        if (sm > 48 * 1024) {
#define CALL(KK)                                                                                  \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bwd_mm_kernel<KK>),   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm))
            KB_DISPATCH(c->k, CALL);
#undef CALL
        }
    }
    if (c->n > 72)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_big_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)mid_big_lds(c->n)));
    if (c->n <= 72 && mid_fused_lds(c->n) > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_fused_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)mid_fused_lds(c->n)));
    return EXPLAINN_OK;
}
