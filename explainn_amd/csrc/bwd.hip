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
    {
        // (all the loads of the block's three passes over [100][n] in flight before the first LDS store:
        // as a loop with a run-time trip count every pass was its own memory round trip)
        // (three passes per batch: 9 values + 3 LDS addresses, inside the 64 registers that let two
        // blocks share a CU; n <= 30 is one batch)
        constexpr int NPASS = 3;
        for (int e0 = tid; e0 < FC_H * n; e0 += 1024 * NPASS) {
            float v1[NPASS], a2[NPASS], vc[NPASS];
            int dst[NPASS];
#pragma unroll
            for (int it = 0; it < NPASS; ++it) {
                const int e = e0 + 1024 * it;
                const bool on = e < FC_H * n;
                const int ec = on ? e : 0, r = ec / n, w = ec - r * n;
                const uint32_t ch = (uint32_t)(u * FC_H + r);
                dst[it] = on ? r * ld + w : -1;
                v1[it] = fc1_w[(size_t)ch * n + w];
                a2[it] = A2[(size_t)ch * NS + w];
                vc[it] = VC[(size_t)ch * NS + w];
            }
#pragma unroll
            for (int it = 0; it < NPASS; ++it) { KEEP(v1[it]); KEEP(a2[it]); KEEP(vc[it]); }
#pragma unroll
            for (int it = 0; it < NPASS; ++it)
                if (dst[it] >= 0) { V1s[dst[it]] = v1[it]; A2s[dst[it]] = a2[it]; VCl[dst[it]] = vc[it]; }
        }
    }
    // passA's partial sums are stored w-major (EQp[..][w][r]): r is the fast index here
    for (int e0 = tid; e0 < FC_H * n; e0 += 1024 * 3) {
        // three passes x four partials in flight, fixed-order sums (one round trip at ACH <= 4, n <= 30)
        double eq[3] = {0, 0, 0};
        for (int c0 = 0; c0 < ACH; c0 += 4) {
            float pv[3][4];
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int e = e0 + 1024 * it, ec = e < FC_H * n ? e : 0;
                const int w = ec / FC_H, r = ec % FC_H;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    pv[it][i] = EQp[(((size_t)u * ACH + min(c0 + i, ACH - 1)) * NS + w) * FC_H + r];
            }
#pragma unroll
            for (int it = 0; it < 3; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) KEEP(pv[it][i]);
#pragma unroll
            for (int it = 0; it < 3; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) eq[it] += (c0 + i < ACH) ? (double)pv[it][i] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int e = e0 + 1024 * it;
            if (e < FC_H * n) EQl[(e % FC_H) * ld + e / FC_H] = (float)eq[it];
        }
    }
    double* qbd = reinterpret_cast<double*>((reinterpret_cast<size_t>(sgl + FC_H) + 7) & ~size_t(7));   // [n] doubles
    for (int w = tid; w < n; w += 1024) { const double v = qbar[(size_t)u * NS + w]; qb[w] = (float)v; qbd[w] = v; }
    for (int r = tid; r < FC_H; r += 1024) {
        double s = 0;
        for (int c0 = 0; c0 < ACH; c0 += 4) {          // four partials in flight (was one round trip each)
            float pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = Sep[((size_t)u * ACH + min(c0 + i, ACH - 1)) * FC_H + r];
#pragma unroll
            for (int i = 0; i < 4; ++i) KEEP(pv[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) s += (c0 + i < ACH) ? (double)pv[i] : 0.0;
        }
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
#define FIN_THREADS 320              // fin_bwd block: 4 groups of 76 taps at k = 19
#define FIN_MAXGRP 8
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
    __shared__ double fin_red[2][16];
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
    // The filter-gradient partials: nt of them per tap (64 at C2).  The block's threads form
    // NGRP = nthr / K4 groups of K4; group q takes partials q, q + NGRP, ... with up to sixteen loads
    // in flight per thread (one memory round trip for 64 partials instead of four), sums them in
    // fp64 in a fixed order, and the groups are combined through LDS in group order.
    __shared__ double fin_part[FIN_MAXGRP][4 * MAX_K];
    const int ngrp = min(FIN_MAXGRP, max(1, nthr / K4));
    const int grp = tid / K4, i = tid - grp * K4;
    if (grp < ngrp) {
        double D = 0;
        for (int t0 = grp; t0 < nt; t0 += 16 * ngrp) {     // sixteen partials in flight, fixed-order sum
            float pv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) pv[q] = Dspp[((size_t)u * NT + min(t0 + q * ngrp, nt - 1)) * K4 + i];
#pragma unroll
            for (int q = 0; q < 16; ++q) KEEP(pv[q]);
#pragma unroll
            for (int q = 0; q < 16; ++q) D += (t0 + q * ngrp < nt) ? (double)pv[q] : 0.0;
        }
        fin_part[grp][i] = D;
    }
    __syncthreads();
    if (tid < K4) {
        double D = 0;
        for (int q = 0; q < ngrp; ++q) D += fin_part[q][tid];
        const double val = a * (D - S1 * f.m[tid] - (S2 / sg) * (f.Gw[(size_t)u * K4 + tid] - mu * f.m[tid]));
        f.g_conv_w[(size_t)u * K4 + tid] = (u < f.freeze_n) ? 0.f : (float)val;
    }
    if (tid == 0) {
        f.g_bn1_b[u] = (float)S1;
        f.g_bn1_w[u] = (float)S2;
        f.g_conv_b[u] = 0.f;
    }
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
#define CBM_CHUNK 14                  // pooling windows per LDS image of the expanded masks (even: parity)
#define CBM_WAVES 4                   // wavefronts per workgroup (one per SIMD): they take the windows of a chunk in turn
#ifndef CBM_PF
#define CBM_PF 1                      // windows of dy / idx a wave keeps in flight (2 and 3 measured slower)
#endif
__host__ __device__ constexpr int cbm_tiles(int K) { return (K + 3) / 4; }   // 16-column tiles = 4 taps x 4 bases
// waves per SIMD the register budget is held to: accumulators and B fragments grow with the tile count
__host__ __device__ constexpr int cbm_occ(int K) { return cbm_tiles(K) <= 5 ? 4 : (cbm_tiles(K) <= 6 ? 3 : 2); }

// Geometry: workgroup = (block of 32 sequences, tile of 16 units, half of the pooling windows), four
// wavefronts = one per SIMD, so a CU's SIMDs always hold equal shares.  At C2 that is 1216 workgroups
// of 28 KB of LDS; the kernel is held to 128 registers so that four are resident per CU (4 waves per
// SIMD) and the dispatcher hands the remaining ones to whichever CU finishes first.  What the stamps
// showed about the alternatives (tools/stampbench, profiles/r03): the main loop saturates the matrix
// pipe, so the time is set by the SIMD with the most waves -- 608 four-wave workgroups over all
// windows (three per CU at 51 KB) and 1216 two-wave ones (2-3 waves per SIMD) both left SIMDs with
// 3 x 10.9 K cycles of MFMA work where the average is 2.4 x.
template <int K>
__global__ __launch_bounds__(64 * CBM_WAVES, cbm_occ(K)) void conv_bwd_mm_kernel(
    const float* __restrict__ dy, const uint8_t* __restrict__ idx,
    const unsigned long long* __restrict__ bm, float* __restrict__ Dspp, int U, int n, int Bs, int B,
    int NT64, int Lp, int NP, int wper) {
    constexpr int K4 = 4 * K, NTL = cbm_tiles(K), NTH = 64 * CBM_WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cbsm[];
    cb_u32x4* X = reinterpret_cast<cb_u32x4*>(cbsm);           // [positions][4 bases][4 slots] x 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * 32, u0 = blockIdx.y * 16;
    const int wlo = blockIdx.z * wper, whi = min(n, wlo + wper);   // this workgroup's windows
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
    typedef __attribute__((address_space(3))) cb_u32x4 lds_u32x4;
    typedef __attribute__((address_space(3))) unsigned char lds_uchar;
    const uint32_t xlds = (uint32_t)(size_t)(const lds_uchar*)cbsm;
    STAMP(0);
    for (int wc = wlo; wc < whi; wc += CBM_CHUNK) {
        const int nwin = min(CBM_CHUNK, whi - wc), q0 = POOLW * wc, nq = POOLW * nwin + K - 1;
        // this wave's first CBM_PF windows of the chunk: their loads fly while the masks are expanded
        // (one window ahead is enough: 29.4 us in-pipeline at C2 against 30.6 with two and 32.0 with
        // three ahead -- the deeper queues only make the burst at kernel start larger)
        int w = wc + wave;
        float4 qd0[CBM_PF], qd1[CBM_PF];
        uint2 qiw[CBM_PF];
#pragma unroll
        for (int f = 0; f < CBM_PF; ++f) {
            const int wf = min(w + f * CBM_WAVES, wc + nwin - 1);
            qd0[f] = *reinterpret_cast<const float4*>(dyu + (size_t)wf * Bs);
            qd1[f] = *reinterpret_cast<const float4*>(dyu + (size_t)wf * Bs + 4);
            qiw[f] = *reinterpret_cast<const uint2*>(ixu + (size_t)wf * Bs);
        }
        __syncthreads();                                        // the previous chunk's X is dead
        // (every mask word of this thread is requested before the first is expanded: one load per
        // loop turn exposed a memory round trip each, 8 K of the kernel's 43 K cycles per wave)
        constexpr int XIT = (4 * (POOLW * CBM_CHUNK + K - 1) + NTH - 1) / NTH;
        unsigned long long mw[XIT];
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = min(tid + it * NTH, 4 * nq - 1);
            mw[it] = bm[((size_t)(e & 3) * NT64 + tile64) * Lp + q0 + (e >> 2)];
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = tid + it * NTH;
            if (e < 4 * nq) {
                const int q = e >> 2, a = e & 3;                 // (position, base): 64 contiguous bytes each
                const uint32_t bits32 = half ? (uint32_t)(mw[it] >> 32) : (uint32_t)mw[it];
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
        }
        __syncthreads();
        if (wc == wlo) STAMP(1);
        for (; w < wc + nwin; w += CBM_WAVES) {
            // pieces of the 8 gradients: x = hi + mid + lo exactly, each piece a bf16
            const float4 d0 = qd0[0], d1 = qd1[0];
            const uint2 iw = qiw[0];
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
            // the queue moves up one and the window CBM_PF turns ahead is requested
#pragma unroll
            for (int f = 0; f + 1 < CBM_PF; ++f) { qd0[f] = qd0[f + 1]; qd1[f] = qd1[f + 1]; qiw[f] = qiw[f + 1]; }
            {
                const int wn = min(w + CBM_PF * CBM_WAVES, wc + nwin - 1);
                qd0[CBM_PF - 1] = *reinterpret_cast<const float4*>(dyu + (size_t)wn * Bs);
                qd1[CBM_PF - 1] = *reinterpret_cast<const float4*>(dyu + (size_t)wn * Bs + 4);
                qiw[CBM_PF - 1] = *reinterpret_cast<const uint2*>(ixu + (size_t)wn * Bs);
            }
            // parity of this window's first position (the chunk starts on an even one: 7 * 14 windows)
            const uint32_t xbase = xlds + (uint32_t)(POOLW * (w - wc) * 256);
            const uint32_t wpar = (uint32_t)((w - wc) & 1) << 5;
            // A fragments of position r: the pieces where the offset equals r (0xffff per half whose
            // bit r is set).  The fragments of position r + 1 are built IN PLACE while the MFMAs of
            // position r run, each piece as soon as the matrix instructions that read it have issued
            // (hi is rewritten under the mid MFMAs, mid under the lo ones, lo at the end), with the
            // mask words computed under the hi MFMAs.  sched_barrier after every step pins that order:
            // a wave issues in order, so vector work behind a block of back-to-back MFMAs would find
            // the matrix pipe idle and vector work in front of it the vector pipe.
            cb_u32x4 ah, am, al;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t mk = __umul24(ip[d] & 0x00010001u, 0xffffu);
                ah[d] = ph[d] & mk; am[d] = pm[d] & mk; al[d] = pl[d] & mk;
            }
#pragma unroll
            for (int r = 0; r < POOLW; ++r) {
                cb_bf16x8 fb[NTL];
#pragma unroll
                for (int t = 0; t < NTL; ++t)
                    fb[t] = __builtin_bit_cast(cb_bf16x8, *(const lds_u32x4*)(size_t)(
                        xbase + ((boff[t] ^ wpar ^ (uint32_t)((r & 1) << 5)) + (uint32_t)(r * 256))));
                uint32_t tz[4] = {0u, 0u, 0u, 0u};
                const bool more = r + 1 < POOLW;
                __builtin_amdgcn_sched_barrier(0);
                // hi piece; the four mask words of the next position (3 instructions each) in between
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cb_bf16x8, ah), fb[t], acc[t], 0, 0, 0);
                    if (more) {
#pragma unroll
                        for (int k = (12 * t) / NTL; k < (12 * (t + 1)) / NTL; ++k) {
                            const int d = k / 3, op = k - 3 * d;
                            if (op == 0) tz[d] = ip[d] >> (r + 1);
                            else if (op == 1) tz[d] &= 0x00010001u;
                            else tz[d] = __umul24(tz[d], 0xffffu);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // mid piece; hi of the next position is rewritten in between
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cb_bf16x8, am), fb[t], acc[t], 0, 0, 0);
                    if (more) {
#pragma unroll
                        for (int d = (4 * t) / NTL; d < (4 * (t + 1)) / NTL; ++d) ah[d] = ph[d] & tz[d];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // lo piece; mid of the next position in between, lo behind the last MFMA
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cb_bf16x8, al), fb[t], acc[t], 0, 0, 0);
                    if (more) {
#pragma unroll
                        for (int d = (4 * t) / NTL; d < (4 * (t + 1)) / NTL; ++d) am[d] = pm[d] & tz[d];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) al[d] = pl[d] & tz[d];
                }
            }
        }
    }
    STAMP(2);
    // the waves' tiles are added in a fixed order through LDS (the mask image is dead), then stored
    // as ONE partial per (sequence block, window half, unit): D[row = unit 4g + i][col c] of tile t
    __syncthreads();
    float* red = reinterpret_cast<float*>(cbsm);                // [waves][NTL][4][64]
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((wave * NTL + t) * 4 + i) * 64 + lane] = acc[t][i];
    __syncthreads();
    const int part = blockIdx.z * gridDim.x + blockIdx.x;
    for (int e = tid; e < NTL * 4 * 64; e += NTH) {
        float v = red[e];
#pragma unroll
        for (int wv = 1; wv < CBM_WAVES; ++wv) v += red[wv * NTL * 256 + e];
        const int l = e & 63, i = (e >> 6) & 3, t = e >> 8;
        const int u = u0 + 4 * (l >> 4) + i, j = 4 * t + ((l & 15) >> 2), a = l & 3;
        if (u < U && j < K) Dspp[((size_t)u * NP + part) * K4 + a * K + j] = v;
    }
    STAMP(3);
}

// windows per workgroup: the pooling windows are cut in two halves (one workgroup each) when there
// are enough of them
static int conv_bwd_mm_split(const explainn_ctx* c) {
    int P = c->n >= 8 ? 2 : 1;
    if (const char* e = getenv("EXPLAINN_CBM_SPLIT")) { const int v = atoi(e); if (v >= 1 && v <= 8 && v <= c->n) P = v; }
    return P;
}
static size_t conv_bwd_mm_lds(const explainn_ctx* c) {
    const int wper = (c->n + conv_bwd_mm_split(c) - 1) / conv_bwd_mm_split(c);
    const int nwin = wper < CBM_CHUNK ? wper : CBM_CHUNK;
    const size_t sm = (size_t)(POOLW * nwin + c->k - 1) * 256;
    const size_t red = (size_t)CBM_WAVES * cbm_tiles(c->k) * 256 * sizeof(float);
    return sm > red ? sm : red;
}

int launch_conv_bwd_mm(explainn_ctx* c, int B, hipStream_t s) {
    const size_t sm = conv_bwd_mm_lds(c);
    const int P = conv_bwd_mm_split(c), wper = (c->n + P - 1) / P, NB = (B + 31) / 32;
    const dim3 grid(NB, (c->U + 15) / 16, P);
#define CALL(KK)                                                                                   \
    hipLaunchKernelGGL(conv_bwd_mm_kernel<KK>, grid, dim3(64 * CBM_WAVES), sm, s, c->dy, c->idx, c->bm, \
                       c->Dspp, c->U, c->n, c->Bs, B, (B + 63) / 64, c->Lp, c->Bs / 4, wper)
    KB_DISPATCH(c->k, CALL);
#undef CALL
    LAUNCH_CHECK();
    c->dsp_stride = c->Bs / 4; c->dsp_count = NB * P;
    return EXPLAINN_OK;
}

int launch_conv_bwd(explainn_ctx* c, int B, hipStream_t s) { return launch_conv_bwd_mm(c, B, s); }

// ---------------------------------------------------------------------------------------------
// One block per unit.  (Folding this into conv_bwd's last-arriving tile was tried: the device-scope
// release every tile then needs -- an L2 write-back on this multi-XCD part -- cost 130 us per step.)
__global__ __launch_bounds__(FIN_THREADS) void fin_bwd_kernel(const fin_args fin,
                                                              const float* __restrict__ Dspp, int Bs, int B) {
    fin_unit(fin, Dspp, blockIdx.x, threadIdx.x, FIN_THREADS, Bs, B);
}

int launch_fin_bwd(explainn_ctx* c, const explainn_params* p, const explainn_grads* g, int B,
                   int freeze_n, hipStream_t s) {
    const fin_args fin = {c->S12p, c->m, c->Gw, c->mug, c->sig1, p->bn1_w, g->conv_w, g->conv_b,
                          g->bn1_w, g->bn1_b, c->K4, freeze_n, fc_ng(c->NQ), c->dsp_stride, c->dsp_count};
    hipLaunchKernelGGL(fin_bwd_kernel, dim3(c->U), dim3(FIN_THREADS), 0, s, fin, c->Dspp, c->Bs, B);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int bwd_configure(explainn_ctx* c) {
    {
        const size_t sm = conv_bwd_mm_lds(c);
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
