// The per-unit two-layer FC (architectures/__init__.py:84-100) and its backward on the exact-fp32
// matrix cores: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain and runs at the fp32
// vector rate, but takes both operands from VGPRs -- no scalar/LDS broadcast stream to feed.
// FC1 is the one dense contraction on this path: per unit (B x n).(n x 100).
//
// Fragment convention (cdna_hip_programming.md section 3), lane l, rc = l&31, kk = l>>5:
//   D[i][j] = sum_k A[i][k] B[k][j] + C[i][j];  A operand = A[rc][2s+kk],  B operand = B[2s+kk][rc]
//   C/D register g of lane l holds D[(g&3) + 8*(g>>2) + 4*kk][rc].
//
//   fc_fwd   D[r][b] = sum_w A2[r][w] q[b][w] + sh2[r].  A = folded FC1 weights (fragments staged
//            in LDS once per workgroup), B = q of 32 sequences (registers).  A lane ends up with 16
//            hidden channels of ITS sequence per r-tile: ReLU, dropout, the FC2 dot product
//            z += V2[r]*a and the 100 "relu'>0 and kept" bits are all lane-local; the two lane
//            halves are merged with one cross-half shuffle.  Nothing of size (B,100U) is stored.
//   passA    D[r][w] = sum_b e[b][r] q[b][w],  e = dz[b]*bit[b][r] built on the fly from the bit
//            words (A operand), q rows from the sequence-major copy (B operand).
//   passB    D[w][b] = sum_r T[r][w] e[b][r] - sum_v M[v][w] q[b][v] - k0'[w]; then dy = dq*q and
//            the two BatchNorm1-backward sums.  A = T and M fragments in LDS.
//   qmom     (prep.hip) the q second-moment matrix, same instruction.
//
// Template parameter NQ >= n (bucketed pooled length); hidden width 100 is padded to 4 tiles of 32.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

#define FC_RT 4                      // r-tiles of 32 covering the 100 hidden channels
#define FC_BTW 2                     // 32-sequence tiles per wavefront in fc_fwd
#define PB_BTW 2                     // ... in passB

// MODE: 0 eval, 1 train without dropout, 2 train + counter-based generator, 3 train + keep-mask
template <int NQ, int MODE>
__global__ __launch_bounds__(256, (NQ <= 32 ? 5 : 1)) void fc_fwd_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, const float* __restrict__ A2f, const float* __restrict__ sh2,
    const float* __restrict__ V2, uint4* __restrict__ bits, float* __restrict__ zout,
    const uint8_t* __restrict__ keep_mask, uint32_t thresh16, float scale, uint32_t seed_lo,
    uint32_t seed_hi, const float* __restrict__ c2, const float* __restrict__ g3,
    const float* __restrict__ b3, const float* __restrict__ rm3, const float* __restrict__ rv3,
    float* __restrict__ oout, int n, int Bs, int B, int U, const uint32_t* __restrict__ seed_dev) {
    constexpr int NS = ns_stride(NQ), NKS = (NQ + 1) / 2;
    constexpr bool TRAIN = MODE != 0;
    // a captured step (hipGraph) reads its dropout seed from device memory, so that replays can
    // use a new one; direct launches pass it by value
    if (MODE == 2 && seed_dev) { seed_lo = seed_dev[0]; seed_hi = seed_dev[1]; }
    __shared__ __attribute__((aligned(16))) float Af[FC_RT * NKS * 64];
    __shared__ __attribute__((aligned(16))) float sh2s[128];
    __shared__ __attribute__((aligned(16))) float v2s[128];
    const int u = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rc = lane & 31, kk = lane >> 5;
    STAMP(0);
    // stage the A fragments (prep2 wrote them in fragment order):
    // Af[(t*NKS+s)*64 + l] = A2[32t + (l&31)][2s + (l>>5)]
    {
        const float4* src = reinterpret_cast<const float4*>(A2f + (size_t)u * FC_RT * NKS * 64);
        float4* dst = reinterpret_cast<float4*>(Af);
        constexpr int N4 = FC_RT * NKS * 16;          // float4 count, all loads issued before stores
        float4 tv[(N4 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i)
            tv[i] = (tid + i * 256 < N4) ? src[tid + i * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i)
            if (tid + i * 256 < N4) dst[tid + i * 256] = tv[i];
    }
    if (tid < 128) {
        sh2s[tid] = tid < FC_H ? sh2[(size_t)u * FC_H + tid] : 0.f;
        v2s[tid] = tid < FC_H ? V2[(size_t)u * FC_H + tid] : 0.f;
    }
    __syncthreads();
    STAMP(1);
    const float a1 = alpha[u], s1 = shift[u];
    for (int it = 0; it < FC_BTW; ++it) {
        const int bt = (blockIdx.x * 4 + wave) * FC_BTW + it;
        const int b = bt * 32 + rc;
        if (bt * 32 >= B) break;                       // wave-uniform
        float qf[NKS];
        // all loads first (unconditional, clamped rows), pinned, then exp: see KEEP in common.h
#pragma unroll
        for (int s = 0; s < NKS; ++s) qf[s] = ext[((size_t)u * n + min(2 * s + kk, n - 1)) * Bs + b];
#pragma unroll
        for (int s = 0; s < NKS; ++s) KEEP(qf[s]);
#pragma unroll
        for (int s = 0; s < NKS; ++s) qf[s] = (2 * s + kk < n) ? qval(a1, qf[s], s1) : 0.f;
        uint32_t rs = 0;
        if (MODE == 2)
            rs = mix32(mix32(seed_lo ^ (uint32_t)(2 * b + kk) * 0x9E3779B9U) ^
                       mix32(seed_hi + (uint32_t)u)) | 1u;
        const uint8_t* km = (MODE == 3) ? keep_mask + (size_t)min(b, B - 1) * FC_H * U + (size_t)u * FC_H
                                        : nullptr;
        if (it == 0) STAMP_AFTER_LOADS(2);
        float zp = 0.f;
        uint32_t words[FC_RT];
#pragma unroll
        for (int t = 0; t < FC_RT; ++t) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(&sh2s[32 * t + 8 * g + 4 * kk]);
                acc[4 * g] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
            }
#pragma unroll
            for (int s = 0; s < NKS; ++s) acc = MFMA32(Af[(t * NKS + s) * 64 + lane], qf[s], acc);
            uint32_t word = 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v2 = *reinterpret_cast<const float4*>(&v2s[32 * t + 8 * g + 4 * kk]);
                const float v2a[4] = {v2.x, v2.y, v2.z, v2.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float y = acc[4 * g + j];
                    bool pos = y > 0.f;
                    if (MODE == 2) {
                        uint32_t rnd;
                        if ((j & 1) == 0) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; rnd = rs & 0xffffU; }
                        else rnd = rs >> 16;
                        pos = pos && (rnd >= thresh16);
                    } else if (MODE == 3) {
                        const int r = 32 * t + 8 * g + 4 * kk + j;
                        pos = pos && (km[r < FC_H ? r : 0] != 0);
                    }
                    const float av = pos ? y * scale : 0.f;
                    zp = fmaf(v2a[j], av, zp);
                    word |= (pos ? 1u : 0u) << (8 * g + 4 * kk + j);
                }
            }
            words[t] = word;
        }
        if (it == 0) STAMP(3);
        zp += __shfl_xor(zp, 32, 64);
#pragma unroll
        for (int t = 0; t < FC_RT; ++t) words[t] |= __shfl_xor(words[t], 32, 64);
        if (kk == 0) {
            if (TRAIN) {
                zout[(size_t)u * Bs + b] = zp;
                bits[(size_t)u * Bs + b] = make_uint4(words[0], words[1], words[2], words[3]);
            } else {
                const float inv = g3[u] / sqrtf(rv3[u] + (float)BN_EPS_D);
                const float y3 = fmaf(inv, zp + c2[u] - rm3[u], b3[u]);
                oout[(size_t)u * Bs + b] = fmaxf(y3, 0.f);
            }
        }
    }
    STAMP(4);
}

int launch_fc_fwd(explainn_ctx* c, const explainn_params* p, int B, bool train,
                  const uint8_t* keep_mask, float drop_p, uint64_t seed, hipStream_t s) {
    const int tiles = (B + 31) / 32;
    const dim3 grid((tiles + 4 * FC_BTW - 1) / (4 * FC_BTW), c->U);
    int mode = train ? 1 : 0;
    float scale = 1.f;
    uint32_t thresh = 0;
    if (train && drop_p > 0.f) {
        mode = keep_mask ? 3 : 2;
        scale = 1.0f / (1.0f - drop_p);
        thresh = (uint32_t)(drop_p * 65536.0 + 0.5);
    }
    if (train) { c->fwd_drop = mode > 1; c->fwd_scale = scale; }
#define ARGS c->ext, c->alpha, c->shift, c->A2f, c->sh2, p->fc2_w, c->bits, c->z, keep_mask, thresh, \
             scale, (uint32_t)seed, (uint32_t)(seed >> 32), p->fc2_b, p->bn3_w, p->bn3_b,          \
             p->bn3_rm, p->bn3_rv, c->o, c->n, c->Bs, B, c->U,                                   \
             (c->capturing ? c->seed_dev : (const uint32_t*)nullptr)
#define CALL(N)                                                                                    \
    switch (mode) {                                                                                \
        case 0: hipLaunchKernelGGL((fc_fwd_kernel<N, 0>), grid, dim3(256), 0, s, ARGS); break;     \
        case 1: hipLaunchKernelGGL((fc_fwd_kernel<N, 1>), grid, dim3(256), 0, s, ARGS); break;     \
        case 2: hipLaunchKernelGGL((fc_fwd_kernel<N, 2>), grid, dim3(256), 0, s, ARGS); break;     \
        default: hipLaunchKernelGGL((fc_fwd_kernel<N, 3>), grid, dim3(256), 0, s, ARGS); break;    \
    }
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
#undef ARGS
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// passA: one wavefront per (unit, batch chunk, w-tile); K dimension = sequences, two per MFMA.
// Super-tiles of 64 sequences are fetched with coalesced loads (lane = sequence): the q rows
// (from ext, through exp), the 128 bit words and dz; they are re-read from a wave-private LDS tile
// in MFMA operand order while the next super-tile's loads are already in flight.
// ---------------------------------------------------------------------------------------------
#define QT_LD 65
template <int NQ>
__global__ __launch_bounds__(64, 3) void passA_kernel(const float* __restrict__ ext,
                                                   const float* __restrict__ alpha,
                                                   const float* __restrict__ shift,
                                                   const float* __restrict__ dz,
                                                   const uint4* __restrict__ bits,
                                                   float* __restrict__ EQp,
                                                   float* __restrict__ Sep, int n, int Bs, int B,
                                                   int ACH) {
    constexpr int NS = ns_stride(NQ);
    __shared__ float tq[32 * QT_LD];
    __shared__ __attribute__((aligned(16))) uint4 tw[64];
    __shared__ float tdz[64];
    const int u = blockIdx.y, ch = blockIdx.x, wt = blockIdx.z, lane = threadIdx.x;
    const int rc = lane & 31, kk = lane >> 5;
    const int per = ((((B + ACH - 1) / ACH) + 63) / 64) * 64;
    const int bbeg = ch * per, bend = min(B, bbeg + per);
    const float a1 = alpha[u], sh1 = shift[u];
    const float* eu = ext + (size_t)u * n * Bs;
    const float* dzu = dz + (size_t)u * Bs;
    const uint4* bu = bits + (size_t)u * Bs;
    f32x16 acc[FC_RT];
#pragma unroll
    for (int t = 0; t < FC_RT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    float se[FC_RT] = {0.f, 0.f, 0.f, 0.f};
    float rq[32], rdz = 0.f;
    uint4 rw = make_uint4(0u, 0u, 0u, 0u);
    auto fetch = [&](int b0) {
        const int b = b0 + lane;
        const bool live = b < bend;
        const int bc = live ? b : bbeg;
#pragma unroll
        for (int i = 0; i < 32; ++i) rq[i] = eu[(size_t)min(wt * 32 + i, n - 1) * Bs + bc];
        rdz = dzu[bc];
        rw = bu[bc];
#pragma unroll
        for (int i = 0; i < 32; ++i) KEEP(rq[i]);
        KEEP(rdz);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int w = wt * 32 + i;
            rq[i] = (live && w < n) ? qval(a1, rq[i], sh1) : 0.f;
        }
        rdz = live ? rdz : 0.f;
    };
    STAMP(0);
    if (bbeg < bend) fetch(bbeg);
    for (int b0 = bbeg; b0 < bend; b0 += 64) {
#pragma unroll
        for (int i = 0; i < 32; ++i) tq[i * QT_LD + lane] = rq[i];
        tw[lane] = rw;
        tdz[lane] = rdz;
        if (b0 == bbeg) STAMP(1);
        if (b0 + 64 < bend) fetch(b0 + 64);           // in flight during the MFMAs below
        const int ks = (min(bend - b0, 64) + 1) >> 1;
        for (int s = 0; s < ks; ++s) {
            const int col = 2 * s + kk;
            const float qv = tq[rc * QT_LD + col];
            const float dzb = tdz[col];
            const uint4 wv = tw[col];
            const uint32_t wds[FC_RT] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int t = 0; t < FC_RT; ++t) {
                const float e = ((wds[t] >> rc) & 1u) ? dzb : 0.f;
                se[t] += e;
                acc[t] = MFMA32(e, qv, acc[t]);
            }
        }
    }
    STAMP(2);
    // D[r][w]: lane holds column w = wt*32+rc, rows r = 32t + (g&3) + 8(g>>2) + 4kk
    const int w = wt * 32 + rc;
    if (w < NS) {
#pragma unroll
        for (int t = 0; t < FC_RT; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int r = 32 * t + (g & 3) + 8 * (g >> 2) + 4 * kk;
                if (r < FC_H) EQp[(((size_t)u * ACH + ch) * FC_H + r) * NS + w] = acc[t][g];
            }
    }
    if (wt == 0) {
#pragma unroll
        for (int t = 0; t < FC_RT; ++t) {
            const float sv = se[t] + __shfl_xor(se[t], 32, 64);
            const int r = 32 * t + rc;
            if (kk == 0 && r < FC_H) Sep[((size_t)u * ACH + ch) * FC_H + r] = sv;
        }
    }
    STAMP(3);
}

int launch_passA(explainn_ctx* c, int B, hipStream_t s) {
#define CALL(N)                                                                                  \
    hipLaunchKernelGGL(passA_kernel<N>, dim3(c->ACH, c->U, (N + 31) / 32), dim3(64), 0, s,       \
                       c->ext, c->alpha, c->shift, c->dz, c->bits, c->EQp, c->Sep, c->n, c->Bs,  \
                       B, c->ACH)
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// passB: workgroup = 4 wavefronts of one unit; T and M fragments staged in LDS
// ---------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(256, (NQ <= 32 ? 5 : 1)) void passB_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, const float* __restrict__ dz, const uint4* __restrict__ bits,
    const float* __restrict__ Ttf, const float* __restrict__ Mff, const float* __restrict__ k0p,
    const double* __restrict__ mug, const double* __restrict__ sig1, float* __restrict__ dy,
    float* __restrict__ S12p, int n, int Bs, int B) {
    constexpr int NS = ns_stride(NQ), NKS = (NQ + 1) / 2, NWT = (NQ + 31) / 32, RKS = FC_H / 2;
    extern __shared__ __attribute__((aligned(16))) float smemB[];
    float* Tf = smemB;                                 // [NWT][RKS][64]
    float* Mf = Tf + NWT * RKS * 64;                   // [NWT][NKS][64]
    float* k0s = Mf + NWT * NKS * 64;                  // [NWT*32]
    const int u = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rc = lane & 31, kk = lane >> 5;
    STAMP(0);
    // A fragments Tf[(wt*RKS+s)*64+l] = T[r=2s+(l>>5)][w=wt*32+(l&31)] and Mf likewise over v are
    // laid out by the mid kernels (Ttf/Mff); copy them with float4, all loads before the stores
    {
        constexpr int NT4 = NWT * RKS * 16, NM4 = NWT * NKS * 16, N4 = NT4 + NM4;
        const float4* srcT = reinterpret_cast<const float4*>(Ttf + (size_t)u * NWT * RKS * 64);
        const float4* srcM = reinterpret_cast<const float4*>(Mff + (size_t)u * NWT * NKS * 64);
        float4* dst = reinterpret_cast<float4*>(Tf);  // Mf follows Tf contiguously
        float4 tv[(N4 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i) {
            const int e = tid + i * 256;
            tv[i] = e < NT4 ? srcT[e] : (e < N4 ? srcM[e - NT4] : make_float4(0.f, 0.f, 0.f, 0.f));
        }
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i)
            if (tid + i * 256 < N4) dst[tid + i * 256] = tv[i];
    }
    for (int i = tid; i < NWT * 32; i += 256) k0s[i] = (i < NS) ? k0p[(size_t)u * NS + i] : 0.f;
    __syncthreads();
    STAMP(1);
    // wave-uniform bases + 32-bit lane offsets (saddr addressing): 64-bit per-row addresses were
    // hoisted out of the tile loop and spilled (200 B/lane of scratch at the 5-waves/SIMD budget)
    const float* __restrict__ eu = ext + (size_t)u * n * Bs;
    float* __restrict__ dyu = dy + (size_t)u * n * Bs;
    const float a1 = alpha[u], s1 = shift[u];
    const float mu = (float)mug[u];
    const float isg = (float)(1.0 / sig1[u]);
    for (int it = 0; it < PB_BTW; ++it) {
        const int bt = (blockIdx.x * 4 + wave) * PB_BTW + it;
        if (bt * 32 >= B) break;                       // wave-uniform
        const int b = bt * 32 + rc;
        const bool live = b < B;
        // raw pooled extremes of this lane's sequence, rows w = 2s + kk: the MFMA B operand (-q) is
        // derived from them, and so is the epilogue's ex -- the D layout wants rows
        // wt*32 + (g&3) + 8(g>>2) + 4kk, which this lane or its partner in the other half-wave holds
        float exr[NKS];
#pragma unroll
        for (int s = 0; s < NKS; ++s) exr[s] = eu[min(2 * s + kk, n - 1) * Bs + b];
#pragma unroll
        for (int s = 0; s < NKS; ++s) KEEP(exr[s]);
        const uint4 wv = bits[(size_t)u * Bs + b];
        const float dzb = dz[(size_t)u * Bs + b];
        const uint32_t wds[4] = {wv.x, wv.y, wv.z, wv.w};
        float sA = 0.f, sB = 0.f;
        if (it == 0) STAMP_AFTER_LOADS(2);
#pragma unroll
        for (int wt = 0; wt < NWT; ++wt) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(&k0s[wt * 32 + 8 * g + 4 * kk]);
                acc[4 * g] = -v.x; acc[4 * g + 1] = -v.y; acc[4 * g + 2] = -v.z; acc[4 * g + 3] = -v.w;
            }
#pragma unroll
            for (int s = 0; s < RKS; ++s) {
                const float e = (((wds[s >> 4] >> ((2 * s) & 31)) >> kk) & 1u) ? dzb : 0.f;
                acc = MFMA32(Tf[(wt * RKS + s) * 64 + lane], e, acc);
            }
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const float nq = (2 * s + kk < n) ? -qval(a1, exr[s], s1) : 0.f;
                acc = MFMA32(Mf[(wt * NKS + s) * 64 + lane], nq, acc);
            }
            if (it == 0 && wt == 0) STAMP(3);
            // D[w][b]: lane holds its sequence b, rows w = wt*32 + (g&3) + 8(g>>2) + 4kk.  Row w
            // lives in exr[w>>1] of the half-wave with kk = w&1 = g&1: own register for that half,
            // the partner's (lane ^ 32) for the other -- no second read of ext.
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int wbase = wt * 32 + (g & 3) + 8 * (g >> 2);     // row of the kk = 0 half
                const int s0 = min(wbase >> 1, NKS - 1), s1i = min((wbase + 4) >> 1, NKS - 1);
                float ex;
                if ((g & 1) == 0) {          // even rows are held by kk = 0 lanes
                    const float fromPartner = __shfl_xor(exr[s1i], 32, 64);
                    ex = kk ? fromPartner : exr[s0];
                } else {                     // odd rows by kk = 1 lanes
                    const float fromPartner = __shfl_xor(exr[s0], 32, 64);
                    ex = kk ? exr[s1i] : fromPartner;
                }
                const int w = wbase + 4 * kk;
                const float qv = qval(a1, ex, s1);
                const float dyv = (live && w < n) ? acc[g] * qv : 0.f;
                sA += dyv;
                sB = fmaf(dyv, (ex - mu) * isg, sB);
                if (w < n) dyu[w * Bs + b] = dyv;
            }
        }
        if (it == 0) STAMP(4);
        sA = wave_sum(sA);
        sB = wave_sum(sB);
        if (lane == 0) {
            float* d = S12p + ((size_t)u * (Bs / 32) + bt) * 2;
            d[0] = sA; d[1] = sB;
        }
    }
    STAMP(5);
}

template <int NQ>
static size_t passB_lds() {
    constexpr int NKS = (NQ + 1) / 2, NWT = (NQ + 31) / 32;
    return (size_t)(NWT * (FC_H / 2) * 64 + NWT * NKS * 64 + NWT * 32) * sizeof(float);
}

int launch_passB(explainn_ctx* c, int B, hipStream_t s) {
    const int tiles = (B + 31) / 32;
    const dim3 grid((tiles + 4 * PB_BTW - 1) / (4 * PB_BTW), c->U);
#define CALL(N)                                                                                  \
    hipLaunchKernelGGL(passB_kernel<N>, grid, dim3(256), passB_lds<N>(), s, c->ext, c->alpha,    \
                       c->shift, c->dz, c->bits, c->Ttf, c->Mff, c->k0p, c->mug, c->sig1, c->dy,    \
                       c->S12p, c->n, c->Bs, B)
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int fc_configure(explainn_ctx* c) {
#define CALL(N)                                                                              \
    if (passB_lds<N>() > 48 * 1024)                                                          \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&passB_kernel<N>),          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,              \
                                    (int)passB_lds<N>()))
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    return EXPLAINN_OK;
}
