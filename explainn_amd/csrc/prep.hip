// Per-unit "small algebra" of the forward pass: fold each BatchNorm into a
// scale/shift that the streaming kernels apply, using batch statistics obtained in closed form
// from moments instead of from a pass over the activations.
//
//   prep1    BN1 (architectures/__init__.py:79): mean/var of the conv output from the input
//            moments (m, G):  mu = cb + w.m,  var = w'Gw - (w.m)^2  -> alpha, shift (prep1_stats);
//            the filter taps laid out for the conv kernel come from prep1_tables.
//   qtrans   q = exp(alpha*ext+shift) re-laid sequence-major for scalar-operand consumers.
//   qmom     first/second moments of q over the batch (shifted by sequence 0 for conditioning).
//   prep2    BN2 (architectures/__init__.py:90): mean/var of FC1's output from the q moments
//            -> folded FC1 weights A2 and shift sh2.
#include "common.h"

__device__ __forceinline__ double block_sum_128(double v, double* red) {
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1];
}

// prep1 is split in two so that the filter-bank kernel does not wait for the input moments:
//   prep1_tables  filter taps re-laid for the export kernels (Wt) and as GEMM fragments (Wf, Wsg)   <- W only
//   prep1_stats   BatchNorm1 fold (alpha, shift), running statistics, G.w for the backward  <- W, m, G
__global__ __launch_bounds__(128) void prep1_tables_kernel(const float* __restrict__ conv_w,
                                                           const float* __restrict__ gamma1,
                                                           float* __restrict__ Wt,
                                                           uint16_t* __restrict__ Wf,
                                                           uint32_t* __restrict__ Wsg, int U, int k) {
    __shared__ float wsh[4 * MAX_K];
    filter_tables_unit(conv_w, gamma1, Wt, Wf, Wsg, U, k, blockIdx.x, threadIdx.x, 128, wsh);
}

template <bool TRAIN>
__global__ __launch_bounds__(128) void prep1_stats_kernel(
    const float* __restrict__ conv_w, const float* __restrict__ conv_b,
    const float* __restrict__ g1, const float* __restrict__ b1, float* __restrict__ rm,
    float* __restrict__ rv, int64_t* nbt, const double* __restrict__ G,
    const double* __restrict__ m, float* __restrict__ alpha, float* __restrict__ shift,
    double* __restrict__ mug, double* __restrict__ sig1, double* __restrict__ Gw, int U, int k,
    int B, int Lo) {
    __shared__ double wsh[4 * MAX_K];
    __shared__ double red[2];
    const int u = blockIdx.x, tid = threadIdx.x, K4 = 4 * k;
    if (u >= U) {
        if (tid == 0) { alpha[u] = 0.f; shift[u] = 0.f; }
        return;
    }
    if (!TRAIN) {
        if (tid == 0) {
            const double a = (double)g1[u] / sqrt((double)rv[u] + BN_EPS_D);
            alpha[u] = (float)a;
            shift[u] = (float)((double)b1[u] + a * ((double)conv_b[u] - (double)rm[u]));
        }
        return;
    }
    // everything the block reads is requested up front: the filter, this thread's entry of m, the
    // unit's BatchNorm parameters (thread 0 uses them at the very end) and G -- K4 x K4 doubles, 46 KB
    // at k = 19, staged in LDS 48 per thread per pass, so that k <= 19 is ONE memory round trip (16 per
    // pass were three, with m and the parameters two more behind the barrier; walking G from global
    // had been ten)
    const int mi = min(tid, K4 - 1);
    const float wme = conv_w[(size_t)u * K4 + mi];
    const double mme = m[mi];
    const float g1u = g1[u], b1u = b1[u], cbu = conv_b[u], rmu = rm[u], rvu = rv[u];
    for (int i = tid + 128; i < K4; i += 128) wsh[i] = (double)conv_w[(size_t)u * K4 + i];
    extern __shared__ double Gs[];
    {
        const int n2 = K4 * K4;
        constexpr int GQ = 48;
        for (int e0 = tid; e0 < n2; e0 += 128 * GQ) {
            double gv[GQ];
#pragma unroll
            for (int q = 0; q < GQ; ++q) gv[q] = G[min(e0 + q * 128, n2 - 1)];
#pragma unroll
            for (int q = 0; q < GQ; ++q)
                if (e0 + q * 128 < n2) Gs[e0 + q * 128] = gv[q];
        }
    }
    if (tid < K4) wsh[tid] = (double)wme;
    __syncthreads();
    double mu_p = 0, q_p = 0;
    for (int i = tid; i < K4; i += 128) {
        // (G w)[i] = sum_c G[c][i] w[c] (G is symmetric): column i, consecutive lanes = consecutive doubles
        double g0 = 0, g1s = 0;
        int c0 = 0;
        for (; c0 + 1 < K4; c0 += 2) {
            g0 = fma(Gs[c0 * K4 + i], wsh[c0], g0);
            g1s = fma(Gs[(c0 + 1) * K4 + i], wsh[c0 + 1], g1s);
        }
        if (c0 < K4) g0 = fma(Gs[c0 * K4 + i], wsh[c0], g0);
        const double gw = g0 + g1s;
        Gw[(size_t)u * K4 + i] = gw;
        mu_p = fma(wsh[i], i == tid ? mme : m[i], mu_p);
        q_p = fma(wsh[i], gw, q_p);
    }
    const double mu = block_sum_128(mu_p, red);
    const double wGw = block_sum_128(q_p, red);
    if (tid == 0) {
        double var = wGw - mu * mu;
        var = var > 0 ? var : 0;
        const double sg = sqrt(var + BN_EPS_D);
        const double a = (double)g1u / sg;
        alpha[u] = (float)a;
        shift[u] = (float)((double)b1u - a * mu);
        mug[u] = mu;
        sig1[u] = sg;
        const double N1 = (double)B * (double)Lo;
        rm[u] = (float)((1 - BN_MOM_D) * (double)rmu + BN_MOM_D * ((double)cbu + mu));
        rv[u] = (float)((1 - BN_MOM_D) * (double)rvu + BN_MOM_D * var * N1 / (N1 - 1));
        if (u == 0 && nbt) *nbt += 1;
    }
}

int launch_prep1_tables(explainn_ctx* c, const explainn_params* p, hipStream_t s) {
    hipLaunchKernelGGL(prep1_tables_kernel, dim3(c->U4), dim3(128), 0, s, p->conv_w, p->bn1_w, c->Wt,
                       c->Wf, c->Wsg, c->U, c->k);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int launch_prep1(explainn_ctx* c, const explainn_params* p, int B, bool train, hipStream_t s) {
    if (train)
        hipLaunchKernelGGL(prep1_stats_kernel<true>, dim3(c->U4), dim3(128),
                           (size_t)c->K4 * c->K4 * sizeof(double), s, p->conv_w,
                           p->conv_b, p->bn1_w, p->bn1_b, p->bn1_rm, p->bn1_rv, p->bn1_nbt, c->G,
                           c->m, c->alpha, c->shift, c->mug, c->sig1, c->Gw, c->U, c->k, B, c->Lo);
    else
        hipLaunchKernelGGL(prep1_stats_kernel<false>, dim3(c->U4), dim3(128), 0, s, p->conv_w,
                           p->conv_b, p->bn1_w, p->bn1_b, p->bn1_rm, p->bn1_rv, (int64_t*)nullptr,
                           c->G, c->m, c->alpha, c->shift, c->mug, c->sig1, c->Gw, c->U, c->k, B,
                           c->Lo);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// qmom: first/second moments of q over the batch on the exact-fp32 matrix core.
//   S1[w'] = sum_b (q[b,w'] - s[w']),   S2[w][w'] = sum_b (q[b,w] - s[w]) (q[b,w'] - s[w']),  s = q of sequence 0
// One wavefront per (unit, batch chunk, 32x32 tile of (w,w')).  The sequence index is the MFMA K
// dimension (two sequences per v_mfma_f32_32x32x2_f32), but ext is stored batch-fastest, so each
// super-tile of 64 sequences is fetched with coalesced row loads (lane = sequence), turned into
// q = exp(alpha*ext+shift) and transposed through a wave-private LDS tile [row][65]; the next
// super-tile's loads are in flight while the current one feeds the matrix core.
// ---------------------------------------------------------------------------------------------
typedef float f32x16q __attribute__((ext_vector_type(16)));
typedef float f32x4q __attribute__((ext_vector_type(4)));
#define QT_LD 65
#define QS_LD 66          // row stride of the small-n tile: operand reads (16 rows x 2 columns per 32 lanes) conflict-free

// n <= 32: v_mfma_f32_16x16x4_f32, the upper triangle of the (at most 2 x 2) tile grid; all 16 k-steps
// of a 64-sequence super-tile unrolled, so the LDS operand reads run far ahead of the matrix core.
// (The 32x32x2 version read two operands per 64-cycle MFMA eight steps ahead and still spent 9 K
// cycles per super-tile for 2 K cycles of MFMA.)  The four rows a lane holds of a tile are
// consecutive, so the mirrored half of the symmetric matrix is written with 16-byte stores.
template <int NQ>
__global__ __launch_bounds__(64) void qmom_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, float* __restrict__ qs0, float* __restrict__ S1p,
    float* __restrict__ S2p, int n, int Bs, int B, int QCH) {
    constexpr int NS = ns_stride(NQ), NW16 = fc_nw16(NQ), ROWS = 16 * NW16;
    constexpr int NP = NW16 * (NW16 + 1) / 2;
    __shared__ float tq[ROWS * QS_LD];
    const int u = blockIdx.y, ch = blockIdx.x, lane = threadIdx.x;
    const int c = lane & 15, g = lane >> 4;
    const int per = ((((B + QCH - 1) / QCH) + 63) / 64) * 64;
    const int bbeg = ch * per, bend = min(B, bbeg + per);
    const float a1 = alpha[u], sh1 = shift[u];
    const float* __restrict__ eu = ext + (size_t)u * n * Bs;
    // shift = q of sequence 0 (for conditioning; any constant per row would do)
    // (n lies in (NQLO, NQ]: rows up to NQLO always exist, rows from NQ on never do, only the rows
    // between need a run-time test -- as a test on every row the compiler made two branches per row;
    // and the row stride sits in a VGPR it cannot see through, or it hoists 32 row pointers out of
    // the loop into 64 SGPRs, spills them to lanes and reads each back with wait states in front
    // of its load)
    constexpr int NQLO = nq_lower(NQ);
    uint32_t rstride = (uint32_t)Bs * 4u;
    asm volatile("" : "+v"(rstride));
    const char* __restrict__ eb = reinterpret_cast<const char*>(eu);
    float s0[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
        s0[i] = i < NQ ? *reinterpret_cast<const float*>(eb + __umul24(rstride, (uint32_t)min(i, n - 1))) : 0.f;
    f32x4q acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = f32x4q{0.f, 0.f, 0.f, 0.f};
    float s1[NW16];
#pragma unroll
    for (int j = 0; j < NW16; ++j) s1[j] = 0.f;
    // a tile's rows are REQUESTED (issue) one tile ahead and TURNED INTO q - s0 (finish) behind the
    // MFMAs of the tile in front: as one lambda the transform sat right behind the loads and the
    // "prefetch" waited on the spot
    float rq[ROWS];
    bool rlive = false;
    auto issue = [&](int b0) {
        const int b = b0 + lane;
        rlive = b < bend;
        const uint32_t boff = (uint32_t)(rlive ? b : bbeg) * 4u;
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
            rq[i] = i < NQ ? *reinterpret_cast<const float*>(eb + (__umul24(rstride, (uint32_t)min(i, n - 1)) + boff)) : 0.f;
    };
    auto finish = [&]() {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) KEEP(rq[i]);
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if (i <= NQLO) rq[i] = rlive ? qval(a1, rq[i], sh1) - s0[i] : 0.f;
            else if (i >= NQ) rq[i] = 0.f;
            else rq[i] = (rlive && i < n) ? qval(a1, rq[i], sh1) - s0[i] : 0.f;
        }
    };
    STAMP(0);
    // (the first tile's rows are requested before the shift row is turned into q: one round trip
    // for both)
    if (bbeg < bend) issue(bbeg);
#pragma unroll
    for (int i = 0; i < ROWS; ++i) KEEP(s0[i]);
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
        s0[i] = i <= NQLO ? qval(a1, s0[i], sh1) : (i >= NQ ? 0.f : (i < n ? qval(a1, s0[i], sh1) : 0.f));
    if (ch == 0 && lane == 0) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
            if (i < NS) qs0[(size_t)u * NS + i] = s0[i];
    }
    if (bbeg < bend) finish();
    for (int b0 = bbeg; b0 < bend; b0 += 64) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) tq[i * QS_LD + lane] = rq[i];
        if (b0 == bbeg) STAMP(1);
        const bool more = b0 + 64 < bend;
        if (more) issue(b0 + 64);                       // in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float av[NW16];
#pragma unroll
            for (int j = 0; j < NW16; ++j) {
                av[j] = tq[(16 * j + c) * QS_LD + 4 * s + g];
                s1[j] += av[j];
            }
            int p = 0;
#pragma unroll
            for (int j = 0; j < NW16; ++j)
#pragma unroll
                for (int j2 = j; j2 < NW16; ++j2, ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], av[j2], acc[p], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) finish();
        __builtin_amdgcn_wave_barrier();
    }
    STAMP(2);
    // D[w][w'] of tile (j, j2): lane holds rows w = 16j + 4g + i, column w' = 16j2 + c
    float* out = S2p + ((size_t)u * QCH + ch) * NS * NS;
    int p = 0;
#pragma unroll
    for (int j = 0; j < NW16; ++j)
#pragma unroll
        for (int j2 = j; j2 < NW16; ++j2, ++p) {
            const int w = 16 * j + 4 * g, wp = 16 * j2 + c;
            if (wp < NS && w < NS) {
                // S2[w'][w .. w+3] (symmetric): one 16-byte store; NS is a multiple of 4
                *reinterpret_cast<float4*>(&out[(size_t)wp * NS + w]) =
                    make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
                if (j2 != j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[(size_t)(w + i) * NS + wp] = acc[p][i];
                }
            }
        }
#pragma unroll
    for (int j = 0; j < NW16; ++j) {
        float sv = s1[j];
        sv += __shfl_xor(sv, 16, 64);
        sv += __shfl_xor(sv, 32, 64);
        const int w = 16 * j + c;
        if (g == 0 && w < NS) S1p[((size_t)u * QCH + ch) * NS + w] = sv;
    }
    STAMP(3);
}

// Large pooled lengths (n > 32): one 4-wave block per (unit, batch chunk).  The block stages the rows
// of a 64-sequence super-tile ONCE in LDS (shifted q values) and its four waves share the NP tile
// pairs of the symmetric moment matrix (upper triangle, mirrored on output), pair p going to wave
// p mod 4.  (One wave per chunk doing all pairs needed 15 accumulator tiles = 240 VGPRs and a 36 KB
// tile per WAVE: one wave per SIMD, 236 us at C4 against a 68 us MFMA roof.)
template <int NQ>
__global__ __launch_bounds__(256, 2) void qmom_big_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, float* __restrict__ qs0, float* __restrict__ S1p,
    float* __restrict__ S2p, int n, int Bs, int B, int QCH) {
    constexpr int NS = ns_stride(NQ), NWT = (NQ + 31) / 32, NP = NWT * (NWT + 1) / 2;
    constexpr int NPW = (NP + 3) / 4;                  // tile pairs (accumulators) per wave
    extern __shared__ float qsm[];                     // rows [NWT*32][65], then s0 [NWT*32]
    float* tile = qsm;
    float* s0s = qsm + NWT * 32 * QT_LD;
    const int u = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rc = lane & 31, kk = lane >> 5;
    const int per = ((((B + QCH - 1) / QCH) + 63) / 64) * 64;
    const int bbeg = ch * per, bend = min(B, bbeg + per);
    const float a1 = alpha[u], sh1 = shift[u];
    const float* eu = ext + (size_t)u * n * Bs;
    for (int w = tid; w < NWT * 32; w += 256) {
        const float raw = eu[(size_t)min(w, n - 1) * Bs];
        const float sv = (w < n) ? qval(a1, raw, sh1) : 0.f;
        s0s[w] = sv;
        if (ch == 0 && w < NS) qs0[(size_t)u * NS + w] = sv;
    }
    __syncthreads();
    f32x16q acc[NPW];
#pragma unroll
    for (int p = 0; p < NPW; ++p)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[p][g] = 0.f;
    float s1[NWT];
#pragma unroll
    for (int t = 0; t < NWT; ++t) s1[t] = 0.f;
    for (int b0 = bbeg; b0 < bend; b0 += 64) {
        const int b = b0 + lane;
        const bool live = b < bend;
        const int bcl = live ? b : bbeg;
        __syncthreads();
        // the four waves stage the row groups of 32 between them (shifted values)
        for (int t = wave; t < 2 * NWT; t += 4) {       // half row groups: 16 loads in flight per lane
            float r[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = eu[(size_t)min(t * 16 + i, n - 1) * Bs + bcl];
#pragma unroll
            for (int i = 0; i < 16; ++i) KEEP(r[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int w = t * 16 + i;
                tile[w * QT_LD + lane] = (live && w < n) ? qval(a1, r[i], sh1) - s0s[w] : 0.f;
            }
        }
        __syncthreads();
        if constexpr (NPW <= 2) {
            // few MFMAs per k-step and wave (n <= 96): the LDS latency would be exposed step by step
            // always the full 32 k-steps (columns past the chunk end hold zeros), four at a time with the
            // LDS operands of this wave's pairs read first
    #pragma unroll 1
            for (int s0 = 0; s0 < 32; s0 += 4) {
                float av[NPW][4], bv[NPW][4];
                int p = 0;
    #pragma unroll
                for (int t = 0; t < NWT; ++t)
    #pragma unroll
                    for (int t2 = t; t2 < NWT; ++t2, ++p) {
                        if ((p & 3) != wave) continue;     // wave-uniform
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int col = 2 * (s0 + q) + kk;
                            av[p >> 2][q] = tile[(t * 32 + rc) * QT_LD + col];
                            bv[p >> 2][q] = (t2 == t) ? av[p >> 2][q] : tile[(t2 * 32 + rc) * QT_LD + col];
                        }
                    }
                p = 0;
    #pragma unroll
                for (int t = 0; t < NWT; ++t)
    #pragma unroll
                    for (int t2 = t; t2 < NWT; ++t2, ++p) {
                        if ((p & 3) != wave) continue;
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (t2 == t) s1[t] += av[p >> 2][q];   // the diagonal pair's owner also sums the rows
                            acc[p >> 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p >> 2][q], bv[p >> 2][q],
                                                                               acc[p >> 2], 0, 0, 0);
                        }
                    }
            }
    
        } else {
            // four MFMAs per k-step keep the pipe busy by themselves (and batching costs occupancy)
            for (int s = 0; s < 32; ++s) {
                const int col = 2 * s + kk;
                int p = 0;
#pragma unroll
                for (int t = 0; t < NWT; ++t)
#pragma unroll
                    for (int t2 = t; t2 < NWT; ++t2, ++p) {
                        if ((p & 3) != wave) continue;     // wave-uniform
                        const float av = tile[(t * 32 + rc) * QT_LD + col];
                        const float bv = (t2 == t) ? av : tile[(t2 * 32 + rc) * QT_LD + col];
                        if (t2 == t) s1[t] += av;          // the diagonal pair's owner also sums the rows
                        acc[p >> 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[p >> 2], 0, 0, 0);
                    }
            }
        }
    }
    // D[w][w']: lane holds column w' = t2*32+rc, rows w = t*32 + (g&3) + 8(g>>2) + 4kk; mirror it
    float* out = S2p + ((size_t)u * QCH + ch) * NS * NS;
    int p = 0;
#pragma unroll
    for (int t = 0; t < NWT; ++t)
#pragma unroll
        for (int t2 = t; t2 < NWT; ++t2, ++p) {
            if ((p & 3) != wave) continue;
            const int wB = t2 * 32 + rc;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int w = t * 32 + (g & 3) + 8 * (g >> 2) + 4 * kk;
                if (w < NS && wB < NS) {
                    out[(size_t)w * NS + wB] = acc[p >> 2][g];
                    if (t2 != t) out[(size_t)wB * NS + w] = acc[p >> 2][g];
                }
            }
            if (t2 == t) {
                const float sv = s1[t] + __shfl_xor(s1[t], 32, 64);
                const int w = t * 32 + rc;
                if (kk == 0 && w < NS) S1p[((size_t)u * QCH + ch) * NS + w] = sv;
            }
        }
}

template <int NQ>
static size_t qmom_big_lds() {
    constexpr int NWT = (NQ + 31) / 32;
    return (size_t)(NWT * 32 * QT_LD + NWT * 32) * sizeof(float);
}

int launch_qmoments(explainn_ctx* c, int B, hipStream_t s) {
#define CALL(N)                                                                                  \
    if constexpr ((N) <= 32)      /* (constexpr: only the form a bucket uses is instantiated) */  \
        hipLaunchKernelGGL(qmom_kernel<N>, dim3(c->QCH, c->U, 1), dim3(64), 0, s, c->ext,        \
                           c->alpha, c->shift, c->qs0, c->qS1p, c->qS2p, c->n, c->Bs, B, c->QCH); \
    else                                                                                         \
        hipLaunchKernelGGL(qmom_big_kernel<N>, dim3(c->QCH, c->U), dim3(256), qmom_big_lds<N>(), s, \
                           c->ext, c->alpha, c->shift, c->qs0, c->qS1p, c->qS2p, c->n, c->Bs, B,  \
                           c->QCH)
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// prep2: one 256-thread block per unit; C and V1 staged in LDS; two threads per hidden channel
// ---------------------------------------------------------------------------------------------
static size_t prep2_lds(int n, int NS) {
    return (size_t)NS * sizeof(double) + ((size_t)n * n + (size_t)FC_H * (n + 1)) * sizeof(float);
}

template <bool TRAIN>
__global__ __launch_bounds__(1024) void prep2_kernel(
    const float* __restrict__ fc1_w, const float* __restrict__ fc1_b,
    const float* __restrict__ g2, const float* __restrict__ b2, float* __restrict__ rm2,
    float* __restrict__ rv2, int64_t* nbt, const float* __restrict__ qs0,
    const float* __restrict__ S1p, const float* __restrict__ S2p, double* __restrict__ qbar,
    float* __restrict__ VC, float* __restrict__ A2, float* __restrict__ A2f,
    float* __restrict__ sh2, float* __restrict__ sig2, int n, int NS, int NK4Q, int B, int QCH,
    uint32_t* __restrict__ A2h, int KS) {
    extern __shared__ double sm[];            // qb[NS] (double) | Cs[n][n] | V1s[100][n+1]
    const int u = blockIdx.x, tid = threadIdx.x;
    constexpr int NT = 1024;
    if (!TRAIN) {
        for (int e = tid; e < FC_H * NS; e += NT) {
            const int r = e / NS, w = e % NS, ch = u * FC_H + r;
            const double inv = (double)g2[ch] / sqrt((double)rv2[ch] + BN_EPS_D);
            A2[(size_t)ch * NS + w] = (w < n) ? (float)(inv * (double)fc1_w[(size_t)ch * n + w]) : 0.f;
            if (w == 0)
                sh2[ch] = (float)((double)b2[ch] + inv * ((double)fc1_b[ch] - (double)rm2[ch]));
        }
    } else {
        double* qb = sm;
        STAMP(0);
        float* Cs = (float*)(sm + NS);
        float* V1s = Cs + n * n;
        const int ld = n + 1;
        const double invB = 1.0 / (double)B;
        if (n * n + n <= NT && FC_H * n <= 4 * NT && QCH <= 8) {
            // small n (the headline shape): ALL of the block's inputs in ONE memory round trip -- V1
            // (four elements per thread), the S1 and S2 chunk partials (eight each) and the shift --
            // held in registers across the barrier that the covariance needs.  They were three
            // dependent round trips (V1 + S1 | S2 | shift), 40 % of the block's cycles by stamps.
            // 21 values per thread: still under the 64 registers that let two blocks share a CU
            // (the all-in-one batch tried earlier also merged the k-loop operands and took 96).
            const float* __restrict__ w1u = fc1_w + (size_t)u * FC_H * n;
            const float* __restrict__ s1u = S1p + (size_t)u * QCH * NS;
            const float* __restrict__ s2u = S2p + (size_t)u * QCH * NS * NS;
            float v[4], pv[8], q0;            // pv: S2 partials (threads < n^2) or S1 partials (the last n)
            const int tot = FC_H * n;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = w1u[(uint32_t)min(tid + q * NT, tot - 1)];
            const int w1 = min(NT - 1 - tid, n - 1);              // the last threads take the S1 sums
            const int e2 = min(tid, n * n - 1), w2 = e2 / n, wp2 = e2 - w2 * n;
            const bool takes1 = NT - 1 - tid < n;
            const float* __restrict__ pbase = takes1 ? s1u + w1 : s2u + (w2 * NS + wp2);
            const uint32_t pstride = takes1 ? (uint32_t)NS : (uint32_t)(NS * NS);
#pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = pbase[(uint32_t)min(i, QCH - 1) * pstride];
            q0 = qs0[(size_t)u * NS + min(tid, n - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q) KEEP(v[q]);
#pragma unroll
            for (int i = 0; i < 8; ++i) KEEP(pv[i]);
            KEEP(q0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + q * NT;
                if (e < tot) V1s[(e / n) * ld + (e % n)] = v[q];
            }
            if (takes1) {
                double s1 = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) s1 += (i < QCH) ? (double)pv[i] : 0.0;
                qb[NT - 1 - tid] = s1 * invB;     // mean of (q - s); the shift is added below
            }
            __syncthreads();
            if (tid < n * n) {
                double s2 = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) s2 += (i < QCH) ? (double)pv[i] : 0.0;
                // S2 holds sum (q_w - s_w)(q_w' - s_w'); qb = mean of (q - s)
                Cs[tid] = (float)(s2 * invB - qb[w2] * qb[wp2]);
            }
            __syncthreads();
            if (tid < n) {
                const double vq = qb[tid] + (double)q0;
                qbar[(size_t)u * NS + tid] = vq;
                qb[tid] = vq;                     // from here on qb = mean of q
            }
        } else {
            // V1 into LDS, four elements per thread per pass with their loads issued together (a plain
            // strided loop with a runtime trip count is one exposed round trip per iteration)
            for (int e0 = tid; e0 < FC_H * n; e0 += 4 * NT) {
                float v[4];
    #pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fc1_w[(size_t)u * FC_H * n + min(e0 + q * NT, FC_H * n - 1)];
    #pragma unroll
                for (int q = 0; q < 4; ++q) KEEP(v[q]);
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = e0 + q * NT;
                    if (e < FC_H * n) V1s[(e / n) * ld + (e % n)] = v[q];
                }
            }
            // combine the chunk partials (fixed order -> deterministic), eight loads in flight; the last
            // threads take this so that it overlaps the staging above.  (Issuing these, the covariance
            // partials below and V1 in ONE batch per thread was tried: shorter waves, longer kernel --
            // +6 us on the step with the same change in mid_fused.)
            for (int w = NT - 1 - tid; w < n; w += NT) {
                double s1 = 0;
                for (int c0 = 0; c0 < QCH; c0 += 8) {
                    float pv[8];
    #pragma unroll
                    for (int i = 0; i < 8; ++i) pv[i] = S1p[((size_t)u * QCH + min(c0 + i, QCH - 1)) * NS + w];
    #pragma unroll
                    for (int i = 0; i < 8; ++i) KEEP(pv[i]);
    #pragma unroll
                    for (int i = 0; i < 8; ++i) s1 += (c0 + i < QCH) ? (double)pv[i] : 0.0;
                }
                qb[w] = s1 * invB;                // mean of (q - s); the shift is added below
            }
            __syncthreads();
            for (int e = tid; e < n * n; e += NT) {
                const int w = e / n, wp = e % n;
                double s2 = 0;
                for (int c0 = 0; c0 < QCH; c0 += 8) {     // eight partials in flight, fixed-order sum
                    float pv[8];
    #pragma unroll
                    for (int i = 0; i < 8; ++i)
                        pv[i] = S2p[(((size_t)u * QCH + min(c0 + i, QCH - 1)) * NS + w) * NS + wp];
    #pragma unroll
                    for (int i = 0; i < 8; ++i) KEEP(pv[i]);
    #pragma unroll
                    for (int i = 0; i < 8; ++i) s2 += (c0 + i < QCH) ? (double)pv[i] : 0.0;
                }
                // S2 holds sum (q_w - s_w)(q_w' - s_w'); qb = mean of (q - s)
                const double cov = s2 * invB - qb[w] * qb[wp];
                Cs[e] = (float)cov;
            }
            __syncthreads();
            for (int w = tid; w < n; w += NT) {
                const double v = qb[w] + (double)qs0[(size_t)u * NS + w];
                qbar[(size_t)u * NS + w] = v;
                qb[w] = v;                        // from here on qb = mean of q
            }
        }
        __syncthreads();
        STAMP(1);
        // eight threads per hidden channel: each takes every 8th column w' of the quadratic form
        const int r = tid >> 3, part = tid & 7;
        const int rr = r < FC_H ? r : FC_H - 1;
        const float* v1 = V1s + rr * ld;
        double var = 0;
        {
            // VC = V1 . C on the matrix cores (fp32 MFMA, 32x32 tiles: 4 row tiles of hidden channels
            // x NWT column tiles, K = n in steps of 2), one tile per wave per pass.  The product is
            // also what the backward needs (hq in the mid kernels), so it goes to global memory.
            // (An fp64 VALU version of this quadratic form was LDS-latency bound: 4 us per step more.)
            const int wave = tid >> 6, lane = tid & 63, rc = lane & 31, kk = lane >> 5;
            const int NWT = (n + 31) >> 5;
            for (int tile = wave; tile < 4 * NWT; tile += NT / 64) {
                const int t = tile / NWT, wt = tile % NWT;
                const int ra = 32 * t + rc, wb = 32 * wt + rc;
                const float* arow = V1s + min(ra, FC_H - 1) * ld;
                const float* bcol = Cs + min(wb, n - 1);
                const bool alive = ra < FC_H, blive = wb < n;
                f32x16q acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.f;
                for (int s0 = 0; s0 < (n + 1) / 2; s0 += 4) {     // four k-steps per pass, operands first
                    float av[4], bv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int wc = min(2 * (s0 + q) + kk, n - 1);
                        av[q] = arow[wc];
                        bv[q] = bcol[wc * n];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool on = 2 * (s0 + q) + kk < n;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((alive && on) ? av[q] : 0.f,
                                                                   (blive && on) ? bv[q] : 0.f, acc, 0, 0, 0);
                    }
                }
                if (blive) {
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int ro = 32 * t + (g & 3) + 8 * (g >> 2) + 4 * kk;
                        if (ro < FC_H) VC[((size_t)u * FC_H + ro) * NS + wb] = acc[g];
                    }
                }
            }
            __syncthreads();                  // this block's VC writes are visible to its own reads
            for (int wp = part; wp < n; wp += 8)
                var = fma((double)VC[((size_t)u * FC_H + rr) * NS + wp], (double)v1[wp], var);
        }
        var += __shfl_xor(var, 1, 64); var += __shfl_xor(var, 2, 64); var += __shfl_xor(var, 4, 64);
        const int ch = u * FC_H + rr;
        var = var > 0 ? var : 0;
        const double sg = sqrt(var + BN_EPS_D);
        const double inv = (double)g2[ch] / sg;
        double shp = 0, meanp = 0;
        for (int w = part; w < NS; w += 8) {
            float a = 0.f;
            if (w < n) {
                a = (float)(inv * (double)v1[w]);
                shp = fma((double)a, qb[w], shp);
                meanp = fma((double)v1[w], qb[w], meanp);
            }
            if (r < FC_H) {
                A2[(size_t)ch * NS + w] = a;
                // this (channel, w) is this thread's alone: V1's LDS copy now carries A2, which the
                // fragment emission below reads instead of going back to global memory
                if (w < n) V1s[rr * ld + w] = a;
            }
        }
        shp += __shfl_xor(shp, 1, 64); shp += __shfl_xor(shp, 2, 64); shp += __shfl_xor(shp, 4, 64);
        meanp += __shfl_xor(meanp, 1, 64); meanp += __shfl_xor(meanp, 2, 64);
        meanp += __shfl_xor(meanp, 4, 64);
        if (r < FC_H && part == 0) {
            const double mean = (double)fc1_b[ch] + meanp;
            sh2[ch] = (float)((double)b2[ch] - shp);
            sig2[ch] = (float)sg;
            rm2[ch] = (float)((1 - BN_MOM_D) * (double)rm2[ch] + BN_MOM_D * mean);
            rv2[ch] = (float)((1 - BN_MOM_D) * (double)rv2[ch] +
                              BN_MOM_D * var * (double)B / (double)(B - 1));
        }
        if (u == 0 && tid == 0 && nbt) *nbt += 1;
        STAMP(2);
    }
    __syncthreads();
    const float* a2lds = TRAIN ? reinterpret_cast<const float*>(sm + NS) + n * n : nullptr;   // = V1s
    if (A2h) {
        // the same weights for fc_fwd's bf16 form: three bf16 pieces (hi + mid + lo = the weight,
        // exactly) in the A-fragment order of v_mfma_f32_16x16x32_bf16, two elements per word:
        // A2h[(((t*KS + ks)*3 + piece)*64 + l)*4 + jp] = pieces of A2[16t + (l&15)][32ks + 8(l>>4) + 2jp (+1)]
        for (int i = tid; i < FC_MT * KS * 256; i += NT) {
            const int jp = i & 3, l = (i >> 2) & 63, ks = (i >> 8) % KS, t = (i >> 8) / KS;
            const int r = 16 * t + (l & 15), w = 32 * ks + 8 * (l >> 4) + 2 * jp;
            uint32_t pc[3] = {0u, 0u, 0u};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v = 0.f;
                if (r < FC_H && w + h < n)
                    v = TRAIN ? a2lds[r * (n + 1) + w + h] : A2[((size_t)u * FC_H + r) * NS + w + h];
                const uint32_t hb = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(hb);
                const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(mb);
                pc[0] |= (hb >> 16) << (16 * h);
                pc[1] |= (mb >> 16) << (16 * h);
                pc[2] |= (__float_as_uint(r2) >> 16) << (16 * h);
            }
            uint32_t* dst = A2h + ((size_t)(u * FC_MT + t) * KS + ks) * 3 * 256 + l * 4 + jp;
            dst[0] = pc[0]; dst[256] = pc[1]; dst[512] = pc[2];
        }
    }
    if (A2f) {
        // ... and in MFMA 16x16x4 A-fragment order (large n, the single-launch eval), four k-steps per float4:
        // A2f[(((t*NK4Q + sq)*64 + l)*4 + e] = A2[16t + (l&15)][4*(4sq+e) + (l>>4)]
        for (int i = tid; i < FC_MT * NK4Q * 256; i += NT) {
            const int e = i & 3, l = (i >> 2) & 63, sq = (i >> 8) % NK4Q, t = (i >> 8) / NK4Q;
            const int r = 16 * t + (l & 15), w = 4 * (4 * sq + e) + (l >> 4);
            float v = 0.f;
            if (r < FC_H && w < n) v = TRAIN ? a2lds[r * (n + 1) + w] : A2[((size_t)u * FC_H + r) * NS + w];
            A2f[(size_t)u * FC_MT * NK4Q * 256 + i] = v;
        }
    }
    STAMP(3);
}

int launch_prep2(explainn_ctx* c, const explainn_params* p, int B, bool train, hipStream_t s) {
    // fragment image of the folded weights: the bf16 pieces where fc_fwd runs on the bf16 matrix
    // core (n <= FC_BF_MAXN), the fp32 fragments for the large-n fc_fwd
    uint32_t* a2h = c->NQ <= FC_BF_MAXN ? reinterpret_cast<uint32_t*>(c->A2h) : nullptr;
    float* a2f = a2h ? nullptr : c->A2f;
    if (train)
        hipLaunchKernelGGL(prep2_kernel<true>, dim3(c->U), dim3(1024), prep2_lds(c->n, c->NS), s,
                           p->fc1_w, p->fc1_b, p->bn2_w, p->bn2_b, p->bn2_rm, p->bn2_rv, p->bn2_nbt,
                           c->qs0, c->qS1p, c->qS2p, c->qbar, c->VC, c->A2, a2f, c->sh2, c->sig2,
                           c->n, c->NS, fc_nk4q(c->NQ), B, c->QCH, a2h, fc_ks32(c->NQ));
    else
        hipLaunchKernelGGL(prep2_kernel<false>, dim3(c->U), dim3(1024), 0, s, p->fc1_w, p->fc1_b,
                           p->bn2_w, p->bn2_b, p->bn2_rm, p->bn2_rv, (int64_t*)nullptr, c->qs0,
                           c->qS1p, c->qS2p, c->qbar, c->VC, c->A2, a2f, c->sh2, c->sig2, c->n,
                           c->NS, fc_nk4q(c->NQ), B, c->QCH, a2h, fc_ks32(c->NQ));
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// dynamic LDS above 64 KiB has to be opted into per kernel (n >= 127 needs it for the C tile)
int prep_configure(explainn_ctx* c) {
#define CALL(N)                                                                               \
    if ((N) > 32 && qmom_big_lds<N>() > 48 * 1024)                                            \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&qmom_big_kernel<N>),        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,               \
                                    (int)qmom_big_lds<N>()))
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    if ((size_t)c->K4 * c->K4 * sizeof(double) > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&prep1_stats_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)((size_t)c->K4 * c->K4 * sizeof(double))));
    const size_t sm = prep2_lds(c->n, c->NS);
    if (sm > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&prep2_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    return EXPLAINN_OK;
}
