// The narrow end of the network: BatchNorm3 + ReLU over the batch (architectures/__init__.py:99-100),
// the final nn.Linear(U,T) combiner (:104), the loss (:446-456) and their backward.  All arrays
// here are (U x B) or (B x T) -- a few MB at most -- so these are plain reduction kernels.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// Many tasks (T > HEAD_GEMM_MIN_T: configs C3-C5 have 50-164): the combiner and its backward are
// real GEMMs -- logits = o^T Wf^T (K = U), d o = dl Wf (K = T), d Wf = dl^T o^T (K = B) -- and run
// on the fp32 MFMA.  One 256-thread block per 32x32 output tile; its four waves split K in four
// contiguous quarters (each half-wave takes half a quarter: the order of a sum is free, so K is
// permuted to give every lane a contiguous run), partial tiles are added in fixed order through LDS.
// Operands are read straight from global in batches of eight per lane (they are L2-resident:
// at most a few MB).  *_KC = "K-contiguous": element (row, k) at ptr[row*ld + k]; otherwise
// element (k, row) at ptr[k*ld + row] (coalesced across the 32 rows of a tile).
// ---------------------------------------------------------------------------------------------
typedef float f32x16h __attribute__((ext_vector_type(16)));
enum { EPI_PLAIN = 0, EPI_BIAS = 1, EPI_GW = 2 };   // EPI_GW: gridDim.y batch chunks, partial tiles to D[chunk][M][N]

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm32_kernel(const float* __restrict__ A, int lda,
                                                     const float* __restrict__ Bm, int ldb,
                                                     float* __restrict__ D, int ldd, int M, int N,
                                                     int K, const float* __restrict__ bias,
                                                     float* __restrict__ extra) {
    __shared__ float part[3][16][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, rc = lane & 31, kk = lane >> 5;
    const int tiles_n = (N + 31) >> 5;
    const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
    const int i = ti * 32 + rc, j = tj * 32 + rc;
    const int ic = min(i, M - 1);
    // EPI_GW: column N-1 is a virtual all-ones row of B (it yields the bias gradient)
    const bool ones_col = EPI == EPI_GW && j == N - 1;
    const int jc = min(j, (EPI == EPI_GW ? N - 2 : N - 1));
    // (EPI_GW: this workgroup's chunk [kb0, kb1) of K; otherwise all of K)
    const int Kc = ((K + (int)gridDim.y - 1) / (int)gridDim.y + 7) & ~7;
    const int kb0 = min((int)blockIdx.y * Kc, K), kb1 = min(kb0 + Kc, K);
    const int Kq = (kb1 - kb0 + 3) >> 2;               // this wave's quarter [kq0, kq1)
    const int kq0 = min(kb0 + wave * Kq, kb1), kq1 = min(kq0 + Kq, kb1);
    const int Kh = (kq1 - kq0 + 1) >> 1;               // this half-wave's run [k0, kend)
    const int k0 = kq0 + kk * Kh, kend = min(k0 + Kh, kq1);
    const float* ap = A_KC ? A + (size_t)ic * lda : A + ic;
    const float* bp = B_KC ? Bm + (size_t)jc * ldb : Bm + jc;
    f32x16h acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    for (int s0 = 0; s0 < Kh; s0 += 8) {               // Kh is wave-uniform
        float a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kc = min(k0 + s0 + q, K - 1);
            a[q] = A_KC ? ap[kc] : ap[(size_t)kc * lda];
            b[q] = B_KC ? bp[kc] : bp[(size_t)kc * ldb];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { KEEP(a[q]); KEEP(b[q]); }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool live = k0 + s0 + q < kend;
            const float av = (live && i < M) ? a[q] : 0.f;
            const float bv = (live && j < N) ? (ones_col ? 1.f : b[q]) : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) part[wave - 1][g][lane] = acc[g];
    }
    __syncthreads();
    if (wave == 0 && j < N) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = ti * 32 + (g & 3) + 8 * (g >> 2) + 4 * kk;
            if (row >= M) continue;
            float v = ((acc[g] + part[0][g][lane]) + part[1][g][lane]) + part[2][g][lane];
            if (EPI == EPI_BIAS) v += bias[j];
            if (EPI == EPI_GW) D[((size_t)blockIdx.y * M + row) * N + j] = v;     // (bias column included)
            else D[(size_t)row * ldd + j] = v;
        }
    }
}

// dlT[t][b] = dl[b][t]   (the GEMMs want the batch index contiguous)
__global__ __launch_bounds__(256) void transpose_dl_kernel(const float* __restrict__ dl,
                                                           float* __restrict__ dlT, int B, int T,
                                                           int Bs) {
    __shared__ float tile[32][33];
    const int b0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (b0 + r < B && t0 + tx < T) ? dl[(size_t)(b0 + r) * T + t0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (t0 + r < T && b0 + tx < B) dlT[(size_t)(t0 + r) * Bs + b0 + tx] = tile[tx][r];
}


__device__ __forceinline__ double block_sum_256(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// three sums behind ONE pair of barriers (head_bwd: the two BatchNorm3 sums and the combiner-weight
// gradient of the register path used to take three round trips)
__device__ __forceinline__ void block_sum3_256(double& a, double& b, double& c, double* red12) {
    a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        red12[w] = a; red12[4 + w] = b; red12[8 + w] = c;
    }
    __syncthreads();
    a = (red12[0] + red12[1]) + (red12[2] + red12[3]);
    b = (red12[4] + red12[5]) + (red12[6] + red12[7]);
    c = (red12[8] + red12[9]) + (red12[10] + red12[11]);
}

// The register path of head_fwd_train: this thread's RBX z values stay in registers through the three
// passes (mean, variance, normalise) -- one batch of loads instead of three passes of dependent round
// trips over the same row.  RBX = 4 (batch <= 1024) or 16 (batch <= 4096: configuration C3).
template <int RBX>
__device__ __forceinline__ void head_fwd_train_regs(const float* __restrict__ zu, float* __restrict__ zhu,
                                                    float* __restrict__ ou, float gam, float bet, int B,
                                                    int tid, double* red, double& mean, double& var,
                                                    double& sg) {
    float zr[RBX];
#pragma unroll
    for (int i = 0; i < RBX; ++i) zr[i] = zu[min(tid + 256 * i, B - 1)];
#pragma unroll
    for (int i = 0; i < RBX; ++i) KEEP(zr[i]);
    double s = 0;
#pragma unroll
    for (int i = 0; i < RBX; ++i) s += (tid + 256 * i < B) ? (double)zr[i] : 0.0;
    mean = block_sum_256(s, red) / (double)B;
    double v = 0;
#pragma unroll
    for (int i = 0; i < RBX; ++i)
        if (tid + 256 * i < B) { const double d = (double)zr[i] - mean; v = fma(d, d, v); }
    var = block_sum_256(v, red) / (double)B;
    sg = sqrt(var + BN_EPS_D);
    const float meanf = (float)mean, isg = (float)(1.0 / sg);
#pragma unroll
    for (int i = 0; i < RBX; ++i) {
        const int b = tid + 256 * i;
        if (b < B) {
            const float zh = (zr[i] - meanf) * isg;
            zhu[b] = zh;
            ou[b] = fmaxf(fmaf(gam, zh, bet), 0.f);
        }
    }
}

// one block per unit: batch statistics of z, then zhat / o
__global__ __launch_bounds__(256) void head_fwd_train_kernel(
    const float* __restrict__ z, const float* __restrict__ c2, const float* __restrict__ g3,
    const float* __restrict__ b3, float* __restrict__ rm3, float* __restrict__ rv3, int64_t* nbt,
    float* __restrict__ zhat, float* __restrict__ o, float* __restrict__ sig3, int Bs, int B) {
    __shared__ double red[4];
    const int u = blockIdx.x, tid = threadIdx.x;
    const float* zu = z + (size_t)u * Bs;
    const float gam = g3[u], bet = b3[u];
    double mean, var, sg;
    if (B <= HEAD_RB * 256) {
        head_fwd_train_regs<HEAD_RB>(zu, zhat + (size_t)u * Bs, o + (size_t)u * Bs, gam, bet, B, tid, red, mean, var, sg);
    } else if (B <= 16 * 256) {
        head_fwd_train_regs<16>(zu, zhat + (size_t)u * Bs, o + (size_t)u * Bs, gam, bet, B, tid, red, mean, var, sg);
    } else {
        double s = 0;
        for (int b = tid; b < B; b += 256) s += (double)zu[b];
        mean = block_sum_256(s, red) / (double)B;
        double v = 0;
        for (int b = tid; b < B; b += 256) { const double d = (double)zu[b] - mean; v = fma(d, d, v); }
        var = block_sum_256(v, red) / (double)B;
        sg = sqrt(var + BN_EPS_D);
        const float meanf = (float)mean, isg = (float)(1.0 / sg);
        for (int b = tid; b < B; b += 256) {
            const float zh = (zu[b] - meanf) * isg;
            zhat[(size_t)u * Bs + b] = zh;
            o[(size_t)u * Bs + b] = fmaxf(fmaf(gam, zh, bet), 0.f);
        }
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
    for (int u0 = wv; u0 < U; u0 += 160) {             // ten units (twenty loads) in flight
        float wq[10], oq[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const int u = min(u0 + 16 * q, U - 1);
            wq[q] = wr[u];
            oq[q] = o[(size_t)u * Bs + b];
        }
#pragma unroll
        for (int q = 0; q < 10; ++q) { KEEP(wq[q]); KEEP(oq[q]); }
#pragma unroll
        for (int q = 0; q < 10; ++q) acc = fmaf(u0 + 16 * q < U ? wq[q] : 0.f, oq[q], acc);
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

// Train forward, few tasks: BatchNorm3's batch statistics are FINISHED here, inside the combiner
// launch, from the per-workgroup fp64 sums of z and z^2 fc_fwd left in z12p -- head_fwd_train's
// launch (~5 us for 300 x 1024 values) disappears.  Every block rebuilds the statistics of all units
// (U x NBLK pairs of doubles, a few KB from L2), normalises its 64 sequences on the fly and sums the
// combiner; the blocks of task 0 also store zhat and o for the backward, block (0, 0) the running
// statistics.  Same grid and summation order as logits_kernel.
__global__ __launch_bounds__(1024) void logits_bn_kernel(
    const float* __restrict__ z, const double* __restrict__ z12p, int nblk, const float* __restrict__ c2, const float* __restrict__ g3, const float* __restrict__ b3,
    float* __restrict__ rm3, float* __restrict__ rv3, int64_t* nbt, float* __restrict__ zhat,
    float* __restrict__ o, float* __restrict__ sig3, const float* __restrict__ Wf,
    const float* __restrict__ bf, float* __restrict__ logits, int U, int T, int Bs, int B) {
    extern __shared__ float4 st4[];                   // [U] {mean, 1/sigma, gamma, beta}
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane, t = blockIdx.y;
    const bool owner = blockIdx.x == 0 && blockIdx.y == 0;
    // the first batch of z loads does not depend on the statistics: requested before them, so that
    // the kernel's two round trips overlap (the combiner weights ride in LDS beside the statistics:
    // with them in the batch too the kernel spilled at its 128 registers)
    constexpr int LQ = 20;
    const float* wr = Wf + (size_t)t * U;
    float* wl = reinterpret_cast<float*>(st4 + U);      // [U] combiner weights of task t
    float zq0[LQ];
#pragma unroll
    for (int q = 0; q < LQ; ++q)
        zq0[q] = z[(uint32_t)min(wv + 16 * q, U - 1) * (uint32_t)Bs + (uint32_t)b];
    for (int u = threadIdx.x; u < U; u += 1024) {
        double s1 = 0, s2 = 0;
        for (int i0 = 0; i0 < nblk; i0 += 4) {           // four partial pairs in flight per round trip
            double2 pv[4];                                // (eight, with the z batch above held, spilled)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                pv[i] = *reinterpret_cast<const double2*>(&z12p[((size_t)u * nblk + min(i0 + i, nblk - 1)) * 2]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i0 + i < nblk) { s1 += pv[i].x; s2 += pv[i].y; }
        }
        const double mean = s1 / (double)B;
        const double var = fmax(s2 / (double)B - mean * mean, 0.0);
        const double sg = sqrt(var + BN_EPS_D);
        st4[u] = make_float4((float)mean, (float)(1.0 / sg), g3[u], b3[u]);
        wl[u] = wr[u];
        if (owner) {
            sig3[u] = (float)sg;
            rm3[u] = (float)((1 - BN_MOM_D) * (double)rm3[u] + BN_MOM_D * (mean + (double)c2[u]));
            rv3[u] = (float)((1 - BN_MOM_D) * (double)rv3[u] + BN_MOM_D * var * (double)B / (double)(B - 1));
            if (u == 0 && nbt) *nbt += 1;
        }
    }
    __syncthreads();
    const bool store = t == 0 && b < B;
    float acc = 0.f;
    // twenty units (forty loads) in flight per wave: the 300 units of the headline shape are one
    // memory round trip per wave (ten at a time made two dependent ones in a kernel that is nothing
    // but round trips)
    for (int u0 = wv; u0 < U; u0 += 16 * LQ) {
        float zq[LQ];
        if (u0 == wv) {
#pragma unroll
            for (int q = 0; q < LQ; ++q) zq[q] = zq0[q];
        } else {
            // (32-bit element offset from the uniform base: a 64-bit address per load took two
            // registers each and spilled)
#pragma unroll
            for (int q = 0; q < LQ; ++q)
                zq[q] = z[(uint32_t)min(u0 + 16 * q, U - 1) * (uint32_t)Bs + (uint32_t)b];
        }
#pragma unroll
        for (int q = 0; q < LQ; ++q) KEEP(zq[q]);
#pragma unroll
        for (int q = 0; q < LQ; ++q) {
            const int u = u0 + 16 * q;
            if (u < U) {                               // wave-uniform
                const float4 sv = st4[u];
                const float zh = (zq[q] - sv.x) * sv.y;
                const float ov = fmaxf(fmaf(sv.z, zh, sv.w), 0.f);
                if (store) {
                    const uint32_t off = (uint32_t)u * (uint32_t)Bs + (uint32_t)b;
                    zhat[off] = zh; o[off] = ov;
                }
                acc = fmaf(wl[u], ov, acc);
            }
        }
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
    if (train && logits && c->T <= HEAD_GEMM_MIN_T && !outs && (size_t)c->U * (sizeof(float4) + sizeof(float)) <= 48 * 1024) {
        hipLaunchKernelGGL(logits_bn_kernel, dim3((B + 63) / 64, c->T), dim3(1024),
                           (size_t)c->U * (sizeof(float4) + sizeof(float)), s, c->z, c->z12p, fc_fwd_blocks(B, c->NQ),
                           p->fc2_b, p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv, p->bn3_nbt, c->zhat,
                           c->o, c->sig3, p->final_w, p->final_b, logits, c->U, c->T, c->Bs, B);
        LAUNCH_CHECK();
        return EXPLAINN_OK;
    }
    if (train) {
        hipLaunchKernelGGL(head_fwd_train_kernel, dim3(c->U), dim3(256), 0, s, c->z, p->fc2_b,
                           p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv, p->bn3_nbt, c->zhat, c->o,
                           c->sig3, c->Bs, B);
        LAUNCH_CHECK();
    }
    if (logits && c->T > HEAD_GEMM_MIN_T) {
        // logits[b][t] = sum_u o[u][b] Wf[t][u] + bf[t]: M = B, N = T, K = U
        const int tiles = ((B + 31) / 32) * ((c->T + 31) / 32);
        hipLaunchKernelGGL((gemm32_kernel<false, true, EPI_BIAS>), dim3(tiles), dim3(256), 0, s, c->o,
                           c->Bs, p->final_w, c->U, logits, c->T, B, c->T, c->U, p->final_b,
                           (float*)nullptr);
        LAUNCH_CHECK();
    } else if (logits) {
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


// loss + dlogits; up to LOSS_BLOCKS blocks write one partial sum each, the last block to finish adds
// them in index order -> deterministic whatever the arrival order
constexpr int LOSS_BLOCKS = 64;
// DEFER (the training step): the blocks leave their partial sums and the head backward that follows
// adds them up -- the last-arriver finish below needs a device-scope fence in every block, and on
// this multi-XCD part that write-back made a kernel with 200 K elements take 17 us.
template <bool DEFER>
__global__ __launch_bounds__(1024) void loss_kernel(const float* __restrict__ logits,
                                                    const float* __restrict__ y, int kind, int N,
                                                    float* __restrict__ loss,
                                                    float* __restrict__ dlogits,
                                                    double* __restrict__ partial,
                                                    int* __restrict__ counter) {
    __shared__ double red[16];
    __shared__ int last;
    const int tid = threadIdx.x;
    const float invN = 1.0f / (float)N;
    double acc = 0;
    // four elements per thread per trip with their loads issued together (a block's eight trips of a
    // run-time loop were eight memory round trips; the order of a thread's additions is unchanged)
    const int stride = gridDim.x * 1024;
    for (int i0 = blockIdx.x * 1024 + tid; i0 < N; i0 += 4 * stride) {
        float xv[4], tv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ic = min(i0 + q * stride, N - 1);
            xv[q] = logits[ic]; tv[q] = y[ic];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { KEEP(xv[q]); KEEP(tv[q]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * stride;
            if (i < N) {
                const float x = xv[q], t = tv[q];
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
        }
    }
    acc = wave_sum_d(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (DEFER) {
        if (tid == 0) {
            double s = 0;
            for (int i = 0; i < 16; ++i) s += red[i];
            partial[blockIdx.x] = s;
        }
        return;
    }
    if (tid == 0) {
        double s = 0;
        for (int i = 0; i < 16; ++i) s += red[i];
        partial[blockIdx.x] = s;
        __threadfence();
        last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        double s = 0;
        for (unsigned i = 0; i < gridDim.x; ++i) s += partial[i];
        *loss = (float)(s / (double)N);
        *counter = 0;                                   // ready for the next launch
    }
}

int launch_loss(explainn_ctx* c, int kind, const float* logits, const float* y, int B, float* loss,
                float* dlogits, hipStream_t s) {
    const int N = B * c->T;
    const int blocks = min(LOSS_BLOCKS, (N + 8191) / 8192);
    hipLaunchKernelGGL(loss_kernel<false>, dim3(blocks), dim3(1024), 0, s, logits, y, kind, N, loss, dlogits,
                       c->lossp, c->flags + 1);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// the training step's form: partial sums only; launch_head_bwd (which must follow) finishes the value
int launch_loss_deferred(explainn_ctx* c, int kind, const float* logits, const float* y, int B, float* loss,
                         float* dlogits, hipStream_t s) {
    const int N = B * c->T;
    // (no last-arriver finish here, so nothing limits the block count: one or two elements per thread)
    const int blocks = min(256, (N + 1023) / 1024);
    hipLaunchKernelGGL(loss_kernel<true>, dim3(blocks), dim3(1024), 0, s, logits, y, kind, N, loss, dlogits,
                       c->lossp, c->flags + 1);
    LAUNCH_CHECK();
    c->loss_blocks = blocks; c->loss_out = loss; c->loss_n = N;
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

// one block per unit: final-layer gradients, BN3 backward -> dz[u][b].
// GEMMED: d o (in dz on entry) and the final-layer gradients came from the MFMA GEMMs above.
template <bool FUSED, bool GEMMED = false>
__global__ __launch_bounds__(256) void head_bwd_kernel(
    const float* __restrict__ dl, const float* __restrict__ logits, const float* __restrict__ y,
    int kind, float* __restrict__ loss_out, const float* __restrict__ Wf,
    const float* __restrict__ g3, const float* __restrict__ o, const float* __restrict__ zhat,
    const float* __restrict__ sig3, float* __restrict__ dz, float* __restrict__ gWf,
    float* __restrict__ gbf, float* __restrict__ gg3, float* __restrict__ gb3,
    float* __restrict__ gc2, int U, int T, int Bs, int B, const float* __restrict__ gWp = nullptr,
    int gwch = 0, const double* __restrict__ lossp = nullptr, int lossb = 0, int lossn = 1) {
    __shared__ double red[4];
    __shared__ double red3[12];
    const int u = blockIdx.x, tid = threadIdx.x;
    const float* ou = o + (size_t)u * Bs;
    const float* zh = zhat + (size_t)u * Bs;
    float* dzu = dz + (size_t)u * Bs;
    const float invN = 1.0f / (float)(B * T);
    const float g3u = g3[u], sig3u = sig3[u];        // (used behind the block sums: requested here)
    double s1 = 0, s2 = 0;
    // small batches and few tasks: d3 and zhat of this thread's sequences stay in registers between
    // the two passes (one batch of loads each instead of a dependent round trip per 256 sequences)
    constexpr int RB = HEAD_RB;
    const bool inreg = !GEMMED && T <= 4 && B <= RB * 256;
    float d3r[RB], zhr[RB];
    double gw[4] = {0, 0, 0, 0}, gb[4] = {0, 0, 0, 0}, lacc = 0;
    if (inreg) {
        float our[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int bc = min(tid + 256 * i, B - 1);
            our[i] = ou[bc];
            zhr[i] = zh[bc];
            const bool live_i = tid + 256 * i < B;
            float dob = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= T) continue;
                const float dlv = dl_at<FUSED>(dl, logits, y, kind, invN, bc * T + t);
                dob = fmaf(dlv, Wf[(size_t)t * U + u], dob);
                // the final-layer gradients ride along: d Wf[t][u] += dl * o, and (unit 0's block)
                // d bf[t] += dl and the loss value
                if (live_i) {
                    gw[t] = fma((double)dlv, (double)our[i], gw[t]);
                    if (u == 0) {
                        gb[t] += (double)dlv;
                        if (FUSED) {
                            const float x = logits[bc * T + t], tt = y[bc * T + t];
                            float l;
                            if (kind == EXPLAINN_LOSS_BCE_WITH_LOGITS) l = fmaxf(x, 0.f) - x * tt + log1pf(expf(-fabsf(x)));
                            else { const float e = x - tt; l = e * e; }
                            lacc += (double)l;
                        }
                    }
                }
            }
            d3r[i] = dob;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const bool live = tid + 256 * i < B;
            d3r[i] = (live && our[i] > 0.f) ? d3r[i] : 0.f;
            s1 += (double)d3r[i];
            s2 = fma((double)d3r[i], live ? (double)zhr[i] : 0.0, s2);
        }
    } else {
        if (GEMMED) {
            // four sequences per thread per trip, their loads issued together (one sequence per trip of
            // a run-time loop was one memory round trip each: sixteen in a row at batch 4096)
            for (int b0 = tid; b0 < B; b0 += 4 * 256) {
                float dv[4], ov[4], zv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int bc = min(b0 + 256 * i, B - 1);
                    dv[i] = dzu[bc]; ov[i] = ou[bc]; zv[i] = zh[bc];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { KEEP(dv[i]); KEEP(ov[i]); KEEP(zv[i]); }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = b0 + 256 * i;
                    if (b < B) {
                        const float d3 = ov[i] > 0.f ? dv[i] : 0.f;
                        dzu[b] = d3;
                        s1 += (double)d3;
                        s2 = fma((double)d3, (double)zv[i], s2);
                    }
                }
            }
        } else {
        for (int b = tid; b < B; b += 256) {
            float dob = 0.f;
            for (int t = 0; t < T; ++t)
                dob = fmaf(dl_at<FUSED>(dl, logits, y, kind, invN, b * T + t), Wf[(size_t)t * U + u], dob);
            const float d3 = ou[b] > 0.f ? dob : 0.f;
            dzu[b] = d3;
            s1 += (double)d3;
            s2 = fma((double)d3, (double)zh[b], s2);
        }
        }
    }
    double S1 = s1, S2 = s2, G0 = gw[0];
    if (inreg) {
        block_sum3_256(S1, S2, G0, red3);          // (inreg is block-uniform: so are the barriers)
    } else {
        S1 = block_sum_256(s1, red);
        S2 = block_sum_256(s2, red);
    }
    const float m1 = (float)(S1 / (double)B), m2 = (float)(S2 / (double)B);
    const float sc = g3u / sig3u;
    if (inreg) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
            if (tid + 256 * i < B) dzu[tid + 256 * i] = sc * (d3r[i] - m1 - zhr[i] * m2);
    } else {
        for (int b0 = tid; b0 < B; b0 += 4 * 256) {
            float dv[4], zv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bc = min(b0 + 256 * i, B - 1);
                dv[i] = dzu[bc]; zv[i] = zh[bc];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { KEEP(dv[i]); KEEP(zv[i]); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (b0 + 256 * i < B) dzu[b0 + 256 * i] = sc * (dv[i] - m1 - zv[i] * m2);
        }
    }
    if (tid == 0) { gg3[u] = (float)S2; gb3[u] = (float)S1; gc2[u] = 0.f; }
    if (inreg) {
        // everything was accumulated in the first pass; only the block-wide sums are left
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= T) continue;                      // T is block-uniform: the barriers inside stay uniform
            const double tot = t == 0 ? G0 : block_sum_256(gw[t], red);
            if (tid == 0) gWf[(size_t)t * U + u] = (float)tot;
            if (u == 0) {
                const double ct = block_sum_256(gb[t], red);
                if (tid == 0) gbf[t] = (float)ct;
            }
        }
        if (FUSED && u == 0) {
            const double tot = block_sum_256(lacc, red);
            if (tid == 0) *loss_out = (float)(tot / (double)(B * T));
        }
        return;
    }
    if (lossb > 0 && u == 0) {
        // the loss value: launch_loss_deferred's block partials (at most 256), one per thread, summed
        // in the block's fixed tree order (block-uniform branch: the barriers inside are safe)
        const double sl = block_sum_256(tid < lossb ? lossp[tid] : 0.0, red);
        if (tid == 0) *loss_out = (float)(sl / (double)lossn);
    }
    if (GEMMED && gWp) {
        // the combiner-weight gradient of this unit (and, in unit 0's block, the bias gradient): the
        // GEMM's batch-chunk partials summed in chunk order, eight loads in flight
        for (int e = tid; e < (u == 0 ? 2 * T : T); e += 256) {
            const int t = e < T ? e : e - T, col = e < T ? u : U;
            float sacc = 0.f;
            for (int c0 = 0; c0 < gwch; c0 += 8) {
                float pv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    pv[i] = gWp[((size_t)min(c0 + i, gwch - 1) * T + t) * (U + 1) + col];
#pragma unroll
                for (int i = 0; i < 8; ++i) KEEP(pv[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) sacc += (c0 + i < gwch) ? pv[i] : 0.f;
            }
            if (e < T) gWf[(size_t)t * U + u] = sacc; else gbf[t] = sacc;
        }
    }
    for (int t = 0; t < (GEMMED ? 0 : T); ++t) {
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
    if (c->T > HEAD_GEMM_MIN_T) {
        const int T = c->T, U = c->U;
        hipLaunchKernelGGL(transpose_dl_kernel, dim3((B + 31) / 32, (T + 31) / 32), dim3(256), 0, s,
                           dlogits, c->dlT, B, T, c->Bs);
        LAUNCH_CHECK();
        // d o[u][b] = sum_t Wf[t][u] dl[b][t]  -> dz (raw; the per-unit kernel finishes it)
        hipLaunchKernelGGL((gemm32_kernel<false, false, EPI_PLAIN>),
                           dim3(((U + 31) / 32) * ((B + 31) / 32)), dim3(256), 0, s, p->final_w, U,
                           c->dlT, c->Bs, c->dz, c->Bs, U, B, T, (const float*)nullptr,
                           (float*)nullptr);
        LAUNCH_CHECK();
        // d Wf[t][u] = sum_b dl[b][t] o[u][b], and d bf[t] as the virtual all-ones unit U
        const int gwch = head_gw_chunks(B);
        hipLaunchKernelGGL((gemm32_kernel<true, true, EPI_GW>),
                           dim3(((T + 31) / 32) * ((U + 1 + 31) / 32), gwch), dim3(256), 0, s, c->dlT, c->Bs,
                           c->o, c->Bs, c->gWp, U + 1, T, U + 1, B, (const float*)nullptr,
                           (float*)nullptr);
        LAUNCH_CHECK();
        hipLaunchKernelGGL((head_bwd_kernel<false, true>), dim3(U), dim3(256), 0, s, dlogits,
                           (const float*)nullptr, (const float*)nullptr, 0,
                           c->loss_blocks ? c->loss_out : (float*)nullptr,
                           p->final_w, p->bn3_w, c->o, c->zhat, c->sig3, c->dz, g->final_w,
                           g->final_b, g->bn3_w, g->bn3_b, g->fc2_b, U, T, c->Bs, B, c->gWp, gwch,
                           c->lossp, c->loss_blocks, c->loss_n);
        LAUNCH_CHECK();
        c->loss_blocks = 0;
        return EXPLAINN_OK;
    }
    hipLaunchKernelGGL(head_bwd_kernel<false>, dim3(c->U), dim3(256), 0, s, dlogits,
                       (const float*)nullptr, (const float*)nullptr, 0, c->loss_blocks ? c->loss_out : (float*)nullptr,
                       p->final_w, p->bn3_w, c->o, c->zhat, c->sig3, c->dz, g->final_w, g->final_b, g->bn3_w,
                       g->bn3_b, g->fc2_b, c->U, c->T, c->Bs, B, (const float*)nullptr, 0, c->lossp,
                       c->loss_blocks, c->loss_n);
    LAUNCH_CHECK();
    c->loss_blocks = 0;
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
