// Filter -> PWM export on the device: what the reference does with a dense float16
// (N,U,Lo) host array and Python loops over FASTA files --
//   test.py:128-166       activations stored as float16,
//   interpret.py:363-373  threshold = 0.5 * max activation over the well-predicted sequences,
//   interpret.py:375-429  every start position whose activation exceeds the threshold is a site
//                         (strand, sequence, position order; at most 1e6 sites per filter),
//   interpret.py:431-459  sites -> A/C/G/T count matrix --
// as two passes of one gather kernel over the packed base codes: nothing of size N*U*Lo is ever
// materialised.  Block = (sequence, unit quad), threads over start positions, so the in-sequence
// order of sites is the thread order and a site's global rank is
//     offset[unit][sequence] (exclusive scan of per-sequence counts) + rank inside the sequence.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

constexpr int SITE_T = 256;

enum { SITE_MAX = 0, SITE_COUNT = 1, SITE_ACCUM = 2 };

// float32 activation -> the float16 value numpy stores (round-to-nearest-even, overflow -> inf)
__device__ __forceinline__ float as_f16(float a) { return __half2float(__float2half_rn(a)); }

template <int MODE>
__global__ __launch_bounds__(SITE_T) void site_kernel(
    const uint8_t* __restrict__ codesT, const float* __restrict__ Wt, const float* __restrict__ alpha,
    const float* __restrict__ shift, const uint8_t* __restrict__ select,
    const float* __restrict__ thr, float* __restrict__ umax, int* __restrict__ cnt,
    const int* __restrict__ off, int* __restrict__ pfm, int cap, int U, int k, int L, int Lo, int Bs) {
    extern __shared__ float4 Wsm[];            // [k][5] | codes [L] bytes (16-aligned) | hist [4][k][4]
    uint8_t* cs = reinterpret_cast<uint8_t*>(Wsm + k * 5);
    int* hist = reinterpret_cast<int*>(cs + ((L + 15) & ~15));
    __shared__ int wtot[4][SITE_T / 64];
    __shared__ float wmax[4][SITE_T / 64];
    const int b = blockIdx.x, quad = blockIdx.y, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const bool selected = select == nullptr || select[b] != 0;
    if (!selected) {                           // block-uniform
        if (MODE == SITE_COUNT && tid < 4 && quad * 4 + tid < U) cnt[(size_t)(quad * 4 + tid) * Bs + b] = 0;
        return;
    }
    const float4* src = reinterpret_cast<const float4*>(Wt) + (size_t)quad * k * 5;
    for (int i = tid; i < k * 5; i += SITE_T) Wsm[i] = src[i];
    for (int p = tid; p < L; p += SITE_T) cs[p] = codesT[(size_t)p * Bs + b];
    if (MODE == SITE_ACCUM)
        for (int i = tid; i < 4 * k * 4; i += SITE_T) hist[i] = 0;
    float al[4], sh[4], th[4];
    int base[4];
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
        const int u = min(quad * 4 + uu, U - 1);
        al[uu] = alpha[u];
        sh[uu] = shift[u];
        th[uu] = MODE == SITE_MAX ? 0.f : thr[u];
        base[uu] = MODE == SITE_ACCUM ? off[(size_t)u * Bs + b] : 0;
    }
    __syncthreads();
    float mx[4] = {0.f, 0.f, 0.f, 0.f};        // activations are exp(.) >= 0
    int run[4] = {0, 0, 0, 0};                 // sites of this sequence in earlier position chunks
    for (int p0 = 0; p0 < Lo; p0 += SITE_T) {
        const int p = p0 + tid;
        const bool live = p < Lo;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
            for (int j = 0; j < k; ++j) {
                const float4 v = Wsm[j * 5 + cs[p + j]];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        const float av[4] = {acc.x, acc.y, acc.z, acc.w};
        bool hit[4];
        int pre[4];
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const float a16 = live ? as_f16(qval(al[uu], av[uu], sh[uu])) : 0.f;
            if (MODE == SITE_MAX) {
                mx[uu] = fmaxf(mx[uu], a16);
            } else {
                hit[uu] = live && a16 > th[uu];
                const unsigned long long bal = __ballot(hit[uu]);
                pre[uu] = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) wtot[uu][wave] = __popcll(bal);
            }
        }
        if (MODE != SITE_MAX) {
            __syncthreads();
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                int before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < SITE_T / 64; ++w) {
                    const int t = wtot[uu][w];
                    before += w < wave ? t : 0;
                    total += t;
                }
                if (MODE == SITE_ACCUM && hit[uu]) {
                    // rank of this site among all sites of the unit (interpret.py:398-425 order)
                    const long long rank = (long long)base[uu] + run[uu] + before + pre[uu];
                    if (rank < cap)
                        for (int t = 0; t < k; ++t) {
                            const int cd = cs[p + t];
                            if (cd < 4) atomicAdd(&hist[(uu * k + t) * 4 + cd], 1);
                        }
                }
                run[uu] += total;
            }
            __syncthreads();                   // wtot is rewritten by the next chunk
        }
    }
    if (MODE == SITE_MAX) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            float m = mx[uu];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) wmax[uu][wave] = m;
        }
        __syncthreads();
        if (tid < 4 && quad * 4 + tid < U) {
            float m = 0.f;
            for (int w = 0; w < SITE_T / 64; ++w) m = fmaxf(m, wmax[tid][w]);
            // non-negative floats (and +inf) order like their bit patterns
            atomicMax(reinterpret_cast<unsigned int*>(umax) + quad * 4 + tid, __float_as_uint(m));
        }
    } else if (MODE == SITE_COUNT) {
        if (tid < 4 && quad * 4 + tid < U) cnt[(size_t)(quad * 4 + tid) * Bs + b] = run[tid];
    } else {
        __syncthreads();
        for (int i = tid; i < 4 * k * 4; i += SITE_T) {
            const int uu = i / (k * 4), u = quad * 4 + uu;
            const int v = hist[i];
            if (v != 0 && u < U) atomicAdd(&pfm[(size_t)u * k * 4 + (i - uu * k * 4)], v);
        }
    }
}

// One wavefront per unit: exclusive scan of the per-sequence site counts (sequence order), on top of
// the unit's running total from earlier batches; both are clamped at the cap so nothing overflows.
__global__ __launch_bounds__(64) void site_scan_kernel(const int* __restrict__ cnt,
                                                       int* __restrict__ off,
                                                       int* __restrict__ site_total,
                                                       uint8_t* __restrict__ hit, int cap, int U,
                                                       int B, int Bs) {
    const int u = blockIdx.x, lane = threadIdx.x;
    long long running = site_total[u];
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        const int v = b < B ? cnt[(size_t)u * Bs + b] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (b < B) {
            const long long o64 = running + (inc - v);
            off[(size_t)u * Bs + b] = (int)(o64 < cap ? o64 : cap);
            if (hit) hit[(size_t)b * U + u] = v > 0;
        }
        running += __shfl(inc, 63, 64);
    }
    if (lane == 0) site_total[u] = (int)(running < cap ? running : cap);
}

size_t site_lds(const explainn_ctx* c, bool hist) {
    return (size_t)c->k * 5 * sizeof(float4) + ((c->L + 15) & ~15) +
           (hist ? (size_t)4 * c->k * 4 * sizeof(int) : 0);
}

}  // namespace

int launch_filter_act_max(explainn_ctx* c, int B, const uint8_t* select, float* umax, hipStream_t s) {
    hipLaunchKernelGGL(site_kernel<SITE_MAX>, dim3(B, c->Uq), dim3(SITE_T), site_lds(c, false), s,
                       c->codesT, c->Wt, c->alpha, c->shift, select, (const float*)nullptr, umax,
                       (int*)nullptr, (const int*)nullptr, (int*)nullptr, 0, c->U, c->k, c->L, c->Lo,
                       c->Bs);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int launch_filter_sites(explainn_ctx* c, int B, const uint8_t* select, const float* thr, int cap,
                        int* site_total, int* pfm, uint8_t* hit, hipStream_t s) {
    hipLaunchKernelGGL(site_kernel<SITE_COUNT>, dim3(B, c->Uq), dim3(SITE_T), site_lds(c, false), s,
                       c->codesT, c->Wt, c->alpha, c->shift, select, thr, (float*)nullptr,
                       c->site_cnt, (const int*)nullptr, (int*)nullptr, cap, c->U, c->k, c->L, c->Lo,
                       c->Bs);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(site_scan_kernel, dim3(c->U), dim3(64), 0, s, c->site_cnt, c->site_off,
                       site_total, hit, cap, c->U, B, c->Bs);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(site_kernel<SITE_ACCUM>, dim3(B, c->Uq), dim3(SITE_T), site_lds(c, true), s,
                       c->codesT, c->Wt, c->alpha, c->shift, select, thr, (float*)nullptr,
                       (int*)nullptr, c->site_off, pfm, cap, c->U, c->k, c->L, c->Lo, c->Bs);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}
