// The filter bank: Conv1d(4U->U, k, groups=U) + BatchNorm1 + exp + MaxPool1d(7,7)
// (architectures/__init__.py:73-81) as ONE gather kernel that never materialises the conv output.
//
// One-hot input makes the convolution a gather: conv[b,u,p] = sum_j W[u, s[b,p+j], j].
// Dinucleotide tables halve the gather: for each pair of taps t, LUT[t][c0 c1] holds the sum of
// the two taps for that 2-mer, so a 19-tap window is 10 LDS reads + 10 adds instead of 19 + 19.
// The table size is chosen for the banks: 16 entries x 8 B (two units as float2, ds_read_b64)
// span exactly 32 banks, so lanes reading 16 different 2-mers never conflict (a 256-entry 4-mer
// table read at random addresses would be ~3-4-way conflicted and give the gain back).
// A lane owns one sequence and TWO units; the 2-mer index is 4 consecutive bits of the lane's
// 2-bit packed sequence, taken from a 96-bit register window that slides 14 bits per pooling window.
// An N (all-zero column) is packed as 'C'; the lanes that have one in their window subtract the C
// tap again (per-tap table W[j][code]), one N base at a time.
// BatchNorm+exp are monotone per unit, so the 7-wide max-pool runs on the raw gather sums with the
// sign of alpha = gamma1/sigma1 choosing max or min; only the pooled extreme (and its offset, for
// the backward routing) leaves the kernel: ext[u][w][b], idx[u][w][b].
#include <cstdlib>

#include "common.h"

// The tables LUT[pair][t][code4] = (W[u0][c0][2t] + W[u0][c1][2t+1], same for u1),
// c_i = (code4 >> 2i) & 3, are written by prep1 (prep.hip) from the current filters.

// LDS address of the table entry of the 2-mer starting x bases into the window (x is compile-time
// after unrolling): the 4 code bits are shifted straight to their place in the byte offset and OR-ed
// onto the table's base (which is aligned to the table size), one shift + one v_and_or_b32
template <int ESZ>
__device__ __forceinline__ uint32_t dimer_addr(uint32_t w0, uint32_t w1, uint32_t w2, int x, uint32_t base) {
    constexpr int LG = ESZ == 16 ? 4 : 3;      // log2(ESZ)
    static_assert(ESZ == 16 || ESZ == 8, "table entries are float4 or float2");
    const int bit = 2 * x;
    uint32_t v;                                 // the 4 code bits at bit positions LG..LG+3
    if (bit + 4 <= 32) v = bit >= LG ? (w0 >> (bit - LG)) : (w0 << (LG - bit));
    else if (bit >= 32 && bit + 4 <= 64) v = (bit - 32) >= LG ? (w1 >> (bit - 32 - LG)) : (w1 << (LG - (bit - 32)));
    else if (bit >= 64) v = (bit - 64) >= LG ? (w2 >> (bit - 64 - LG)) : (w2 << (LG - (bit - 64)));
    else if (bit < 32) v = __funnelshift_r(w0, w1, bit) << LG;
    else v = __funnelshift_r(w1, w2, bit - 32) << LG;
    return (v & (0xfu * ESZ)) | base;
}

// The gather + pooling of one wavefront: lane = sequence b, NU units (2 or 4), pooling windows
// [wbeg, wend).  `sink(w, ext, off)` receives the pooled extremes (raw gather sums, NU of them) and
// their offsets inside the window.  L2/Wp: the units' dinucleotide and per-tap tables in LDS (entries
// of NU floats); pks/nms: this wave's private code tiles ([PWC][64] / [NWC][64] words).
// NU = 4 (the training / default eval filter bank): one ds_read_b128 per (position, tap pair) feeds
// four units, and the 2-mer offsets, the window bookkeeping and the N corrections are shared by four
// units instead of two -- conv_pool is instruction-issue-bound (profiles/r02: 255 VALU + 47 LDS
// instructions per window and unit pair), so instructions per unit are what counts.
template <int NU> struct fvec;
template <> struct fvec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct fvec<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int K, int NU, int CPW, bool IDX = true, typename Sink>
__device__ __forceinline__ void conv_pool_windows(const void* L2, const void* Wp, uint32_t* pks,
                                                  uint32_t* nms, const uint32_t* __restrict__ pk2,
                                                  const uint32_t* __restrict__ nmask,
                                                  const float (&sg)[NU], int b, int lane, int wbeg,
                                                  int wend, int Bs, int PW, int NW, Sink sink,
                                                  bool staged = false) {
    typedef typename fvec<NU>::type fv;
    constexpr int ESZ = NU * 4;                 // bytes per table entry
    constexpr int NT = (K + 1) / 2;             // 2-mer tables
    constexpr int NX = POOLW + 2 * (NT - 1);    // distinct 2-mer start offsets inside a window
    constexpr int SPAN = POOLW + K - 1;         // positions a pooling window reads
    typedef __attribute__((address_space(3))) fv lds_fv;
    typedef __attribute__((address_space(3))) char lds_char;
    constexpr int PWC = ((POOLW * CPW + K + 15) >> 4) + 3, NWC = ((POOLW * CPW + K + 31) >> 5) + 2;
    const char* Lb = reinterpret_cast<const char*>(L2);
    const char* Wb = reinterpret_cast<const char*>(Wp);
    // 32-bit LDS address of the 2-mer tables; the callers put them at the start of the dynamic LDS
    // (aligned to the 16-entry table: the entry offset is OR-ed in)
    const uint32_t lbase = (uint32_t)(size_t)(const lds_char*)Lb;
    if (lbase & (16u * ESZ - 1u)) __builtin_trap();
    const uint32_t* pl = pks + lane;
    const uint32_t* nl = nms + lane;
    for (int wc = wbeg; wc < wend; wc += CPW) {
    // chunk origin in words; columns are lane-private, so no barrier is needed between chunks
    const int w_lo = (POOLW * wc) >> 4, n_lo = (POOLW * wc) >> 5;
    // (staged: the caller's windows fit one chunk and it has filled the tiles already -- the fused
    // eval kernel walks many unit pairs over the same codes)
    if (!staged)
        stage_columns2<PWC, NWC>(pks + lane, pk2 + (size_t)w_lo * Bs + b, min(PWC, PW - w_lo),
                                 nms + lane, nmask + (size_t)n_lo * Bs + b, min(NWC, NW - n_lo), Bs);
    // window words of the chunk's first pooling window (the next one is prefetched inside the loop)
    const int wi0 = ((POOLW * wc) >> 4) - w_lo, ni0 = ((POOLW * wc) >> 5) - n_lo;
    uint32_t c0 = pl[wi0 * 64], c1 = pl[(wi0 + 1) * 64], c2 = pl[(wi0 + 2) * 64], c3 = pl[(wi0 + 3) * 64];
    uint32_t m0 = nl[ni0 * 64], m1 = nl[(ni0 + 1) * 64], m2 = nl[(ni0 + 2) * 64];
    const int wcend = min(wend, wc + CPW);
    for (int w = wc; w < wcend; ++w) {
        const int p0 = POOLW * w;
        const int sh = (p0 & 15) * 2, nsh = p0 & 31;
        const uint32_t w0 = __funnelshift_r(c0, c1, sh), w1 = __funnelshift_r(c1, c2, sh),
                       w2 = __funnelshift_r(c2, c3, sh);
        const uint32_t nm0 = __funnelshift_r(m0, m1, nsh), nm1 = __funnelshift_r(m1, m2, nsh);
        {   // prefetch the next window's words (past the chunk they are unused: the chunk restages)
            const int q0 = p0 + POOLW, wi = (q0 >> 4) - w_lo, ni = (q0 >> 5) - n_lo;
            c0 = pl[wi * 64]; c1 = pl[(wi + 1) * 64]; c2 = pl[(wi + 2) * 64]; c3 = pl[(wi + 3) * 64];
            m0 = nl[ni * 64]; m1 = nl[(ni + 1) * 64]; m2 = nl[(ni + 2) * 64];
        }
        constexpr uint32_t HIMASK = SPAN > 32 ? ((SPAN >= 64) ? 0xffffffffu : ((1u << (SPAN - 32)) - 1u)) : 0u;
        constexpr uint32_t LOMASK = SPAN >= 32 ? 0xffffffffu : ((1u << SPAN) - 1u);
        // the first tap pair initialises the sums (no zero + add), the table addresses are formed as
        // 32-bit LDS addresses with the table offset in the instruction's immediate
        fv acc[POOLW];
        {
            uint32_t a8[NX];
#pragma unroll
            for (int x = 0; x < NX; ++x) a8[x] = dimer_addr<ESZ>(w0, w1, w2, x, lbase);
#pragma unroll
            for (int i = 0; i < POOLW; ++i) acc[i] = *reinterpret_cast<const lds_fv*>((const lds_char*)(size_t)a8[i]);
#pragma unroll
            for (int t = 1; t < NT; ++t) {
#pragma unroll
                for (int i = 0; i < POOLW; ++i)
                    acc[i] += *reinterpret_cast<const lds_fv*>((const lds_char*)(size_t)a8[i + 2 * t] + t * 16 * ESZ);
            }
        }
        // N bases are packed as 'C': take the C tap back out wherever the mask says N.  Per lane and
        // per N base (a loop over the set bits; lanes without an N idle through it), instead of
        // sending the whole wavefront down a 19-reads-per-position path because one lane saw an N.
        // The corrections are summed apart and subtracted once: with the sums themselves carried
        // through the loop the compiler copied all of them twice per window, N or not.
        uint32_t r0 = nm0 & LOMASK, r1 = nm1 & HIMASK;
        if (__any((r0 | r1) != 0u)) {
            fv corr[POOLW];
#pragma unroll
            for (int i = 0; i < POOLW; ++i) corr[i] = fv(0.f);
            do {
                if ((r0 | r1) != 0u) {
                    int x;
                    if (r0) { x = __ffs(r0) - 1; r0 &= r0 - 1u; }
                    else { x = 32 + __ffs(r1) - 1; r1 &= r1 - 1u; }
#pragma unroll
                    for (int i = 0; i < POOLW; ++i) {
                        const int j = x - i;                 // base x of the window is tap j of position i
                        if (j >= 0 && j < K)
                            corr[i] += *reinterpret_cast<const fv*>(Wb + j * 5 * ESZ + ESZ);   // code 1 = C
                    }
                }
            } while (__any((r0 | r1) != 0u));
#pragma unroll
            for (int i = 0; i < POOLW; ++i) acc[i] -= corr[i];
        }
        // pooled extreme: max or min by the (wave-uniform) sign, then the first position that holds it
        // (the index chains of the NU units are interleaved: a compare and the select that reads
        // its lane mask back to back cost two idle issue slots each)
        float ex[NU];
        int bi[NU];
#pragma unroll
        for (int uu = 0; uu < NU; ++uu) {
            float hi = acc[0][uu], lo = acc[0][uu];
#pragma unroll
            for (int i = 1; i < POOLW; ++i) { hi = fmaxf(hi, acc[i][uu]); lo = fminf(lo, acc[i][uu]); }
            ex[uu] = sg[uu] > 0.f ? hi : lo;
            bi[uu] = POOLW - 1;
        }
        // (IDX = false -- eval: only the backward routes gradients by the argmax position)
        if (IDX)
#pragma unroll
        for (int i = POOLW - 2; i >= 0; --i) {
            bool eq[NU];
#pragma unroll
            for (int uu = 0; uu < NU; ++uu) eq[uu] = acc[i][uu] == ex[uu];
#pragma unroll
            for (int uu = 0; uu < NU; ++uu) bi[uu] = eq[uu] ? i : bi[uu];       // first index wins ties
        }
        sink(w, ex, bi);
    }
    }
}

template <int K, int CPW, bool IDX>
__global__ __launch_bounds__(64) void conv_pool_kernel(const uint32_t* __restrict__ pk2,
                                                       const uint32_t* __restrict__ nmask,
                                                       const float4* __restrict__ lut,
                                                       const float* __restrict__ Wt,
                                                       const float* __restrict__ gamma1, int U,
                                                       float* __restrict__ ext,
                                                       uint8_t* __restrict__ idx, int n, int Bs,
                                                       int PW, int NW, int wsplit) {
    constexpr int NT = (K + 1) / 2;
    extern __shared__ __attribute__((aligned(256))) uint32_t csm[];
    float4* L2 = reinterpret_cast<float4*>(csm);            // [NT][16]  2-mer sums of the unit quad
    float4* Wp = L2 + NT * 16;                              // [K][5]    per-tap table (N path)
    // the packed codes are staged per chunk of CPW pooling windows: ~7 KB of LDS per wave at any
    // sequence length (the whole of a 1000-bp sequence was 26 KB and cost two thirds of the occupancy);
    // CPW = 8 where a wave's share of the windows fits one such chunk (3.3 KB: the 32-window tile
    // capped the C2 launch at 13 of its 19 waves per CU)
    constexpr int PWC = ((POOLW * CPW + K + 15) >> 4) + 3;
    uint32_t* pks = reinterpret_cast<uint32_t*>(Wp + K * 5); // [PWC][64]
    uint32_t* nms = pks + (size_t)PWC * 64;                  // [NWC][64]
    const int quad = blockIdx.y, lane = threadIdx.x;
    const int b = (blockIdx.x / wsplit) * 64 + lane;
    // the pooling windows of a (tile, quad) are split over `wsplit` wavefronts: more waves per
    // SIMD to hide the LDS latency (a wave issues at most one instruction per 4 cycles)
    const int wper = (n + wsplit - 1) / wsplit;
    const int wbeg = (blockIdx.x % wsplit) * wper, wend = min(n, wbeg + wper);
    STAMP(0);
    {
        // both tables with all their loads in flight before the first LDS store
        const float4* src = lut + (size_t)quad * NT * 16;
        const float4* wsrc = reinterpret_cast<const float4*>(Wt) + (size_t)quad * K * 5;
        constexpr int NL = (NT * 16 + 63) / 64, NWP = (K * 5 + 63) / 64;
        float4 lv[NL], wv[NWP];
#pragma unroll
        for (int j = 0; j < NL; ++j) lv[j] = src[min(lane + 64 * j, NT * 16 - 1)];
#pragma unroll
        for (int j = 0; j < NWP; ++j) wv[j] = wsrc[min(lane + 64 * j, K * 5 - 1)];
#pragma unroll
        for (int j = 0; j < NL; ++j) { KEEP(lv[j].x); KEEP(lv[j].y); KEEP(lv[j].z); KEEP(lv[j].w); }
#pragma unroll
        for (int j = 0; j < NWP; ++j) { KEEP(wv[j].x); KEEP(wv[j].y); KEEP(wv[j].z); KEEP(wv[j].w); }
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (lane + 64 * j < NT * 16) L2[lane + 64 * j] = lv[j];
#pragma unroll
        for (int j = 0; j < NWP; ++j)
            if (lane + 64 * j < K * 5) Wp[lane + 64 * j] = wv[j];
    }
    // sign(alpha) = sign(gamma1): the pooling direction does not need the BatchNorm statistics
    float sg[4];
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) sg[uu] = (quad * 4 + uu < U && gamma1[quad * 4 + uu] < 0.f) ? -1.f : 1.f;
    __syncthreads();
    STAMP(1);
    conv_pool_windows<K, 4, CPW, IDX>(L2, Wp, pks, nms, pk2, nmask, sg, b, lane, wbeg, wend, Bs, PW, NW,
                            [&](int w, const float (&e)[4], const int (&i)[4]) {
                                // wave-uniform row pointers + one 32-bit lane offset (saddr stores)
                                const uint32_t o = (uint32_t)(w * Bs + b), o4 = o * 4u;
#pragma unroll
                                for (int uu = 0; uu < 4; ++uu) {
                                    const size_t row = (size_t)(quad * 4 + uu) * n * Bs;
                                    *reinterpret_cast<float*>(reinterpret_cast<char*>(ext + row) + o4) = e[uu];
                                    if (IDX) (idx + row)[o] = (uint8_t)i[uu];
                                }
                            });
    STAMP(2);
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

// window split and chunk length of the launch: four units per lane leave half the waves of a
// two-unit version per window split, so the windows are split four ways where there are enough
static int conv_pool_wsplit(const explainn_ctx* c) { return c->n >= 16 ? 4 : (c->n >= 8 ? 2 : 1); }
static int conv_pool_cpw(const explainn_ctx* c) {
    const int ws = conv_pool_wsplit(c);
    return (c->n + ws - 1) / ws <= 8 ? 8 : 32;
}

static size_t conv_pool_lds(const explainn_ctx* c, int cpw) {
    const int NT = (c->k + 1) / 2;
    // tables + the chunk tiles [PWC + NWC][64] (see the kernel)
    const int pwc = ((POOLW * cpw + c->k + 15) >> 4) + 3, nwc = ((POOLW * cpw + c->k + 31) >> 5) + 2;
    return (size_t)(NT * 16 + c->k * 5) * sizeof(float4) + (size_t)(pwc + nwc) * 64 * 4;
}

int launch_conv_pool(explainn_ctx* c, const explainn_params* p, int B, bool want_idx, hipStream_t s) {
    const int wsplit = conv_pool_wsplit(c), cpw = conv_pool_cpw(c);
    const dim3 grid(((B + 63) / 64) * wsplit, c->Uq);
    const size_t sm = conv_pool_lds(c, cpw);
#define ARGS grid, dim3(64), sm, s, c->pk2, c->nmask, reinterpret_cast<const float4*>(c->lut), c->Wt, \
             p->bn1_w, c->U, c->ext, c->idx, c->n, c->Bs, c->PW, c->NW, wsplit
    // (the argmax offsets are the backward's: eval launches skip them -- 52 of ~260 instructions per window)
#define CALL(KK)                                                                               \
    if (cpw == 8) {                                                                            \
        if (want_idx) hipLaunchKernelGGL((conv_pool_kernel<KK, 8, true>), ARGS);               \
        else hipLaunchKernelGGL((conv_pool_kernel<KK, 8, false>), ARGS);                       \
    } else {                                                                                   \
        if (want_idx) hipLaunchKernelGGL((conv_pool_kernel<KK, 32, true>), ARGS);              \
        else hipLaunchKernelGGL((conv_pool_kernel<KK, 32, false>), ARGS);                      \
    }
    K_DISPATCH(c->k, CALL);
#undef CALL
#undef ARGS
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int conv_configure(explainn_ctx* c) {
    const int cpw = conv_pool_cpw(c);
    const size_t sm = conv_pool_lds(c, cpw);
    if (sm > 48 * 1024) {
#define CALL(KK)                                                                               \
        if (cpw == 8) {                                                                        \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pool_kernel<KK, 8, true>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pool_kernel<KK, 8, false>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); \
        } else {                                                                               \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pool_kernel<KK, 32, true>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pool_kernel<KK, 32, false>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); \
        }
        K_DISPATCH(c->k, CALL);
#undef CALL
    }
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
