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
#include <type_traits>

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

// ---------------------------------------------------------------------------------------------
// The same filter bank on the matrix core.  conv[u][(b,p)] = sum_k Wf[u][k] X[k][(b,p)] with
// k = 4 tap + base and X the one-hot bit "sequence b has base a at position p + tap": a GEMM whose A
// operand is the filters as three bf16 pieces (sum = the fp32 weight exactly, every product with a
// 0/1 bit exact, fp32 accumulation) and whose B operand is generated from the 2-bit codes -- a lane
// needs the one-hot images of TWO consecutive bases (8 bf16 = one 16-byte entry of a 16-entry LDS
// table indexed by the 4 code bits; the 64-entry table also zeroes the taps that see an N).
// v_mfma_f32_32x32x16_bf16: tile = 32 units x 32 sequences at ONE position, 4 taps per k-step.
// A wave (= a workgroup; one per SIMD, 512 registers) owns 32 sequences, UT unit tiles whose
// fragments stay in registers, and a range of pooling windows; it walks the positions in order and
// keeps the running maximum and the first position that reached it (strict >), so the conv output
// never exists.  Units with gamma1 < 0 pool the minimum: their filters are negated in Wf and the
// sign comes back at the store.
// The instruction stream is laid out by hand (sched_barrier after every MFMA): an MFMA holds the
// SIMD's vector issue for 8 of its 32 cycles, so behind each one go its share of the vector work of
// the NEIGHBOURING positions -- the max/argmax update of the previous position (three accumulator
// sets rotate X A B A B A B so that the window length 7 needs no parity), the table addresses and
// LDS reads of the next position's operands (three operand buffers, same rotation), and the stores
// of the previous window.
// Roof: 3 pieces x KS k-steps x 32 clk per (32 units, 32 sequences, position) on the matrix pipe.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 cbf16x8;
typedef float cf32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t cu32x4 __attribute__((ext_vector_type(4)));

template <int KS, int UT> struct cpm_state {
    cu32x4 Wr[UT][KS][3];        // A fragments (pieces hi, mid, lo)
    cf32x16 acc[3][UT];          // accumulator sets X, A, B
    float best[UT][16];
    uint32_t arg[UT][16];
    cu32x4 bop[3][KS];           // B operands of three positions
    uint32_t xs[3], nx[2];       // the window's codes / N bits, shifted to the lane's k-half
    uint32_t gw[4], mw[3];       // words of the NEXT window (global loads in flight)
    uint32_t smask[UT][16];      // 0x80000000 where that unit pools the minimum
    uint32_t so, so4;            // running offsets (idx bytes, ext bytes) of the stores of the previous window
    uint32_t r1, r5;             // its row strides in elements (0 while there is no previous window)
};
struct cpm_args {
    const uint32_t* pk2; const uint32_t* nmask; float* ext; uint8_t* idx;
    int n, Bs, PW, NW, t0, b, kh; uint32_t tbase;
    uint32_t row1, row5;         // n Bs and 5 n Bs: element strides between the rows a lane stores
    __amdgpu_buffer_rsrc_t rext, ridx;   // raw buffer descriptors of ext (bytes) and idx
};
__device__ __forceinline__ constexpr int cpm_role(int i) { return i == 0 ? 0 : ((i & 1) ? 1 : 2); }

template <int KS, int UT>
__device__ __forceinline__ void cpm_fetch(cpm_state<KS, UT>& S, const cpm_args& A, int w) {
    const int p0 = POOLW * w, wi = p0 >> 4, ni = p0 >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) S.gw[i] = A.pk2[(size_t)min(wi + i, A.PW - 1) * A.Bs + A.b];
#pragma unroll
    for (int i = 0; i < 3; ++i) S.mw[i] = A.nmask[(size_t)min(ni + i, A.NW - 1) * A.Bs + A.b];
}
// window w's words -> xs / nx (the fetched words are consumed; the fetch of window w + 1 follows)
template <int KS, int UT>
__device__ __forceinline__ void cpm_window_words(cpm_state<KS, UT>& S, const cpm_args& A, int w) {
    const int p0 = POOLW * w;
    const int sh = (p0 & 15) * 2, nsh = p0 & 31;
    const uint32_t w0 = __funnelshift_r(S.gw[0], S.gw[1], sh), w1 = __funnelshift_r(S.gw[1], S.gw[2], sh),
                   w2 = __funnelshift_r(S.gw[2], S.gw[3], sh);
    // the lane's k-half starts two taps further on
    S.xs[0] = __funnelshift_r(w0, w1, 4 * A.kh); S.xs[1] = __funnelshift_r(w1, w2, 4 * A.kh);
    S.xs[2] = w2 >> (4 * A.kh);
    const uint32_t nm0 = __funnelshift_r(S.mw[0], S.mw[1], nsh), nm1 = __funnelshift_r(S.mw[1], S.mw[2], nsh);
    S.nx[0] = __funnelshift_r(nm0, nm1, 2 * A.kh); S.nx[1] = nm1 >> (2 * A.kh);
    cpm_fetch(S, A, w + 1);
}
// B operand of (position i of the window in xs, k-step ks) into buffer `buf`
template <int KS, int UT, int I, int KSI, int BUF>
__device__ __forceinline__ void cpm_operand(cpm_state<KS, UT>& S, const cpm_args& A) {
    typedef __attribute__((address_space(3))) cu32x4 lds_u4;
    typedef __attribute__((address_space(3))) char lds_char;
    constexpr int off = 2 * I + 8 * KSI, wd = off >> 5, o = off & 31;
    constexpr int noff = I + 4 * KSI, nwd = noff >> 5, no = noff & 31;
    uint32_t cs, ns;
    if constexpr (o + 4 <= 32) {
        if constexpr (o >= 4) cs = S.xs[wd] >> (o - 4); else cs = S.xs[wd] << (4 - o);
    } else {
        cs = __funnelshift_r(S.xs[wd], S.xs[wd + 1], o) << 4;
    }
    if constexpr (no + 2 <= 32) {
        if constexpr (no >= 8) ns = S.nx[nwd] >> (no - 8); else ns = S.nx[nwd] << (8 - no);
    } else {
        ns = __funnelshift_r(S.nx[nwd], S.nx[nwd + 1], no) << 8;
    }
    const uint32_t a = ((cs & 0xF0u) | A.tbase) | (ns & 0x300u);
    S.bop[BUF][KSI] = *reinterpret_cast<const lds_u4*>((const lds_char*)(size_t)a);
}
// max / argmax update of element (t, r) with the sums of position PI held in set ROLE, in two halves:
// the compare (lane mask into an SGPR pair) and, two elements later, the two selects that read it --
// a select within two instructions of the compare that writes its mask costs wait states, and a
// mask the compiler carries itself across the MFMA blocks gets materialised in a VGPR.  Inline
// assembly, so the distance is this file's business (cpm_position keeps three masks in rotation).
template <int KS, int UT, bool IDX, int PI, int ROLE>
__device__ __forceinline__ void cpm_update_cmp(cpm_state<KS, UT>& S, int t, int r, unsigned long long& gt) {
    const float v = S.acc[ROLE][t][r];
    if (PI == 0) {
        S.best[t][r] = v;                  // (the offset starts with position 1's select: 0 or 1)
    } else if (IDX) {
        asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(gt) : "v"(v), "v"(S.best[t][r]));
    } else {
        // (fmaxf would canonicalise both operands first: three v_max per element)
        float bv = S.best[t][r];
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(bv) : "v"(v));
        S.best[t][r] = bv;
    }
}
template <int KS, int UT, bool IDX, int PI, int ROLE>
__device__ __forceinline__ void cpm_update_sel(cpm_state<KS, UT>& S, int t, int r, unsigned long long gt) {
    if (PI != 0 && IDX) {
        const float v = S.acc[ROLE][t][r];
        float bv = S.best[t][r];
        uint32_t av = PI == 1 ? 0u : S.arg[t][r];
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(bv) : "v"(v), "s"(gt));
        if (PI == 1) asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(av) : "s"(gt));
        else asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(av) : "n"(PI), "s"(gt));
        S.best[t][r] = bv; S.arg[t][r] = av;
    }
}
template <int KS, int UT, bool IDX, int PI, int ROLE>
__device__ __forceinline__ void cpm_update(cpm_state<KS, UT>& S, int t, int r) {
    unsigned long long gt = 0;
    cpm_update_cmp<KS, UT, IDX, PI, ROLE>(S, t, r, gt);
    asm volatile("s_nop 1");
    cpm_update_sel<KS, UT, IDX, PI, ROLE>(S, t, r, gt);
}
// store element (t, r) of window w: ext[u][w][b] (+ idx), 32 consecutive sequences per half-wave.
// Raw buffer stores: one descriptor per array + a 32-bit lane offset that walks the rows (so = the
// element offset of the lane's row, advanced as the stores go: 64 precomputed row pointers were 128
// SGPRs, spilled to lanes and read back with five wait states in front of every store).  The
// descriptor's range check is what makes the wave's first window cheap: it has no predecessor to
// store, so its offsets are parked beyond num_records with stride 0 and the hardware drops the writes
// (a dump word in memory instead had 960 waves hammering the same 4 cache lines: 9 K cycles each).
template <int KS, int UT, bool IDX>
__device__ __forceinline__ void cpm_store(cpm_state<KS, UT>& S, const cpm_args& A, int t, int r) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S.best[t][r]) ^ S.smask[t][r], A.rext, (int)S.so4, 0, 0);
    if (IDX) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)S.arg[t][r], A.ridx, (int)S.so, 0, 0);
    // next element: row + 1, or + 5 at the end of a group of four (rows 8(r/4) + 4 kh + r%4)
    if (IDX) S.so += (r & 3) == 3 ? S.r5 : S.r1;
    S.so4 += (r & 3) == 3 ? 4u * S.r5 : 4u * S.r1;
}
#define CPM_OOB 0xFFFFFF00u

// element offset of the lane's first row (tile 0 of the group, r = 0) of window w
__device__ __forceinline__ uint32_t cpm_store_origin(const cpm_args& A, int w) {
    return (uint32_t)(((32 * A.t0 + 4 * A.kh) * A.n + w) * A.Bs + A.b);
}

// One position: its MFMAs with the neighbouring positions' vector work dealt out behind them.
// Vector units of position I of window w, in order:
//   I == 6: the next window's words; then the operands of position I + 1 (of the next window for I == 6);
//   I == 0: update with position 6 of window w - 1 (set B), then the stores of the first half of the
//           tiles of window w - 1 (the wave's first window has no predecessor: it runs the same code
//           on don't-care registers and its stores are dropped by the range check);
//   I == 1: the stores of the other tiles of window w - 1, then the update (an assignment) with position 0;
//   I >= 2: the update with position I - 1.
// The MFMAs are inline assembly: accumulators in arch VGPRs (the update reads them directly; the
// compiler's own choice at one wave per SIMD is AccVGPRs, one v_accvgpr_read per element and
// position), filter fragments and B operands in AccVGPRs.  What the compiler then cannot see is the
// matrix-pipe hazard "MFMA result read by a vector instruction": the updates of a position come
// behind at least CPM_GUARD MFMAs of the next one (>= 64 cycles of matrix pipe), and the wave's last
// position is followed by explicit wait states.
#define CPM_GUARD 2
// compile-time loop (the bodies index register arrays: nothing may be left to the loop unroller,
// whose size limit once left a position rolled and the whole state in scratch)
template <int K0, int K1, typename F>
__device__ __forceinline__ void cpm_static_for(F&& f) {
    if constexpr (K0 < K1) {
        f(std::integral_constant<int, K0>{});
        cpm_static_for<K0 + 1, K1>(f);
    }
}
// the vector units of position I and the MFMA each one is issued behind
template <int KS, int UT, int I> struct cpm_sched {
    static constexpr int NM = 3 * KS * UT;
    static constexpr int T0 = (UT + 1) / 2;                     // tiles stored at position 0
    static constexpr int NX = I == 6 ? 1 : 0, NG = KS, NE = 16 * UT;
    static constexpr int NS0 = I == 0 ? 16 * T0 : 0;            // after the updates
    static constexpr int NS1 = I == 1 ? 16 * (UT - T0) : 0;     // before the updates
    // weights in units of ~5 issue cycles (tools/mfma_probe.hip: a vector instruction behind an MFMA
    // ~5, a store ~30): words 12, operand 5, update 3 (assignment 1), a pair of stores + offsets 9
    static constexpr int WX = 12, WG = 5, WE = (I == 1) ? 1 : 3, WS = 9;
    static constexpr int TOT = NX * WX + NG * WG + NE * WE + (NS0 + NS1) * WS;
    static constexpr int NU = NX + NG + NS1 + NE + NS0;
    static constexpr int E0 = NX + NG + NS1;                    // first update unit
    static constexpr int weight(int k) {
        return k < NX ? WX : k < NX + NG ? WG : k < E0 ? WS : k < E0 + NE ? WE : WS;
    }
    static constexpr int slot(int k) {
        int cum = 0;
        for (int j = 0; j < k; ++j) cum += weight(j);
        int sl = (cum * NM) / (TOT > 0 ? TOT : 1);
        // the updates (and the stores behind them) not before CPM_GUARD MFMAs of this position
        if (k >= E0 && sl < CPM_GUARD) sl = CPM_GUARD < NM ? CPM_GUARD : NM - 1;
        return sl < NM ? sl : NM - 1;
    }
    static constexpr int first_unit(int m) {                    // first unit with slot >= m
        int k = 0;
        while (k < NU && slot(k) < m) ++k;
        return k;
    }
};

template <int KS, int UT, bool IDX, int I>
__device__ __forceinline__ void cpm_position(cpm_state<KS, UT>& S, const cpm_args& A, int w) {
    typedef cpm_sched<KS, UT, I> SC;
    constexpr int NM = SC::NM;
    constexpr int RM = cpm_role(I), RN = cpm_role((I + 1) % POOLW), RP = cpm_role((I + POOLW - 1) % POOLW);
    constexpr int PI = (I + POOLW - 1) % POOLW;
    unsigned long long gt[3] = {0, 0, 0};
    cpm_static_for<0, NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int pc = 2 - m / (KS * UT), ks = (m / UT) % KS, t = m % UT;
        if constexpr (m < UT)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0"
                         : "=&v"(S.acc[RM][t]) : "a"(S.Wr[t][ks][pc]), "a"(S.bop[RM][ks]));
        else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                         : "+v"(S.acc[RM][t]) : "a"(S.Wr[t][ks][pc]), "a"(S.bop[RM][ks]));
        cpm_static_for<SC::first_unit(m), SC::first_unit(m + 1)>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < SC::NX) {
                cpm_window_words(S, A, w + 1);
            } else if constexpr (k < SC::NX + SC::NG) {
                cpm_operand<KS, UT, (I + 1) % POOLW, k - SC::NX, RN>(S, A);
            } else if constexpr (k < SC::E0) {
                constexpr int e = k - SC::NX - SC::NG;
                cpm_store<KS, UT, IDX>(S, A, SC::T0 + e / 16, e % 16);
            } else if constexpr (k < SC::E0 + SC::NE) {
                constexpr int e = k - SC::E0;
                if constexpr (e == 0 && NM <= CPM_GUARD) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                cpm_update_cmp<KS, UT, IDX, PI, RP>(S, e / 16, e % 16, gt[e % 3]);
                if constexpr (e > 1) cpm_update_sel<KS, UT, IDX, PI, RP>(S, (e - 2) / 16, (e - 2) % 16, gt[(e - 2) % 3]);
                if constexpr (e == SC::NE - 1) {
                    cpm_update_sel<KS, UT, IDX, PI, RP>(S, (e - 1) / 16, (e - 1) % 16, gt[(e - 1) % 3]);
                    cpm_update_sel<KS, UT, IDX, PI, RP>(S, e / 16, e % 16, gt[e % 3]);
                }
            } else {
                constexpr int e = k - SC::E0 - SC::NE;
                cpm_store<KS, UT, IDX>(S, A, e / 16, e % 16);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <int KS, int UT, bool IDX>
__device__ __forceinline__ void cpm_window(cpm_state<KS, UT>& S, const cpm_args& A, int w) {
    cpm_position<KS, UT, IDX, 0>(S, A, w);
    cpm_position<KS, UT, IDX, 1>(S, A, w);
    cpm_position<KS, UT, IDX, 2>(S, A, w);
    cpm_position<KS, UT, IDX, 3>(S, A, w);
    cpm_position<KS, UT, IDX, 4>(S, A, w);
    cpm_position<KS, UT, IDX, 5>(S, A, w);
    cpm_position<KS, UT, IDX, 6>(S, A, w);
}

// The filter fragments go straight into AccVGPRs (through the compiler they came as 30 loads into
// VGPRs + 120 v_accvgpr_write); nothing else the compiler sees depends on them, so the kernel waits
// for them explicitly (s_waitcnt vmcnt(0) before the first window).
template <int KS, int UT>
__device__ __forceinline__ void cpm_load_fragments(cpm_state<KS, UT>& S, const cu32x4* wsrc) {
    cpm_static_for<0, UT * KS * 3>([&S, wsrc](auto ic) {
        constexpr int i = decltype(ic)::value;           // (t, ks, pc) in table order
        cu32x4& dst = S.Wr[i / (KS * 3)][(i / 3) % KS][i % 3];
        const cu32x4* src = wsrc + 64 * i;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(dst) : "v"(src));
    });
}

template <int KS, int UT, bool IDX>
__global__ __launch_bounds__(64, (UT == 1 && KS <= 5) ? 2 : 1) void conv_pool_mm_kernel(
    const uint32_t* __restrict__ pk2, const uint32_t* __restrict__ nmask,
    const cu32x4* __restrict__ Wf, const cu32x4* __restrict__ Wsg,
    float* __restrict__ ext, uint8_t* __restrict__ idx, int n, int Bs, int PW, int NW, int wper) {
    __shared__ __attribute__((aligned(1024))) cu32x4 oh[64];      // [N bits of the two taps][code pair]
    typedef __attribute__((address_space(3))) char lds_char;
    const int lane = threadIdx.x;
    STAMP(0);
    {
        const int c4 = lane & 15, nb = lane >> 4;
        uint32_t r[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cde = (c4 >> (2 * h)) & 3;
            const uint32_t one = ((nb >> h) & 1) ? 0u : 0x3F80u;     // bf16 1.0
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q == 2 * h + (cde >> 1)) r[q] = one << (16 * (cde & 1));
        }
        oh[lane] = cu32x4{r[0], r[1], r[2], r[3]};
    }
    cpm_state<KS, UT> S;
    cpm_args A;
    A.pk2 = pk2; A.nmask = nmask; A.ext = ext; A.idx = idx; A.n = n; A.Bs = Bs; A.PW = PW; A.NW = NW;
    A.t0 = blockIdx.y * UT; A.b = blockIdx.x * 32 + (lane & 31); A.kh = lane >> 5;
    A.tbase = (uint32_t)(size_t)(const lds_char*)oh;
    A.row1 = (uint32_t)(n * Bs); A.row5 = 5u * A.row1;
    {
        const int rows = 32 * (int)gridDim.y * UT;                 // the launcher checked 4 rows n Bs < 2^31
        A.rext = __builtin_amdgcn_make_buffer_rsrc(ext, 0, 4 * rows * n * Bs, 0x00020000);
        A.ridx = __builtin_amdgcn_make_buffer_rsrc(idx, 0, rows * n * Bs, 0x00020000);
    }
    const int wbeg = blockIdx.z * wper, wend = min(n, wbeg + wper);
    if (wbeg >= wend) return;
    cpm_load_fragments(S, Wf + (size_t)A.t0 * KS * 3 * 64 + lane);
    cpm_fetch(S, A, wbeg);
    {
        const cu32x4* ssrc = Wsg + ((size_t)A.t0 * 2 + A.kh) * 4;
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const cu32x4 v = ssrc[t * 8 + q];
#pragma unroll
                for (int i = 0; i < 4; ++i) S.smask[t][4 * q + i] = v[i];
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(1);
    cpm_window_words(S, A, wbeg);
    // operands of the first position
    cpm_static_for<0, KS>([&](auto kc) { cpm_operand<KS, UT, 0, decltype(kc)::value, 0>(S, A); });
    STAMP(2);
    // (one copy of the window code for every window: a separate first-window body doubled the code
    // every wave runs through cold)
#pragma unroll 1
    for (int w = wbeg; w < wend; ++w) {
        const bool first = w == wbeg;
        S.so = first ? CPM_OOB : cpm_store_origin(A, w - 1);
        S.so4 = first ? CPM_OOB : 4u * S.so;
        S.r1 = first ? 0u : A.row1; S.r5 = first ? 0u : A.row5;
        cpm_window<KS, UT, IDX>(S, A, w);
        if (first) STAMP(3);
    }
    STAMP(4);
    // the last window: its position 6 sits in set B (results of the last MFMAs: wait them out)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) cpm_update<KS, UT, IDX, POOLW - 1, 2>(S, t, r);
    S.so = cpm_store_origin(A, wend - 1);
    S.so4 = 4u * S.so;
    S.r1 = A.row1; S.r5 = A.row5;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) cpm_store<KS, UT, IDX>(S, A, t, r);
    STAMP(5);
}

static int conv_pool_mm_ut(const explainn_ctx* c) {
    if (const char* e = getenv("EXPLAINN_CPM_UT")) { const int v = atoi(e); if (v == 1 || (v == 2 && conv_ksteps(c->k) <= 5)) return v; }
    return conv_ut(c->k);
}

static int conv_pool_mm_parts(const explainn_ctx* c, int B, int ut) {
    // window ranges per (32 sequences, unit group): all the waves resident in one round
    if (const char* e = getenv("EXPLAINN_CPM_PARTS")) { const int v = atoi(e); if (v >= 1) return min(v, c->n); }
    const int groups = ut == 1 ? (c->U + 31) / 32 : conv_tiles_padded(c->U, c->k) / ut, sb = (B + 31) / 32;
    int parts = (ut == 1 ? 2048 : 1024) / (groups * sb);
    if (parts > c->n) parts = c->n;
    if (parts < 1) parts = 1;
    return parts;
}

int launch_conv_pool_mm(explainn_ctx* c, const explainn_params* p, int B, bool want_idx, hipStream_t s) {
    const int ut = conv_pool_mm_ut(c);
    const int parts = conv_pool_mm_parts(c, B, ut);
    const int wper = (c->n + parts - 1) / parts;
    const dim3 grid((B + 31) / 32, ut == 1 ? (c->U + 31) / 32 : conv_tiles_padded(c->U, c->k) / ut, (c->n + wper - 1) / wper);
#define ARGS grid, dim3(64), 0, s, c->pk2, c->nmask, reinterpret_cast<const cu32x4*>(c->Wf), \
             reinterpret_cast<const cu32x4*>(c->Wsg), c->ext, c->idx, c->n, c->Bs, c->PW, c->NW, wper
#define CALLKS(KSv, UTv)                                                                        \
    if (want_idx) hipLaunchKernelGGL((conv_pool_mm_kernel<KSv, UTv, true>), ARGS);              \
    else hipLaunchKernelGGL((conv_pool_mm_kernel<KSv, UTv, false>), ARGS);
#define CALLUT(KSv) if (ut == 2) { CALLKS(KSv, 2); } else { CALLKS(KSv, 1); }
    switch (conv_ksteps(c->k)) {
#ifndef CPM_ONLY5
        case 1: { CALLUT(1); } break;
        case 2: { CALLUT(2); } break;
        case 3: { CALLUT(3); } break;
        case 4: { CALLUT(4); } break;
        case 6: { CALLKS(6, 1); } break;
        case 7: { CALLKS(7, 1); } break;
        case 8: { CALLKS(8, 1); } break;
#endif
        case 5: { CALLUT(5); } break;
        default: explainn_set_error("kernel_size %d not instantiated (2..32)", c->k); return EXPLAINN_E_UNSUPPORTED;
    }
#undef CALLUT
#undef CALLKS
#undef ARGS
    LAUNCH_CHECK();
    return EXPLAINN_OK;
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
    static const bool mm = [] { const char* e = getenv("EXPLAINN_CONV_MM"); return !e || atoi(e) != 0; }();
    // (the GEMM form addresses ext through a raw buffer descriptor: 32-bit byte offsets)
    const bool fits = (int64_t)32 * conv_tiles_padded(c->U, c->k) * c->n * c->Bs * 4 < (int64_t)1 << 31;
    if (mm && fits) return launch_conv_pool_mm(c, p, B, want_idx, s);
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
