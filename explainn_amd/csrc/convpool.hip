// The filter bank: Conv1d(4U->U, k, groups=U) + BatchNorm1 + exp + MaxPool1d(7,7)
// (architectures/__init__.py:73-81) as ONE kernel that never materialises the conv output.
//
// One-hot input makes the convolution a sum of one table entry per tap: conv[b,u,p] = sum_j W[u, s[b,p+j], j].
// Rounds 1-2 did that as a gather (dinucleotide tables in LDS, 40 B of LDS traffic per unit and
// position, 0.61 of the LDS array); round 3 puts it on the matrix core as a GEMM against the one-hot
// bits.  BatchNorm+exp are monotone per unit, so the 7-wide max-pool runs on the raw sums with the
// sign of alpha = gamma1/sigma1 choosing max or min; only the pooled extreme (and its offset, for
// the backward routing) leaves the kernel: ext[u][w][b], idx[u][w][b].
#include <cstdlib>
#include <type_traits>

#include "common.h"

// ---------------------------------------------------------------------------------------------
// The filter bank on the matrix core.  conv[u][(b,p)] = sum_k Wf[u][k] X[k][(b,p)] with
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

// element offset, inside the group's rows, of the lane's first row (tile 0, r = 0) of window w
__device__ __forceinline__ uint32_t cpm_store_origin(const cpm_args& A, int w) {
    return (uint32_t)((4 * A.kh * A.n + w) * A.Bs + A.b);
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
    {   // descriptors of the group's own 32 UT rows: offsets stay far below 2^31 at any model size
        // (made wave-uniform by hand: a descriptor the compiler cannot prove uniform gets a
        // readfirstlane waterfall loop around every store -- seen in the stamped build of this kernel)
        const size_t row0 = (size_t)32 * A.t0 * n * Bs;
        const int span = __builtin_amdgcn_readfirstlane(32 * UT * n * Bs);
        auto uniform = [](auto* q) {
            const unsigned long long v = reinterpret_cast<unsigned long long>(q);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return reinterpret_cast<decltype(q)>(((unsigned long long)hi << 32) | lo);
        };
        A.rext = __builtin_amdgcn_make_buffer_rsrc(uniform(ext + row0), 0, 4 * span, 0x00020000);
        A.ridx = __builtin_amdgcn_make_buffer_rsrc(uniform(idx + row0), 0, span, 0x00020000);
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
    // Window ranges per (32 sequences, unit group).  A wave costs about one window's worth of fixed
    // work (fragments, table, the cold first pass through the code, the last stores) plus its windows,
    // and the launch takes as many rounds as its waves need SIMD slots: pick the split that minimises
    // rounds x (windows per wave + 1).
    if (const char* e = getenv("EXPLAINN_CPM_PARTS")) { const int v = atoi(e); if (v >= 1) return min(v, c->n); }
    const int groups = ut == 1 ? (c->U + 31) / 32 : conv_tiles_padded(c->U, c->k) / ut, sb = (B + 31) / 32;
    const int cols = groups * sb, slots = (ut == 1 && conv_ksteps(c->k) <= 5) ? 2048 : 1024;
    int best = 1;
    long best_cost = -1;
    for (int p = 1; p <= c->n; ++p) {
        const int wper = (c->n + p - 1) / p, np = (c->n + wper - 1) / wper;
        if (np != p) continue;                                    // (same split as a smaller p)
        const long rounds = ((long)cols * np + slots - 1) / slots;
        const long cost = rounds * (wper + 1);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = p; }
    }
    return best;
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
#ifndef CPM_ONLY5     /* (development builds: -DCPM_ONLY5 compiles the k = 17..20 forms only) */
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

int launch_conv_pool(explainn_ctx* c, const explainn_params* p, int B, bool want_idx, hipStream_t s) {
    return launch_conv_pool_mm(c, p, B, want_idx, s);
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
