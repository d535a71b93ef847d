// The per-unit two-layer FC (architectures/__init__.py:84-100) and its backward on the exact-fp32
// matrix cores: v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain and runs at the fp32
// vector rate, but takes both operands from VGPRs -- no scalar/LDS broadcast stream to feed.
// FC1 is the one dense contraction on this path: per unit (B x n).(n x 100).
//
// 16x16x4 tiles (round 2; round 1 used 32x32x2): the hidden width 100 pads to 7 x 16 = 112 instead of
// 4 x 32 = 128, the accumulator tile is 4 registers instead of 16 (all 7 channel tiles of a sequence
// tile stay live, so the operand of a k-step is built once and used 7 times), and the dependent
// latency is 40 cycles against an issue interval of 32: two independent chains saturate the pipe.
//
// Fragment convention (cdna_hip_programming.md section 3), lane l, c = l&15, g = l>>4:
//   D[i][j] = sum_k A[i][k] B[k][j] + C[i][j];  A operand = A[c][4s+g],  B operand = B[4s+g][c]
//   C/D register i of lane l holds D[4g + i][c].
//
//   fc_fwd   D[r][b] = sum_w A2[r][w] q[b][w] + sh2[r].  A = folded FC1 weights (fragments staged
//            in LDS once per workgroup), B = q of 16 sequences (registers).  A lane ends up with 4
//            hidden channels of ITS sequence per channel tile: ReLU, dropout, the FC2 dot product
//            z += V2[r]*a and the "relu'>0 and kept" bits are lane-local; the four lane groups of a
//            sequence are merged with two cross-group shuffles.  Nothing of size (B,100U) is stored.
//   passA    D[r][w] = sum_b e[b][r] q[b][w],  e = dz[b]*bit[b][r] built on the fly from the bit
//            words (A operand), q rows from a wave-private LDS tile (B operand); K = sequences.
//   passB    D[w][b] = dz[b] sum_r T[r][w] bit[b][r] - sum_v M[v][w] q[b][v] - k0'[w]; then dy = dq*q
//            and the two BatchNorm1-backward sums.  A = T (bf16 pieces) and M fragments in LDS.  The k
//            order of the M.q product is chosen so that a lane's q operands ARE the q values of its D rows.
//   passA / the T.bit part of passB have a BIT matrix as one operand: they run on the bf16 matrix
//   core with the real operand split exactly into three bf16 pieces (see passA).
//   qmom     (prep.hip) the q second-moment matrix.
//
// Template parameter NQ >= n (bucketed pooled length).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// Lane-local epilogue of one 16-channel tile: the lane holds channels r = 16t + 4g + j (j = 0..3) of ITS
// sequence.  ReLU, dropout (MODE 2: the built-in generator; MODE 3: the caller's keep mask), the FC2
// partial dot product (v2s = V2 * 1/(1-p): the dropout scale is folded into the weights when they
// are staged) and the "relu' > 0 and kept" bits.
//
// fc_fwd is bound by its vector instructions (10.1 M of them at C2, 19.6 us of vector-pipe time in
// a 27.5 us kernel, profiles/r03_mid_*), three quarters of them this epilogue, so it is written for
// instruction count, per channel:
//   * the conditions live in SGPR pairs (one v_cmp each, combined on the scalar unit);
//   * the accumulation zp += v2 * y runs under that mask as EXEC (s_and_saveexec / v_fmac /
//     s_mov): one vector instruction instead of a select and an FMA -- the scalar instructions
//     issue on their own port;
//   * the bit is shifted into the lane's word by an add-with-carry that takes the mask as its
//     carry-in (nib = nib + nib + bit): one instruction instead of a select and an or;
//   * the generator is one 24-bit multiply-add per draw (below) instead of half a six-instruction
//     xorshift step, and the 16-bit draw is compared in place (SDWA) as before.
// 11 -> 5 vector instructions per channel.
//
// Generator (MODE 2): per (seed, sequence, lane quarter, unit) a linear congruential stream
// s' = (s mod 2^24) * 0xFD43FD + 0xC39EC3 (full period 2^24: multiplier = 5 mod 8, odd increment), started from
// a three-round integer hash of the key; the draw is bits 16..31 of the 32-bit multiply-add, i.e. the
// top byte of the next state under the product's overflow byte -- the well-mixed end of an LCG.
// Keep rate, scaling and independence across channels / sequences / units / seeds are measured by
// tests/test_gpu_parity.py::test_builtin_dropout_generator_statistics on the mask itself.
__device__ __forceinline__ void fmac_under_mask(float& acc, float a, float b, unsigned long long m) {
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %0, %4\n\tv_fmac_f32 %1, %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved), "+v"(acc) : "v"(a), "v"(b), "s"(m) : "scc");
}
__device__ __forceinline__ uint32_t shift_in_bit(uint32_t nib, unsigned long long m) {
    unsigned long long cout;
    asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(nib), "=s"(cout) : "s"(m));
    return nib;
}

template <int MODE>
__device__ __forceinline__ void fc_epilogue_tile(const f32x4& acc, int t, int g, const float* v2s,
                                                 uint32_t& rs, uint32_t thresh16, const uint8_t* km,
                                                 float& zp, uint32_t (&words)[4]) {
    const float4 v2 = *reinterpret_cast<const float4*>(&v2s[16 * t + 4 * g]);
    const float v2a[4] = {v2.x, v2.y, v2.z, v2.w};
    if (MODE == 0) {
        // eval: no bits to keep (the shift-ins are volatile assembly: they stayed in the eval kernel,
        // 28 per 16 sequences, until this branch), ReLU as a plain max
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = acc[j];
            asm("v_max_f32 %0, 0, %0" : "+v"(y));          // (fmaxf would canonicalise first)
            zp = fmaf(v2a[j], y, zp);
        }
        return;
    }
    unsigned long long pm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float y = acc[j];
        pm[j] = __builtin_amdgcn_ballot_w64(y > 0.f);
        if (MODE == 2) {
            rs = __umul24(rs, 0xFD43FDu) + 0xC39EC3u;
            pm[j] &= __builtin_amdgcn_ballot_w64((rs >> 16) >= thresh16);
        } else if (MODE == 3) {
            const int r = 16 * t + 4 * g + j;
            pm[j] &= __builtin_amdgcn_ballot_w64(km[r < FC_H ? r : 0] != 0);
        }
        fmac_under_mask(zp, v2a[j], y, pm[j]);
    }
    // bit j of the nibble = channel 4g + j of the tile: shifted in from the top
    uint32_t nib = 0u;
#pragma unroll
    for (int j = 3; j >= 0; --j) nib = shift_in_bit(nib, pm[j]);
    // channel r = 16t + 4g + j is bit (r & 31) of word r >> 5
    words[t >> 1] |= nib << (16 * (t & 1) + 4 * g);
}

// Workgroup total of the z moments (fixed order: lanes by butterfly, waves 0..3) -> z12p[u][bx]
template <int W>
__device__ __forceinline__ void fc_zmom_finish(double s1, double s2, double* __restrict__ z12p, int u,
                                               int bx, int nbx, int wave, int lane) {
    __shared__ double zred[W][2];
    // only lanes 0..15 (g = 0) hold sums: four butterfly steps bring their total to lane 0
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    if (lane == 0) { zred[wave][0] = s1; zred[wave][1] = s2; }
    __syncthreads();
    if (wave == 0 && lane == 0) {
        double a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) { a += zred[i][0]; b += zred[i][1]; }
        double* dst = z12p + ((size_t)u * nbx + bx) * 2;
        dst[0] = a; dst[1] = b;
    }
}

// FC_BTW (common.h): 16-sequence tiles per wavefront in fc_fwd
#ifndef PB_BTW
#define PB_BTW 4                     // 16-sequence tiles per wavefront in passB
#endif
#define FC_AHEAD 2                   // A-fragment reads in flight ahead of their MFMAs in the bf16 fc_fwd (3 spills one register at 96)

// MODE: 0 eval, 1 train without dropout, 2 train + counter-based generator, 3 train + keep-mask
template <int NQ, int MODE>
__global__ __launch_bounds__(64 * fc_fwd_waves(NQ)) void fc_fwd_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, const float* __restrict__ A2f, const float* __restrict__ sh2,
    const float* __restrict__ V2, uint4* __restrict__ bits, float* __restrict__ zout,
    const uint8_t* __restrict__ keep_mask, uint32_t thresh16, float scale, uint32_t seed_lo,
    uint32_t seed_hi, const float* __restrict__ c2, const float* __restrict__ g3,
    const float* __restrict__ b3, const float* __restrict__ rm3, const float* __restrict__ rv3,
    float* __restrict__ oout, int n, int Bs, int B, int U,
    double* __restrict__ z12p) {
    constexpr int NK4 = fc_nk4(NQ), NK4Q = fc_nk4q(NQ), FW = fc_fwd_waves(NQ), NTH = 64 * FW;
    constexpr bool TRAIN = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float4* Af = reinterpret_cast<float4*>(fsm);            // [FC_MT][NK4Q][64] float4 (4 k-steps each)
    float* sh2s = fsm + FC_MT * NK4Q * 64 * 4;               // [112]
    float* v2s = sh2s + FC_MT * 16;                          // [112]
    int u, bx;
    if (!unit_chunk_of_block(U, u, bx)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const float* __restrict__ eu = ext + (size_t)u * n * Bs;
    const int bt0 = (bx * FW + wave) * FC_BTW;
    STAMP(0);
    // the first tile's raw pooled extremes are requested before the weight fragments are staged:
    // one memory round trip covers both
    float raw[NK4];
    {
        const int b = min(bt0 * 16 + c, Bs - 1);
#pragma unroll
        for (int s = 0; s < NK4; ++s) raw[s] = eu[min(4 * s + g, n - 1) * Bs + b];
    }
    {
        // A fragments (prep2 wrote them in fragment order):
        // Af[(t*NK4Q + sq)*64 + l].{x,y,z,w} = A2[16t + (l&15)][4*(4sq+e) + (l>>4)], e = 0..3
        const float4* src = reinterpret_cast<const float4*>(A2f) + (size_t)u * FC_MT * NK4Q * 64;
        constexpr int N4 = FC_MT * NK4Q * 64;
        float4 tv[(N4 + NTH - 1) / NTH];
#pragma unroll
        for (int i = 0; i < (N4 + NTH - 1) / NTH; ++i)
            tv[i] = (tid + i * NTH < N4) ? src[tid + i * NTH] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < (N4 + NTH - 1) / NTH; ++i)
            if (tid + i * NTH < N4) Af[tid + i * NTH] = tv[i];
    }
    if (tid < FC_MT * 16) {
        sh2s[tid] = tid < FC_H ? sh2[(size_t)u * FC_H + tid] : 0.f;
        v2s[tid] = tid < FC_H ? V2[(size_t)u * FC_H + tid] * scale : 0.f;
    }
#pragma unroll
    for (int s = 0; s < NK4; ++s) KEEP(raw[s]);
    __syncthreads();
    STAMP(1);
    const float a1 = alpha[u], s1 = shift[u];
    // train: sums of z and z^2 over this workgroup's sequences for BatchNorm3's batch statistics
    // (head.hip finishes them inside the combiner launch); fp64, unshifted: a shift would have to be
    // state (the running mean) and make the step's rounding depend on it
    double zs1 = 0, zs2 = 0;
    for (int it = 0; it < FC_BTW; ++it) {
        const int bt = bt0 + it;
        if (bt * 16 >= B) break;                       // wave-uniform
        const int b = bt * 16 + c;
        float qf[NK4];
#pragma unroll
        for (int s = 0; s < NK4; ++s) qf[s] = (4 * s + g < n) ? qval(a1, raw[s], s1) : 0.f;
        if (it + 1 < FC_BTW && (bt + 1) * 16 < B) {    // next tile's loads fly under this tile's MFMAs
            const int bn = min(b + 16, Bs - 1);
#pragma unroll
            for (int s = 0; s < NK4; ++s) raw[s] = eu[min(4 * s + g, n - 1) * Bs + bn];
        }
        uint32_t rs = 0;
        if (MODE == 2)
            rs = mix32(mix32(seed_lo ^ (uint32_t)(4 * b + g) * 0x9E3779B9U) ^
                       mix32(seed_hi + (uint32_t)u));
        const uint8_t* km = (MODE == 3) ? keep_mask + (size_t)min(b, B - 1) * FC_H * U + (size_t)u * FC_H
                                        : nullptr;
        if (it == 0) STAMP(2);
        float zp = 0.f;
        uint32_t words[4] = {0u, 0u, 0u, 0u};
        // all seven channel tiles first (seven independent accumulator chains, k-step outermost: one
        // operand read per tile and step), then the lane-local epilogue of all 28 channels
        f32x4 acc[FC_MT];
#pragma unroll
        for (int t = 0; t < FC_MT; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(&sh2s[16 * t + 4 * g]);
            acc[t][0] = v.x; acc[t][1] = v.y; acc[t][2] = v.z; acc[t][3] = v.w;
        }
#pragma unroll
        for (int sq = 0; sq < NK4Q; ++sq) {
#pragma unroll
            for (int t = 0; t < FC_MT; ++t) {
                const float4 a4 = Af[(t * NK4Q + sq) * 64 + lane];
                const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * sq + e < NK4) acc[t] = MFMA16(aa[e], qf[4 * sq + e], acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < FC_MT; ++t)
            fc_epilogue_tile<MODE>(acc[t], t, g, v2s, rs, thresh16, km, zp, words);
        if (it == 0) STAMP(3);
        // the four lane groups of a sequence hold disjoint channel sets: merge across g
        zp += __shfl_xor(zp, 16, 64);
        zp += __shfl_xor(zp, 32, 64);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            words[k] |= __shfl_xor(words[k], 16, 64);
            words[k] |= __shfl_xor(words[k], 32, 64);
        }
        if (TRAIN && g == 0 && b < B) { const double d = (double)zp; zs1 += d; zs2 = fma(d, d, zs2); }
        if (g == 0 && b < Bs) {
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
    if (TRAIN) fc_zmom_finish<FW>(zs1, zs2, z12p, u, bx, gridDim.x, wave, lane);
    STAMP(4);
}

// ---------------------------------------------------------------------------------------------
// fc_fwd on the bf16 matrix core (n <= FC_BF_MAXN).  Measured (profiles/r02): the fp32 MFMA and the
// vector instructions of the epilogue do not overlap on a SIMD -- the kernel took the SUM of its
// MFMA-busy and vector-issue times -- while a bf16 MFMA holds the vector issue for 8 of its 16
// cycles.  Both operands are real here, so both are split exactly into three bf16 pieces
// (x = hi + mid + lo with hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = x - hi - mid: every
// subtraction exact, 8+8+8 mantissa bits) and all nine piece products are accumulated: each is exact
// in fp32 (8 x 8 bits), so D = sh2 + sum_w sum_pieces a_i q_j is an fp32 sum of exact products -- the
// same function as the fmaf chain up to summation order -- at 9/16 of the fp32 MFMA's cycles and one
// 32-wide k-step for the 26 pooled positions of the headline shape.
//   A[row c][k = 8g + j] = A2 pieces (prep2's A2h image, staged in LDS), B[k = 8g + j][col c] = q pieces of
//   sequence c built in registers; D layout as above, so the epilogue is the fp32 kernel's.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 fbf16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NQ, int MODE>
__global__ __launch_bounds__(64 * fc_fwd_waves(NQ), fc_ks32(NQ) == 1 ? 5 : 2) void fc_fwd_bf_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, const uint32_t* __restrict__ A2h, const float* __restrict__ sh2,
    const float* __restrict__ V2, uint4* __restrict__ bits, float* __restrict__ zout,
    const uint8_t* __restrict__ keep_mask, uint32_t thresh16, float scale, uint32_t seed_lo,
    uint32_t seed_hi, const float* __restrict__ c2, const float* __restrict__ g3,
    const float* __restrict__ b3, const float* __restrict__ rm3, const float* __restrict__ rv3,
    float* __restrict__ oout, int n, int Bs, int B, int U,
    double* __restrict__ z12p) {
    constexpr int KS = fc_ks32(NQ), FW = fc_fwd_waves(NQ), NTH = 64 * FW;
    constexpr bool TRAIN = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    u32x4* Ah = reinterpret_cast<u32x4*>(fsm);               // [FC_MT][KS][3][64] x 8 bf16
    float* sh2s = fsm + FC_MT * KS * 3 * 256;                // [112]
    float* v2s = sh2s + FC_MT * 16;                          // [112]
    int u, bx;
    if (!unit_chunk_of_block(U, u, bx)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const float* __restrict__ eu = ext + (size_t)u * n * Bs;
    const int bt0 = (bx * FW + wave) * FC_BTW;
    STAMP(0);
    // the first tile's raw pooled extremes are requested before the weight fragments are staged
    float raw[KS][8];
    {
        const int b = min(bt0 * 16 + c, Bs - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[ks][j] = eu[min(32 * ks + 8 * g + j, n - 1) * Bs + b];
    }
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(A2h) + (size_t)u * FC_MT * KS * 3 * 64;
        constexpr int N4 = FC_MT * KS * 3 * 64;
        u32x4 tv[(N4 + NTH - 1) / NTH];
#pragma unroll
        for (int i = 0; i < (N4 + NTH - 1) / NTH; ++i) tv[i] = src[min(tid + i * NTH, N4 - 1)];
#pragma unroll
        for (int i = 0; i < (N4 + NTH - 1) / NTH; ++i)
            if (tid + i * NTH < N4) Ah[tid + i * NTH] = tv[i];
    }
    if (tid < FC_MT * 16) {
        sh2s[tid] = tid < FC_H ? sh2[(size_t)u * FC_H + tid] : 0.f;
        v2s[tid] = tid < FC_H ? V2[(size_t)u * FC_H + tid] * scale : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) KEEP(raw[ks][j]);
    __syncthreads();
    STAMP(1);
    const float a1 = alpha[u], s1 = shift[u];
    // train: sums of z and z^2 over this workgroup's sequences for BatchNorm3's batch statistics
    // (head.hip finishes them inside the combiner launch); fp64, unshifted: a shift would have to be
    // state (the running mean) and make the step's rounding depend on it
    double zs1 = 0, zs2 = 0;
    for (int it = 0; it < FC_BTW; ++it) {
        const int bt = bt0 + it;
        if (bt * 16 >= B) break;                       // wave-uniform
        const int b = bt * 16 + c;
        // q of this tile, split into its three bf16 pieces, two elements per word
        u32x4 bq[KS][3];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                float x[2], r1[2], r2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    x[h] = (32 * ks + 8 * g + 2 * jp + h < n) ? qval(a1, raw[ks][2 * jp + h], s1) : 0.f;
                    r1[h] = x[h] - __uint_as_float(__float_as_uint(x[h]) & 0xffff0000u);
                    r2[h] = r1[h] - __uint_as_float(__float_as_uint(r1[h]) & 0xffff0000u);
                }
                // upper halves of the two values -> one word (element 2jp low, 2jp+1 high)
                bq[ks][0][jp] = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
                bq[ks][1][jp] = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                bq[ks][2][jp] = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
            }
        }
        if (it + 1 < FC_BTW && (bt + 1) * 16 < B) {    // next tile's loads fly under this tile's MFMAs
            const int bn = min(b + 16, Bs - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[ks][j] = eu[min(32 * ks + 8 * g + j, n - 1) * Bs + bn];
        }
        uint32_t rs = 0;
        if (MODE == 2)
            rs = mix32(mix32(seed_lo ^ (uint32_t)(4 * b + g) * 0x9E3779B9U) ^
                       mix32(seed_hi + (uint32_t)u));
        const uint8_t* km = (MODE == 3) ? keep_mask + (size_t)min(b, B - 1) * FC_H * U + (size_t)u * FC_H
                                        : nullptr;
        if (it == 0) STAMP(2);
        float zp = 0.f;
        uint32_t words[4] = {0u, 0u, 0u, 0u};
        f32x4 acc[FC_MT];
#pragma unroll
        for (int t = 0; t < FC_MT; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(&sh2s[16 * t + 4 * g]);
            acc[t][0] = v.x; acc[t][1] = v.y; acc[t][2] = v.z; acc[t][3] = v.w;
        }
        // nine piece products per channel tile and k-step, smallest pieces first.  The A fragments
        // come from LDS FC_AHEAD groups ahead of the three MFMAs that use them (a group = one
        // fragment, three dependent MFMAs = 48 cycles; a ds_read_b128 takes ~130)
        {
            constexpr int NGRP = KS * 3 * FC_MT;
            auto frag = [&](int i) -> u32x4 {
                const int t = i % FC_MT, pa = 2 - (i / FC_MT) % 3, ks = i / (3 * FC_MT);
                return Ah[((t * KS + ks) * 3 + pa) * 64 + lane];
            };
            u32x4 fr[FC_AHEAD + 1];
#pragma unroll
            for (int i = 0; i < FC_AHEAD; ++i) fr[i] = frag(i < NGRP ? i : NGRP - 1);
#pragma unroll
            for (int i = 0; i < NGRP; ++i) {
                if (i + FC_AHEAD < NGRP) fr[(i + FC_AHEAD) % (FC_AHEAD + 1)] = frag(i + FC_AHEAD);
                const int t = i % FC_MT, ks = i / (3 * FC_MT);
                const fbf16x8 afr = __builtin_bit_cast(fbf16x8, fr[i % (FC_AHEAD + 1)]);
#pragma unroll
                for (int pb = 2; pb >= 0; --pb)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        afr, __builtin_bit_cast(fbf16x8, bq[ks][pb]), acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int t = 0; t < FC_MT; ++t)
            fc_epilogue_tile<MODE>(acc[t], t, g, v2s, rs, thresh16, km, zp, words);
        if (it == 0) STAMP(3);
        zp += __shfl_xor(zp, 16, 64);
        zp += __shfl_xor(zp, 32, 64);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            words[k] |= __shfl_xor(words[k], 16, 64);
            words[k] |= __shfl_xor(words[k], 32, 64);
        }
        if (TRAIN && g == 0 && b < B) { const double d = (double)zp; zs1 += d; zs2 = fma(d, d, zs2); }
        if (g == 0 && b < Bs) {
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
    if (TRAIN) fc_zmom_finish<FW>(zs1, zs2, z12p, u, bx, gridDim.x, wave, lane);
    STAMP(4);
}

template <int NQ>
static size_t fc_fwd_bf_lds() {
    return (size_t)(FC_MT * fc_ks32(NQ) * 3 * 256 + 2 * FC_MT * 16) * sizeof(float);
}

template <int NQ>
static size_t fc_fwd_lds() {
    return (size_t)(FC_MT * fc_nk4q(NQ) * 64 * 4 + 2 * FC_MT * 16) * sizeof(float);
}

// (templates, so that only the form a bucket uses is instantiated)
template <int N, int MODE>
static void fc_fwd_launch_nm(explainn_ctx* c, const explainn_params* p, int B, dim3 grid,
                             const uint8_t* keep_mask, uint32_t thresh, float scale, uint64_t seed,
                             hipStream_t s) {
    if constexpr (N <= FC_BF_MAXN)
        hipLaunchKernelGGL((fc_fwd_bf_kernel<N, MODE>), grid, dim3(64 * fc_fwd_waves(N)), fc_fwd_bf_lds<N>(), s, c->ext,
                           c->alpha, c->shift, reinterpret_cast<const uint32_t*>(c->A2h), c->sh2,
                           p->fc2_w, c->bits, c->z, keep_mask, thresh, scale, (uint32_t)seed,
                           (uint32_t)(seed >> 32), p->fc2_b, p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv,
                           c->o, c->n, c->Bs, B, c->U, c->z12p);
    else
        hipLaunchKernelGGL((fc_fwd_kernel<N, MODE>), grid, dim3(64 * fc_fwd_waves(N)), fc_fwd_lds<N>(), s, c->ext,
                           c->alpha, c->shift, c->A2f, c->sh2, p->fc2_w, c->bits, c->z, keep_mask,
                           thresh, scale, (uint32_t)seed, (uint32_t)(seed >> 32), p->fc2_b, p->bn3_w,
                           p->bn3_b, p->bn3_rm, p->bn3_rv, c->o, c->n, c->Bs, B, c->U, c->z12p);
}

template <int N, int MODE>
static int fc_fwd_configure_nm() {
    if constexpr (N <= FC_BF_MAXN) {
        if (fc_fwd_bf_lds<N>() > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_fwd_bf_kernel<N, MODE>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)fc_fwd_bf_lds<N>()));
    } else {
        if (fc_fwd_lds<N>() > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_fwd_kernel<N, MODE>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)fc_fwd_lds<N>()));
    }
    return EXPLAINN_OK;
}

int launch_fc_fwd(explainn_ctx* c, const explainn_params* p, int B, bool train,
                  const uint8_t* keep_mask, float drop_p, uint64_t seed, hipStream_t s) {
    const int tiles = (B + 15) / 16;
    const dim3 grid(fc_fwd_blocks(B, c->NQ), units_grid(c->U));
    int mode = train ? 1 : 0;
    float scale = 1.f;
    uint32_t thresh = 0;
    if (train && drop_p > 0.f) {
        mode = keep_mask ? 3 : 2;
        scale = 1.0f / (1.0f - drop_p);
        thresh = (uint32_t)(drop_p * 65536.0 + 0.5);
    }
    if (train) { c->fwd_drop = mode > 1; c->fwd_scale = scale; }
#define CALL(N)                                                                                    \
    switch (mode) {                                                                                \
        case 0: fc_fwd_launch_nm<N, 0>(c, p, B, grid, keep_mask, thresh, scale, seed, s); break;   \
        case 1: fc_fwd_launch_nm<N, 1>(c, p, B, grid, keep_mask, thresh, scale, seed, s); break;   \
        case 2: fc_fwd_launch_nm<N, 2>(c, p, B, grid, keep_mask, thresh, scale, seed, s); break;   \
        default: fc_fwd_launch_nm<N, 3>(c, p, B, grid, keep_mask, thresh, scale, seed, s); break;  \
    }
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// passA: EQ[r][w] = sum_b e[b][r] q[b][w] with e[b][r] = dz[b] * bit[b][r], and Se[r] = sum_b e[b][r].
// One wavefront per (unit, batch chunk, group of w tiles); K dimension = sequences.
//
// One operand of this product is a BIT matrix.  Written as  EQ = bit^T . X  with X[b][w] = dz[b] q[b][w]
// (and one extra column X[b][n] = dz[b], which makes Se the (n+1)-th column of the same product), the
// bits are exact in bf16 (0.0 / 1.0) and X is exact as the sum of three bf16 pieces (hi = the top 16
// bits of the float, mid = the top 16 bits of x - hi, lo = x - hi - mid: 8 + 8 + 8 significant
// bits).  Every product bit * piece is exact and the matrix core accumulates in fp32: the result is
// an fp32 dot product, at 3/16 of the fp32-MFMA cost (v_mfma_f32_16x16x32_bf16: 32 sequences per
// instruction at 16 cycles against 4 at 32).  Round 1/2a ran this on v_mfma_f32_32x32x2 / 16x16x4 and
// was MFMA-pipe-bound at 38 us (C2).
//
// Super-tiles of 64 sequences are fetched with coalesced loads (lane = sequence): the q rows (from
// ext, through exp), the bit words and dz.  X is split and written to a wave-private LDS image
// [piece][w][sequence] (row stride 144 B: the 16-byte operand reads of 16 rows fall on distinct
// banks); the bit operand A[r][k = sequence] is assembled from the per-sequence bit words, eight
// 4-byte LDS reads per (channel-tile pair, 32 sequences), most of them broadcasts.  The next
// super-tile's loads are in flight during the MFMAs.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// The head backward (final-layer gradients, BatchNorm3 backward -> dz) for few tasks, as passA's
// prologue: it was a launch of its own (one block per unit, ~10 us at C2 for 300 x 1024 values).
// Every passA wave needs dz of its own batch chunk, and dz needs two sums over the unit's WHOLE
// batch (BatchNorm3's backward), so every wave makes that pass itself (o, zhat, and the loss
// gradient recomputed from logits/targets or read: a few loads per sequence out of L2), then
// writes dz for its chunk -- the main loop below (and passB) read it from memory; a lane reads back
// what it wrote itself.  The chunk-0 wave of a unit also owns the unit's parameter gradients, the
// (unit 0, chunk 0) wave the combiner bias gradient and the loss value.  fp64 sums, lanes by
// butterfly: fixed order.
// ---------------------------------------------------------------------------------------------
// d loss / d logit from (x = dlogit given | x = logit, t = target)
__device__ __forceinline__ float pa_dl_of(const pa_head_args& h, float invN, float x, float t) {
    if (h.mode == 1) return x;
    if (h.kind == EXPLAINN_LOSS_BCE_WITH_LOGITS) return (1.0f / (1.0f + expf(-x)) - t) * invN;
    return 2.0f * (x - t) * invN;
}
__device__ __forceinline__ float pa_dl(const pa_head_args& h, float invN, int i) {
    return h.mode == 1 ? h.dl[i] : pa_dl_of(h, invN, h.logits[i], h.y[i]);
}

__device__ __forceinline__ void pa_head_prologue(const pa_head_args& h, int u, int ch, int grp, int lane,
                                              int bbeg, int bend, int Bs, int B, int U) {
    const int T = h.T;
    const float invN = 1.0f / (float)(B * T);
    const float* __restrict__ ou = h.o + (size_t)u * Bs;
    const float* __restrict__ zh = h.zhat + (size_t)u * Bs;
    float wf[PA_HEAD_MAX_T];
#pragma unroll
    for (int t = 0; t < PA_HEAD_MAX_T; ++t) wf[t] = t < T ? h.Wf[(size_t)t * U + u] : 0.f;
    const bool owner = ch == 0 && grp == 0;
    const bool first = owner && u == 0;
    double s1 = 0, s2 = 0, gw[PA_HEAD_MAX_T], gb[PA_HEAD_MAX_T], lacc = 0;
#pragma unroll
    for (int t = 0; t < PA_HEAD_MAX_T; ++t) { gw[t] = 0; gb[t] = 0; }
    // eight 64-sequence slices per round trip: every load of a batch is issued before the first use
    constexpr int PF = 8;
    for (int b0 = 0; b0 < B; b0 += 64 * PF) {
        float ovq[PF], zvq[PF], xq[PF][PA_HEAD_MAX_T], yq[PF][PA_HEAD_MAX_T];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int bc = min(b0 + 64 * q + lane, B - 1);
            ovq[q] = ou[bc]; zvq[q] = zh[bc];
#pragma unroll
            for (int t = 0; t < PA_HEAD_MAX_T; ++t) {
                xq[q][t] = 0.f; yq[q][t] = 0.f;
                if (t < T) {
                    if (h.mode == 1) xq[q][t] = h.dl[bc * T + t];
                    else { xq[q][t] = h.logits[bc * T + t]; yq[q][t] = h.y[bc * T + t]; }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            KEEP(ovq[q]); KEEP(zvq[q]);
#pragma unroll
            for (int t = 0; t < PA_HEAD_MAX_T; ++t) { KEEP(xq[q][t]); KEEP(yq[q][t]); }
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const bool live = b0 + 64 * q + lane < B;
            const float ov = ovq[q], zv = zvq[q];
            float dob = 0.f;
#pragma unroll
            for (int t = 0; t < PA_HEAD_MAX_T; ++t) {
                if (t >= T) continue;
                const float dlv = pa_dl_of(h, invN, xq[q][t], yq[q][t]);
                dob = fmaf(dlv, wf[t], dob);
                if (owner && live) {
                    gw[t] = fma((double)dlv, (double)ov, gw[t]);
                    if (first) {
                        gb[t] += (double)dlv;
                        if (h.mode == 2) {
                            const float x = xq[q][t], tt = yq[q][t];
                            float l;
                            if (h.kind == EXPLAINN_LOSS_BCE_WITH_LOGITS) l = fmaxf(x, 0.f) - x * tt + log1pf(expf(-fabsf(x)));
                            else { const float e = x - tt; l = e * e; }
                            lacc += (double)l;
                        }
                    }
                }
            }
            const float d3 = (live && ov > 0.f) ? dob : 0.f;
            s1 += (double)d3;
            s2 = fma((double)d3, live ? (double)zv : 0.0, s2);
        }
    }
    const double S1 = wave_sum_d(s1), S2 = wave_sum_d(s2);
    const float m1 = (float)(S1 / (double)B), m2 = (float)(S2 / (double)B);
    const float sc = h.g3[u] / h.sig3[u];
    float* dzu = h.dz + (size_t)u * Bs;
    for (int b0 = bbeg; b0 < bend; b0 += 64) {
        const int b = b0 + lane;
        if (b < bend) {
            const float ov = ou[b], zv = zh[b];
            float dob = 0.f;
#pragma unroll
            for (int t = 0; t < PA_HEAD_MAX_T; ++t)
                if (t < T) dob = fmaf(pa_dl(h, invN, b * T + t), wf[t], dob);
            const float d3 = ov > 0.f ? dob : 0.f;
            dzu[b] = sc * (d3 - m1 - zv * m2);
        }
    }
    if (owner) {
        if (lane == 0) { h.gg3[u] = (float)S2; h.gb3[u] = (float)S1; h.gc2[u] = 0.f; }
#pragma unroll
        for (int t = 0; t < PA_HEAD_MAX_T; ++t) {
            if (t >= T) continue;
            const double tot = wave_sum_d(gw[t]);
            if (lane == 0) h.gWf[(size_t)t * U + u] = (float)tot;
            if (first) {
                const double ct = wave_sum_d(gb[t]);
                if (lane == 0) h.gbf[t] = (float)ct;
            }
        }
        if (first && h.mode == 2) {
            const double tot = wave_sum_d(lacc);
            if (lane == 0) *h.loss_out = (float)(tot / (double)(B * T));
        }
    }
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define PA_LD 72                     // bf16 elements per row of the X image (64 sequences + 8 pad)
__host__ __device__ constexpr int pa_nw16(int NQ) { return (NQ + 1 + 15) / 16; }   // + the dz column
__host__ __device__ constexpr int pa_wgt(int NQ) { return pa_nw16(NQ) <= 2 ? pa_nw16(NQ) : 3; }
__host__ __device__ constexpr int pa_ng(int NQ) { return (pa_nw16(NQ) + pa_wgt(NQ) - 1) / pa_wgt(NQ); }

#define PA_WAVES 2                    // wavefronts per passA workgroup: their tiles are added in LDS before the store
// HEAD: the instantiation that carries the head backward in its prologue (few tasks, batch <= 512).
// The plain one -- what the headline shape launches -- does not hold pa_head_args' 18 pointers in
// SGPRs across the main loop: with them passA<26> spilled 88 SGPRs and 12 VGPRs (20 B of scratch per
// lane, tools/check_resources.py now fails the build on that).
// passA's rows w = ROWS G + i turned into X[b][w] = dz q (w < n), dz (w == n: the Se column), 0 beyond.
// n lies in (NQLO, NQ], the bucket the kernel is instantiated for, and G is a template argument, so
// every row is classed at compile time: always there (w <= NQLO), never (w > NQ), or one of the few
// in between that need a run-time test.  As nested selects on a run-time row index this was two
// scalar branches per row (61 per 64 sequences at n = 26; with several row groups 226 branches and
// their conditions spilled to lanes).
template <int NQ, int G, int ROWS>
__device__ __forceinline__ void pa_transform(float (&rq)[ROWS], float dzv, float a1, float sh1, int n) {
    static_assert(pa_ng(NQ) <= 4, "passA dispatches at most four row groups");
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        constexpr int W0 = G * ROWS;
        const int w = W0 + i;
        if (w <= nq_lower(NQ)) rq[i] = dzv * qval(a1, rq[i], sh1);
        else if (w > NQ) rq[i] = 0.f;
        else rq[i] = w < n ? dzv * qval(a1, rq[i], sh1) : (w == n ? dzv : 0.f);
    }
}

template <int NQ, bool HEAD>
__global__ __launch_bounds__(64 * PA_WAVES, pa_wgt(NQ) <= 2 ? 3 : 2) void passA_kernel(const float* __restrict__ ext,
                                                   const float* __restrict__ alpha,
                                                   const float* __restrict__ shift,
                                                   const float* __restrict__ dz,
                                                   const uint4* __restrict__ bits,
                                                   float* __restrict__ EQp,
                                                   float* __restrict__ Sep, int n, int Bs, int B,
                                                   int ACH, const pa_head_args h, int U) {
    constexpr int NS = ns_stride(NQ), WGT = pa_wgt(NQ), ROWS = 16 * WGT;
    // per wave: the X image [piece][row][sequence] bf16 and the bit words [sequence][4]; the region of
    // wave 1 doubles as the buffer its accumulator tile crosses to wave 0 in at the end
    // (the bit words are double-buffered: the next super-tile's arrive by LDS-DMA during the MFMAs)
    constexpr int XT_HALFS = 3 * ROWS * PA_LD, REGION = XT_HALFS * 2 + 2 * 64 * 4 * 4;
    static_assert(REGION >= FC_MT * WGT * 64 * 16, "accumulator tile must fit the wave's LDS region");
    __shared__ __attribute__((aligned(16))) unsigned char pal[PA_WAVES][REGION];
    // (plain blockIdx mapping: this kernel stages no per-unit tables, and it sits at its register cap)
    const int u = blockIdx.y, ch = blockIdx.x, grp = blockIdx.z, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: LDS bases stay scalar
    uint16_t* xt = reinterpret_cast<uint16_t*>(pal[wave]);
    uint32_t* tw0 = reinterpret_cast<uint32_t*>(pal[wave] + XT_HALFS * 2);      // [2][64 sequences][4 words]
    const int c = lane & 15, g = lane >> 4;
    const int w0 = grp * ROWS;                         // first column of this group
    // a workgroup owns one of the ACH batch chunks; its PA_WAVES waves split it in 64-sequence tiles
    const int per = ((((B + ACH - 1) / ACH) + 64 * PA_WAVES - 1) / (64 * PA_WAVES)) * (64 * PA_WAVES);
    const int bbeg = min(B, ch * per + wave * (per / PA_WAVES)), bend = min(B, bbeg + per / PA_WAVES);
    const float a1 = alpha[u], sh1 = shift[u];
    const float* __restrict__ eu = ext + (size_t)u * n * Bs;
    const float* __restrict__ dzu = dz + (size_t)u * Bs;
    const uint4* __restrict__ bu = bits + (size_t)u * Bs;
    if (HEAD) pa_head_prologue(h, u, ch + wave, grp, lane, bbeg, bend, Bs, B, U);     // (owner: chunk 0, wave 0)
    f32x4 acc[FC_MT][WGT];
#pragma unroll
    for (int t = 0; t < FC_MT; ++t)
#pragma unroll
        for (int j = 0; j < WGT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][j][i] = 0.f;
    float rq[ROWS], rdz = 0.f;
    // fetch = loads only.  Nothing here USES a loaded value: a use (even an empty asm pin) makes the
    // compiler wait for the data on the spot, and the point of issuing the next super-tile's loads
    // before the MFMAs is that they return while the matrix core works.  The values are turned
    // into X at the top of the next iteration, behind one s_waitcnt.
    auto fetch = [&](int b0, int slot) {
        const int b = b0 + lane;
        const int bc = b < bend ? b : bbeg;
        // the 16 bytes of bit words per sequence go global -> LDS directly (global_load_lds_dwordx4:
        // destination = wave-uniform base + 16 * lane, exactly the [sequence][4] image): held in
        // registers across the MFMA phase they were the four registers the allocator spilled
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(bu + bc),
            (__attribute__((address_space(3))) void*)(tw0 + slot * 256), 16, 0, 0);
        // one scalar base + a 32-bit byte offset per row (global_load ... saddr): per-row 64-bit
        // vector addresses cost 2 x ROWS registers
        const char* __restrict__ eb = reinterpret_cast<const char*>(eu);
        // (the row stride sits in a VGPR the compiler cannot see through: with a visible uniform
        // stride it hoists the 32 row offsets out of the loop -- as SGPRs spilled to lanes, or as
        // 64-bit vector addresses spilled to scratch, depending on the rest of the loop)
        const uint32_t boff = (uint32_t)bc * 4u;
        uint32_t rstride = (uint32_t)Bs * 4u;
        asm volatile("" : "+v"(rstride));
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
            rq[i] = *reinterpret_cast<const float*>(eb + (__umul24(rstride, (uint32_t)min(w0 + i, n - 1)) + boff));
        rdz = dzu[bc];
    };
    STAMP(0);
    if (bbeg < bend) fetch(bbeg, 0);
    int slot = 0;
    for (int b0 = bbeg; b0 < bend; b0 += 64, slot ^= 1) {
        const uint32_t* tw = tw0 + slot * 256;
        // everything of fetch(b0) has landed: the q rows are about to be used, and the LDS-DMA of the
        // bit words is ordered before this wave's reads of them by this wait alone (wave-private image)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            // X[b][w] = dz q (w < n), dz (w == n: the Se column), 0 beyond; dead lanes carry zeros
            const float dzv = (b0 + lane < bend) ? rdz : 0.f;
            // one copy of the row loop per row group, chosen by ONE branch: inside a copy the row
            // index is a compile-time constant (see pa_transform)
            if constexpr (pa_ng(NQ) == 1) pa_transform<NQ, 0, ROWS>(rq, dzv, a1, sh1, n);
            else if (grp == 0) pa_transform<NQ, 0, ROWS>(rq, dzv, a1, sh1, n);
            else if (grp == 1) pa_transform<NQ, (pa_ng(NQ) > 1 ? 1 : 0), ROWS>(rq, dzv, a1, sh1, n);
            else if (grp == 2) pa_transform<NQ, (pa_ng(NQ) > 2 ? 2 : 0), ROWS>(rq, dzv, a1, sh1, n);
            else pa_transform<NQ, (pa_ng(NQ) > 3 ? 3 : 0), ROWS>(rq, dzv, a1, sh1, n);
        }
        // three bf16 pieces of every X value into the LDS image (the store takes the upper half of
        // the register)
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const float x = rq[i];
            const uint32_t hb = __float_as_uint(x) & 0xffff0000u;
            const float r1 = x - __uint_as_float(hb);
            const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mb);
            xt[(0 * ROWS + i) * PA_LD + lane] = (uint16_t)(hb >> 16);
            xt[(1 * ROWS + i) * PA_LD + lane] = (uint16_t)(mb >> 16);
            xt[(2 * ROWS + i) * PA_LD + lane] = (uint16_t)(__float_as_uint(r2) >> 16);
        }
        if (b0 == bbeg) STAMP(1);
        if (b0 + 64 < bend) fetch(b0 + 64, slot ^ 1); // in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);            // (issued HERE, not wherever the scheduler likes)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {               // 32 sequences per MFMA
            bf16x8 bx[3][WGT];
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3)
#pragma unroll
                for (int j = 0; j < WGT; ++j)
                    bx[p3][j] = *reinterpret_cast<const bf16x8*>(&xt[(p3 * ROWS + 16 * j + c) * PA_LD + 32 * kk + 8 * g]);
#pragma unroll
            for (int tp = 0; tp < (FC_MT + 1) / 2; ++tp) {
                // channels 32tp .. 32tp+31 live in word tp of every sequence: eight sequences' words
                uint32_t wd[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wd[i] = tw[(32 * kk + 8 * g + i) * 4 + tp];
#pragma unroll
                for (int th = 0; th < 2; ++th) {
                    const int t = 2 * tp + th;
                    if (t >= FC_MT) break;
                    // A[row r = 16t + c][k = sequence 8g + i]: bit 16th + c of word i -> bf16 1.0 / 0.0
                    // (as bit | bit << 16 times 0x3f80 in ONE 24-bit multiply: written as masks and-ed
                    // with the constants, the compiler turned every bit into test + compare + select
                    // behind a branch -- 457 vector instructions, 119 wait states and 61 branches per
                    // 64 sequences for what is 224 instructions)
                    u32x4 av;
                    const uint32_t sh = (uint32_t)(16 * th + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t two = __builtin_amdgcn_ubfe(wd[2 * i], sh, 1u) |
                                             (__builtin_amdgcn_ubfe(wd[2 * i + 1], sh, 1u) << 16);
                        av[i] = __umul24(two, 0x3f80u);
                    }
                    const bf16x8 afr = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                    for (int p3 = 0; p3 < 3; ++p3)
#pragma unroll
                        for (int j = 0; j < WGT; ++j)
                            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bx[p3][j], acc[t][j], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    STAMP(2);
    // the tile of wave 1 is added to wave 0's through LDS (fixed order), halving the partial sums the
    // small-algebra kernel has to fetch: its 300 blocks pull them all in one burst at kernel start
    __syncthreads();
    {
        float4* red = reinterpret_cast<float4*>(pal[1]);
        if (wave == 1) {
#pragma unroll
            for (int t = 0; t < FC_MT; ++t)
#pragma unroll
                for (int j = 0; j < WGT; ++j)
                    red[(t * WGT + j) * 64 + lane] = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int t = 0; t < FC_MT; ++t)
#pragma unroll
            for (int j = 0; j < WGT; ++j) {
                const float4 v = red[(t * WGT + j) * 64 + lane];
                acc[t][j][0] += v.x; acc[t][j][1] += v.y; acc[t][j][2] += v.z; acc[t][j][3] += v.w;
                if ((t * WGT + j) % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // four reads in flight, not 14
            }
    }
    // D[r][w]: lane holds column w = w0 + 16j + c, rows r = 16t + 4g + i (i = 0..3 consecutive);
    // column n carries Se.  The partial sums are stored w-major, EQp[..][w][r], so that a lane's four
    // rows are one 16-byte store (r-major they were 56 four-byte stores per lane, each instruction
    // touching four rows: a quarter of the wave's life.  Staging the tile through LDS instead was
    // tried and was twice as slow).
#pragma unroll
    for (int j = 0; j < WGT; ++j) {
        const int w = w0 + 16 * j + c;
#pragma unroll
        for (int t = 0; t < FC_MT; ++t) {
            const int r = 16 * t + 4 * g;
            if (r < FC_H) {                            // FC_H is a multiple of 4: all four rows or none
                const float4 v = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
                if (w == n) *reinterpret_cast<float4*>(&Sep[((size_t)u * ACH + ch) * FC_H + r]) = v;
                else if (w < NS) *reinterpret_cast<float4*>(&EQp[(((size_t)u * ACH + ch) * NS + w) * FC_H + r]) = v;
            }
        }
    }
    STAMP(3);
}

int launch_passA(explainn_ctx* c, int B, const pa_head_args* head, hipStream_t s) {
    pa_head_args h = {};
    if (head) h = *head;
#define CALL(N)                                                                                  \
    if (h.mode)                                                                                  \
        hipLaunchKernelGGL((passA_kernel<N, true>), dim3(c->ACH, c->U, pa_ng(N)), dim3(64 * PA_WAVES), 0, s, \
                           c->ext, c->alpha, c->shift, c->dz, c->bits, c->EQp, c->Sep, c->n, c->Bs, \
                           B, c->ACH, h, c->U);                                                  \
    else                                                                                         \
        hipLaunchKernelGGL((passA_kernel<N, false>), dim3(c->ACH, c->U, pa_ng(N)), dim3(64 * PA_WAVES), 0, s, \
                           c->ext, c->alpha, c->shift, c->dz, c->bits, c->EQp, c->Sep, c->n, c->Bs, \
                           B, c->ACH, h, c->U)
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

// ---------------------------------------------------------------------------------------------
// passB: workgroup = 4 wavefronts of one (unit, group of w tiles); T and M fragments staged in LDS.
// k order of the M.q product: step (j', i') of lane (g, c) carries v = 16j' + 4g + i' -- exactly the
// rows 16j + 4g + i the lane's accumulators hold, so the epilogue's q values are the operand
// registers themselves.
// ---------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(256, 2) void passB_kernel(
    const float* __restrict__ ext, const float* __restrict__ alpha,
    const float* __restrict__ shift, const float* __restrict__ dz, const uint4* __restrict__ bits,
    const float* __restrict__ Ttf, const float* __restrict__ Mff, const float* __restrict__ k0p,
    const double* __restrict__ mug, const double* __restrict__ sig1, float* __restrict__ dy,
    float* __restrict__ S12p, int n, int Bs, int B, int U) {
    constexpr int NS = ns_stride(NQ), NW16 = fc_nw16(NQ), WGT = fc_wgt(NQ), NG = fc_ng(NQ);
    constexpr int TBW = 3 * 4 * 256, MK = 4 * NW16;    // floats of one w tile's T pieces; k-steps of M.q
    extern __shared__ __attribute__((aligned(16))) float smemB[];
    float* Tf = smemB;                                 // [WGT][3 pieces][4 k-steps][64 lanes][8 bf16]
    float* Mf = Tf + WGT * TBW;                        // [WGT][MK][64]
    float* k0s = Mf + WGT * MK * 64;                   // [WGT*16]
    int u, bx;
    if (!unit_chunk_of_block(U, u, bx)) return;
    const int grp = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int j0 = grp * WGT;                          // first w tile of this group
    const int ntile = min(WGT, NW16 - j0);             // w tiles this group really has
    STAMP(0);
    // A fragments: T as bf16 pieces (Tf) and Mf[(j*MK+kq)*64+l] = M[v=16(kq>>2)+4(l>>4)+(kq&3)][w=16(j0+j)+(l&15)]
    // are laid out by the mid kernels; copy them with float4, all loads before the stores
    {
        constexpr int NT4 = WGT * TBW / 4, NM4 = WGT * MK * 16, N4 = NT4 + NM4;
        const float4* srcT = reinterpret_cast<const float4*>(Ttf + ((size_t)u * NW16 + j0) * TBW);
        const float4* srcM = reinterpret_cast<const float4*>(Mff + ((size_t)u * NW16 + j0) * MK * 64);
        const int lim_t = ntile * TBW / 4, lim_m = ntile * MK * 16;
        float4* dst = reinterpret_cast<float4*>(Tf);  // Mf follows Tf contiguously
        float4 tv[(N4 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i) {
            const int e = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < NT4) { if (e < lim_t) v = srcT[e]; }
            else if (e < N4) { if (e - NT4 < lim_m) v = srcM[e - NT4]; }
            tv[i] = v;
        }
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i)
            if (tid + i * 256 < N4) dst[tid + i * 256] = tv[i];
    }
    for (int i = tid; i < WGT * 16; i += 256) {
        const int w = 16 * j0 + i;
        k0s[i] = (w < NS) ? k0p[(size_t)u * NS + w] : 0.f;
    }
    __syncthreads();
    STAMP(1);
    const float* __restrict__ eu = ext + (size_t)u * n * Bs;
    float* __restrict__ dyu = dy + (size_t)u * n * Bs;
    const float a1 = alpha[u], s1 = shift[u];
    const float mu = (float)mug[u];
    const float isg = (float)(1.0 / sig1[u]);
    for (int it = 0; it < PB_BTW; ++it) {
        const int bt = (bx * 4 + wave) * PB_BTW + it;
        if (bt * 16 >= B) break;                       // wave-uniform
        const int b = bt * 16 + c;
        const bool live = b < B;
        const int bc = min(b, Bs - 1);
        // raw pooled extremes of this lane's sequence, rows v = 16j' + 4g + i' (register kq = 4j'+i')
        float exr[MK];
#pragma unroll
        for (int kq = 0; kq < MK; ++kq) exr[kq] = eu[min(16 * (kq >> 2) + 4 * g + (kq & 3), n - 1) * Bs + bc];
        const uint4 wv = bits[(size_t)u * Bs + bc];
        const float dzb = dz[(size_t)u * Bs + bc];
#pragma unroll
        for (int kq = 0; kq < MK; ++kq) KEEP(exr[kq]);
        float nq[MK];
#pragma unroll
        for (int kq = 0; kq < MK; ++kq)
            nq[kq] = (16 * (kq >> 2) + 4 * g + (kq & 3) < n) ? -qval(a1, exr[kq], s1) : 0.f;
        float sA = 0.f, sB = 0.f;
        if (it == 0) STAMP_AFTER_LOADS(2);
        // P[w][b] = sum_r T[r][w] bit[b][r] on the bf16 matrix core: the bits are exact in bf16 and T
        // is the exact sum of its three bf16 pieces, so every product is exact and the sum is an fp32
        // accumulation -- at 3/16 of the fp32-MFMA cost (25 k-steps of 16x16x4 became 4 x 3 of
        // 16x16x32).  B operand: element i of lane (g, c) = bit of channel 32kk + 8g + i of sequence c.
        f32x4 accP[WGT];
#pragma unroll
        for (int j = 0; j < WGT; ++j) accP[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t wsv[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint32_t wg = wsv[kk] >> (8 * g);
            u32x4 bv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m0 = __builtin_amdgcn_sbfe(wg, 2 * i, 1), m1 = __builtin_amdgcn_sbfe(wg, 2 * i + 1, 1);
                bv[i] = ((uint32_t)m0 & 0x00003f80u) | ((uint32_t)m1 & 0x3f800000u);
            }
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
            for (int j = 0; j < WGT; ++j)
#pragma unroll
                for (int p3 = 0; p3 < 3; ++p3) {
                    const bf16x8 afr = *reinterpret_cast<const bf16x8*>(
                        reinterpret_cast<const uint16_t*>(Tf) + (((j * 3 + p3) * 4 + kk) * 64 + lane) * 8);
                    accP[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, accP[j], 0, 0, 0);
                }
        }
        // - sum_v M[v][w] q[b][v] - k0'[w] stays on the exact-fp32 MFMA (both operands are real-valued)
        f32x4 acc[WGT];
#pragma unroll
        for (int j = 0; j < WGT; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(&k0s[16 * j + 4 * g]);
            acc[j][0] = -v.x; acc[j][1] = -v.y; acc[j][2] = -v.z; acc[j][3] = -v.w;
        }
#pragma unroll
        for (int kq = 0; kq < MK; ++kq)
#pragma unroll
            for (int j = 0; j < WGT; ++j) acc[j] = MFMA16(Mf[(j * MK + kq) * 64 + lane], nq[kq], acc[j]);
#pragma unroll
        for (int j = 0; j < WGT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = fmaf(dzb, accP[j][i], acc[j][i]);      // e = dz * bit
        if (it == 0) STAMP(3);
        // D[w][b]: lane holds its sequence b, rows w = 16(j0+j) + 4g + i = the v of register 4(j0+j)+i
#pragma unroll
        for (int j = 0; j < WGT; ++j) {
            if (j >= ntile) break;                     // block-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // register index 4(j0+j)+i: j0 is a multiple of WGT and the loop is unrolled, but j0
                // is a runtime value -> select among the NG candidates (compile-time indices)
                float nqv = 0.f, ex = 0.f;
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) {
                    const int kq = 4 * (gg * WGT + j) + i;
                    if (kq < MK && gg == grp) { nqv = nq[kq]; ex = exr[kq]; }
                }
                const int w = 16 * (j0 + j) + 4 * g + i;
                const float dyv = (live && w < n) ? -acc[j][i] * nqv : 0.f;      // dq * q
                sA += dyv;
                sB = fmaf(dyv, (ex - mu) * isg, sB);
                if (w < n && b < Bs) dyu[w * Bs + b] = dyv;
            }
        }
        if (it == 0) STAMP(4);
        sA = wave_sum(sA);
        sB = wave_sum(sB);
        if (lane == 0) {
            float* d = S12p + (((size_t)u * NG + grp) * (Bs / 16) + bt) * 2;
            d[0] = sA; d[1] = sB;
        }
    }
    STAMP(5);
}

template <int NQ>
static size_t passB_lds() {
    constexpr int WGT = fc_wgt(NQ), MK = 4 * fc_nw16(NQ);
    return (size_t)(WGT * 3 * 4 * 256 + WGT * MK * 64 + WGT * 16) * sizeof(float);
}

int launch_passB(explainn_ctx* c, int B, hipStream_t s) {
    const int tiles = (B + 15) / 16;
    const int nx = (tiles + 4 * PB_BTW - 1) / (4 * PB_BTW);
#define CALL(N)                                                                                  \
    hipLaunchKernelGGL(passB_kernel<N>, dim3(nx, units_grid(c->U), fc_ng(N)), dim3(256), passB_lds<N>(), s,  \
                       c->ext, c->alpha, c->shift, c->dz, c->bits, c->Ttf, c->Mff, c->k0p, c->mug,  \
                       c->sig1, c->dy, c->S12p, c->n, c->Bs, B, c->U)
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
                                    (int)passB_lds<N>()));                                   \
    { int rc;                                                                                \
      if ((rc = fc_fwd_configure_nm<N, 0>()) || (rc = fc_fwd_configure_nm<N, 1>()) ||        \
          (rc = fc_fwd_configure_nm<N, 2>()) || (rc = fc_fwd_configure_nm<N, 3>())) return rc; }
    NQ_DISPATCH(c->NQ, CALL);
#undef CALL
    return EXPLAINN_OK;
}
