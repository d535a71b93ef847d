// Stage 0 of the pipeline: the dense fp32 one-hot the reference hands to forward()
// (train.py:289, sequence/__init__.py:19-26) becomes base codes, and -- for train mode -- the
// unit-independent input moments that give BatchNorm1 its batch statistics in closed form.
//
//   pack_onehot   x (B,4,L) fp32  ->  codesT [L][Bs] u8   (0..3, 4 = N / padding lanes),
//                 pk2 [PW][Bs] (2 bit/base) and nmask [NW][Bs] (1 bit/base) from the same tile
//                 and (train mode) bm [4][tiles][Lp] u64: the batch as bit masks per (base, position)
//   moments       cnt[d][q][a,a'] = #{b : s[b,q]=a and s[b,q+d]=a'} = popc(bm[a][q] & bm[a'][q+d])
//                 (exact, int);  G[(a,j),(a',j')] = (1/N) sum_{q=j}^{j+Lo-1} cnt[j'-j][q][a,a'],  m = diag(G)
//
// This is the only stage that touches the HBM-resident input: 16*L bytes per sequence.
#include "common.h"

// CODES = false: x is the fp32 one-hot (B,4,L).  CODES = true: `codes_in` is a
// (B,L) byte matrix of base codes 0..3 = A,C,G,T, 4 = N -- 16x less input traffic, no fp32 one-hot
// in HBM at all -- optionally reverse-complemented on the fly (code' [p] = 3 - code[L-1-p], N stays
// N: sequence/__init__.py:59-61 flips both axes of the one-hot).
__device__ __forceinline__ int gridDim_x_tiles(int B) { return (B + 63) / 64; }
#define PACK_WAVES 8        // wavefronts per pack block (64 sequences x 64 positions)

template <bool CODES>
__device__ __forceinline__ void pack_tile(const float* __restrict__ x,
                                          const uint8_t* __restrict__ codes_in, int rc,
                                          uint8_t* __restrict__ codesT, uint32_t* __restrict__ pk2,
                                          uint32_t* __restrict__ nmask, int B, int L, int Bs, int PW,
                                          int NW, int* __restrict__ flags, int bx, int by,
                                          unsigned long long* __restrict__ bm, int Lp) {
    __shared__ uint8_t tile[64][68];
    const int b0 = bx * 64, p0 = by * 64;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    int bad = 0;
    const int p = p0 + lane;
    const int pc = min(p, L - 1);
    if (CODES) {
        const int ps = rc ? L - 1 - pc : pc;            // source position of output position p
        for (int i0 = q; i0 < 64; i0 += 4 * PACK_WAVES) {
            int v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = codes_in[(size_t)min(b0 + i0 + PACK_WAVES * r, B - 1) * L + ps];
#pragma unroll
            for (int r = 0; r < 4; ++r) KEEP(v[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + PACK_WAVES * r, b = b0 + i;
                uint8_t code = 0;                       // padding lanes / past the end: see below
                if (b < B && p < L) {
                    if (v[r] < 4) code = rc ? 3 - v[r] : v[r];
                    else { code = 4; if (v[r] != 4) bad = 1; }   // not a base code: N, flagged
                }
                tile[i][lane] = code;
            }
        }
    } else {
    // eight sequences (32 loads) per wave, all issued before any is used (see KEEP in common.h): with
    // PACK_WAVES = 8 waves the tile's 64 sequences are ONE pass -- one memory round trip (four-wave
    // blocks made two dependent ones; the 80 blocks of the headline shape are latency, not bandwidth)
    for (int i0 = q; i0 < 64; i0 += 8 * PACK_WAVES) {
        float v[8][4];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int bc = min(b0 + i0 + PACK_WAVES * r, B - 1);
            const float* xp = x + (size_t)bc * 4 * L + pc;
#pragma unroll
            for (int a = 0; a < 4; ++a) v[r][a] = xp[(size_t)a * L];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a) KEEP(v[r][a]);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = i0 + PACK_WAVES * r, b = b0 + i;
            // padding lanes (b >= B) and positions past the end are 'A': nothing reads their results,
            // and as N they would drag their wavefront through the N corrections of the conv kernels
            uint8_t code = 0;
            if (b < B && p < L) {
                const float v0 = v[r][0], v1 = v[r][1], v2 = v[r][2], v3 = v[r][3];
                const int ones = (v0 == 1.f) + (v1 == 1.f) + (v2 == 1.f) + (v3 == 1.f);
                const int zeros = (v0 == 0.f) + (v1 == 0.f) + (v2 == 0.f) + (v3 == 0.f);
                if (ones == 1 && zeros == 3) code = v0 == 1.f ? 0 : (v1 == 1.f ? 1 : (v2 == 1.f ? 2 : 3));
                else { code = 4; if (zeros != 4) bad = 1; }   // all-zero column = N; anything else: N, flagged
            }
            tile[i][lane] = code;
        }
    }
    }
    __syncthreads();
    for (int pp = q; pp < 64; pp += PACK_WAVES) {
        const int p = p0 + pp;
        if (p < L) codesT[(size_t)p * Bs + b0 + lane] = tile[lane][pp];
    }
    // packed forms of the same tile: wave q < 4 packs positions [16q, 16q+16) into one 2-bit word per
    // sequence; waves 0 and 1 also build the two 32-position N-mask words; waves 4..7 take the bit
    // masks of positions [16(q-4), ...) below
    if (q < 4) {
        uint32_t w2 = 0, nm = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t c = tile[lane][16 * q + i];
            w2 |= (c > 3u ? 1u : c) << (2 * i);       // N is packed as 'C'; nmask marks it (see conv_bwd)
        }
        const int wi = (p0 >> 4) + q;
        if (wi < PW) pk2[(size_t)wi * Bs + b0 + lane] = w2;
        if (q < 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) nm |= (tile[lane][32 * q + i] > 3u ? 1u : 0u) << i;
            const int ni = (p0 >> 5) + q;
            if (ni < NW) nmask[(size_t)ni * Bs + b0 + lane] = nm;
        }
    }
    // train mode: the batch as bit masks per (base, position) -- bit b of bm[a][tile][p] says
    // "sequence 64*tile + b has base a at position p" -- one ballot per (position, base) of the
    // tile; the moment kernel counts base pairs with AND + popcount on them.  N and the padding
    // lanes set no bit.  Wave q takes positions [16q, 16q+16): lane 16a + i keeps ballot (a, i).
    if (bm != nullptr && q >= PACK_WAVES - 4) {
        const int q = (threadIdx.x >> 6) - (PACK_WAVES - 4);     // (the last four waves)
        const bool live = b0 + lane < B;
        unsigned long long mine = 0ull;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t c = tile[lane][16 * q + i];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const unsigned long long bal = __ballot(live && c == (uint32_t)a);
                mine = (lane == 16 * a + i) ? bal : mine;
            }
        }
        const int a = lane >> 4, pq = p0 + 16 * q + (lane & 15);
        if (pq < Lp) bm[((size_t)a * gridDim_x_tiles(B) + bx) * Lp + pq] = mine;
    }
    if (bad) atomicOr(flags, 1);
}

template <bool CODES>
__global__ __launch_bounds__(64 * PACK_WAVES) void pack_onehot_kernel(const float* __restrict__ x,
                                                          const uint8_t* __restrict__ codes_in,
                                                          int rc,
                                                          uint8_t* __restrict__ codesT,
                                                          uint32_t* __restrict__ pk2,
                                                          uint32_t* __restrict__ nmask, int B,
                                                          int L, int Bs, int PW, int NW,
                                                          int* __restrict__ flags,
                                                          unsigned long long* __restrict__ bm, int Lp) {
    pack_tile<CODES>(x, codes_in, rc, codesT, pk2, nmask, B, L, Bs, PW, NW, flags, blockIdx.x, blockIdx.y,
                     bm, Lp);
}

// Train forward: the pack tiles and the per-unit filter tables (which depend only on the weights)
// in ONE launch -- the first gx*gy blocks pack, the following U4 blocks build tables; every launch
// saved is ~4.5 us of this latency-bound pipeline.
__global__ __launch_bounds__(64 * PACK_WAVES) void pack_tables_kernel(const float* __restrict__ x,
                                                          uint8_t* __restrict__ codesT,
                                                          uint32_t* __restrict__ pk2,
                                                          uint32_t* __restrict__ nmask, int B,
                                                          int L, int Bs, int PW, int NW,
                                                          int* __restrict__ flags, int gx, int gy,
                                                          const float* __restrict__ conv_w,
                                                          const float* __restrict__ gamma1,
                                                          float* __restrict__ Wt,
                                                          uint16_t* __restrict__ Wf,
                                                          uint32_t* __restrict__ Wsg, int U, int k,
                                                          unsigned long long* __restrict__ bm, int Lp) {
    __shared__ float wsh[4 * MAX_K];
    const int blk = blockIdx.x;                        // block-uniform role
    if (blk < gx * gy)
        pack_tile<false>(x, nullptr, 0, codesT, pk2, nmask, B, L, Bs, PW, NW, flags, blk % gx, blk / gx,
                         bm, Lp);
    else
        filter_tables_unit(conv_w, gamma1, Wt, Wf, Wsg, U, k, blk - gx * gy, threadIdx.x, 64 * PACK_WAVES, wsh);
}

// Input moments in ONE launch (round 1: 3800 thin ballot waves for the pair counts + a Gram kernel,
// 24 + 9 us on the side stream, stretching the filter bank that runs beside them).
// One block per (gap d, base pair (a, a')): thread <-> position q,
//   cnt[q] = #{b : s[b,q] = a and s[b,q+d] = a'} = sum over 64-sequence tiles of popc(bm[a][q] & bm[a'][q+d])
// (exact integers), then every entry of G on this block's diagonal is a sum of Lo consecutive cnt:
//   G[(a,j),(a',j+d)] = (1/N) sum_{q=j}^{j+Lo-1} cnt[q]  = (T0 - sum_{q<j} cnt[q] + sum_{q=Lo}^{Lo+j-1} cnt[q]) / N,
// T0 = the first window's sum (block reduction), and m = diag(G).
__global__ __launch_bounds__(256) void moments_kernel(const unsigned long long* __restrict__ bm,
                                                      double* __restrict__ G, double* __restrict__ m,
                                                      int B, int L, int k, int Lp) {
    extern __shared__ int cq[];                        // [L] pair counts of this (d, a, a'), then 4 partial sums
    const int d = blockIdx.x >> 4, a = (blockIdx.x >> 2) & 3, a2 = blockIdx.x & 3;
    const int tid = threadIdx.x, NT = (B + 63) / 64, Lo = L - k + 1, K4 = 4 * k;
    const unsigned long long* __restrict__ m0 = bm + (size_t)a * NT * Lp;
    const unsigned long long* __restrict__ m1 = bm + (size_t)a2 * NT * Lp + d;
    long long part = 0;
    for (int q = tid; q < L; q += 256) {
        int c = 0;
        if (q + d < L) {
            for (int t0 = 0; t0 < NT; t0 += 8) {           // sixteen 8-byte loads in flight
                unsigned long long x0[8], x1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const size_t o = (size_t)min(t0 + i, NT - 1) * Lp + q;
                    x0[i] = m0[o]; x1[i] = m1[o];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) c += (t0 + i < NT) ? __popcll(x0[i] & x1[i]) : 0;
            }
        }
        cq[q] = c;
        if (q < Lo) part += c;
    }
    long long* red = reinterpret_cast<long long*>(cq + ((L + 1) & ~1));
    part = (long long)wave_sum_d((double)part);        // exact: the sums stay far below 2^53
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const long long T0 = red[0] + red[1] + red[2] + red[3];
    if (tid < k - d) {
        const int j = tid;
        long long sj = T0;
        for (int q = 0; q < j; ++q) sj += cq[Lo + q] - cq[q];
        const double v = (double)sj / ((double)B * (double)Lo);
        const int row = a * k + j, col = a2 * k + j + d;
        G[(size_t)row * K4 + col] = v;
        G[(size_t)col * K4 + row] = v;
        if (row == col) m[row] = v;
    }
}

// counts: also write the bit masks the moment kernel reads (train mode)
int launch_pack(explainn_ctx* c, const float* x, int B, bool counts, hipStream_t s) {
    const int gb = (B + 63) / 64;
    if (x == nullptr) {
        // x == NULL: run on the base codes explainn_stage_codes() left in the context
        if (c->staged_B != B) {
            explainn_set_error("x is NULL but %s (batch %d)", c->staged_B ? "the staged codes hold "
                               "another batch size" : "no codes are staged", B);
            return EXPLAINN_E_STATE;
        }
    } else {
        c->staged_B = 0;
        // position tiles cover the padded tail too, so the packed words past L are written (as zeros)
        hipLaunchKernelGGL(pack_onehot_kernel<false>, dim3(gb, (c->NW * 32 + 63) / 64), dim3(64 * PACK_WAVES), 0,
                           s, x, (const uint8_t*)nullptr, 0, c->codesT, c->pk2, c->nmask, B, c->L,
                           c->Bs, c->PW, c->NW, c->flags, counts ? c->bm : nullptr, c->Lp);
        LAUNCH_CHECK();
    }
    return EXPLAINN_OK;
}

int launch_pack_tables(explainn_ctx* c, const float* x, const explainn_params* p, int B,
                       hipStream_t s) {
    const int gx = (B + 63) / 64, gy = (c->NW * 32 + 63) / 64;
    c->staged_B = 0;
    hipLaunchKernelGGL(pack_tables_kernel, dim3(gx * gy + c->U4), dim3(64 * PACK_WAVES), 0, s, x, c->codesT,
                       c->pk2, c->nmask, B, c->L, c->Bs, c->PW, c->NW, c->flags, gx, gy, p->conv_w,
                       p->bn1_w, c->Wt, c->Wf, c->Wsg, c->U, c->k, c->bm, c->Lp);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int launch_pack_codes(explainn_ctx* c, const uint8_t* codes, int B, int rc, hipStream_t s) {
    hipLaunchKernelGGL(pack_onehot_kernel<true>, dim3((B + 63) / 64, (c->NW * 32 + 63) / 64),
                       dim3(64 * PACK_WAVES), 0, s, (const float*)nullptr, codes, rc, c->codesT, c->pk2, c->nmask,
                       B, c->L, c->Bs, c->PW, c->NW, c->flags, c->bm, c->Lp);
    LAUNCH_CHECK();
    c->staged_B = B;
    return EXPLAINN_OK;
}

// input moments (train mode): bit masks (written by the pack kernel) -> Gram matrix of the window
// indicator, one launch
int launch_moments(explainn_ctx* c, int B, hipStream_t s) {
    const size_t sm = (size_t)((c->L + 1) & ~1) * sizeof(int) + 4 * sizeof(long long);
    hipLaunchKernelGGL(moments_kernel, dim3(c->k * 16), dim3(256), sm, s, c->bm, c->G, c->m, B, c->L,
                       c->k, c->Lp);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}
