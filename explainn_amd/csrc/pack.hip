// Stage 0 of the pipeline: the dense fp32 one-hot the reference hands to forward()
// (train.py:289, sequence/__init__.py:19-26) becomes base codes, and -- for train mode -- the
// unit-independent input moments that give BatchNorm1 its batch statistics in closed form.
//
//   pack_onehot   x (B,4,L) fp32  ->  codesT [L][Bs] u8   (0..3, 4 = N / padding lanes),
//                 pk2 [PW][Bs] (2 bit/base) and nmask [NW][Bs] (1 bit/base) from the same tile
//   pair_counts   cnt[d][q][a*4+a'] = #{b : s[b,q]=a and s[b,q+d]=a'}           (exact, int)
//   gram          G[(a,j),(a',j')] = (1/N) sum_{q=j}^{j+Lo-1} cnt[j'-j][q][a,a'],  m = diag(G)
//
// This is the only stage that touches the HBM-resident input: 16*L bytes per sequence.
#include "common.h"

// CODES = false: x is the fp32 one-hot (B,4,L).  CODES = true: `codes_in` is a
// (B,L) byte matrix of base codes 0..3 = A,C,G,T, 4 = N -- 16x less input traffic, no fp32 one-hot
// in HBM at all -- optionally reverse-complemented on the fly (code' [p] = 3 - code[L-1-p], N stays
// N: sequence/__init__.py:59-61 flips both axes of the one-hot).
template <bool CODES>
__device__ __forceinline__ void pack_tile(const float* __restrict__ x,
                                          const uint8_t* __restrict__ codes_in, int rc,
                                          uint8_t* __restrict__ codesT, uint32_t* __restrict__ pk2,
                                          uint32_t* __restrict__ nmask, int B, int L, int Bs, int PW,
                                          int NW, int* __restrict__ flags, int bx, int by) {
    __shared__ uint8_t tile[64][68];
    const int b0 = bx * 64, p0 = by * 64;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    int bad = 0;
    const int p = p0 + lane;
    const int pc = min(p, L - 1);
    if (CODES) {
        const int ps = rc ? L - 1 - pc : pc;            // source position of output position p
        for (int i0 = q; i0 < 64; i0 += 16) {
            int v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = codes_in[(size_t)min(b0 + i0 + 4 * r, B - 1) * L + ps];
#pragma unroll
            for (int r = 0; r < 4; ++r) KEEP(v[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 4 * r, b = b0 + i;
                uint8_t code = 0;                       // padding lanes / past the end: see below
                if (b < B && p < L) {
                    if (v[r] < 4) code = rc ? 3 - v[r] : v[r];
                    else { code = 4; if (v[r] != 4) bad = 1; }   // not a base code: N, flagged
                }
                tile[i][lane] = code;
            }
        }
    } else {
    // eight sequences (32 loads) per pass, all issued before any is used (see KEEP in common.h)
    for (int i0 = q; i0 < 64; i0 += 32) {
        float v[8][4];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int bc = min(b0 + i0 + 4 * r, B - 1);
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
            const int i = i0 + 4 * r, b = b0 + i;
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
    for (int pp = q; pp < 64; pp += 4) {
        const int p = p0 + pp;
        if (p < L) codesT[(size_t)p * Bs + b0 + lane] = tile[lane][pp];
    }
    // packed forms of the same tile: wave q packs positions [16q, 16q+16) into one 2-bit word per
    // sequence; waves 0 and 1 also build the two 32-position N-mask words
    {
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
    if (bad) atomicOr(flags, 1);
}

template <bool CODES>
__global__ __launch_bounds__(256) void pack_onehot_kernel(const float* __restrict__ x,
                                                          const uint8_t* __restrict__ codes_in,
                                                          int rc,
                                                          uint8_t* __restrict__ codesT,
                                                          uint32_t* __restrict__ pk2,
                                                          uint32_t* __restrict__ nmask, int B,
                                                          int L, int Bs, int PW, int NW,
                                                          int* __restrict__ flags) {
    pack_tile<CODES>(x, codes_in, rc, codesT, pk2, nmask, B, L, Bs, PW, NW, flags, blockIdx.x, blockIdx.y);
}

// Train forward: the pack tiles and the per-unit filter tables (which depend only on the weights)
// in ONE launch -- the first gx*gy blocks pack, the following U4 blocks build tables; every launch
// saved is ~4.5 us of this latency-bound pipeline.
__global__ __launch_bounds__(256) void pack_tables_kernel(const float* __restrict__ x,
                                                          uint8_t* __restrict__ codesT,
                                                          uint32_t* __restrict__ pk2,
                                                          uint32_t* __restrict__ nmask, int B,
                                                          int L, int Bs, int PW, int NW,
                                                          int* __restrict__ flags, int gx, int gy,
                                                          const float* __restrict__ conv_w,
                                                          float* __restrict__ Wt,
                                                          float* __restrict__ lut, int U, int k) {
    __shared__ float wsh[4 * MAX_K];
    const int blk = blockIdx.x;                        // block-uniform role
    if (blk < gx * gy)
        pack_tile<false>(x, nullptr, 0, codesT, pk2, nmask, B, L, Bs, PW, NW, flags, blk % gx, blk / gx);
    else
        filter_tables_unit(conv_w, Wt, lut, U, k, blk - gx * gy, threadIdx.x, 256, wsh);
}

// one wavefront per (gap d, position q); lanes stride over the batch, the 16 pair bins are
// counted with ballots so the counters live in scalar registers
__global__ __launch_bounds__(256) void pair_counts_kernel(const uint8_t* __restrict__ codesT,
                                                          int* __restrict__ cnt, int B, int L,
                                                          int k, int Bs) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= k * L) return;
    const int d = wid / L, q = wid % L;
    int c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0;
    if (q + d < L) {
        const uint8_t* r0 = codesT + (size_t)q * Bs;
        const uint8_t* r1 = codesT + (size_t)(q + d) * Bs;
        for (int bb = lane; bb < ((B + 63) & ~63); bb += 256) {
            int s0v[4], s1v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // eight byte loads in flight
                const int bc = min(bb + 64 * r, B - 1);
                s0v[r] = r0[bc];
                s1v[r] = r1[bc];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { KEEP(s0v[r]); KEEP(s1v[r]); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int id = -1;
                if (bb + 64 * r < B && s0v[r] < 4 && s1v[r] < 4) id = s0v[r] * 4 + s1v[r];
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] += __popcll(__ballot(id == i));
            }
        }
    }
    int mine = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) mine = (lane == i) ? c[i] : mine;
    if (lane < 16) cnt[(size_t)wid * 16 + lane] = mine;
}

// one wavefront per entry of G (lanes stride over the Lo positions, shuffle reduce);
// row/col index (a,j) -> a*k + j (the flattening of a (4,k) filter)
__global__ __launch_bounds__(256) void gram_kernel(const int* __restrict__ cnt,
                                                   double* __restrict__ G, double* __restrict__ m,
                                                   int B, int L, int k) {
    const int K4 = 4 * k;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= K4 * K4) return;
    const int row = e / K4, col = e % K4;
    int a = row / k, j = row % k, a2 = col / k, j2 = col % k;
    if (j > j2) { int t = a; a = a2; a2 = t; t = j; j = j2; j2 = t; }
    const int d = j2 - j, Lo = L - k + 1;
    const int* src = cnt + ((size_t)d * L + j) * 16 + a * 4 + a2;
    long long s = 0;
    for (int q = lane; q < Lo; q += 64) s += src[(size_t)q * 16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
        const double v = (double)s / ((double)B * (double)Lo);
        G[e] = v;
        if (row == col) m[row] = v;
    }
}

int launch_moments(explainn_ctx* c, int B, hipStream_t s);

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
        hipLaunchKernelGGL(pack_onehot_kernel<false>, dim3(gb, (c->NW * 32 + 63) / 64), dim3(256), 0,
                           s, x, (const uint8_t*)nullptr, 0, c->codesT, c->pk2, c->nmask, B, c->L,
                           c->Bs, c->PW, c->NW, c->flags);
        LAUNCH_CHECK();
    }
    if (counts) return launch_moments(c, B, s);
    return EXPLAINN_OK;
}

int launch_pack_tables(explainn_ctx* c, const float* x, const explainn_params* p, int B,
                       hipStream_t s) {
    const int gx = (B + 63) / 64, gy = (c->NW * 32 + 63) / 64;
    c->staged_B = 0;
    hipLaunchKernelGGL(pack_tables_kernel, dim3(gx * gy + c->U4), dim3(256), 0, s, x, c->codesT,
                       c->pk2, c->nmask, B, c->L, c->Bs, c->PW, c->NW, c->flags, gx, gy, p->conv_w,
                       c->Wt, c->lut, c->U, c->k);
    LAUNCH_CHECK();
    return EXPLAINN_OK;
}

int launch_pack_codes(explainn_ctx* c, const uint8_t* codes, int B, int rc, hipStream_t s) {
    hipLaunchKernelGGL(pack_onehot_kernel<true>, dim3((B + 63) / 64, (c->NW * 32 + 63) / 64),
                       dim3(256), 0, s, (const float*)nullptr, codes, rc, c->codesT, c->pk2, c->nmask,
                       B, c->L, c->Bs, c->PW, c->NW, c->flags);
    LAUNCH_CHECK();
    c->staged_B = B;
    return EXPLAINN_OK;
}

// input moments (train mode): pair counts -> Gram matrix of the window indicator
int launch_moments(explainn_ctx* c, int B, hipStream_t s) {
    {
        const int waves = c->k * c->L;
        hipLaunchKernelGGL(pair_counts_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, c->codesT,
                           c->cnt, B, c->L, c->k, c->Bs);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(gram_kernel, dim3((c->K4 * c->K4 + 3) / 4), dim3(256), 0, s, c->cnt,
                           c->G, c->m, B, c->L, c->k);
        LAUNCH_CHECK();
    }
    return EXPLAINN_OK;
}
