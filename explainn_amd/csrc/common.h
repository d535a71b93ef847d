// Internal definitions shared by the gfx950 kernels of libexplainn_hip.so.
// Data layout in HBM: everything per-sequence is stored with the batch
// index fastest ("lane = sequence"), so a 64-wide wavefront reads/writes 256 contiguous bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/explainn_hip.h"

#define FC_H 100          // hidden width of the per-unit FC (architectures/__init__.py:86)
#define HEAD_RB 8            // head kernels keep up to HEAD_RB*256 sequences per unit in registers
#define HEAD_GEMM_MIN_T 8   // more tasks than this: combiner forward/backward as MFMA GEMMs (head.hip)
// d Wf = dl^T o^T has K = batch and only ceil(T/32) x ceil((U+1)/32) output tiles (20 at C3): the batch is
// cut into chunks of >= 128 sequences, one workgroup per (tile, chunk), summed per unit in head_bwd
#define HEAD_GW_MAXCH 32
__host__ inline int head_gw_chunks(int B) { int c = (B + 127) / 128; return c < 1 ? 1 : (c > HEAD_GW_MAXCH ? HEAD_GW_MAXCH : c); }
#define POOLW 7           // MaxPool1d(7,7)        (architectures/__init__.py:81)
#define BN_EPS_D 1e-5     // architectures/__init__.py:79,90,99
#define BN_MOM_D 0.1
#define WAVE 64
#define MAX_K 32          // kernel sizes instantiated for the conv kernels
#define MAX_NQ 160        // largest pooled length with an instantiated FC kernel

struct explainn_ctx {
    int U, k, L, T, maxB, device;
    int Lo, n;            // conv output length, pooled length
    int U4, Uq;           // units rounded up to 4, number of unit quads
    int NQ, NS;           // instantiated pooled-length bucket (>= n), its row stride (mult. of 4)
    int Bs;               // batch stride of every [..][b] array (= maxB rounded up to 64)
    int K4;               // 4*k
    int QCH;              // b-chunks of the q-moment kernel
    int ACH;              // b-chunks of passA
    // per-stage timing (explainn_stage_timing): events bracketing every stage of the training step
    bool timing; unsigned timed;
    hipEvent_t ev0[16], ev1[16];
    // ---- state of the step in flight ----
    int fwd_B;            // batch of the last train forward (0 = none)
    int tail_B;           // batch of a train_step_fc whose train_step_conv is still due (0 = none)
    int fwd_drop;         // dropout was active
    float fwd_scale;      // 1/(1-p)
    // ---- device scratch ----
    char* base; int64_t bytes;
    uint8_t* codesT;      // [L][Bs]          base code per position, 0..3, 4 = N
    uint32_t* pk2;        // [PW][Bs]         2-bit codes, 16 positions per word (N -> 0)
    uint32_t* nmask;      // [NW][Bs]         1 bit per position, set where N
    int PW, NW;
    unsigned long long* bm;  // [4][tiles][Lp] the batch as bit masks per (base, position): bit b of tile t
                          //                = sequence 64t+b has that base there (train mode, pack.hip)
    int Lp;               // positions per bm row (the pack grid's coverage: NW*32 rounded up to 64)
    double* G;            // [4k][4k]         mean window-indicator second moment
    double* m;            // [4k]             mean window indicator
    float* alpha;         // [U4]             BN1 scale   gamma1/sigma1
    float* shift;         // [U4]             BN1 shift   (bias and mean folded in)
    double* mug;          // [U4]             mean of the raw conv sum
    double* sig1;         // [U4]
    double* Gw;           // [U4][4k]
    float* Wt;            // [Uq][k][5][4]    filter taps, unit-quad interleaved, code 4 -> 0
    uint32_t* Wsg;        // [tiles32][2 k-halves][16]: 0x80000000 where the unit pools the minimum, in the
                          // order a lane of the filter-bank GEMM holds its 16 rows
    uint16_t* Wf;         // [tiles32 (padded to whole unit groups)][KS][3 pieces][64 lanes][8] bf16: the filters as
                          // A fragments of v_mfma_f32_32x32x16_bf16 (k = 4 tap + base), sign(gamma1) folded in
    float* ext;           // [U4 (padded to 64-unit groups)][n][Bs]  pooled extreme of the raw conv sum
    uint8_t* idx;         // [U4][n][Bs]      argmax offset 0..6 inside the pooling window
    float* qs0;           // [U][NS]          shift for the q moments (q of sequence 0)
    float* qS1p;          // [U][QCH][NS]
    float* qS2p;          // [U][QCH][NS][NS]
    double* qbar;         // [U][NS]
    float* VC;            // [U][100][NS]     V1 . C  (BN2 variance in prep2, BN2 backward in mid)
    float* A2;            // [U][100][NS]     FC1 weights with BN2 folded in
    float* A2f;           // [U][7][NK4Q][64][4] the same in MFMA 16x16x4 A-fragment order (fc_fwd stages it)
    float* A2h;           // [U][7][KS][3][64][8] bf16: the same as three bf16 pieces in the A-fragment order of
                          // v_mfma_f32_16x16x32_bf16 (fc_fwd for n <= FC_BF_MAXN; KS = 32-wide k-steps)
    float* sh2;           // [U][100]
    float* sig2;          // [U][100]
    float* z;             // [U][Bs]          FC2 output (without its bias)
    float* zhat;          // [U][Bs]
    float* o;             // [U][Bs]          unit outputs
    float* sig3;          // [U]
    double* z12p;         // [U][ZBLK][2] per-workgroup fp64 sums of z and z^2 from fc_fwd (train)
    int ZBLK;             // fc_fwd workgroups per unit at the largest batch
    uint4* bits;          // [U][Bs]          100 bits per (unit, sequence): relu'>0 and kept
    float* dz;            // [U][Bs]
    float* EQp;           // [U][ACH][NS][100]  passA's partial sums, w-major (four rows r = one 16-byte store)
    float* Sep;           // [U][ACH][100]
    float* EQs;           // [U][100][NS]
    float* Ttf;           // [U][NW16][3][4][64][8] bf16: T as three bf16 pieces in the A-fragment order of
                          //                        v_mfma_f32_16x16x32_bf16 (passB copies it to LDS)
    float* Mff;           // [U][NW16][4 NW16][64] M in A-fragment order, k order (j',i') -> v = 16j'+4g+i' 
    float* k0p;           // [U][NS]
    float* dy;            // [U4][n][Bs]
    float* S12p;          // [U][NG][Bs/16][2] per (w-tile group, 16-sequence tile): sum dy, sum dy*chat
    float* Dspp;          // [U][Bs/16][4k]     filter-gradient partials (one per 32-sequence block and window half)
    int dsp_stride, dsp_count;   // partial slots per unit / how many of them the last conv_bwd wrote
    float* dlogits;       // [maxB][T]         (train_step only)
    float* dlT;           // [T][Bs]   d loss / d logits, task-major (head GEMMs, T > HEAD_GEMM_MIN_T)
    int loss_blocks, loss_n; float* loss_out;   // a deferred loss value (launch_loss_deferred -> launch_head_bwd)
    float* gWp;           // [HEAD_GW_CHUNKS][T][U+1]  batch-chunk partials of the combiner-weight gradient
    double* lossp;        // [64]      per-block partial sums of the loss
    int staged_B;         // batch size of the codes explainn_stage_codes() staged, 0 = none
    // eval-mode tables (filter tables, BatchNorm1/2 folds, FC1 fragments) held in the scratch are
    // those of parameter version eval_version; a train-mode forward overwrites them
    bool eval_valid;
    uint64_t eval_version;
    // soft (not one-hot) input: explainn_dense_input switches the stages that touch x to dense.hip
    bool dense;
    const float* dense_x;  // x of the train forward in flight (its backward reads it again)
    int* flags;           // [1]
    int* site_cnt;        // [U4][Bs]  sites per (unit, sequence) of the current batch (filter->PWM export)
    int* site_off;        // [U4][Bs]  their exclusive scan in sequence order, plus the running total
};

enum explainn_stage {
    ST_PACK, ST_MOMENTS, ST_PREP1, ST_CONV_POOL, ST_QMOM, ST_PREP2, ST_FC_FWD, ST_HEAD_FWD, ST_LOSS,
    ST_HEAD_BWD, ST_PASSA, ST_MID, ST_PASSB, ST_CONV_BWD, ST_FIN, ST_COUNT
};

// ---- error plumbing (api.hip) ----
void explainn_set_error(const char* fmt, ...);
#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            explainn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                \
            return EXPLAINN_E_HIP;                                                 \
        }                                                                          \
    } while (0)
#define LAUNCH_CHECK()  HIP_TRY(hipGetLastError())

// ---- launchers, one per pipeline stage (defined next to their kernels) ----
int launch_pack(explainn_ctx* c, const float* x, int B, bool counts, hipStream_t s);
int launch_prep1(explainn_ctx* c, const explainn_params* p, int B, bool train, hipStream_t s);
int launch_moments(explainn_ctx* c, int B, hipStream_t s);
int launch_prep1_tables(explainn_ctx* c, const explainn_params* p, hipStream_t s);
int launch_conv_pool(explainn_ctx* c, const explainn_params* p, int B, bool want_idx, hipStream_t s);
int launch_pack_tables(explainn_ctx* c, const float* x, const explainn_params* p, int B, hipStream_t s);
int launch_pack_codes(explainn_ctx* c, const uint8_t* codes, int B, int rc, hipStream_t s);
int launch_conv_act(explainn_ctx* c, int B, float* acts, hipStream_t s);
int launch_filter_act_max(explainn_ctx* c, int B, const uint8_t* select, float* umax, hipStream_t s);
int launch_filter_sites(explainn_ctx* c, int B, const uint8_t* select, const float* thr, int cap,
                        int* site_total, int* pfm, uint8_t* hit, hipStream_t s);
int launch_qmoments(explainn_ctx* c, int B, hipStream_t s);
int launch_prep2(explainn_ctx* c, const explainn_params* p, int B, bool train, hipStream_t s);
int launch_fc_fwd(explainn_ctx* c, const explainn_params* p, int B, bool train,
                  const uint8_t* keep_mask, float drop_p, uint64_t seed, hipStream_t s);
int launch_head_fwd(explainn_ctx* c, const explainn_params* p, int B, bool train, float* logits,
                    float* outs, hipStream_t s);
int launch_loss_deferred(explainn_ctx* c, int kind, const float* logits, const float* y, int B, float* loss,
                         float* dlogits, hipStream_t s);
int launch_loss(explainn_ctx* c, int kind, const float* logits, const float* y, int B,
                float* loss, float* dlogits, hipStream_t s);
int launch_head_bwd(explainn_ctx* c, const explainn_params* p, const explainn_grads* g,
                    const float* dlogits, int B, hipStream_t s);
int launch_head_bwd_fused_loss(explainn_ctx* c, const explainn_params* p, const explainn_grads* g,
                               int kind, const float* logits, const float* y, float* loss_out, int B,
                               hipStream_t s);
// The head backward of a few-task model (T <= PA_HEAD_MAX_T) rides in passA's prologue (fc.hip):
// mode 1 = d loss / d logits given (dl), mode 2 = recomputed from logits and targets (+ the loss value)
#define PA_HEAD_MAX_T 4
struct pa_head_args {
    int mode, T, kind;
    const float *dl, *logits, *y;
    float* loss_out;
    const float *Wf, *g3, *o, *zhat, *sig3;
    float *dz, *gWf, *gbf, *gg3, *gb3, *gc2;
};
int launch_passA(explainn_ctx* c, int B, const pa_head_args* head, hipStream_t s);
int launch_mid_bwd(explainn_ctx* c, const explainn_params* p, const explainn_grads* g, int B,
                   hipStream_t s);
int launch_passB(explainn_ctx* c, int B, hipStream_t s);
int launch_conv_bwd(explainn_ctx* c, int B, hipStream_t s);
int launch_fin_bwd(explainn_ctx* c, const explainn_params* p, const explainn_grads* g, int B,
                   int freeze_n, hipStream_t s);

int launch_dense_moments(explainn_ctx* c, const float* x, int B, hipStream_t s);
int launch_dense_conv_pool(explainn_ctx* c, const float* x, const explainn_params* p, int B, hipStream_t s);
int launch_dense_conv_bwd(explainn_ctx* c, const float* x, int B, hipStream_t s);
int launch_dense_conv_act(explainn_ctx* c, const float* x, int B, float* acts, hipStream_t s);

int prep_configure(explainn_ctx* c);
int bwd_configure(explainn_ctx* c);
int fc_configure(explainn_ctx* c);

// In-kernel stamps (tools/stampbench.hip defines EXPLAINN_STAMP; the library build compiles them out)
#ifdef EXPLAINN_STAMP
extern __device__ unsigned long long g_stamps[];
#define STAMP(i)                                                                                  \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if ((threadIdx.x & 63) == 0)                                                              \
            g_stamps[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * (blockDim.x / 64) + threadIdx.x / 64) * 8 + (i)] = t_; \
    } while (0)
#define STAMP_AFTER_LOADS(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(i); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP_AFTER_LOADS(i) do { } while (0)
#endif

// KEEP(x): pins a loaded value in a VGPR at this point.  hipcc otherwise sinks a global load into the
// predicated block of its only use (`cond ? f(load) : 0`), which yields one `s_cbranch_execz;
// global_load; s_waitcnt vmcnt(0)` per element -- a fully serialised fetch (22 K cycles for 26 rows).
// Pattern: issue all loads into an array, KEEP() each element in a second loop, then compute.
#define KEEP(x) asm volatile("" : "+v"(x))

// The filter bank as a GEMM (convpool.hip) on v_mfma_f32_32x32x16_bf16: k-steps of 16 (4 taps x 4
// bases), 32-unit tiles, two tiles per wave while their fragments and three operand buffers fit
__host__ __device__ inline int conv_ksteps(int k) { return (k + 3) / 4; }
__host__ __device__ inline int conv_ut(int k) { return conv_ksteps(k) <= 5 ? 2 : 1; }
__host__ __device__ inline int conv_tiles_padded(int U, int k) {
    const int ut = conv_ut(k);
    return (((U + 31) / 32 + ut - 1) / ut) * ut;
}

// The filter bank's tables for unit u (all threads of the block call it): Wt (per-tap table,
// unit-quad interleaved, entry 4 = N = zero: the export and dense kernels), Wf / Wsg (the GEMM's A
// fragments and pooling signs) from the current filters.  wsh: 4*MAX_K floats of LDS.
__device__ __forceinline__ void filter_tables_unit(const float* __restrict__ conv_w,
                                                   const float* __restrict__ gamma1,
                                                   float* __restrict__ Wt,
                                                   uint16_t* __restrict__ Wf, uint32_t* __restrict__ Wsg,
                                                   int U, int k, int u, int tid, int nthreads,
                                                   float* wsh) {
    const int K4 = 4 * k;
    for (int i = tid; i < K4; i += nthreads) {
        const int a = i / k, j = i % k;
        const float wv = (u < U) ? conv_w[(size_t)u * K4 + i] : 0.f;
        Wt[((size_t)(u >> 2) * k + j) * 20 + a * 4 + (u & 3)] = wv;
        wsh[i] = wv;
    }
    for (int j = tid; j < k; j += nthreads) Wt[((size_t)(u >> 2) * k + j) * 20 + 16 + (u & 3)] = 0.f;
    __syncthreads();
    // Wf: the unit's row of the A operand of the filter-bank GEMM (convpool.hip), k = 4 tap + base,
    // as three bf16 pieces whose sum is the fp32 weight exactly.  The pooling direction is folded in:
    // a unit with gamma1 < 0 pools the minimum, so its row is negated (exact) and the kernel takes the
    // maximum for everybody and gives the sign back when it stores.
    const int KS = conv_ksteps(k);
    const float sgn = (u < U && gamma1[u] < 0.f) ? -1.f : 1.f;
    // D row of unit u: 8 (r/4) + 4 kh + r%4 within its tile
    if (tid == 0) Wsg[((u >> 5) * 2 + ((u >> 2) & 1)) * 16 + 4 * ((u & 31) >> 3) + (u & 3)] = sgn < 0.f ? 0x80000000u : 0u;
    for (int kk = tid; kk < 16 * KS; kk += nthreads) {
        const int j = kk >> 2, a = kk & 3;
        const float wv = j < k ? sgn * wsh[a * k + j] : 0.f;
        const uint32_t hb = __float_as_uint(wv) & 0xffff0000u;
        const float r1 = wv - __uint_as_float(hb);
        const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(mb);
        // A fragment of v_mfma_f32_32x32x16_bf16: lane 32 (k/8 mod 2) + row, element k mod 8
        const int ks = kk >> 4, lane = 32 * ((kk >> 3) & 1) + (u & 31), e = kk & 7;
        uint16_t* dst = Wf + ((((size_t)(u >> 5) * KS + ks) * 3) * 64 + lane) * 8 + e;
        dst[0] = (uint16_t)(hb >> 16);
        dst[512] = (uint16_t)(mb >> 16);
        dst[1024] = (uint16_t)(__float_as_uint(r2) >> 16);
    }
}

// (unit, chunk) of a workgroup of a per-unit kernel launched on a grid (chunks, units rounded up to 8
// [, groups]).  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), so
// blockIdx-order (chunk, unit) puts the G chunks of one unit on G different XCDs and every one of
// them pulls the unit's weight fragments through its own L2.  This mapping gives the workgroups
// with equal (linear id mod 8) -- one XCD -- the chunks of the same unit: the fragments cross the
// fabric once and the other chunks hit that L2.  Placement is a speed matter only (the hardware
// promises nothing): results do not depend on it.  Returns false for the padding units.
#ifndef EXPLAINN_XCD_MAP
#define EXPLAINN_XCD_MAP 1
#endif
__device__ __forceinline__ bool unit_chunk_of_block(int U, int& u, int& chunk) {
#if EXPLAINN_XCD_MAP
    const int G = gridDim.x, L = blockIdx.y * G + blockIdx.x;
    const int slot = L >> 3;
    u = (slot / G) * 8 + (L & 7);
    chunk = slot - (slot / G) * G;
#else
    u = blockIdx.y; chunk = blockIdx.x;
#endif
    return u < U;
}
__host__ inline int units_grid(int U) { return (U + 7) & ~7; }

// q = exp(alpha*ext + shift): every consumer must evaluate it identically
__device__ __forceinline__ float qval(float alpha, float ext, float shift) {
    // v_exp_f32 path: ~2 ulp, far inside the 1e-4 parity budget, and 10x fewer instructions
    return __expf(fmaf(alpha, ext, shift));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Row stride NS of every per-unit table with one entry per pooled position (q, T, M, EQ, VC rows):
// NQ rounded up in chunks of at most 36 floats, each a multiple of 4 (float4-aligned rows; the
// chunking dates from the scalar-operand FC of profiles/r01_b and is kept as the layout).
__host__ __device__ constexpr int chunk_count(int len) { return (len + 35) / 36; }
__host__ __device__ constexpr int chunk_len(int len) {
    return (((len + chunk_count(len) - 1) / chunk_count(len)) + 3) & ~3;
}
// q / T / M / EQ rows: NQ payload floats
__host__ __device__ constexpr int ns_stride(int NQ) { return chunk_count(NQ) * chunk_len(NQ); }

// pooled-length buckets with instantiated FC kernels (0 if unsupported)
static inline int nq_bucket(int n) {
    static const int b[] = {4, 8, 12, 16, 20, 24, 26, 28, 32, 40, 48, 56, 64, 72, 84, 96,
                            112, 128, 140, 160};
    for (unsigned i = 0; i < sizeof(b) / sizeof(b[0]); ++i)
        if (b[i] >= n) return b[i];
    return 0;
}
// the bucket below NQ (n > nq_lower(NQ) for every n that dispatches to NQ)
__host__ __device__ constexpr int nq_lower(int NQ) {
    constexpr int b[] = {0, 4, 8, 12, 16, 20, 24, 26, 28, 32, 40, 48, 56, 64, 72, 84, 96, 112, 128, 140, 160};
    int lo = 0;
    for (int i = 1; i < 21; ++i) if (b[i] == NQ) lo = b[i - 1];
    return lo;
}
// ---- tiling of the FC kernels (fc.hip) on v_mfma_f32_16x16x4_f32 ----
#define FC_MT 7                                          // 16-channel tiles covering the 100 hidden channels
__host__ __device__ constexpr int fc_nk4(int NQ) { return (NQ + 3) / 4; }      // k-steps over pooled positions
__host__ __device__ constexpr int fc_nk4q(int NQ) { return (fc_nk4(NQ) + 3) / 4; }  // ... in float4 groups of 4
__host__ __device__ constexpr int fc_nw16(int NQ) { return (NQ + 15) / 16; }   // 16-wide tiles of pooled positions
// workgroups per unit of an fc_fwd launch (fc.hip: fc_fwd_waves x FC_BTW tiles of 16 sequences)
#ifndef FC_BTW
#define FC_BTW 4
#endif
// wavefronts per fc_fwd workgroup.  The weight fragments are staged once per workgroup (21 KB in the
// bf16 form: a quarter of the kernel at C2, tools/stampbench), but 8-wave workgroups need 6 waves
// per SIMD to be resident at once (600 workgroups on 256 CUs) and at 80 registers the train kernels
// spill; fewer, longer waves (8 tiles each) lost more in latency hiding than the staging saved
__host__ __device__ constexpr int fc_fwd_waves(int NQ) { return 4; }
__host__ __device__ constexpr int fc_fwd_blocks(int B, int NQ) {
    return ((B + 15) / 16 + fc_fwd_waves(NQ) * FC_BTW - 1) / (fc_fwd_waves(NQ) * FC_BTW);
}
#define FC_BF_MAXN 96                                     // fc_fwd runs on the bf16 matrix core (exact 3x3 split) up to here
__host__ __device__ constexpr int fc_ks32(int NQ) { return (NQ + 31) / 32; }   // 32-wide k-steps of the bf16 form
// passA / passB split the w tiles into groups of WGT (one wave / workgroup per group)
__host__ __device__ constexpr int fc_wgt(int NQ) { return fc_nw16(NQ) <= 2 ? fc_nw16(NQ) : 3; }
__host__ __device__ constexpr int fc_ng(int NQ) { return (fc_nw16(NQ) + fc_wgt(NQ) - 1) / fc_wgt(NQ); }

#define NQ_DISPATCH(NQv, CALL)                                                        \
    switch (NQv) {                                                                    \
        case 4: { CALL(4); } break;     case 8: { CALL(8); } break;                   \
        case 12: { CALL(12); } break;   case 16: { CALL(16); } break;                 \
        case 20: { CALL(20); } break;   case 24: { CALL(24); } break;                 \
        case 26: { CALL(26); } break;   case 28: { CALL(28); } break;                 \
        case 32: { CALL(32); } break;   case 40: { CALL(40); } break;                 \
        case 48: { CALL(48); } break;   case 56: { CALL(56); } break;                 \
        case 64: { CALL(64); } break;   case 72: { CALL(72); } break;                 \
        case 84: { CALL(84); } break;   case 96: { CALL(96); } break;                 \
        case 112: { CALL(112); } break; case 128: { CALL(128); } break;               \
        case 140: { CALL(140); } break; case 160: { CALL(160); } break;               \
        default: explainn_set_error("pooled length bucket %d not instantiated", NQv); \
                 return EXPLAINN_E_UNSUPPORTED;                                       \
    }
