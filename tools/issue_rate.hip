// Vector / LDS issue-rate probe for gfx950: how many wave-instructions per
// cycle does ONE SIMD issue when 1, 2, 4 or 5 waves are resident on it, on every CU at once?
// DESIGN.md section 5 priced the "issue roof" of the gather kernels at 4 clk per wave-instruction per
// SIMD; MI355X_MICROARCH.md says a wave64 VALU instruction takes 2 cycles on the SIMD-32 and that
// ONE wave alone sustains one per 4.  This settles which holds for the instruction mixes the
// kernels are made of.
//
//   hipcc -O3 --offload-arch=gfx950 tools/issue_rate.hip -o gpurun_out/issue_rate && gpurun_out/issue_rate
//
// Geometry: workgroups of 256 threads (4 waves, one per SIMD: waves of a workgroup are dealt to the
// SIMDs cyclically), W workgroups per CU forced by the LDS each declares (floor(160 KiB / W)), grid
// = 256 CUs x W, so every SIMD of the chip holds exactly W waves for the whole run.  Each wave runs
// REPS x UNROLL instructions of the mix on independent registers between two s_memtime stamps.
// Every wave also records WHERE it ran (HW_ID: XCC, SE, CU, SIMD).  The report groups the waves by
// SIMD and computes, per SIMD, (waves on it) x (instructions per wave) / (last end - first start):
// the issue rate that SIMD actually delivered with the waves it actually held -- placement is the
// dispatcher's, so the table is keyed by the resident-wave count that was OBSERVED (the LDS
// declaration only caps it).  The clock is measured too (s_memtime ticks per 100 MHz s_memrealtime
// tick), so "clk" is shader cycles.  Reported per (mix, waves per SIMD): median over SIMDs of
// wave-instructions per clk, and its inverse -- the number bench.py quotes as `issue_clk_per_inst`.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MIX_FMA, MIX_PK_FMA, MIX_PK_ADD, MIX_AND_OR, MIX_DS128, MIX_CONVPOOL, MIX_CONVBWD, MIX_EXP,
       MIX_XOR, MIX_LSHL, MIX_LSHL_ADD, MIX_CMP_CND, MIX_CND_VCC, MIX_MUL_U24, MIX_PERM, MIX_BFE, MIX_MAX, MIX_ADD3, MIX_COUNT };
static const char* kMixName[MIX_COUNT] = {
    "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_and_or_b32", "ds_read_b128 (16 rows)",
    "conv_pool mix (bfe, and_or, ds128, 2 pk_add)", "conv_bwd mix (lshr, and_or, ds128, 2 pk_fma)",
    "v_exp_f32", "v_xor_b32 (VOP2)", "v_lshlrev_b32 (VOP2)", "v_lshl_add_u32 (VOP3)", "v_cmp_gt_f32 sgpr + v_cndmask_e64",
    "v_cmp_gt_f32 vcc + v_cndmask_e32", "v_mul_u32_u24", "v_perm_b32", "v_bfe_u32", "v_max_f32", "v_add3_u32"};
// wave-instructions per inner-loop body of each mix (what the rate is counted in)
static const int kMixInsts[MIX_COUNT] = {128, 128, 128, 128, 64, 160, 160, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};

struct Rec { unsigned long long t0, t1, r0, r1, hw; };

template <int MIX>
__global__ __launch_bounds__(256) void issue_probe(Rec* rec, float* sink, int reps, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // a 16-row x 16-byte table at LDS offset 0 (conflict-free: 16 distinct rows = 64 distinct banks)
    for (int i = tid; i < 1024; i += 256) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    float a[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = (float)(lane + i); p[i] = f32x2{(float)i, (float)(lane - i)}; }
    float x = 1.0001f, y = 0.9999f;
    f32x2 px = {1.0001f, 0.9999f}, py = {0.5f, 0.25f};
    unsigned bits = (unsigned)lane * 0x9E3779B9u, m = 0xf0u, base = 0u;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = bits + i;
    f32x4 row[4] = {};
    // the table address of lane l: row (hash & 15) -> byte offset 16 * row, all 16 rows in use
    unsigned addr[4];
    for (int i = 0; i < 4; ++i) addr[i] = (((unsigned)lane * 7u + i * 5u) & 15u) * 16u;
    __builtin_amdgcn_s_barrier();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (MIX == MIX_FMA) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if (MIX == MIX_PK_FMA) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(px), "v"(py));
        } else if (MIX == MIX_PK_ADD) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(px));
        } else if (MIX == MIX_AND_OR) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(base));
        } else if (MIX == MIX_EXP) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (MIX == MIX_XOR || MIX == MIX_LSHL || MIX == MIX_LSHL_ADD || MIX == MIX_MUL_U24 || MIX == MIX_PERM ||
                   MIX == MIX_BFE || MIX == MIX_ADD3) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MIX == MIX_XOR) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(m));
                    else if (MIX == MIX_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
                    else if (MIX == MIX_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(m));
                    else if (MIX == MIX_MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(m));
                    else if (MIX == MIX_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(base));
                    else if (MIX == MIX_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[i]));
                    else asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(base));
                }
        } else if (MIX == MIX_MAX) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        } else if (MIX == MIX_CMP_CND) {
            // the ReLU / dropout select of the FC epilogues: a compare into an SGPR pair and a select on it
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned long long sm;
                    asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(sm) : "v"(a[i]), "v"(x));
                    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "s"(sm));
                }
        } else if (MIX == MIX_CND_VCC) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(a[i]), "v"(x), "v"(y) : "vcc");
        } else if (MIX == MIX_DS128) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(row[i]) : "v"(addr[i]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else if (MIX == MIX_CONVPOOL) {
            // conv_pool's inner body per tap pair and unit quad: one field extract of the packed codes,
            // the table address by and-or, one 16-byte row read, two packed adds into the four sums
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                unsigned f, ad;
                asm volatile("v_bfe_u32 %0, %1, %2, 8" : "=v"(f) : "v"(bits), "v"((unsigned)((k & 7) * 3)));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(f), "v"(m), "v"(base));
                asm volatile("ds_read_b128 %0, %1" : "=v"(row[k & 3]) : "v"(ad));
                if ((k & 3) == 3) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(2 * k) & 7]) : "v"(px));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(2 * k + 1) & 7]) : "v"(py));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MIX == MIX_CONVBWD) {
            // conv_bwd's body per (window, tap): a shift of the code word, the row address by and-or, one
            // one-hot row read (16 bytes), two packed FMAs of the pooled gradient into the tap's sums
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                unsigned f, ad;
                asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(f) : "v"((unsigned)(2 * (k & 7))), "v"(bits));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(f), "v"(m), "v"(base));
                asm volatile("ds_read_b128 %0, %1" : "=v"(row[k & 3]) : "v"(ad));
                if ((k & 3) == 3) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[(2 * k) & 7]) : "v"(px), "v"(py));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[(2 * k + 1) & 7]) : "v"(py), "v"(px));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float out = row[0][0] + row[1][1] + row[2][2] + row[3][3];
    for (int i = 0; i < 8; ++i) out += a[i] + p[i][0] + p[i][1] + (float)u[i];
    sink[(size_t)blockIdx.x * 256 + tid] = out;
    if (lane == 0) {
        Rec& r = rec[(size_t)blockIdx.x * 4 + (tid >> 6)];
        r.t0 = t0; r.t1 = t1; r.r0 = r0; r.r1 = r1;
        // HW_REG_HW_ID (4): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; HW_REG_XCC_ID (20)
        r.hw = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
               ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}

template <int MIX>
static void run(int W, Rec* rec, float* sink, int reps) {
    const int cus = 256, grid = cus * W;
    // force W workgroups per CU: each declares floor(160 KiB / W) of LDS (W = 1: 160 KiB itself)
    size_t lds = (size_t)(160 * 1024 / W) & ~(size_t)255;
    if (lds < 4096) lds = 4096;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&issue_probe<MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(issue_probe<MIX>, dim3(grid), dim3(256), lds, 0, rec, sink, reps, (int)(lds / 4));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<Rec> h((size_t)grid * 4);
    CK(hipMemcpy(h.data(), rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
    const double insts = (double)reps * kMixInsts[MIX];
    struct Simd { int n = 0; unsigned long long a = ~0ull, b = 0; };
    std::map<unsigned long long, Simd> simds;
    std::vector<double> clk;
    for (auto& r : h) {
        const unsigned hw = (unsigned)r.hw, xcc = (unsigned)(r.hw >> 32) & 15;
        const unsigned long long key = ((unsigned long long)xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) |
                                       (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3);
        Simd& sd = simds[key];
        sd.n++; sd.a = std::min(sd.a, r.t0); sd.b = std::max(sd.b, r.t1);
        if (r.r1 > r.r0) clk.push_back((double)(r.t1 - r.t0) / (double)(r.r1 - r.r0) * 0.1);   // GHz
    }
    std::sort(clk.begin(), clk.end());
    std::map<int, std::vector<double>> rate;          // waves on the SIMD -> wave-inst per clk
    for (auto& kv : simds) rate[kv.second.n].push_back(kv.second.n * insts / (double)(kv.second.b - kv.second.a));
    printf("%-46s cap W=%d  event %7.1f us  clock %.2f GHz  %zu SIMDs |", kMixName[MIX], W, ms * 1e3,
           clk.empty() ? 0.0 : clk[clk.size() / 2], simds.size());
    for (auto& kv : rate) {
        auto& v = kv.second; std::sort(v.begin(), v.end());
        if (v.size() * 20 < simds.size()) continue;                  // (fewer than 5 % of the SIMDs: noise)
        printf("  %d waves/SIMD (%zu SIMDs): %.3f inst/clk = %.2f clk/inst", kv.first, v.size(), v[v.size() / 2], 1.0 / v[v.size() / 2]);
    }
    printf("\n");
}

int main() {
    Rec* rec; float* sink;
    CK(hipMalloc(&rec, (size_t)256 * 8 * 4 * sizeof(Rec)));
    CK(hipMemset(rec, 0, (size_t)256 * 8 * 4 * sizeof(Rec)));
    CK(hipMalloc(&sink, (size_t)256 * 8 * 256 * sizeof(float)));
    const int reps = 500;
    const int Ws[] = {1, 2, 4, 8};
    for (int W : Ws) {
        run<MIX_FMA>(W, rec, sink, reps);
        run<MIX_PK_FMA>(W, rec, sink, reps);
        run<MIX_PK_ADD>(W, rec, sink, reps);
        run<MIX_AND_OR>(W, rec, sink, reps);
        run<MIX_EXP>(W, rec, sink, reps);
        run<MIX_DS128>(W, rec, sink, reps);
        run<MIX_CONVPOOL>(W, rec, sink, reps);
        run<MIX_CONVBWD>(W, rec, sink, reps);
        run<MIX_XOR>(W, rec, sink, reps);
        run<MIX_LSHL>(W, rec, sink, reps);
        run<MIX_LSHL_ADD>(W, rec, sink, reps);
        run<MIX_BFE>(W, rec, sink, reps);
        run<MIX_ADD3>(W, rec, sink, reps);
        run<MIX_MUL_U24>(W, rec, sink, reps);
        run<MIX_PERM>(W, rec, sink, reps);
        run<MIX_MAX>(W, rec, sink, reps);
        run<MIX_CMP_CND>(W, rec, sink, reps);
        run<MIX_CND_VCC>(W, rec, sink, reps);
        printf("\n");
    }
    return 0;
}
