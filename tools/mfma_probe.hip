// Matrix-pipe probe for gfx950: cycles per v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 on ONE wave per
// SIMD, as a function of (a) where the accumulator lives (arch VGPRs or AccVGPRs), (b) how many
// independent accumulator chains alternate, (c) how many vector instructions are issued behind each
// MFMA.  Written to size the software pipeline of conv_pool_mm (convpool.hip).
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/_bin/mfma_probe && tools/_bin/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 b8;
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
struct Rec { unsigned long long t0, t1, r0, r1; };

#define VALU(n) do { _Pragma("unroll") for (int q_ = 0; q_ < (n); ++q_) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[q_ & 7]) : "v"(one)); } while (0)

template <int V>
__global__ __launch_bounds__(256, 1) void probe(Rec* rec, float* sink, int reps) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    b8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    f16v c0 = {}, c1 = {}, c2 = {};
    f4v d0 = {};
    float f[8], one = 1.f, g = 0.f;
    unsigned laddr = (lane & 15) * 16, goff = (blockIdx.x * 256 + threadIdx.x) * 4 + 1024, u0 = lane, u1 = lane * 3;
    for (int i = 0; i < 8; ++i) f[i] = (float)i;
    constexpr int NV = V == 4 || V == 5 || V == 6 || V == 7 || V == 8 ? 5 : (V == 9 ? 2 : (V == 10 || V == 11 ? 4 : 0));
    __builtin_amdgcn_s_barrier();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (V == 0 || V == 4) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            } else if (V == 1 || V == 5 || V == 10) {
                if (k & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            } else if (V == 2 || V == 6) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
            } else if (V == 3 || V == 7 || V == 8 || V == 11) {
                if (k & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
            } else if (V == 9) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d0) : "v"(a), "v"(b));
            }
            VALU(NV);
            if (V == 8 || V == 11) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(g) : "a"(c2[k & 15]));
            if (V == 10) asm volatile("v_max_f32 %0, %1, %0" : "+v"(g) : "v"(c2[k & 15]));
            if (V == 17) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "a"(a), "a"(b));
            if (V == 18) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "a"(a), "v"(b));
            if (V == 19) { if (k & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "a"(a), "a"(b)); else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "a"(a), "a"(b)); }
            if (V >= 12 && V <= 16) {
                // chain + the update pattern of conv_pool_mm: compares into SGPR pairs, selects from them
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
                if (V == 12 || V == 14 || V == 15 || V == 16) {
                    unsigned long long m0, m1;
                    asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(c2[k & 15]), "v"(f[0]));
                    asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(c2[(k + 1) & 15]), "v"(f[1]));
                    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[0]) : "v"(c2[k & 15]), "s"(m0));
                    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[2]) : "v"(one), "s"(m0));
                    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[1]) : "v"(c2[(k + 1) & 15]), "s"(m1));
                    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[3]) : "v"(one), "s"(m1));
                }
                if (V == 13) {
                    asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %1, %1, %0, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc"
                                 : : "v"(c2[k & 15]), "v"(f[0]), "v"(f[2]), "v"(one) : "vcc");
                    asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %1, %1, %0, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc"
                                 : : "v"(c2[(k + 1) & 15]), "v"(f[1]), "v"(f[3]), "v"(one) : "vcc");
                }
                if (V == 14 && (k % 3) == 0) {
                    f4v t;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(laddr));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    g += t[0];
                }
                if (V == 15 && (k % 3) == 0) {
                    asm volatile("global_store_dword %0, %1, %2" : : "v"(goff), "v"(f[4]), "s"(sink) : "memory");
                }
                if (V == 16) { asm volatile("v_and_or_b32 %0, %0, %1, %1\n v_lshrrev_b32 %0, 3, %0\n v_and_b32 %0, 255, %0\n v_lshlrev_b32 %0, 2, %0" : "+v"(u0) : "v"(u1)); }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = g + d0[0];
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 1234.5f) sink[0] = s + (float)u0;
    if (lane == 0) rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = Rec{t0, t1, r0, r1};
}

template <int V> void run(const char* name, Rec* drec, float* sink) {
    const int reps = 2000, nblk = 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipLaunchKernelGGL(probe<V>, dim3(nblk), dim3(256), 100 * 1024, 0, drec, sink, reps);
    hipLaunchKernelGGL(probe<V>, dim3(nblk), dim3(256), 100 * 1024, 0, drec, sink, reps);
    CK(hipDeviceSynchronize());
    std::vector<Rec> rec(nblk * 4);
    CK(hipMemcpy(rec.data(), drec, rec.size() * sizeof(Rec), hipMemcpyDeviceToHost));
    std::vector<double> v, mhz;
    for (auto& r : rec) {
        mhz.push_back(100.0 * (double)(r.t1 - r.t0) / (double)(r.r1 - r.r0));   // s_memtime ticks per 100 MHz tick
        v.push_back((double)(r.t1 - r.t0) / (reps * 32.0));
    }
    std::sort(v.begin(), v.end()); std::sort(mhz.begin(), mhz.end());
    // s_memtime runs at 100 MHz on this part; convert with the shader clock measured by the caller
    printf("%-58s memtime ticks/MFMA median %.3f  (p10 %.3f p90 %.3f)  clock %.0f MHz\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], mhz[mhz.size() / 2]);
}

int main() {
    Rec* drec; float* sink;
    CK(hipMalloc(&drec, 1024 * sizeof(Rec))); CK(hipMalloc(&sink, 1 << 20));
    run<0>("32x32x16 1 chain, VGPR acc", drec, sink);
    run<1>("32x32x16 2 chains, VGPR acc", drec, sink);
    run<2>("32x32x16 1 chain, AGPR acc", drec, sink);
    run<3>("32x32x16 2 chains, AGPR acc", drec, sink);
    run<4>("32x32x16 1 chain, VGPR acc + 5 v_add", drec, sink);
    run<5>("32x32x16 2 chains, VGPR acc + 5 v_add", drec, sink);
    run<6>("32x32x16 1 chain, AGPR acc + 5 v_add", drec, sink);
    run<7>("32x32x16 2 chains, AGPR acc + 5 v_add", drec, sink);
    run<8>("32x32x16 2 chains, AGPR acc + 5 v_add + accvgpr_read", drec, sink);
    run<9>("16x16x32 1 chain, VGPR acc + 2 v_add", drec, sink);
    run<10>("32x32x16 2 chains, VGPR acc + 4 v_add + v_max(third set)", drec, sink);
    run<11>("32x32x16 2 chains, AGPR acc + 4 v_add + accvgpr_read", drec, sink);
    run<12>("32x32x16 1 chain + 2 x (cmp->sgpr, 2 cndmask_e64)", drec, sink);
    run<13>("32x32x16 1 chain + 2 x (cmp->vcc, 2 cndmask_e32)", drec, sink);
    run<14>("32x32x16 1 chain + 2 x update + ds_read_b128 every 3rd", drec, sink);
    run<15>("32x32x16 1 chain + 2 x update + global_store every 3rd", drec, sink);
    run<16>("32x32x16 1 chain + 2 x update + 4 int ops", drec, sink);
    run<17>("32x32x16 1 chain, VGPR acc, A and B in AccVGPRs", drec, sink);
    run<18>("32x32x16 1 chain, VGPR acc, A in AccVGPRs, B in VGPRs", drec, sink);
    run<19>("32x32x16 2 chains, VGPR acc, A and B in AccVGPRs", drec, sink);
    return 0;
}
