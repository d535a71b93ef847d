// Micro-benchmarks behind DESIGN.md's latency model (run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o gpurun_out/microbench && gpurun_out/microbench)
// Each test launches G one-wave workgroups (like the library's lane=sequence kernels) and records,
// per wave, its start time (s_memrealtime, 100 MHz) and its duration in shader cycles (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Rec { unsigned long long t0_real, t1_real, cyc; };

template <int MODE>
__global__ __launch_bounds__(64) void probe(const float* __restrict__ src, float* __restrict__ dst,
                                            Rec* rec, int rows, int stride, int reps, int nmfma) {
    __shared__ float tile[32 * 65];
    const int lane = threadIdx.x, wid = blockIdx.x + blockIdx.y * gridDim.x;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    float acc1 = 0.f;
    f32x16 acc;
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    const float* base = src + (size_t)blockIdx.y * rows * stride + (size_t)blockIdx.x * 64 + lane;
    for (int r = 0; r < reps; ++r) {
        float v[32];
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = (i < rows) ? base[(size_t)i * stride + (size_t)r * 64 * 0] : 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) tile[i * 65 + lane] = v[i] + acc1;
        }
        if (MODE >= 2) {
            for (int s = 0; s < nmfma; ++s) {
                const float a = tile[(lane & 31) * 65 + ((2 * s + (lane >> 5)) & 63)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a - 1.f, acc, 0, 0, 0);
            }
            acc1 = acc[0] * 1e-30f;      // next repetition's loads depend on this one's result
        } else if (MODE == 1) {
            acc1 = tile[lane] * 1e-30f;
        }
        base += (size_t)(acc1 != 12345.f ? 0 : 1);
    }
    float out = acc1;
    for (int g = 0; g < 16; ++g) out += acc[g];
    dst[(size_t)wid * 64 + lane] = out;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { rec[wid].t0_real = r0; rec[wid].t1_real = r1; rec[wid].cyc = c1 - c0; }
}

template <int MODE>
void run(const char* name, int gx, int gy, int rows, int stride, int reps, int nmfma, float* src, float* dst, Rec* rec) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int G = gx * gy;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(gx, gy), dim3(64), 0, 0, src, dst, rec, rows, stride, reps, nmfma);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Rec> h(G); CK(hipMemcpy(h.data(), rec, G * sizeof(Rec), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0; std::vector<double> dur, start;
    for (auto& r : h) { t0 = std::min(t0, r.t0_real); t1 = std::max(t1, r.t1_real); }
    for (auto& r : h) { dur.push_back((double)r.cyc); start.push_back((r.t0_real - t0) * 0.01); }
    std::sort(dur.begin(), dur.end()); std::sort(start.begin(), start.end());
    printf("%-44s grid=%dx%d event=%.1f us  span(realtime)=%.1f us  wave cycles p50=%.0f p90=%.0f max=%.0f  start p50=%.1f p90=%.1f max=%.1f us\n",
           name, gx, gy, ms * 1e3, (t1 - t0) * 0.01, dur[G / 2], dur[G * 9 / 10], dur[G - 1], start[G / 2], start[G * 9 / 10], start[G - 1]);
}

int main() {
    const int U = 300, n = 26, Bs = 1088;
    float *src, *dst; Rec* rec;
    CK(hipMalloc(&src, (size_t)U * 32 * Bs * 4 + 4096)); CK(hipMalloc(&dst, (size_t)16 * 1024 * 1024)); CK(hipMalloc(&rec, 65536 * sizeof(Rec)));
    CK(hipMemset(src, 0, (size_t)U * 32 * Bs * 4));
    run<0>("empty, 2400 one-wave WGs", 8, 300, n, Bs, 1, 0, src, dst, rec);
    run<0>("empty, 4800 one-wave WGs", 16, 300, n, Bs, 1, 0, src, dst, rec);
    run<1>("26 coalesced row loads + LDS, 1 rep", 16, 300, n, Bs, 1, 0, src, dst, rec);
    run<1>("26 row loads, 4 dependent reps", 16, 300, n, Bs, 4, 0, src, dst, rec);
    run<2>("26 row loads + 32 MFMA, 1 rep", 16, 300, n, Bs, 1, 32, src, dst, rec);
    run<2>("26 row loads + 32 MFMA, 2 reps (qmom-like)", 8, 300, n, Bs, 2, 32, src, dst, rec);
    run<2>("26 row loads + 128 MFMA, 2 reps", 8, 300, n, Bs, 2, 128, src, dst, rec);
    run<2>("0 loads + 256 MFMA, 1 rep", 8, 300, 0, Bs, 1, 256, src, dst, rec);
    run<2>("0 loads + 1024 MFMA, 1 rep, 1024 waves", 4, 256, 0, Bs, 1, 1024, src, dst, rec);
    return 0;
}
