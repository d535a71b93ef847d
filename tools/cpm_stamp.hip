// Stamp harness for conv_pool_mm alone (C2 shape): per-wave phase cycles and SIMD placement.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/cpm_stamp.hip -o tools/_bin/cpm_stamp
#define EXPLAINN_STAMP 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <algorithm>
#include <map>
#include <vector>
__device__ unsigned long long g_stamps[1 << 20];
void explainn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#include "../explainn_amd/csrc/common.h"
#undef STAMP
#define STAMP(i)                                                                                  \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if ((threadIdx.x & 63) == 0) {                                                            \
            const size_t s_ = ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * (blockDim.x / 64) + threadIdx.x / 64) * 8; \
            g_stamps[s_ + (i)] = t_;                                                              \
            if ((i) == 0) g_stamps[s_ + 6] = __builtin_amdgcn_s_memrealtime();                       \
            if ((i) == 5) g_stamps[s_ + 6] = __builtin_amdgcn_s_memrealtime() - g_stamps[s_ + 6];     \
            if ((i) == 0) g_stamps[s_ + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | \
                                             ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
        }                                                                                         \
    } while (0)
#include "../explainn_amd/csrc/convpool.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int UT, bool IDX> static void run(int parts) {
    const int U = 300, k = 19, L = 200, n = 26, B = 1024, Bs = 1088, NW = (L + 31) / 32 + 2, PW = 2 * NW, KS = 5;
    auto dalloc = [](size_t bytes) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); return p; };
    const int tiles = conv_tiles_padded(U, k);
    uint32_t* pk2 = (uint32_t*)dalloc((size_t)PW * Bs * 4); uint32_t* nm = (uint32_t*)dalloc((size_t)NW * Bs * 4);
    std::vector<uint32_t> hp((size_t)PW * Bs); for (auto& v : hp) v = (uint32_t)rand() * 2654435761u;
    CK(hipMemcpy(pk2, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    cu32x4* Wf = (cu32x4*)dalloc((size_t)tiles * KS * 3 * 1024);
    std::vector<uint16_t> hw((size_t)tiles * KS * 3 * 512); for (auto& v : hw) v = (uint16_t)(0x3c00 + (rand() & 0xff));
    CK(hipMemcpy(Wf, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    cu32x4* g1 = (cu32x4*)dalloc((size_t)tiles * 32 * 4); float* ext = (float*)dalloc(((size_t)32 * tiles * n * Bs + 64 + 64 * 8192) * 4); uint8_t* idx = (uint8_t*)dalloc((size_t)32 * tiles * n * Bs + 64 + 64 * 8192);
    const int wper = (n + parts - 1) / parts;
    const dim3 grid((B + 31) / 32, UT == 1 ? (U + 31) / 32 : tiles / UT, (n + wper - 1) / wper);
    const int waves = grid.x * grid.y * grid.z;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms = 0;
    void* sp = nullptr; CK(hipGetSymbolAddress(&sp, HIP_SYMBOL(g_stamps)));
    for (int rep = 0; rep < 30; ++rep) {
        CK(hipMemset(sp, 0, sizeof(unsigned long long) << 20));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((conv_pool_mm_kernel<5, UT, IDX>), grid, dim3(64), 0, 0, pk2, nm, Wf, g1, ext, idx, n, Bs, PW, NW, wper);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h((size_t)waves * 8);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8));
    printf("UT=%d IDX=%d parts=%d (wper %d): %d waves, event %.1f us\n", UT, (int)IDX, parts, wper, waves, ms * 1e3);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < waves; ++w) { if (h[w * 8]) tmin = std::min(tmin, h[w * 8]); tmax = std::max(tmax, h[w * 8 + 5]); }
    printf("  kernel span (first stamp 0 -> last stamp 5): %.0f cyc\n", (double)(tmax - tmin));
    const char* ph[5] = {"table", "prologue", "first window", "other windows", "drain"};
    for (int p = 1; p <= 5; ++p) {
        std::vector<double> d;
        for (int w = 0; w < waves; ++w) if (h[w * 8 + p] && h[w * 8 + p - 1]) d.push_back((double)(h[w * 8 + p] - h[w * 8 + p - 1]));
        std::sort(d.begin(), d.end());
        if (!d.empty()) printf("  %-14s p10 %.0f p50 %.0f p90 %.0f max %.0f\n", ph[p - 1], d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10], d.back());
    }
    {
        std::vector<double> mhz;
        for (int w = 0; w < waves; ++w) if (h[w * 8 + 6] && h[w * 8 + 5] > h[w * 8]) mhz.push_back(100.0 * (double)(h[w * 8 + 5] - h[w * 8]) / (double)h[w * 8 + 6]);
        std::sort(mhz.begin(), mhz.end());
        if (!mhz.empty()) printf("  shader clock over the wave's life: p10 %.0f p50 %.0f p90 %.0f MHz\n", mhz[mhz.size() / 10], mhz[mhz.size() / 2], mhz[mhz.size() * 9 / 10]);
    }
    std::map<unsigned, int> simd; std::vector<double> skew, tot;
    for (int w = 0; w < waves; ++w) {
        const unsigned long long id = h[w * 8 + 7]; const unsigned hw_ = (unsigned)id, xcc = (unsigned)(id >> 32) & 15;
        simd[(xcc << 16) | (((hw_ >> 13) & 7) << 12) | (((hw_ >> 8) & 15) << 4) | ((hw_ >> 4) & 3)]++;
        skew.push_back((double)(h[w * 8] - tmin)); tot.push_back((double)(h[w * 8 + 5] - h[w * 8]));
    }
    std::sort(skew.begin(), skew.end()); std::sort(tot.begin(), tot.end());
    std::map<int, int> hist; for (auto& kv : simd) hist[kv.second]++;
    printf("  SIMDs used %zu; waves per SIMD histogram:", simd.size()); for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
    printf("\n  start skew p50 %.0f p90 %.0f max %.0f; wave total p50 %.0f max %.0f\n", skew[skew.size() / 2], skew[skew.size() * 9 / 10], skew.back(), tot[tot.size() / 2], tot.back());
}
int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    run<2, true>(6); run<2, false>(6); run<2, true>(3); run<2, false>(3);
    return 0;
}
