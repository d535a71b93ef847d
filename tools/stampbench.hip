// In-kernel stamp harness: compiles the REAL kernels of explainn_amd/csrc with EXPLAINN_STAMP and
// prints, per kernel, the median shader cycles each wave spends in each phase (C2 shapes).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/stampbench.hip -o gpurun_out/stampbench
#define EXPLAINN_STAMP 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <algorithm>
#include <vector>
__device__ unsigned long long g_stamps[1 << 20];
void explainn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#include "../explainn_amd/csrc/common.h"
#ifdef BALANCE
// -DBALANCE: stamp 0 also records where the wave runs (slot 7: HW_ID | XCC_ID << 32) and the report
// becomes a load-balance table: waves per SIMD, wave time by co-residency, start skew
#undef STAMP
#define STAMP(i)                                                                                  \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if ((threadIdx.x & 63) == 0) {                                                            \
            const size_t s_ = ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * (blockDim.x / 64) + threadIdx.x / 64) * 8; \
            g_stamps[s_ + (i)] = t_;                                                              \
            if ((i) == 0) g_stamps[s_ + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | \
                                             ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
        }                                                                                         \
    } while (0)
#endif
#include "../explainn_amd/csrc/prep.hip"
#include "../explainn_amd/csrc/fc.hip"
#include "../explainn_amd/csrc/bwd.hip"
#include "../explainn_amd/csrc/convpool.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#ifdef BALANCE
#include <map>
static void balance(const char* name, const std::vector<unsigned long long>& h, int waves, int nst) {
    // group waves by (xcc, se, cu, simd); gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    struct W { unsigned long long t0, t1; };
    std::map<unsigned, std::vector<W>> simd;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < waves; ++w) {
        const unsigned long long t0 = h[w * 8], t1 = h[w * 8 + nst - 1], id = h[w * 8 + 7];
        if (!t0 || !t1) continue;
        const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 15;
        const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3);
        simd[key].push_back({t0, t1});
        tmin = std::min(tmin, t0); tmax = std::max(tmax, t1);
    }
    std::map<int, std::vector<double>> by_n;          // waves on the SIMD -> wave durations
    std::map<int, int> nsimd;
    std::vector<double> busy, skew;
    for (auto& kv : simd) {
        const int nw = (int)kv.second.size();
        nsimd[nw]++;
        unsigned long long a = ~0ull, b = 0;
        for (auto& w : kv.second) { by_n[nw].push_back((double)(w.t1 - w.t0)); a = std::min(a, w.t0); b = std::max(b, w.t1); skew.push_back((double)(w.t0 - tmin)); }
        busy.push_back((double)(b - a));
    }
    std::sort(busy.begin(), busy.end()); std::sort(skew.begin(), skew.end());
    printf("%-10s BALANCE: %zu SIMDs used, kernel span %.0f cyc; SIMD busy span p50 %.0f max %.0f; wave start skew p50 %.0f p90 %.0f max %.0f\n",
           name, simd.size(), (double)(tmax - tmin), busy[busy.size() / 2], busy.back(), skew[skew.size() / 2], skew[skew.size() * 9 / 10], skew.back());
    for (auto& kv : by_n) {
        auto& d = kv.second; std::sort(d.begin(), d.end());
        printf("           %d waves/SIMD: %d SIMDs, wave time p50 %.0f p90 %.0f max %.0f\n", kv.first, nsimd[kv.first], d[d.size() / 2], d[d.size() * 9 / 10], d.back());
    }
}
#endif

// every launch starts from zeroed stamps: a wave that leaves before a stamp (passA's second wave
// hands its tile over and returns) must read as "no stamp", not as the previous kernel's value
static void clear_stamps() {
    void* p = nullptr;
    CK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)));
    CK(hipMemset(p, 0, sizeof(unsigned long long) << 20));
}

static void report(const char* name, int waves, int nst, float ms) {
    std::vector<unsigned long long> h((size_t)waves * 8);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8));
#ifdef BALANCE
    balance(name, h, waves, nst);
#endif
    printf("%-10s event %.1f us | median cycles per phase:", name, ms * 1e3);
    for (int p = 1; p < nst; ++p) {
        std::vector<double> d;
        for (int w = 0; w < waves; ++w) if (h[w * 8 + p] && h[w * 8 + p - 1]) d.push_back((double)(h[w * 8 + p] - h[w * 8 + p - 1]));
        std::sort(d.begin(), d.end());
        printf("  [%d->%d] %.0f (p90 %.0f)", p - 1, p, d.empty() ? -1.0 : d[d.size() / 2], d.empty() ? -1.0 : d[d.size() * 9 / 10]);
    }
    std::vector<double> tot;
    for (int w = 0; w < waves; ++w) if (h[w * 8 + nst - 1] && h[w * 8]) tot.push_back((double)(h[w * 8 + nst - 1] - h[w * 8]));   // (waves that exit early leave no last stamp)
    std::sort(tot.begin(), tot.end());
    printf("  | total p50 %.0f max %.0f\n", tot[tot.size() / 2], tot.back());
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);      // a faulting kernel must not take the earlier reports with it
    const int U = 300, n = 26, B = 1024, Bs = 1088, NQ = 26, NS = ns_stride(NQ), NKS = 13, QCH = 8, ACH = 4, NK4Q = fc_nk4q(NQ), NW16 = fc_nw16(NQ);
    auto dalloc = [](size_t bytes) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); return p; };
    float* ext = (float*)dalloc((size_t)U * n * Bs * 4); float* alpha = (float*)dalloc(U * 4); float* shift = (float*)dalloc(U * 4);
    std::vector<float> he((size_t)U * n * Bs); for (auto& v : he) v = (rand() % 2000) * 1e-3f - 1.f;
    CK(hipMemcpy(ext, he.data(), he.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ha(U, 0.7f); CK(hipMemcpy(alpha, ha.data(), U * 4, hipMemcpyHostToDevice));
    float* qs0 = (float*)dalloc(U * NS * 4); float* S1p = (float*)dalloc((size_t)U * QCH * NS * 4); float* S2p = (float*)dalloc((size_t)U * QCH * NS * NS * 4);
    float* A2f = (float*)dalloc((size_t)U * FC_MT * NK4Q * 256 * 4); float* sh2 = (float*)dalloc(U * 100 * 4); float* V2 = (float*)dalloc(U * 100 * 4);
    std::vector<float> hw((size_t)U * FC_MT * NK4Q * 256); for (auto& v : hw) v = (rand() % 2000) * 1e-3f - 1.f;
    CK(hipMemcpy(A2f, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    uint32_t* A2h = (uint32_t*)dalloc((size_t)U * FC_MT * fc_ks32(NQ) * 3 * 256 * 4); double* z12p = (double*)dalloc((size_t)U * 8 * 16);
    uint4* bits = (uint4*)dalloc((size_t)U * Bs * 16 + 64); float* z = (float*)dalloc((size_t)U * Bs * 4); float* o = (float*)dalloc((size_t)U * Bs * 4);
    float* dz = (float*)dalloc((size_t)U * Bs * 4 + 64); float* EQp = (float*)dalloc((size_t)U * ACH * 100 * NS * 4); float* Sep = (float*)dalloc((size_t)U * ACH * 100 * 4);
    float* Tt = (float*)dalloc((size_t)U * NW16 * 3072 * 4 + (size_t)(U * 100 + 2) * NS * 8); float* M = (float*)dalloc((size_t)(U * NS + 2) * NS * 8); float* k0p = (float*)dalloc(U * NS * 4);
    double* mug = (double*)dalloc(U * 8); double* sig1 = (double*)dalloc(U * 8); std::vector<double> one(U, 1.0); CK(hipMemcpy(sig1, one.data(), U * 8, hipMemcpyHostToDevice));
    float* dy = (float*)dalloc((size_t)U * n * Bs * 4); float* S12p = (float*)dalloc((size_t)U * (Bs / 16) * 2 * 4);
    float* fz = (float*)dalloc(U * 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(fz, 0, 4));
        clear_stamps();
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(qmom_kernel<26>, dim3(QCH, U, 1), dim3(64), 0, 0, ext, alpha, shift, qs0, S1p, S2p, n, Bs, B, QCH);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) report("qmom", QCH * U, 4, ms);
        clear_stamps();
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fc_fwd_bf_kernel<26, 2>), dim3(4, units_grid(U)), dim3(256), fc_fwd_bf_lds<26>(), 0, ext, alpha, shift, A2h, sh2, V2, bits, z, (const uint8_t*)nullptr, 19661u, 1.4285715f, 1u, 2u, fz, fz, fz, fz, fz, o, n, Bs, B, U, z12p);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) report("fc_fwd", 4 * U * 4, 5, ms);
        clear_stamps();
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((passA_kernel<26, false>), dim3(ACH, U, 1), dim3(64 * PA_WAVES), 0, 0, ext, alpha, shift, dz, bits, EQp, Sep, n, Bs, B, ACH, pa_head_args{}, U);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) report("passA", ACH * U * PA_WAVES, 4, ms);
        clear_stamps();
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(passB_kernel<26>, dim3(4, units_grid(U), 1), dim3(256), passB_lds<26>(), 0, ext, alpha, shift, dz, bits, Tt, M, k0p, mug, sig1, dy, S12p, n, Bs, B, U); // (tables passed as fragment-ordered stand-ins)
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) report("passB", 4 * U * 4, 6, ms);
    }
    // ---- second group: conv_pool, conv_bwd, prep2, mid_fused ----
    {
        const int k = 19, L = 200, NW = (L + 31) / 32 + 2, PW = 2 * NW, NT = 10, U4 = 300;
        uint32_t* pk2 = (uint32_t*)dalloc((size_t)PW * Bs * 4); uint32_t* nmask = (uint32_t*)dalloc((size_t)NW * Bs * 4);
        std::vector<uint32_t> hp((size_t)PW * Bs); for (auto& v : hp) v = (uint32_t)rand() * 65537u;
        CK(hipMemcpy(pk2, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
        // (the filter bank has its own harness with placement and clock: tools/cpm_stamp.hip; here its
        // five phases only.  ext of this program has U rows: the kernel gets its own padded arrays)
        const int tiles = conv_tiles_padded(U, k), KS = conv_ksteps(k);
        cu32x4* Wf = (cu32x4*)dalloc((size_t)tiles * KS * 3 * 1024); cu32x4* Wsg = (cu32x4*)dalloc((size_t)tiles * 32 * 4);
        float* extp = (float*)dalloc((size_t)32 * tiles * n * Bs * 4); uint8_t* idxp = (uint8_t*)dalloc((size_t)32 * tiles * n * Bs);
        uint8_t* idx = (uint8_t*)dalloc((size_t)U * n * Bs);
        float* Dspp = (float*)dalloc((size_t)U * (Bs / 16) * 76 * 4);   // (partial stride Bs/16: common.h)
        float* fc1_w = (float*)dalloc((size_t)U * 100 * n * 4); float* VC = (float*)dalloc((size_t)U * 100 * NS * 4);
        double* qbar = (double*)dalloc((size_t)U * NS * 8); float* A2 = (float*)dalloc((size_t)U * 100 * NS * 4);
        float* sig2 = (float*)dalloc(U * 100 * 4); std::vector<float> ones(U * 100, 1.f); CK(hipMemcpy(sig2, ones.data(), U * 100 * 4, hipMemcpyHostToDevice));
        float* md = (float*)dalloc((size_t)U * 100 * n * 4 + 4096);
        for (int rep = 0; rep < 2; ++rep) {
            clear_stamps();
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((conv_pool_mm_kernel<5, 2, true>), dim3(B / 32, tiles / 2, 6), dim3(64), 0, 0, pk2, nmask, Wf, Wsg, extp, idxp, n, Bs, PW, NW, 5);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("conv_pool", (B / 32) * (tiles / 2) * 6, 6, ms);
            {   // the matrix-core filter gradient: (32-sequence block, 16-unit tile, window half), two waves
                static unsigned long long* bmask = nullptr; static float* Dmm = nullptr;
                const int Lp = ((NW * 32 + 63) / 64) * 64, NT64 = (B + 63) / 64;
                if (!bmask) {
                    std::vector<unsigned long long> hb((size_t)4 * NT64 * Lp);
                    for (auto& v : hb) v = ((unsigned long long)rand() << 33) ^ ((unsigned long long)rand() << 11) ^ rand();
                    bmask = (unsigned long long*)dalloc(hb.size() * 8);
                    CK(hipMemcpy(bmask, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
                    Dmm = (float*)dalloc((size_t)U * (Bs / 16) * 76 * 4);
                    std::vector<float> hd((size_t)U * n * Bs); for (auto& v : hd) v = (rand() % 2000) * 1e-3f - 1.f;
                    CK(hipMemcpy(dy, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
                    std::vector<uint8_t> hi((size_t)U * n * Bs); for (auto& v : hi) v = rand() % 7;
                    CK(hipMemcpy(idx, hi.data(), hi.size(), hipMemcpyHostToDevice));
                }
                const int wper = (n + 1) / 2;
                const size_t smm = (size_t)(7 * wper + 19 - 1) * 256;
                clear_stamps();
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(conv_bwd_mm_kernel<19>, dim3((B + 31) / 32, (U + 15) / 16, 2), dim3(64 * CBM_WAVES), smm, 0,
                                   dy, idx, bmask, Dmm, U, n, Bs, B, NT64, Lp, Bs / 16, wper);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipGetLastError());
                if (rep) report("conv_bwd_mm", ((B + 31) / 32) * ((U + 15) / 16) * 2 * CBM_WAVES, 4, ms);
            }
            clear_stamps();
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(prep2_kernel<true>, dim3(U), dim3(1024), prep2_lds(n, NS), 0, fc1_w, sh2, sh2, sh2, md, md + U * 100, (int64_t*)nullptr, qs0, S1p, S2p, qbar, VC, A2, (float*)nullptr, sh2, sig2, n, NS, NK4Q, B, QCH, A2h, fc_ks32(NQ));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("prep2", U * 16, 4, ms);
            clear_stamps();
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mid_fused_kernel, dim3(U), dim3(1024), mid_fused_lds(n), 0, EQp, Sep, A2, sh2, sig2, fc1_w, V2, sh2, qbar, VC, Tt, M, k0p, md, md, md, md, md, n, NS, NW16, B, ACH, 1.4285715f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("mid_fused", U * 16, 6, ms);
        }
    }
    return 0;
}
