// Stand-alone harness for the split-f16 inference attention kernels (some_amd/csrc/attention_f16x3.hip is #included as is): random
// operands in the kernels' own layouts at the benchmark shape (32 clips x 2584 frames, 2 streams), HIP-event timing of the placed
// kernel and of the round-1..4 kernel (SOME_AMD_ATTN_V1), max |difference| of their outputs, and - when the kernel is built with
// -DSOME_ATTN_DBG - a per-wavefront timeline (s_memtime at every barrier arrival / release) of one workgroup.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Isome_amd/csrc -Iinclude [-DSOME_ATTN_DBG] [-DSOME_ATTN_ABL=..] \
//         -o tools/_bin/attn_probe tools/attn_probe.hip
//   tools/_bin/attn_probe [clips] [frames] [iters]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#ifdef SOME_ATTN_DBG
__device__ unsigned long long* g_attn_dbg;      // [tiles][4 waves][2]: barrier arrival, release (workgroup SOME_ATTN_DBG_WG)
#endif
#include "../some_amd/csrc/attention_f16x3.hip"

#ifdef SOME_ATTN_DBG
constexpr size_t DBG_LDS = 4096;
#else
constexpr size_t DBG_LDS = 0;
#endif
#ifndef PROBE_DMA
#define PROBE_DMA true
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 2584, iters = argc > 3 ? atoi(argv[3]) : 10;
    const int groups = 2;
    std::vector<int32_t> fo(B + 1), pad(B + 1);
    for (int b = 0; b <= B; ++b) { fo[b] = b * T; pad[b] = b * ((T + 15) / 16 * 16); }
    const int M = B * T;
    const int64_t Mc = attn_rows_cover(M, B);
    const int ldv = vt_ld(Mc);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    // Q / K: SPLIT32 rows [Mc, 512]; V^T: f16 planes [1024, ldv].  Values ~N(0, 1) split into hi / lo halves.
    std::vector<_Float16> q((size_t)Mc * 1024), k((size_t)Mc * 1024), vt((size_t)1024 * ldv);
    auto fill_split = [&](std::vector<_Float16>& a) {
        for (size_t r = 0; r < (size_t)Mc; ++r)
            for (int kb = 0; kb < 16; ++kb)
                for (int j = 0; j < 32; ++j) {
                    const float x = nd(rng);
                    const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
                    a[r * 1024 + kb * 64 + j] = hi;
                    a[r * 1024 + kb * 64 + 32 + j] = lo;
                }
    };
    fill_split(q);
    fill_split(k);
    for (size_t r = 0; r < 512; ++r)
        for (int c = 0; c < ldv; ++c) {
            const float x = nd(rng);
            const _Float16 hi = (_Float16)x;
            vt[r * ldv + c] = hi;
            vt[(512 + r) * ldv + c] = (_Float16)(x - (float)hi);
        }
    Attn3Args a{};
    int32_t *fo_d, *pad_d;
    CK(hipMalloc(&fo_d, (B + 1) * 4));
    CK(hipMalloc(&pad_d, (B + 1) * 4));
    CK(hipMemcpy(fo_d, fo.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(pad_d, pad.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    float* out[2][2];
    for (int g = 0; g < groups; ++g) {
        void *qd, *kd, *vd;
        CK(hipMalloc(&qd, q.size() * 2));
        CK(hipMalloc(&kd, k.size() * 2));
        CK(hipMalloc(&vd, vt.size() * 2));
        CK(hipMemcpy(qd, q.data(), q.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(kd, k.data(), k.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(vd, vt.data(), vt.size() * 2, hipMemcpyHostToDevice));
        a.q[g] = (const float*)qd; a.k[g] = (const float*)kd; a.vt[g] = vd;
        for (int v = 0; v < 2; ++v) { CK(hipMalloc(&out[v][g], (size_t)M * 512 * 4)); CK(hipMemset(out[v][g], 0, (size_t)M * 512 * 4)); }
    }
    a.frame_offsets = fo_d; a.pad_offsets = pad_d; a.groups = groups; a.B = B; a.max_frames = T; a.M = (int)Mc; a.ldv = ldv;
#ifdef SOME_ATTN_DBG
    unsigned long long* dbg_d;
    const int n_tiles = (T + 63) / 64;
    CK(hipMalloc(&dbg_d, (size_t)(n_tiles + 2) * 8 * 8));
    CK(hipMemset(dbg_d, 0, (size_t)(n_tiles + 2) * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), &dbg_d, sizeof(dbg_d)));
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flops = 4.0 * 64 * 8 * groups * (double)B * T * T;
    for (int v = 0; v < 2; ++v) {           // v = 0: the placed kernel, 1: the round-1..4 kernel
        for (int g = 0; g < groups; ++g) a.out[g] = out[v][g];
        const unsigned nqb = (T + 127) / 128, units = B * 8 * groups, slots = (units + 7) / 8;
        auto launch = [&]() {
            if (v == 0) hipLaunchKernelGGL(attention3i_kernel<PROBE_DMA>, dim3(slots * nqb * 8), dim3(256), LDS_BYTES + DBG_LDS, 0, a, (int)nqb);
            else hipLaunchKernelGGL(attention3_kernel<false>, dim3(slots * nqb * 8), dim3(256), LDS_BYTES, 0, a, (int)nqb);
        };
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3i_kernel<PROBE_DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_BYTES + DBG_LDS)));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        for (int w = 0; w < 3; ++w) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) launch();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.4f ms per launch, %.1f TFLOP/s algorithmic (%.3f of 2500 dense f16)\n", v == 0 ? "placed" : "round-4", ms / iters,
               flops / (ms / iters) * 1e-9, flops / (ms / iters) * 1e-9 / 2500.0);
    }
    // outputs are SPLIT32: compare hi + lo as floats
    {
        const size_t n = (size_t)M * 512;
        std::vector<_Float16> h0(n * 2), h1(n * 2);
        CK(hipMemcpy(h0.data(), out[0][0], n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), out[1][0], n * 4, hipMemcpyDeviceToHost));
        double worst = 0, ref = 0;
        size_t bad = 0;
        for (size_t r = 0; r < (size_t)M; ++r)
            for (int kb = 0; kb < 16; ++kb)
                for (int j = 0; j < 32; ++j) {
                    const size_t o = r * 1024 + kb * 64 + j;
                    const double x0 = (double)(float)h0[o] + (double)(float)h0[o + 32], x1 = (double)(float)h1[o] + (double)(float)h1[o + 32];
                    if (!(std::isfinite(x0) && std::isfinite(x1))) { ++bad; continue; }
                    worst = std::fmax(worst, std::fabs(x0 - x1));
                    ref = std::fmax(ref, std::fabs(x1));
                }
        printf("stream 0: max |placed - round-4| = %.3e (max |out| %.3e), non-finite %zu\n", worst, ref, bad);
    }
#ifdef SOME_ATTN_DBG
    {
        std::vector<unsigned long long> h((size_t)(n_tiles + 2) * 8);
        CK(hipMemcpy(h.data(), dbg_d, h.size() * 8, hipMemcpyDeviceToHost));
        printf("timeline of one workgroup (cycles since its first record): tile | arrival at the end of the step w0..w3 | wait for the tile DMA (vmcnt 0) w0..w3 | "
               "barrier release w0..w3 | step length (release to release, w0)\n");
        unsigned long long t0 = ~0ull;
        for (size_t i = 0; i < h.size(); i += 2) if (h[i] && h[i] < t0) t0 = h[i];
        const unsigned long long low = 0xFFFFFFFFFFull;
        for (int t = 0; t < n_tiles; ++t) {
            printf("%3d |", t);
            for (int w = 0; w < 4; ++w) printf(" %7llu", h[(size_t)t * 8 + w * 2] ? h[(size_t)t * 8 + w * 2] - t0 : 0ull);
            printf(" |");
            for (int w = 0; w < 4; ++w) printf(" %5llu", h[(size_t)t * 8 + w * 2 + 1] >> 40);
            printf(" |");
            for (int w = 0; w < 4; ++w) printf(" %7llu", (h[(size_t)t * 8 + w * 2 + 1] & low) ? (h[(size_t)t * 8 + w * 2 + 1] & low) - (t0 & low) : 0ull);
            if (t > 0) printf(" | %lld", (long long)((h[(size_t)t * 8 + 1] & low) - (h[(size_t)(t - 1) * 8 + 1] & low)));
            printf("\n");
        }
    }
#endif
    return 0;
}
