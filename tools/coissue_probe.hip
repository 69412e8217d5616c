// Probe (gfx950): does VALU work hide in the shadow of MFMAs on the same SIMD?
//  A. one wavefront per SIMD: loop of 4 independent v_mfma_f32_32x32x16_f16, each followed by NV independent v_fma_f32 -> time per
//     iteration vs NV (hidden: flat until 2 NV cycles approach 32; not hidden: +2 NV cycles per MFMA from NV = 1)
//  B. two wavefronts per SIMD (512 threads per workgroup, one workgroup per CU): wavefronts 0-3 pure MFMA, wavefronts 4-7 pure VALU
//     (the same SIMD holds w and w + 4) -> does the pair finish in max(a, b) or a + b?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void k_mix(float* sink, int iters) {
    f16v acc[4] = {};
    h8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f - threadIdx.x * 0.002f); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.01f + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[u], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[(u * NV + j) & 15] = __builtin_fmaf(v[(u * NV + j) & 15], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    for (int j = 0; j < 16; ++j) s += v[j];
    if (s == 12345.678f) *sink = s;
}

// role 0: MFMA only, 1: VALU only, 2: wavefronts 0-3 MFMA and 4-7 VALU
__global__ __launch_bounds__(512) void k_pair(float* sink, int iters, int role) {
    const int wave = threadIdx.x >> 6;
    f16v acc[4] = {};
    h8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f - threadIdx.x * 0.002f); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.01f + j;
    const bool do_m = role == 0 || (role == 2 && wave < 4), do_v = role == 1 || (role == 2 && wave >= 4);
    if ((role == 0 || role == 1) && wave >= 4) return;
    if (do_m) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[u], 0, 0, 0);
    }
    if (do_v) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 64; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], 1.0001f, 0.5f);       // 64 VALU ~ the 128 cycles of 4 MFMAs
    }
    float s = 0;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    for (int j = 0; j < 16; ++j) s += v[j];
    if (s == 12345.678f) *sink = s;
}

template <typename F>
double timed(F launch) {
    launch(100);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    launch(20000);
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
    float* sink;
    (void)hipMalloc(&sink, 4);
    const int blocks = 256;
    printf("A. one wavefront per SIMD, 4 MFMAs per iteration, NV fp32 FMAs behind each (ns per iteration; 4 MFMAs alone = 128 cycles):\n");
#define RUN(NV) { double t = timed([&](int it) { hipLaunchKernelGGL(k_mix<NV>, dim3(blocks), dim3(256), 0, 0, sink, it); }); printf("   NV = %2d: %.1f ns\n", NV, t * 1e9 / 20000); }
    RUN(0) RUN(1) RUN(2) RUN(4) RUN(8) RUN(12) RUN(16) RUN(24)
    printf("B. two wavefronts per SIMD (ns per iteration of 4 MFMAs | 64 FMAs):\n");
    for (int role = 0; role < 3; ++role) {
        double t = timed([&](int it) { hipLaunchKernelGGL(k_pair, dim3(blocks), dim3(512), 0, 0, sink, it, role); });
        printf("   %s: %.1f ns\n", role == 0 ? "MFMA wavefronts only" : role == 1 ? "VALU wavefronts only" : "both, one of each per SIMD", t * 1e9 / 20000);
    }
    return 0;
}
