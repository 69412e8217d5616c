// Box calibration (measurement only, not on the hot path): what THIS GPU sustains on the two resources the hot path is bound by, so
// that a bench line measured on one box can be compared with a line measured on another (the same binary has measured 81.5 - 85.4 ms
// per step across the boxes of rounds 2 - 5: power-limited clocks differ from package to package).
//   * matrix pipe under its power limit: a pure v_mfma_f32_32x32x16_f16 stream on random operands, the B operand shared by four
//     consecutive issues and A rotating (the reuse pattern of a GEMM k-loop; tools/mfma_energy.hip variant 1, profiles/r01_mfma_power_ceiling.txt)
//   * HBM: a float4 copy of 1 GiB
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/some_amd.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void cal_fill_kernel(_Float16* dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u + 12345u;          // integer hash -> uniform in [-0.05, 0.05): full-entropy mantissas
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    dst[i] = (_Float16)(((float)(x & 0xffffu) * (1.0f / 65536.0f) - 0.5f) * 0.1f);
}

__global__ __launch_bounds__(256) void cal_mfma_kernel(const half8* __restrict__ src, float* __restrict__ dst, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 0xFFFF]; b[i] = src[(tid * 8 + 4 + i) & 0xFFFF]; }
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[j >> 2], c[j & 3], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[i][r];
    if (s == 12345.678f) dst[tid] = s;
}

__global__ __launch_bounds__(256) void cal_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
    // four independent 16-byte loads in flight per thread before the first store (n is a multiple of 4 x the grid's threads)
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
}

}  // namespace

extern "C" int some_box_calibrate(double seconds, double* mfma_tflops, double* mfma_mhz, double* copy_gbs, void* stream_) {
    if (!mfma_tflops || !mfma_mhz || !copy_gbs || !(seconds > 0.0) || seconds > 30.0) return SOME_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    constexpr int kSrc = 65536 * 8;                              // halves
    constexpr size_t kCopy = (size_t)1 << 30;                    // bytes per direction
    constexpr int kBlocks = 512, kIters = 4000;
    _Float16* src = nullptr; float* sink = nullptr; char* buf = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SOME_EHIP;
    float ms = 0.f;
    do {
        if (hipMalloc(&src, kSrc * 2) != hipSuccess || hipMalloc(&sink, (size_t)kBlocks * 256 * 4) != hipSuccess ||
            hipMalloc(&buf, 2 * kCopy) != hipSuccess) { rc = SOME_ENOMEM; break; }
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
        hipLaunchKernelGGL(cal_fill_kernel, dim3(kSrc / 256), dim3(256), 0, s, src, kSrc);
        if (hipMemsetAsync(buf, 0x3c, 2 * kCopy, s) != hipSuccess) break;
        // ---- matrix pipe: one launch to size the run, a quarter of the time as warm-up (the package settles on its power-limited
        // clock within ~50 ms), then the timed launches back to back
        hipLaunchKernelGGL(cal_mfma_kernel, dim3(kBlocks), dim3(256), 0, s, (const half8*)src, sink, kIters);
        if (hipEventRecord(e0, s) != hipSuccess) break;
        hipLaunchKernelGGL(cal_mfma_kernel, dim3(kBlocks), dim3(256), 0, s, (const half8*)src, sink, kIters);
        if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) break;
        const double want_ms = seconds * 0.5e3;                  // half of the budget for the matrix pipe, half for the copy
        int n = (int)(want_ms / (ms > 1e-3f ? ms : 1e-3f)) + 1;
        const int warm = n / 4 + 1;
        for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(cal_mfma_kernel, dim3(kBlocks), dim3(256), 0, s, (const half8*)src, sink, kIters);
        if (hipEventRecord(e0, s) != hipSuccess) break;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(cal_mfma_kernel, dim3(kBlocks), dim3(256), 0, s, (const half8*)src, sink, kIters);
        if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) break;
        const double flop = (double)n * kBlocks * 4 /* waves */ * kIters * 16 * 32768.0;
        *mfma_tflops = flop / (ms * 1e-3) / 1e12;
        *mfma_mhz = *mfma_tflops * 1e12 / (1024.0 /* SIMDs */ * 1024.0 /* FLOP per SIMD-cycle */) / 1e6;
        // ---- HBM copy
        const size_t n4 = kCopy / 16;
        hipLaunchKernelGGL(cal_copy_kernel, dim3(256 * 16), dim3(256), 0, s, (const f32x4*)buf, (f32x4*)(buf + kCopy), n4);
        if (hipEventRecord(e0, s) != hipSuccess) break;
        hipLaunchKernelGGL(cal_copy_kernel, dim3(256 * 16), dim3(256), 0, s, (const f32x4*)buf, (f32x4*)(buf + kCopy), n4);
        if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) break;
        n = (int)(want_ms / (ms > 1e-3f ? ms : 1e-3f)) + 1;
        if (hipEventRecord(e0, s) != hipSuccess) break;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(cal_copy_kernel, dim3(256 * 16), dim3(256), 0, s, (const f32x4*)buf, (f32x4*)(buf + kCopy), n4);
        if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) break;
        *copy_gbs = 2.0 * (double)kCopy * n / (ms * 1e-3) / 1e9;      // bytes read + bytes written
        rc = hipGetLastError() == hipSuccess ? SOME_OK : SOME_EHIP;
    } while (false);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (sink) (void)hipFree(sink);
    if (buf) (void)hipFree(buf);
    return rc;
}
