// Tuning aid: does the ORDER of MFMA issues (operand reuse between consecutive instructions) change the
// power-limited throughput of v_mfma_f32_32x32x16_f16 on gfx950?   hipcc --offload-arch=gfx950 -O3 mfma_energy.hip
// Each variant runs ~2.5 s of back-to-back launches on random f16 operands; rocm-smi is sampled while the queue drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// VARIANT 0: a and b both change every instruction      1: b shared by 4 consecutive, a rotates
//         2: a and b shared by 4 consecutive (only the accumulator rotates)    3: one a, one b for everything
//         4: like 1 but operands are (hi, lo, hi) style: every third product uses a small-magnitude operand
template <int VARIANT>
__global__ __launch_bounds__(256) void mfma_loop(const half8* __restrict__ src, float* __restrict__ dst, int iters) {
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
        for (int j = 0; j < 16; ++j) {
            const int ci = j & 3;
            int ai, bi;
            if (VARIANT == 0) { ai = (j + (j >> 2)) & 3; bi = j & 3; }
            else if (VARIANT == 1 || VARIANT == 4) { ai = j & 3; bi = j >> 2; }
            else if (VARIANT == 2) { ai = j >> 2; bi = j >> 2; }
            else { ai = 0; bi = 0; }
            c[ci] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ai], b[bi], c[ci], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[i][r];
    if (s == 12345.678f) dst[tid] = s;
}

static std::string smi() {
    std::string out;
    FILE* f = popen("rocm-smi -P -c 2>/dev/null | grep -E 'sclk|Power \\(W\\)' | tr '\\n' ' '", "r");
    if (!f) return out;
    char buf[512];
    while (fgets(buf, sizeof buf, f)) out += buf;
    pclose(f);
    return out;
}

template <int V>
void run(const half8* src, float* dst, const char* name) {
    const int iters = 4000, blocks = 256 * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mfma_loop<V>, dim3(blocks), dim3(256), 0, 0, src, dst, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<V>, dim3(blocks), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms1 = 0;
    hipEventElapsedTime(&ms1, e0, e1);
    const int n = (int)(2500.f / ms1) + 1;
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(mfma_loop<V>, dim3(blocks), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1, 0);
    std::string s1 = smi(), s2 = smi();
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)n * blocks * 4 * iters * 16 * 32768.0;
    printf("%-34s %8.1f TF issued  (%.3f ms/launch)\n    %s\n    %s\n", name, flop / (ms * 1e-3) / 1e12, ms / n, s1.c_str(), s2.c_str());
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1 && std::string(argv[1]) == "zeros";
    const bool lo3 = argc > 1 && std::string(argv[1]) == "lo3";
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
        float v = zeros ? 0.f : ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
        if (lo3 && (i / 8) % 3 == 1) v *= 1.0f / 2048.f;      // "lo"-magnitude operand values in a third of the fragments
        h[i] = (_Float16)v;
    }
    half8* src;
    float* dst;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&dst, 256 * 2 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0>(src, dst, "a,b change every issue");
    run<1>(src, dst, "b shared x4, a rotates");
    run<2>(src, dst, "a,b shared x4 (acc rotates)");
    run<3>(src, dst, "single a, single b");
    return 0;
}
