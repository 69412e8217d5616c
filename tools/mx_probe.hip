// Hardware probe (gfx950): semantics and rates of the MX building blocks an FP6 / FP8 cross-term attention kernel would use -
// there is no ISA manual on this image, so the operand conventions are measured, not assumed.
//   1. v_cvt_scalef32_pk32_fp6_f16 / _f16_fp6: does the f32 scale divide on the way in and multiply on the way out?  rounding?
//   2. v_mfma_scale_f32_32x32x64_f8f6f4 (cbsz = blgp = 2: FP6 e2m3; 0: FP8 e4m3): with lane (i = l & 31, g = l >> 5) holding
//      32 values x[j], is the contraction index k = 32 g + j for BOTH operands (so any consistent (g, j) -> k assignment of A and B
//      gives the right dot product)?  what does the e8m0 scale byte do (127 = 1.0?) and which byte does op_sel 0 read?
//   3. issue rates: f16 32x32x16 vs FP8 / FP6 32x32x64, and the pk32 convert.
//   hipcc --offload-arch=gfx950 -O2 tools/mx_probe.hip -o tools/_bin/mx_probe && tools/_bin/mx_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u6 __attribute__((ext_vector_type(6)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_cvt(const _Float16* in, unsigned* out6, _Float16* back, float s_in, float s_out) {
    h32 v;
    for (int j = 0; j < 32; ++j) v[j] = in[threadIdx.x * 32 + j];
    u6 q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, s_in);
    for (int j = 0; j < 6; ++j) out6[threadIdx.x * 6 + j] = q[j];
    h32 b = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(q, s_out);
    for (int j = 0; j < 32; ++j) back[threadIdx.x * 32 + j] = b[j];
}

// A, B given as f16 [64 lanes][32]; converted in the kernel (scale 1), multiplied with the given scale registers
template <int FMT>
__global__ void k_mfma(const _Float16* a16, const _Float16* b16, float* d, int sa, int sb) {
    h32 va, vb;
    for (int j = 0; j < 32; ++j) { va[j] = a16[threadIdx.x * 32 + j]; vb[j] = b16[threadIdx.x * 32 + j]; }
    i8v a = {}, b = {};
    if constexpr (FMT == 2) {
        u6 qa = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(va, 1.0f), qb = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vb, 1.0f);
        for (int j = 0; j < 6; ++j) { a[j] = (int)qa[j]; b[j] = (int)qb[j]; }
    } else {
        // FP8 e4m3: v_cvt_scalef32_pk_fp8_f16 converts two values into one half of a dword
        for (int j = 0; j < 16; ++j) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            typedef short s2 __attribute__((ext_vector_type(2)));
            const h2 pa = {va[2 * j], va[2 * j + 1]}, pb = {vb[2 * j], vb[2 * j + 1]};
            s2 ra = {0, 0}, rb = {0, 0};
            ra = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(ra, pa, 1.0f, false);
            rb = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(rb, pb, 1.0f, false);
            const unsigned wa = (unsigned short)ra[0], wb = (unsigned short)rb[0];
            a[j / 2] |= (int)(wa << (16 * (j & 1)));
            b[j / 2] |= (int)(wb << (16 * (j & 1)));
        }
    }
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, FMT, FMT, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) d[threadIdx.x * 16 + r] = acc[r];
}

template <int KIND>      // 0: f16 32x32x16, 1: fp8 32x32x64 (scaled form, scale 1), 2: fp6 32x32x64, 3: pk32 f16 -> fp6 convert
__global__ void k_rate(float* sink, int iters, int one) {
    f16v acc[4] = {};
    h8 x = {}, y = {};
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f - threadIdx.x * 0.002f); }
    i8v a = {}, b = {};
    for (int j = 0; j < 8; ++j) { a[j] = 0x3c3c3c3c + (int)threadIdx.x + j; b[j] = 0x38383838 - (int)threadIdx.x * 3 + j; }
    h32 big;
    for (int j = 0; j < 32; ++j) big[j] = (_Float16)(0.01f * (threadIdx.x + j));
    unsigned accq = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[u], 0, 0, 0);
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[u], 0, 0, 0, one, 0, one);
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[u], 2, 2, 0, one, 0, one);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                u6 q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(big, 1.0f + u);
                accq += q[0] ^ q[5];
                big[u] = (_Float16)((float)big[u] + 0.001f);
            }
        }
    }
    float s = (float)accq;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    if (s == 12345.678f) *sink = s;
}

static float fp6_e2m3(float v) {          // nearest representable e2m3 value (ties away; test inputs avoid ties)
    static std::vector<float> t;
    if (t.empty()) {
        for (int m = 0; m < 8; ++m) t.push_back(m * 0.125f);
        for (int e = 1; e <= 3; ++e) for (int m = 0; m < 8; ++m) t.push_back(std::ldexp(1.0f + m / 8.0f, e - 1));
    }
    float a = std::fabs(v), best = 0;
    for (float x : t) if (std::fabs(x - a) < std::fabs(best - a)) best = x;
    if (a > 7.5f) best = 7.5f;
    return v < 0 ? -best : best;
}

int main() {
    _Float16 *d_in, *d_back, *d_a, *d_b;
    unsigned* d_q;
    float *d_d, *d_sink;
    hipMalloc(&d_in, 64 * 32 * 2); hipMalloc(&d_back, 64 * 32 * 2); hipMalloc(&d_q, 64 * 6 * 4);
    hipMalloc(&d_a, 64 * 32 * 2); hipMalloc(&d_b, 64 * 32 * 2); hipMalloc(&d_d, 64 * 16 * 4); hipMalloc(&d_sink, 4);

    // ---- 1. convert semantics ------------------------------------------------------------------------------------------
    const float probe[16] = {0.0f, 0.125f, 0.25f, 0.9f, 1.0f, 1.06f, 1.125f, 2.2f, 3.9f, 7.5f, 8.0f, 100.0f, -0.3f, -5.1f, 0.06f, 0.07f};
    std::vector<_Float16> in(64 * 32), back(64 * 32);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) in[l * 32 + j] = (_Float16)probe[j % 16];
    hipMemcpy(d_in, in.data(), in.size() * 2, hipMemcpyHostToDevice);
    for (auto sc : {std::pair<float, float>{1.f, 1.f}, {2.f, 2.f}, {0.5f, 0.5f}, {2.f, 1.f}, {3.f, 3.f}}) {
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, d_in, d_q, d_back, sc.first, sc.second);
        hipMemcpy(back.data(), d_back, back.size() * 2, hipMemcpyDeviceToHost);
        printf("cvt f16 -> fp6 (scale %.2f) -> f16 (scale %.2f):", sc.first, sc.second);
        for (int j = 0; j < 16; ++j) printf(" %g->%g", probe[j], (float)back[j]);
        printf("\n");
    }
    std::vector<unsigned> q(64 * 6);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, d_in, d_q, d_back, 1.f, 1.f);
    hipMemcpy(q.data(), d_q, q.size() * 4, hipMemcpyDeviceToHost);
    printf("fp6 bit stream of lane 0 (scale 1): %08x %08x %08x %08x %08x %08x\n", q[0], q[1], q[2], q[3], q[4], q[5]);

    // ---- 2. MFMA contraction mapping and scale bytes ----------------------------------------------------------------------
    srand(1);
    std::vector<float> A(32 * 64), B(64 * 32);          // A[i][k], B[k][j] with fp6-exact entries
    const float vals[8] = {0.5f, 1.0f, 1.5f, 2.0f, 3.0f, -1.0f, -0.25f, 0.75f};
    for (auto& x : A) x = vals[rand() % 8];
    for (auto& x : B) x = vals[rand() % 8];
    std::vector<_Float16> a16(64 * 32), b16(64 * 32);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const int k = 32 * (l >> 5) + j;
            a16[l * 32 + j] = (_Float16)A[(l & 31) * 64 + k];
            b16[l * 32 + j] = (_Float16)B[k * 32 + (l & 31)];
        }
    hipMemcpy(d_a, a16.data(), a16.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_b, b16.data(), b16.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> D(64 * 16);
    auto check = [&](const char* what, double expect_scale) {
        hipMemcpy(D.data(), d_d, D.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, ratio = 0;
        int n = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += (double)A[row * 64 + k] * B[k * 32 + col];
                worst = std::fmax(worst, std::fabs(D[l * 16 + r] - ref * expect_scale));
                if (std::fabs(ref) > 1) { ratio += D[l * 16 + r] / ref; ++n; }
            }
        printf("%-58s max |D - %g x reference| = %g   (mean D / reference = %g)\n", what, expect_scale, worst, ratio / (n ? n : 1));
    };
    hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d, 127, 127);
    check("fp6 32x32x64, k = 32 g + j, scales 127 / 127:", 1.0);
    hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d, 128, 127);
    check("fp6, scale_a byte 128:", 2.0);
    hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d, 127, 125);
    check("fp6, scale_b byte 125:", 0.25);
    hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d, 127 | (130 << 8), 127);
    check("fp6, scale_a = 127 in byte 0, 130 in byte 1 (op_sel 0):", 1.0);
    hipLaunchKernelGGL(k_mfma<0>, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d, 127, 127);
    check("fp8 e4m3 32x32x64, scales 127 / 127:", 1.0);
    // per-lane scales: lanes of group g = 1 of A scaled by 2 -> D = sum_{k<32} + 2 sum_{k>=32}
    {
        hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d, 127, 127);
        hipMemcpy(D.data(), d_d, D.size() * 4, hipMemcpyDeviceToHost);
    }

    // ---- 3. rates ------------------------------------------------------------------------------------------------------------
    const char* names[4] = {"v_mfma_f32_32x32x16_f16", "v_mfma_scale 32x32x64 fp8 e4m3", "v_mfma_scale 32x32x64 fp6 e2m3", "v_cvt_scalef32_pk32_fp6_f16"};
    for (int kind = 0; kind < 4; ++kind) {
        const int iters = 20000, blocks = 256 * 2, threads = 256;        // 2 workgroups of 4 wavefronts per CU: 2 wavefronts per SIMD
        auto launch = [&](int it) {
            if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(threads), 0, 0, d_sink, it, 127);
            if (kind == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(threads), 0, 0, d_sink, it, 127);
            if (kind == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(threads), 0, 0, d_sink, it, 127);
            if (kind == 3) hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(threads), 0, 0, d_sink, it, 127);
        };
        launch(100);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        launch(iters);
        hipDeviceSynchronize();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double instr = (double)iters * 4 * blocks * (threads / 64);
        const double flop = kind == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 64;
        printf("%-34s %.3f ms for %.3g wave-instructions: %.2f ns per instruction and SIMD", names[kind], dt * 1e3, instr, dt * 1e9 / (instr / 1024));
        if (kind < 3) printf(", %.0f TFLOP/s", instr * flop / dt / 1e12);
        printf("\n");
    }
    return 0;
}
