// Probe 2 (gfx950): which LANE's scale operand applies to which (row, K block) of v_mfma_scale_f32_32x32x64_f8f6f4, and how
// v_cvt_scalef32_2xpk16_fp6_f32 orders its 32 results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));

__global__ void k_scale(float* d, int lane_a, int lane_b, int opsel_case) {
    f16v ones;
    for (int j = 0; j < 16; ++j) ones[j] = 1.0f;
    const u6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ones, ones, 1.0f);
    i8v a = {(int)q[0], (int)q[1], (int)q[2], (int)q[3], (int)q[4], (int)q[5], 0, 0};
    const int sa = 127 + ((int)threadIdx.x == lane_a ? 1 : 0), sb = 127 + ((int)threadIdx.x == lane_b ? 1 : 0);
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, a, acc, 2, 2, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) d[threadIdx.x * 16 + r] = acc[r];
}

__global__ void k_order(unsigned* out) {
    f16v a, b;
    // distinct fp6-exact magnitudes: a[j] = table value j + 1, b[j] = table value j + 17 (codes 1..16 and 17..32 -> use 1..15 / 16..31)
    const float t[32] = {0, .125f, .25f, .375f, .5f, .625f, .75f, .875f, 1, 1.125f, 1.25f, 1.375f, 1.5f, 1.625f, 1.75f, 1.875f,
                         2, 2.25f, 2.5f, 2.75f, 3, 3.25f, 3.5f, 3.75f, 4, 4.5f, 5, 5.5f, 6, 6.5f, 7, 7.5f};
    for (int j = 0; j < 16; ++j) { a[j] = t[j]; b[j] = t[16 + j]; }
    const u6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, 1.0f);
    if (threadIdx.x == 0) for (int j = 0; j < 6; ++j) out[j] = q[j];
}

int main() {
    float* d_d; unsigned* d_o;
    hipMalloc(&d_d, 64 * 16 * 4); hipMalloc(&d_o, 24);
    std::vector<float> D(64 * 16);
    auto rows_cols = [&](const char* what) {
        hipMemcpy(D.data(), d_d, D.size() * 4, hipMemcpyDeviceToHost);
        float M[32][32];
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) M[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31] = D[l * 16 + r];
        printf("%s: D[0][0] = %g;", what, M[0][0]);
        printf(" rows != 64:");
        for (int i = 0; i < 32; ++i) { bool any = false; for (int j = 0; j < 32; ++j) any |= M[i][j] != 64.f; if (any) printf(" %d(%g)", i, M[i][0] != 64.f ? M[i][0] : M[i][31]); }
        printf(" | cols != 64:");
        for (int j = 0; j < 32; ++j) { bool any = false; for (int i = 0; i < 32; ++i) any |= M[i][j] != 64.f; if (any) printf(" %d(%g)", j, M[0][j] != 64.f ? M[0][j] : M[31][j]); }
        printf("\n");
    };
    for (int la : {-1, 0, 1, 5, 31, 32, 33, 63}) {
        hipLaunchKernelGGL(k_scale, dim3(1), dim3(64), 0, 0, d_d, la, -1, 0);
        char buf[64]; snprintf(buf, 64, "scale_a x2 in lane %2d only", la);
        rows_cols(buf);
    }
    for (int lb : {0, 7, 32, 40}) {
        hipLaunchKernelGGL(k_scale, dim3(1), dim3(64), 0, 0, d_d, -1, lb, 0);
        char buf[64]; snprintf(buf, 64, "scale_b x2 in lane %2d only", lb);
        rows_cols(buf);
    }
    hipLaunchKernelGGL(k_order, dim3(1), dim3(64), 0, 0, d_o);
    unsigned q[6];
    hipMemcpy(q, d_o, 24, hipMemcpyDeviceToHost);
    printf("2xpk16(a = codes 0..15, b = codes 16..31) little-endian 6-bit fields:");
    unsigned long long lo = q[0] | ((unsigned long long)q[1] << 32), mid = q[2] | ((unsigned long long)q[3] << 32), hi = q[4] | ((unsigned long long)q[5] << 32);
    for (int j = 0; j < 32; ++j) {
        const int bit = 6 * j;
        unsigned v;
        if (bit + 6 <= 64) v = (lo >> bit) & 63;
        else if (bit < 64) v = ((lo >> bit) | (mid << (64 - bit))) & 63;
        else if (bit + 6 <= 128) v = (mid >> (bit - 64)) & 63;
        else if (bit < 128) v = ((mid >> (bit - 64)) | (hi << (128 - bit))) & 63;
        else v = (hi >> (bit - 128)) & 63;
        printf(" %u", v);
    }
    printf("\n");
    return 0;
}
