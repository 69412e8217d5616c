// Probe 2 (gfx950): how many single-issue instructions hide in the gap behind a v_mfma_f32_32x32x16_f16?
//
// Round 3's coissue_probe.hip concluded "MFMA and VALU exclude each other on a SIMD".  Its ISA shows why that measurement
// does not answer the question: hipcc hoisted the four MFMAs of an iteration in front of ALL the FMAs (sched_barrier(0)
// does not pin builtins to a position relative to plain arithmetic), packed part of them into v_pk_fma_f32, and the
// fillers ended up bunched behind the last MFMA of the iteration.  Here every instruction is its own `asm volatile`
// (volatile asm statements keep their program order), the stream is  MFMA, NF fillers, MFMA, NF fillers, ...  with four
// rotating accumulators, and time is read with s_memtime (shader cycles) AND the wall clock (effective clock = the ratio),
// so a DVFS change is not mistaken for an issue cost.  MI355X_MICROARCH.md, "Per-instruction cycle constants": <= 5
// single-issue instructions hidden per gap at ONE wavefront per SIMD, 32.4 cycles / MFMA.
//
//   A. one wavefront per SIMD (256-thread workgroups, one per CU), accumulators in AGPRs or in arch VGPRs,
//      filler kind x NF = 0..8  ->  cycles per MFMA
//   B. two wavefronts per SIMD running the SAME stream (512-thread workgroups) -> cycles per MFMA per SIMD
//   C. two wavefronts per SIMD, one pure MFMA and one pure filler stream (the round-3 "pair" test, asm-placed)
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/coissue_probe2 tools/coissue_probe2.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

enum Kind { K_FMA, K_MUL, K_EXP, K_CVT, K_AND, K_MAX3, K_DSR, K_MIX, K_PKMUL, K_ACCRD, K_MIXLO, K_SOFTMAX, K_COUNT };
static const char* kind_name[] = {"v_fma_f32", "v_mul_f32", "v_exp_f32", "v_cvt_pkrtz", "v_and_b32", "v_max3_f32", "ds_read_b128",
                                  "v_fma_mix_f32", "v_pk_mul_f32", "v_accvgpr_read", "v_fma_mixlo_f16", "softmax mix"};

struct Fill {
    float f[8];
    f4v d[4];
    f2v p[4];
    float ag[8];
    uint32_t lds_addr;
    float c1, c2;
};

template <int KIND>
__device__ __forceinline__ void filler(Fill& s, int j) {
    float& f = s.f[j & 7];
    const float g = s.f[(j + 3) & 7];
    if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(s.c1), "v"(s.c2));
    else if constexpr (KIND == K_MUL) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f) : "v"(s.c1));
    else if constexpr (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(f));
    else if constexpr (KIND == K_CVT) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(f) : "v"(s.c1));
    else if constexpr (KIND == K_AND) asm volatile("v_and_b32 %0, %1, %0" : "+v"(f) : "v"(s.c1));
    else if constexpr (KIND == K_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f) : "v"(s.c1), "v"(s.c2));
    else if constexpr (KIND == K_DSR) asm volatile("ds_read_b128 %0, %1" : "=v"(s.d[j & 3]) : "v"(s.lds_addr));
    else if constexpr (KIND == K_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(f) : "v"(s.c1), "v"(s.c2));
    else if constexpr (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(s.p[j & 3]) : "v"(s.p[(j + 1) & 3]));
    else if constexpr (KIND == K_ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f) : "a"(s.ag[j & 7]));
    else if constexpr (KIND == K_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(f) : "v"(g), "v"(s.c1), "v"(s.c2));
    else if constexpr (KIND == K_SOFTMAX) {
        // the attention kernel's own mix per two probabilities, cyclically: fmamk-like fma, exp, add, and, sub, cvt_pkrtz, mul, max3
        switch (j & 7) {
            case 0: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(s.c1), "v"(s.c2)); break;
            case 1: asm volatile("v_exp_f32 %0, %0" : "+v"(f)); break;
            case 2: asm volatile("v_add_f32 %0, %1, %0" : "+v"(f) : "v"(s.c1)); break;
            case 3: asm volatile("v_and_b32 %0, %1, %0" : "+v"(f) : "v"(s.c1)); break;
            case 4: asm volatile("v_sub_f32 %0, %1, %0" : "+v"(f) : "v"(s.c1)); break;
            case 5: asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(f) : "v"(s.c1)); break;
            case 6: asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f) : "v"(s.c1)); break;
            default: asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f) : "v"(s.c1), "v"(s.c2)); break;
        }
    }
}

template <bool AGPR>
__device__ __forceinline__ void mfma(f16v& acc, const h8& x, const h8& y) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
}

__device__ __forceinline__ void init(Fill& s, f16v (&acc)[4], h8& x, h8& y) {
    const int t = threadIdx.x;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(((t * 37 + j * 11) % 64) * 0.03125f - 1.f); y[j] = (_Float16)(((t * 13 + j * 7) % 64) * 0.03125f - 1.f); }
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    for (int j = 0; j < 8; ++j) { s.f[j] = 0.5f + 0.001f * ((t + j) % 97); s.ag[j] = 1.f + j; }
    for (int j = 0; j < 4; ++j) { s.d[j] = f4v{0, 0, 0, 0}; s.p[j] = f2v{1.0001f, 0.9999f}; }
    s.lds_addr = (uint32_t)(t & 255) * 16u;
    s.c1 = 0.99993f; s.c2 = 0.0001f;
}

__device__ __forceinline__ float fold(const Fill& s, const f16v (&acc)[4]) {
    float r = 0;
    for (int u = 0; u < 4; ++u) for (int i = 0; i < 16; ++i) r += acc[u][i];
    for (int j = 0; j < 8; ++j) r += s.f[j];
    for (int j = 0; j < 4; ++j) r += s.d[j][0] + s.d[j][3] + s.p[j][0] + s.p[j][1];
    return r;
}

// ROLE 0: every wavefront runs MFMA + NF fillers per gap.  ROLE 1: wavefronts 0-3 pure MFMA, wavefronts 4-7 (same SIMDs)
// 8 x NF fillers per iteration and no MFMA.  out[block * 8 + wave] = cycles of the timed loop.
template <int KIND, int NF, bool AGPR, int THREADS, int ROLE>
__global__ __launch_bounds__(THREADS) void k_gap(uint64_t* out, float* sink, int iters) {
    __shared__ f4v lds_buf[256];
    lds_buf[threadIdx.x & 255] = f4v{1, 2, 3, 4};
    Fill s;
    f16v acc[4];
    h8 x, y;
    init(s, acc, x, y);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if constexpr (ROLE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                mfma<AGPR>(acc[u & 3], x, y);
#pragma unroll
                for (int j = 0; j < NF; ++j) filler<KIND>(s, u * NF + j);
            }
            if (KIND == K_DSR && NF > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (wave < 4) {          // two separate loops: no per-instruction branch in either stream
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) mfma<AGPR>(acc[u & 3], x, y);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int j = 0; j < NF; ++j) filler<KIND>(s, u * NF + j);
            }
        }
    }
    asm volatile("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    const float r = fold(s, acc);
    if (r == 12345.678f) *sink = r;
}

struct Result { double cyc, ns; };

template <typename K>
Result run(K kernel, int threads, uint64_t* out_d, float* sink, int waves_used_lo, int waves_used_hi) {
    const int blocks = 256, iters = 4000;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out_d, sink, 64);
    (void)hipDeviceSynchronize();
    auto w0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out_d, sink, iters);
    (void)hipDeviceSynchronize();
    const double ns = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() * 1e9;
    std::vector<uint64_t> h(blocks * 8);
    (void)hipMemcpy(h.data(), out_d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = waves_used_lo; w < waves_used_hi; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    return {sum / n / (iters * 8.0), ns / (iters * 8.0)};
}

// section C: the MFMA wavefronts' cycles per MFMA, the filler wavefronts' cycles per 8 NF fillers, and the wall time - with the
// filler stream sized from "far shorter than" to "longer than" the 256 cycles of the 8 MFMAs
template <int KIND>
void sweep_pair(uint64_t* out_d, float* sink) {
    Result m[8], f[8];
#define ONE(i, NF) m[i] = run(k_gap<KIND, NF, true, 512, 1>, 512, out_d, sink, 0, 4); f[i] = run(k_gap<KIND, NF, true, 512, 1>, 512, out_d, sink, 4, 8);
    ONE(0, 0) ONE(1, 2) ONE(2, 4) ONE(3, 8) ONE(4, 12) ONE(5, 16) ONE(6, 24) ONE(7, 32)
#undef ONE
    printf("%-16s MFMA wavefront, cycles / MFMA      :", kind_name[KIND]);
    for (int i = 0; i < 8; ++i) printf(" %6.1f", m[i].cyc);
    printf("\n%-16s filler wavefront, cycles / NF fillers:", kind_name[KIND]);
    for (int i = 0; i < 8; ++i) printf(" %6.1f", f[i].cyc);
    printf("\n%-16s wall ns / (MFMA + NF fillers)       :", kind_name[KIND]);
    for (int i = 0; i < 8; ++i) printf(" %6.1f", m[i].ns);
    printf("\n");
    fflush(stdout);
}

template <int KIND, bool AGPR, int THREADS, int ROLE>
void sweep(uint64_t* out_d, float* sink, const char* label) {
    const int w = THREADS / 64;
    printf("%-16s %-5s %s", kind_name[KIND], AGPR ? "agpr" : "vgpr", label);
    Result r[8];
#define ONE(i, NF) r[i] = run(k_gap<KIND, NF, AGPR, THREADS, ROLE>, THREADS, out_d, sink, ROLE == 1 ? 0 : 0, ROLE == 1 ? 4 : w);
    ONE(0, 0) ONE(1, 1) ONE(2, 2) ONE(3, 3) ONE(4, 4) ONE(5, 5) ONE(6, 6) ONE(7, 8)
#undef ONE
    for (int i = 0; i < 8; ++i) printf(" %6.1f", r[i].cyc);
    printf("   | wall ns/MFMA-slot:");
    for (int i = 0; i < 8; ++i) printf(" %5.1f", r[i].ns);
    printf("\n");
    fflush(stdout);
}

int main() {
    uint64_t* out_d;
    float* sink;
    (void)hipMalloc(&out_d, 256 * 8 * 8);
    (void)hipMalloc(&sink, 4);
    printf("cycles (s_memtime) per MFMA slot of ONE wavefront; columns NF = 0 1 2 3 4 5 6 8 fillers behind every v_mfma_f32_32x32x16_f16\n");
    printf("A. one wavefront per SIMD\n");
#define ROW(K) sweep<K, true, 256, 0>(out_d, sink, "1w/SIMD"); sweep<K, false, 256, 0>(out_d, sink, "1w/SIMD");
    ROW(K_FMA) ROW(K_MUL) ROW(K_EXP) ROW(K_CVT) ROW(K_AND) ROW(K_MAX3) ROW(K_DSR) ROW(K_MIX) ROW(K_PKMUL) ROW(K_ACCRD) ROW(K_MIXLO) ROW(K_SOFTMAX)
#undef ROW
    printf("B. two wavefronts per SIMD, same stream in both (cycles per MFMA slot of one wavefront: 64 = the pair issues one MFMA per 32 cycles)\n");
#define ROW(K) sweep<K, true, 512, 0>(out_d, sink, "2w/SIMD"); sweep<K, false, 512, 0>(out_d, sink, "2w/SIMD");
    ROW(K_FMA) ROW(K_EXP) ROW(K_CVT) ROW(K_DSR) ROW(K_SOFTMAX)
#undef ROW
    printf("C. two wavefronts per SIMD: wavefronts 0-3 pure MFMA (8 per iteration), 4-7 pure fillers (8 NF per iteration); NF = 0 2 4 8 12 16 24 32\n");
    sweep_pair<K_FMA>(out_d, sink);
    sweep_pair<K_EXP>(out_d, sink);
    sweep_pair<K_SOFTMAX>(out_d, sink);
    return 0;
}
