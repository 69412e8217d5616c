// Hardware probe (gfx950) for the round-2 open finding (profiles/r02_experiments.md, "SGPR wave index"): the failing FFN1 epilogue
// contains, per column tile,
//     v_mad_u64_u32 v[146:147], s[12:13], s6, v146, v[0:1]     ; VALU, carry-out mask -> s[12:13] (dead)
//     s_lshl_b32 s5, s4, 2
//     s_or_b32  s7,  s5, 16
//     s_or_b32  s12, s5, 64                                     ; SALU re-uses s12 / s13 at once: store offsets of the LO halves
//     s_or_b32  s13, s5, 0x50
//     ... ~120 VALU instructions ...
//     buffer_store_dwordx4 ..., s12 offen                       ; lo halves
// Question: can the (multi-pass, quarter-rate) VALU instruction's SGPR write land AFTER the SALU writes that follow it in
// program order - a write-after-write on an SGPR across the two pipes that neither the hardware nor hipcc's hazard recogniser
// orders?  Each variant below issues that pair with G wait states between the VALU instruction and the SALU write and reads
// the SGPR back D cycles later; `bad` counts read-backs that are not the SALU value.  Other wavefronts of the SIMD run the
// same loop (VALU contention as in the epilogue: transcendental chains).
//
//   hipcc --offload-arch=gfx950 -O2 tools/hazard_probe.hip -o tools/_bin/hazard_probe && tools/_bin/hazard_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define NOP_0 ""
#define NOP_1 "s_nop 0\n\t"
#define NOP_2 "s_nop 1\n\t"
#define NOP_4 "s_nop 3\n\t"
#define NOP_8 "s_nop 7\n\t"
#define NOP_16 "s_nop 15\n\t"
#define NOP_32 "s_nop 15\n\ts_nop 15\n\t"

// KIND 0: v_mad_u64_u32 (carry -> s[20:21]);  1: v_add_co_u32 (carry -> s[20:21]);  2: v_cmp_lt_u32 (mask -> s[20:21]);
//      3: v_readfirstlane_b32 (-> s20)
#define PROBE_BODY(VALU_TEXT, GAP, DELAY)                                                                      \
    asm volatile(VALU_TEXT GAP "s_mov_b32 s20, 0x12345678\n\t"                                                    \
                               "s_mov_b32 s21, 0x0badcafe\n\t" DELAY "v_mov_b32 %[g0], s20\n\t"                   \
                               "v_mov_b32 %[g1], s21\n\t"                                                         \
                 : [g0] "=&v"(g0), [g1] "=&v"(g1), [o] "+v"(o), [t] "=&v"(t32)                                                      \
                 : [a] "s"(sa), [b] "v"(vb), [c] "v"(c64)                                                          \
                 : "s20", "s21", "vcc")

template <int KIND, int GAP, int DELAY>
__global__ void probe(uint32_t* bad, uint32_t* seen, int iters, float* sink) {
    uint32_t nbad = 0, last0 = 0, last1 = 0;
    float f = threadIdx.x * 0.001f + 0.5f;
    uint32_t sa = __builtin_amdgcn_readfirstlane(blockIdx.x * 8192u + 12345u);
    uint32_t vb = threadIdx.x * 7u + 3u;
    uint64_t c64 = (uint64_t)threadIdx.x * 32u;
    uint64_t o = 0;
    for (int it = 0; it < iters; ++it) {
        // a transcendental chain in front, as in the SiLU epilogue
        f = __expf(-f) + 1.0f;
        f = __builtin_amdgcn_rcpf(f) + 0.25f;
        uint32_t g0, g1, t32;
#define VALU_MAD "v_mad_u64_u32 %[o], s[20:21], %[a], %[b], %[c]\n\t"
#define VALU_ADD "v_add_co_u32 %[t], s[20:21], %[a], %[b]\n\t"
#define VALU_CMP "v_cmp_lt_u32 s[20:21], %[a], %[b]\n\t"
#define VALU_RFL "v_readfirstlane_b32 s20, %[b]\n\t"
        if constexpr (KIND == 0) {
            if constexpr (GAP == 0 && DELAY == 0) PROBE_BODY(VALU_MAD, NOP_0, NOP_0);
            else if constexpr (GAP == 0 && DELAY == 8) PROBE_BODY(VALU_MAD, NOP_0, NOP_8);
            else if constexpr (GAP == 0 && DELAY == 32) PROBE_BODY(VALU_MAD, NOP_0, NOP_32);
            else if constexpr (GAP == 1 && DELAY == 32) PROBE_BODY(VALU_MAD, NOP_1, NOP_32);
            else if constexpr (GAP == 2 && DELAY == 32) PROBE_BODY(VALU_MAD, NOP_2, NOP_32);
            else if constexpr (GAP == 4 && DELAY == 32) PROBE_BODY(VALU_MAD, NOP_4, NOP_32);
            else if constexpr (GAP == 8 && DELAY == 32) PROBE_BODY(VALU_MAD, NOP_8, NOP_32);
            else if constexpr (GAP == 16 && DELAY == 32) PROBE_BODY(VALU_MAD, NOP_16, NOP_32);
            else PROBE_BODY(VALU_MAD, NOP_32, NOP_32);
        } else if constexpr (KIND == 1) {
            if constexpr (DELAY == 0) PROBE_BODY(VALU_ADD, NOP_0, NOP_0);
            else PROBE_BODY(VALU_ADD, NOP_0, NOP_32);
        } else if constexpr (KIND == 2) {
            if constexpr (DELAY == 0) PROBE_BODY(VALU_CMP, NOP_0, NOP_0);
            else PROBE_BODY(VALU_CMP, NOP_0, NOP_32);
        } else {
            if constexpr (DELAY == 0) PROBE_BODY(VALU_RFL, NOP_0, NOP_0);
            else PROBE_BODY(VALU_RFL, NOP_0, NOP_32);
        }
        if (g0 != 0x12345678u || g1 != 0x0badcafeu) { ++nbad; last0 = g0; last1 = g1; }
        vb += (uint32_t)o + (KIND == 1 ? t32 : 0u);
    }
    if (nbad) {
        atomicAdd(bad, nbad);
        seen[0] = last0;
        seen[1] = last1;
    }
    if (f == 123.456f) *sink = f + (float)o;
}

template <int KIND, int GAP, int DELAY>
void run(const char* what, uint32_t* d_bad, uint32_t* d_seen, float* d_sink) {
    for (int threads : {64, 256, 512, 1024}) {
        (void)hipMemset(d_bad, 0, 4);
        (void)hipMemset(d_seen, 0, 8);
        const int blocks = 2048, iters = 2000;
        hipLaunchKernelGGL((probe<KIND, GAP, DELAY>), dim3(blocks), dim3(threads), 0, 0, d_bad, d_seen, iters, d_sink);
        (void)hipDeviceSynchronize();
        uint32_t bad = 0, seen[2] = {0, 0};
        (void)hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(seen, d_seen, 8, hipMemcpyDeviceToHost);
        const double total = (double)blocks * threads * iters;
        printf("%-18s gap %2d delay %2d  threads/wg %4d : wrong read-backs %u of %.3g lane-trials", what, GAP, DELAY, threads, bad, total);
        if (bad) printf("   (last wrong values s20 = 0x%08x  s21 = 0x%08x)", seen[0], seen[1]);
        printf("\n");
    }
}

int main() {
    uint32_t *d_bad, *d_seen;
    float* d_sink;
    (void)hipMalloc(&d_bad, 4);
    (void)hipMalloc(&d_seen, 8);
    (void)hipMalloc(&d_sink, 4);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    printf("device: %s (%s)\n", prop.name, prop.gcnArchName);
    run<0, 0, 0>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<0, 0, 8>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<0, 0, 32>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<0, 1, 32>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<0, 2, 32>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<0, 4, 32>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<0, 8, 32>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<0, 16, 32>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<0, 32, 32>("v_mad_u64_u32", d_bad, d_seen, d_sink);
    run<1, 0, 0>("v_add_co_u32", d_bad, d_seen, d_sink);
    run<1, 0, 32>("v_add_co_u32", d_bad, d_seen, d_sink);
    run<2, 0, 0>("v_cmp_lt_u32", d_bad, d_seen, d_sink);
    run<2, 0, 32>("v_cmp_lt_u32", d_bad, d_seen, d_sink);
    run<3, 0, 0>("v_readfirstlane", d_bad, d_seen, d_sink);
    run<3, 0, 32>("v_readfirstlane", d_bad, d_seen, d_sink);
    return 0;
}
