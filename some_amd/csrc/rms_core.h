// Summation order of the silence slicer's frame RMS (utils/slicer2.py:5-38: np.mean(np.abs(x) ** 2) over a strided
// frame view, float32).  numpy reduces the contiguous frame axis with its pairwise scheme: blocks of at most 128
// elements are summed into 8 interleaved accumulators that are folded as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a
// scalar tail, and longer ranges are split at n/2 rounded down to a multiple of 8.  Chunk boundaries depend on
// rms < threshold and argmin, so the device result has to be the same float32 bit pattern; this header restates that
// order once for the HIP kernel (ingest.hip) and for a host build (tests/host/host_emu.cpp) that is checked against
// numpy itself on a machine without a GPU.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RMS_HD __host__ __device__ __forceinline__
#else
#define RMS_HD inline
#endif
#define RMS_ADD(a, b) ((a) + (b))
#define RMS_MUL(a, b) ((a) * (b))

// x * x + acc must stay two roundings.  HIP contracts by default, and __fmul_rn / __fadd_rn do not help: in this
// toolchain's headers they are plain * and + compiled WITH the contract flag, so they fuse after inlining.  Plain
// operators under contract(off) carry no such flag (g++ host build: -ffp-contract=off on the command line).
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

constexpr int kRmsBlock = 128;                    // numpy PW_BLOCKSIZE
constexpr int kRmsMaxDepth = 40;                  // explicit stack: 2 entries per level, frames up to 2^19 samples

// sum of ld(i)^2 for i in [lo, lo + m), m <= 128
template <class Load>
RMS_HD float rms_leaf(const Load& ld, int lo, int m) {
    if (m < 8) {
        float r = 0.f;
        for (int i = 0; i < m; ++i) { const float v = ld(lo + i); r = RMS_ADD(r, RMS_MUL(v, v)); }
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = ld(lo + j); r[j] = RMS_MUL(v, v); }
    int i = 8;
    for (; i < m - (m % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = ld(lo + i + j); r[j] = RMS_ADD(r[j], RMS_MUL(v, v)); }
    }
    float res = RMS_ADD(RMS_ADD(RMS_ADD(r[0], r[1]), RMS_ADD(r[2], r[3])), RMS_ADD(RMS_ADD(r[4], r[5]), RMS_ADD(r[6], r[7])));
    for (; i < m; ++i) { const float v = ld(lo + i); res = RMS_ADD(res, RMS_MUL(v, v)); }
    return res;
}

// numpy's pairwise sum of squares over n samples (recursion unrolled onto an explicit stack; m < 0 marks "add the
// two partial sums below")
template <class Load>
RMS_HD float rms_pairwise_sumsq(const Load& ld, int n) {
    int lo_stk[kRmsMaxDepth], m_stk[kRmsMaxDepth];
    float v_stk[kRmsMaxDepth];
    int sp = 0, vp = 0;
    lo_stk[0] = 0;
    m_stk[0] = n;
    sp = 1;
    while (sp > 0) {
        --sp;
        const int lo = lo_stk[sp], m = m_stk[sp];
        if (m < 0) {
            const float b = v_stk[--vp], a = v_stk[--vp];
            v_stk[vp++] = RMS_ADD(a, b);
        } else if (m <= kRmsBlock) {
            v_stk[vp++] = rms_leaf(ld, lo, m);
        } else {
            int n2 = m / 2;
            n2 -= n2 % 8;
            m_stk[sp++] = -1;
            lo_stk[sp] = lo + n2; m_stk[sp++] = m - n2;
            lo_stk[sp] = lo; m_stk[sp++] = n2;
        }
    }
    return v_stk[0];
}
