// 3-term split-f16 GEMM family for gfx950: fp32-equivalent results on the f16 matrix pipe
// (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate; 3 MFMAs per logical product => 5.3x the f32-MFMA roof).
//
//   C[g] = epilogue( A[g] * W[g]^T ),  A [M,K] and W [N,K] both in the SPLIT32 format of split.h
//   (per 32-element k-block: 32 f16 hi | 32 f16 lo, same bytes as fp32),  acc += ah*bh + ah*bl + al*bh  in fp32.
//
// Same role, epilogues and block->tile map as gemm.hip (exact-f32 path); what changes is the inner product and
// therefore the balance: one k-block (128 B per row) now costs 24 MFMA issue slots of 32 cycles per 64x64 wave
// tile instead of 64 slots of 64 cycles, so staging bandwidth per flop matters 5x more and the tile grows:
// default 256 x 256 per workgroup, 8 waves (4 x 2), each wave 64 x 128 (2 x 4 MFMA tiles, 128 accumulator
// VGPRs), 2 LDS stages of (256 + 256) rows x 144 B = 147 KB -> 1 workgroup / CU, 2 waves / SIMD.
// Because a SPLIT32 row segment is byte-for-byte a 128-byte run in HBM, operands are staged exactly like fp32
// rows (16-byte global loads -> registers -> ds_write_b128 into [rows][36 dwords], conflict-free ds_read_b128);
// neither operand is converted in this kernel: weights are split once at pack time, activations by the
// epilogue of whichever kernel produced them (LayerNorm, SiLU epilogue below, attention, dwconv).
#include "internal.h"
#include "split.h"

namespace {

constexpr int LDT = 36;      // LDS row in dwords: 16 (32 hi halves) + 16 (32 lo halves) + 4 pad

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool OUT_SPLIT>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void hgemm3_kernel(GemmArgs a) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int STAGE = (BM + BN) * LDT;               // dwords
    constexpr int NLD = (BM + BN) * 8 / NT;              // 16-byte chunks per thread per k-block
    static_assert((BM + BN) * 8 % NT == 0, "staging must divide evenly");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const GemmGroup g = a.g[blockIdx.y];
    const int n_tiles = a.n_tiles;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int m_tile = (j / n_tiles) * 8 + xcd;
    const int n_tile = j % n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (m0 >= a.M || n0 >= g.N) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, kg = lane >> 5;
    const int nk = a.K >> 5;

    // ---- staging roles
    const char* src[NLD];
    int dst[NLD];
    bool ok[NLD];
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
        const int c = tid + p * NT;
        const int row = c >> 3, col = c & 7;
        dst[p] = row * LDT + col * 4;
        if (row < BM) {
            ok[p] = (m0 + row) < a.M;
            src[p] = reinterpret_cast<const char*>(g.A) + ((size_t)(ok[p] ? m0 + row : 0) * a.lda) * 4 + col * 16;
        } else {
            const int r = row - BM;
            ok[p] = (n0 + r) < g.N;
            src[p] = reinterpret_cast<const char*>(g.W) + ((size_t)(ok[p] ? n0 + r : 0) * a.K) * 4 + col * 16;
        }
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 stage[NLD];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < NLD; ++p)
            stage[p] = ok[p] ? *reinterpret_cast<const f32x4*>(src[p] + (size_t)kt * 128) : zero4;
    };
    auto lstore = [&](int buf) {
        float* base = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < NLD; ++p) *reinterpret_cast<f32x4*>(base + dst[p]) = stage[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();

    const int a_off = (wm * TM * 32 + l31) * LDT + kg * 4;
    const int w_off = (BM + wn * TN * 32 + l31) * LDT + kg * 4;

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float* As = lds + buf * STAGE + a_off;
        const float* Ws = lds + buf * STAGE + w_off;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                    // two k = 16 slabs per 32-element k-block
            half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(As + i * 32 * LDT + s * 8);
                al[i] = *reinterpret_cast<const half8*>(As + i * 32 * LDT + 16 + s * 8);
            }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                bh[jn] = *reinterpret_cast<const half8*>(Ws + jn * 32 * LDT + s * 8);
                bl[jn] = *reinterpret_cast<const half8*>(Ws + jn * 32 * LDT + 16 + s * 8);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[jn], acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[jn], acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jn], acc[i][jn], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    const int hi = kg;
    if constexpr (EPI == EPI_GLU || EPI == EPI_GLU_RES) {
        static_assert(TN % 2 == 0, "GLU pairs adjacent 32-column tiles");
#pragma unroll
        for (int jp = 0; jp < TN / 2; ++jp) {
            const int np = n0 + (wn * TN + 2 * jp) * 32 + l31;        // packed column of the `a` half
            if (np >= g.N) continue;
            const int oc = (np - l31) / 2 + l31;                       // output column: packed 64-block -> 32 outputs
            const float ba = g.bias[np], bg = g.bias[np + 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (m < a.M) {
                        float v = (acc[i][2 * jp][r] + ba) * sigmoidf_(acc[i][2 * jp + 1][r] + bg);
                        if constexpr (EPI == EPI_GLU_RES) {
                            v += g.res[(size_t)m * a.ldr + oc];
                            if (g.mask != nullptr && g.mask[m] == 0) v = 0.f;
                        }
                        g.C[(size_t)m * a.ldc + oc] = v;
                    }
                }
        }
    } else if constexpr (EPI == EPI_QKV) {
        // N = 1536: columns [0,512) -> Q plane, [512,1024) -> K plane (both SPLIT32 rows of 512),
        // [1024,1536) -> V^T f16 planes with the frame index contiguous (what attention_f16x3.hip stages)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const int region = n >> 9;                                  // uniform per 32-column tile
            if (region < 2) {
                char* plane = reinterpret_cast<char*>(region == 0 ? g.C : g.C2);
                const int nn = n & 511;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        half_t h, l;
                        split_f16(acc[i][jn][r], h, l);
                        const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
                        const uint32_t other = __shfl_xor(mine, 1, 64);
                        const uint32_t word = (lane & 1) ? ((other >> 16) | (mine & 0xffff0000u))
                                                         : ((mine & 0xffffu) | (other << 16));
                        if (m < a.M) {
                            char* rowp = plane + (size_t)m * 2048 + (size_t)((nn - l31) >> 5) * 128;
                            *reinterpret_cast<uint32_t*>(rowp + ((lane & 1) ? 64 + (l31 - 1) * 2 : l31 * 2)) = word;
                        }
                    }
            } else {
                const int d = n - 1024;
                char* vh = reinterpret_cast<char*>(g.C3) + (size_t)d * g.ldv * 2;
                char* vl = vh + (size_t)kDim * g.ldv * 2;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int m = m0 + (wm * TM + i) * 32 + 8 * rq + 4 * hi;     // 4 consecutive frames
                        half4 hh, ll;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { half_t h, l; split_f16(acc[i][jn][4 * rq + e], h, l); hh[e] = h; ll[e] = l; }
                        if (m < g.ldv) {      // rows >= M carry exact zeros (zero-filled A rows, no bias): finite padding
                            *reinterpret_cast<half4*>(vh + (size_t)m * 2) = hh;
                            *reinterpret_cast<half4*>(vl + (size_t)m * 2) = ll;
                        }
                    }
            }
        }
    } else {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const bool nv = n < g.N;
            float bias = 0.f;
            if constexpr (EPI != EPI_NONE) bias = nv ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float v = acc[i][jn][r] + bias;
                    if constexpr (EPI == EPI_BIAS) {
                        if (g.act == 1) v = sigmoidf_(v);
                        if (g.mask != nullptr && m < a.M && g.mask[m] == 0) v = 0.f;
                    } else if constexpr (EPI == EPI_BIAS_SILU) {
                        v = v * sigmoidf_(v);
                    } else if constexpr (EPI == EPI_BIAS_RES) {
                        if (nv && m < a.M) v = g.res[(size_t)m * a.ldr + n] + a.alpha * v;
                    }
                    if constexpr (OUT_SPLIT) {
                        // SPLIT32 output: this tile's 32 columns are one k-block of the consumer GEMM.  Lane pairs
                        // exchange halves so each lane stores one packed dword: even lane -> two hi, odd -> two lo.
                        half_t h, l;
                        split_f16(v, h, l);
                        const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
                        const uint32_t other = __shfl_xor(mine, 1, 64);
                        const uint32_t word = (lane & 1) ? ((other >> 16) | (mine & 0xffff0000u))
                                                         : ((mine & 0xffffu) | (other << 16));
                        if (nv && m < a.M) {
                            char* rowp = reinterpret_cast<char*>(g.C) + (size_t)m * a.ldc * 4 + (size_t)((n - l31) >> 5) * 128;
                            *reinterpret_cast<uint32_t*>(rowp + ((lane & 1) ? 64 + (l31 - 1) * 2 : l31 * 2)) = word;
                        }
                    } else {
                        if (nv && m < a.M) g.C[(size_t)m * a.ldc + n] = v;
                    }
                }
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool OUT_SPLIT>
hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t LDS_BYTES = 2 * (size_t)(BM + BN) * LDT * sizeof(float);
    static bool attr_set = false;
    auto kern = &hgemm3_kernel<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int n_max = 0;
    for (int g = 0; g < a.groups; ++g) n_max = a.g[g].N > n_max ? a.g[g].N : n_max;
    const int m_tiles = (a.M + BM - 1) / BM, n_tiles = (n_max + BN - 1) / BN;
    GemmArgs b = a;
    b.n_tiles = n_tiles;
    dim3 grid((unsigned)((m_tiles + 7) / 8 * 8 * n_tiles), (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), LDS_BYTES, s, b);
    return hipGetLastError();
}

template <int EPI, bool OUT_SPLIT>
hipError_t launch_epi(const GemmArgs& a, int tile, hipStream_t s) {
    switch (tile) {
        case 0: return launch_cfg<2, 2, 2, 2, EPI, OUT_SPLIT>(a, s);    // 128 x 128, 4 waves
        case 1: return launch_cfg<4, 2, 2, 2, EPI, OUT_SPLIT>(a, s);    // 256 x 128, 8 waves
        default: return launch_cfg<4, 2, 2, 4, EPI, OUT_SPLIT>(a, s);   // 256 x 256, 8 waves
    }
}

}  // namespace

hipError_t launch_gemm_f16x3(GemmEpi epi, const GemmArgs& a, bool out_split, int tile, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if ((a.K & 31) || (a.lda & 31)) return hipErrorInvalidValue;
    if (out_split && epi != EPI_BIAS_SILU) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_NONE: return launch_epi<EPI_NONE, false>(a, tile, s);
        case EPI_BIAS: return launch_epi<EPI_BIAS, false>(a, tile, s);
        case EPI_BIAS_SILU: return out_split ? launch_epi<EPI_BIAS_SILU, true>(a, tile, s) : launch_epi<EPI_BIAS_SILU, false>(a, tile, s);
        case EPI_BIAS_RES: return launch_epi<EPI_BIAS_RES, false>(a, tile, s);
        case EPI_GLU: return launch_epi<EPI_GLU, false>(a, tile, s);
        case EPI_GLU_RES: return launch_epi<EPI_GLU_RES, false>(a, tile, s);
        case EPI_QKV:
            for (int g = 0; g < a.groups; ++g)
                if (a.g[g].N != 3 * kDim || !a.g[g].C2 || !a.g[g].C3 || (a.g[g].ldv & 255) || a.g[g].ldv < a.M) return hipErrorInvalidValue;
            return launch_epi<EPI_QKV, false>(a, tile, s);
    }
    return hipErrorInvalidValue;
}
