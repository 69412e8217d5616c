// Training GEMMs on operands STORED in 16 bits (mixed-precision training: pl_trainer_precision 'bf16' / '16-mixed',
// reference configs/midi_conformer.yaml:35 - under autocast nn.Linear reads and writes 16-bit tensors, training/base_task.py:260-283).
//
//   C [M, N] = A16 [M, K] * B16 [N, K]^T        both operands row-major with the CONTRACTION index contiguous, bf16 or f16
//
// Why a second kernel beside gemm16_kernel (gemm_f16x3.hip), which converts fp32 arrays in its staging path: measured per layer
// shape (tools/train_gemm_bench.py, 8 x 2584 frames) that kernel moves 11 - 14 TB/s from L2 into the CUs and still runs the matrix
// pipe at 16 - 22 % - one product per staged element makes it latency-bound (bytes in flight / load latency), not MFMA- or HBM-bound.
// Operands that already lie in memory as 16-bit values need half the bytes per product AND no register round trip: they go
// global -> LDS by DMA (global_load_lds, 16 B per lane) into a ring filled two k-steps ahead, exactly the ring of
// hgemm3_ring_kernel with the [hi | lo] halves of a 64-byte LDS row replaced by k 0-15 | k 16-31 of a 32-element k-step.
// The weights' 16-bit images (W16 [N, K] for the forward, W16T [K, N] for the data gradient) are refreshed once per optimiser step
// (transpose16_kernel); activations are written in 16 bits by the producing epilogue:
//   G16S_FFN1   bias + h16 = rn16(acc), a16 = rn16(dropout(silu(h16)))   - conform_ffn.forward ln1 / act / drop1 (Gconform.py:29-32)
//   G16S_DSILU  dh16 = rn16(acc * keep * silu'(h16))                      - the data gradient through drop1 / act into ln1
//   G16S_F32    fp32 output (+ bias)
//   G16S_RESDROP  fp32 output = residual + alpha * dropout(acc + bias)    - the block's `x = ffn(x) * 0.5 + x` (Gconform.py:57,60) with
//               conform_ffn's output dropout, folded into the second linear
// so the FFN's [M, 2048] intermediates exist ONLY as 16-bit arrays (h16 and a16 forward, dh16 backward) and no element-wise pass
// runs between the four GEMMs.
#include <cstdlib>

#include "internal.h"
#include "split.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
constexpr uint32_t kOob = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000);
}
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// two fp32 -> one dword of 16-bit values (element 0 in the low half)
template <bool BF16>
__device__ __forceinline__ uint32_t pack16(float a, float b) {
    const f32x2 v = {a, b};
    if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2_t));
}
template <bool BF16>
__device__ __forceinline__ float from16(uint32_t bits) {          // the low 16 bits of `bits`
    if constexpr (BF16) return __builtin_bit_cast(float, bits << 16);
    else return (float)__builtin_bit_cast(half_t, (uint16_t)bits);
}
__device__ __forceinline__ uint32_t swap_pair(uint32_t x) {        // value of lane ^ 1 (DPP quad permute [1, 0, 3, 2])
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
}

// Dropout bits: 32 pseudo-random bits per PAIR of rows (2 q, 2 q + 1) of one column, 16 per element - a pure function of
// (key, row, column), so the backward epilogue regenerates the forward's mask.  Two-multiply integer finaliser with the second key
// word injected between the rounds (restated in numpy by some_amd/training/dropout_bits.py for the keep-rate / independence checks).
__device__ __forceinline__ uint32_t drop_bits(uint32_t idx, uint32_t k0, uint32_t k1) {
    uint32_t x = idx + k0;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= k1;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

struct G16sArgs {
    const char* A; const char* B; const float* bias; char* C; const char* H;
    int M, N, K;
    int lda, ldb, ldc, ldh;      // in elements of the respective arrays
    int n_tiles;
    uint32_t plane;              // G16S_FFN1: byte distance from the h16 plane to the a16 plane
    uint32_t thr; float keep;    // dropout: an element is zeroed when its 16 random bits < thr, kept values are scaled by `keep`
    uint32_t k0, k1;
    float alpha;                 // G16S_RESDROP
};
enum { G16S_F32 = 0, G16S_FFN1 = 1, G16S_DSILU = 2, G16S_RESDROP = 3 };

// ---- epilogues.  C/D layout of v_mfma_f32_32x32x16: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
// 16-bit outputs leave as DWORDS: lanes 2 j and 2 j + 1 (adjacent columns) exchange halves by DPP, after which the even lane holds
// the pair (n, n + 1) of one array / row and the odd lane the pair of ANOTHER array / row - one 4-byte store per lane and register,
// 64 contiguous bytes per row and half-wave.
template <int EPI, bool BF16, bool FULL, int TM, int TN>
__device__ __forceinline__ void epilogue16s(const G16sArgs& a, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    const bool odd = (lane & 1) != 0;
    auto rk = [](int r) { return (uint32_t)((r & 3) + 8 * (r >> 2)); };
    if constexpr (EPI == G16S_F32) {
        const uint32_t row_c = (uint32_t)a.ldc * 4u;
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.C, (size_t)a.M * row_c);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const bool nv = FULL || n < a.N;
            const float bias = (a.bias != nullptr && nv) ? a.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const uint32_t row0 = (uint32_t)(m0 + (wm * TM + i) * 32 + 4 * hi);
                const uint32_t vc = nv ? row0 * row_c + (uint32_t)n * 4u : kOob;
#pragma unroll
                for (int r = 0; r < 16; ++r)      // rows >= M of a partial row tile are masked explicitly (not left to the descriptor's range check)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, acc[i][jn][r] + bias), rc,
                                                          (FULL || row0 + rk(r) < (uint32_t)a.M) ? vc : kOob, rk(r) * row_c, 0);
            }
        }
    } else if constexpr (EPI == G16S_RESDROP) {
        const uint32_t row_c = (uint32_t)a.ldc * 4u, row_h = (uint32_t)a.ldh * 4u;
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.C, (size_t)a.M * row_c);
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(a.H, (size_t)a.M * row_h);
        const float ak = a.alpha * a.keep;
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const bool nv = FULL || n < a.N;
            const float bias = (a.bias != nullptr && nv) ? a.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row0 = m0 + (wm * TM + i) * 32 + 4 * hi;
                const uint32_t vc = nv ? (uint32_t)row0 * row_c + (uint32_t)n * 4u : kOob;
                const uint32_t vr = nv ? (uint32_t)row0 * row_h + (uint32_t)n * 4u : kOob;
                const uint32_t cell0 = ((uint32_t)row0 >> 1) * (uint32_t)a.N + (uint32_t)n;
                float res[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (FULL || row0 + (int)rk(r) < a.M) ? vr : kOob, rk(r) * row_h, 0));
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    const uint32_t bits = drop_bits(cell0 + (rk(2 * rp) >> 1) * (uint32_t)a.N, a.k0, a.k1);   // rows (m, m + 1), m even
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 2 * rp + e;
                        const float k = (e ? bits >> 16 : bits & 0xffffu) >= a.thr ? ak : 0.f;
                        const float v = __fadd_rn(res[r], __fmul_rn(k, acc[i][jn][r] + bias));       // alpha * dropout(y) + x, no contraction
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rc, (FULL || row0 + (int)rk(r) < a.M) ? vc : kOob, rk(r) * row_c, 0);
                    }
                }
            }
        }
    } else if constexpr (EPI == G16S_FFN1) {
        const uint32_t row_c = (uint32_t)a.ldc * 2u;
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.C, (size_t)a.plane + (size_t)a.M * row_c);
        const uint32_t sel = odd ? 0x03020706u : 0x05040100u;        // v_perm_b32 (other, mine): even lanes low halves, odd lanes high halves
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const bool nv = FULL || n < a.N;
            const float bias = nv ? a.bias[n] : 0.f;
            const uint32_t col = odd ? a.plane + (uint32_t)(n - 1) * 2u : (uint32_t)n * 2u;      // even lanes: h16, odd lanes: a16
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row0 = m0 + (wm * TM + i) * 32 + 4 * hi;
                const uint32_t vbase = nv ? (uint32_t)row0 * row_c + col : kOob;
                const uint32_t cell0 = ((uint32_t)row0 >> 1) * (uint32_t)a.N + (uint32_t)n;
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    const uint32_t bits = drop_bits(cell0 + (rk(2 * rp) >> 1) * (uint32_t)a.N, a.k0, a.k1);   // rows (m, m + 1), m even
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float v = acc[i][jn][2 * rp + e] + bias;
                        const float h = from16<BF16>(pack16<BF16>(v, 0.f));                       // silu on the STORED (rounded) value
                        const float k = (e ? bits >> 16 : bits & 0xffffu) >= a.thr ? a.keep : 0.f;
                        const uint32_t mine = pack16<BF16>(v, h * sigmoidf_(h) * k);             // h16 | a16 << 16
                        const uint32_t word = __builtin_amdgcn_perm(swap_pair(mine), mine, sel);
                        const uint32_t vc = (FULL || row0 + (int)rk(2 * rp + e) < a.M) ? vbase : kOob;
                        __builtin_amdgcn_raw_buffer_store_b32(word, rc, vc, rk(2 * rp + e) * row_c, 0);
                    }
                }
            }
        }
    } else {
        const uint32_t row_c = (uint32_t)a.ldc * 2u, row_h = (uint32_t)a.ldh * 2u;
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.C, (size_t)a.M * row_c);
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.H, (size_t)a.M * row_h);
        const uint32_t sel = odd ? 0x03020706u : 0x05040100u;
        // even lanes own row m of the column pair (n, n + 1), odd lanes row m + 1 of (n - 1, n), for the load and for the store.
        // All 64 h16 dwords of the lane are requested before the first is used: one memory round trip for the epilogue, not 64.
        uint32_t hq[TN][TM][8];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const bool nv = FULL || n < a.N;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row0 = m0 + (wm * TM + i) * 32 + 4 * hi + (odd ? 1 : 0);
                const uint32_t hbase = nv ? (uint32_t)row0 * row_h + (uint32_t)(odd ? n - 1 : n) * 2u : kOob;
#pragma unroll
                for (int rp = 0; rp < 8; ++rp)
                    hq[jn][i][rp] = __builtin_amdgcn_raw_buffer_load_b32(rh, (FULL || row0 + (int)rk(2 * rp) < a.M) ? hbase : kOob, rk(2 * rp) * row_h, 0);
            }
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const bool nv = FULL || n < a.N;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row0 = m0 + (wm * TM + i) * 32 + 4 * hi;
                const int rowl = row0 + (odd ? 1 : 0);
                const uint32_t cbase = nv ? (uint32_t)rowl * row_c + (uint32_t)(odd ? n - 1 : n) * 2u : kOob;
                const uint32_t cell0 = ((uint32_t)row0 >> 1) * (uint32_t)a.N + (uint32_t)n;
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    const uint32_t L = hq[jn][i][rp];
                    const uint32_t hh = __builtin_amdgcn_perm(swap_pair(L), L, sel);             // h16[m][n] | h16[m + 1][n] << 16
                    const float h0 = from16<BF16>(hh), h1 = from16<BF16>(hh >> 16);
                    const uint32_t bits = drop_bits(cell0 + (rk(2 * rp) >> 1) * (uint32_t)a.N, a.k0, a.k1);
                    const float s0 = sigmoidf_(h0), s1 = sigmoidf_(h1);
                    const float k0 = (bits & 0xffffu) >= a.thr ? a.keep : 0.f, k1 = (bits >> 16) >= a.thr ? a.keep : 0.f;
                    const float v0 = acc[i][jn][2 * rp] * k0 * s0 * (1.f + h0 * (1.f - s0));
                    const float v1 = acc[i][jn][2 * rp + 1] * k1 * s1 * (1.f + h1 * (1.f - s1));
                    const uint32_t mine = pack16<BF16>(v0, v1);
                    const uint32_t word = __builtin_amdgcn_perm(swap_pair(mine), mine, sel);
                    const uint32_t vc = (FULL || rowl + (int)rk(2 * rp) < a.M) ? cbase : kOob;
                    __builtin_amdgcn_raw_buffer_store_b32(word, rc, vc, rk(2 * rp) * row_c, 0);
                }
            }
        }
    }
}

// 16 bytes per lane global -> LDS.  A plain function on purpose: with the builtin written inside the kernel TEMPLATE (dependent context),
// hipcc (ROCm 7.2) silently drops the kernel's host stub and the library fails to load with an undefined kernel symbol.
__device__ __forceinline__ void dma16(const char* src, char* lds_dst) {
    typedef __attribute__((address_space(3))) void lds_void;
    __builtin_amdgcn_global_load_lds(src, (lds_void*)(uintptr_t)lds_dst, 16, 0, 0);
}

// ---- the kernel: (64 WM) x (64 TN) tile, 2 WM waves (WM x 2, each 64 x 32 TN), NSTAGE-deep DMA ring of 32-element k-steps ----
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI, bool BF16, int WM, int TN, int NSTAGE, int MINB>
__global__ __launch_bounds__(128 * WM, MINB) void gemm16s_kernel(G16sArgs a) {
    constexpr int WAVES_N = 2, TM = 2;
    constexpr int NT = 128 * WM, BM = 64 * WM, BN = 64 * TN;
    constexpr int ROWS = BM + BN;                       // rows x 64 B per stage
    constexpr int STAGE_B = ROWS * 64;
    constexpr int AHEAD = NSTAGE - 1;                   // k-steps in flight beyond the one being multiplied
    constexpr int NDMA = ROWS * 4 / NT;                 // chunks of 16 B per thread and k-step
    static_assert(ROWS * 4 % NT == 0 && AHEAD >= 2 && AHEAD <= 4 && AHEAD * NDMA < 64, "ring shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* lbase = reinterpret_cast<char*>(lds);

    const int n_tiles = a.n_tiles;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int m_tile = (j / n_tiles) * 8 + xcd;
    const int n_tile = j % n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (m0 >= a.M || n0 >= a.N) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, kg = lane >> 5;
    const int ns = a.K >> 5;                            // k-steps of 32

    // DMA roles: pass p moves linear chunk L = p * NT + tid -> LDS byte 16 L = row (L >> 2), slot (L & 3); slot s of row r holds
    // the logical chunk s ^ ((r >> 2) & 3) (8 k values each) - applied to the per-lane SOURCE address (the DMA destination is
    // lane-linear) and again on the fragment reads: every ds_read_b128 is conflict-free.  Rows past M / N are clamped to the last
    // valid row (their products are never stored).
    const char* src[NDMA];
#pragma unroll
    for (int p = 0; p < NDMA; ++p) {
        const int L = p * NT + tid;
        const int row = L >> 2, slot = L & 3;
        const int c = slot ^ ((row >> 2) & 3);
        if (row < BM) src[p] = a.A + (size_t)min(m0 + row, a.M - 1) * a.lda * 2 + c * 16;
        else src[p] = a.B + (size_t)min(n0 + row - BM, a.N - 1) * a.ldb * 2 + c * 16;
    }
    auto dma = [&](int t) {                             // k-step t -> ring slot t % NSTAGE
        char* stage = lbase + (t % NSTAGE) * STAGE_B;
#pragma unroll
        for (int p = 0; p < NDMA; ++p) dma16(src[p] + t * 64, stage + (p * NT + wave * 64) * 16);
    };
    // wait until at most `stages` k-steps (NDMA instructions each, issued in order) are still in flight
    auto wait_all_but = [&](int stages) {
        if (stages <= 0) wait_vm<0>();
        else if (stages == 1) wait_vm<NDMA>();
        else if (stages == 2) wait_vm<2 * NDMA>();
        else wait_vm<3 * NDMA>();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    // fragment addressing: row = tile base (multiple of 32) + l31  ->  swz = (l31 >> 2) & 3 for every tile
    const int swz = (l31 >> 2) & 3;
    const int off0 = (kg ^ swz) * 16, off1 = ((2 + kg) ^ swz) * 16;      // k 8 kg .. + 7 of the first / second 16-element slab
    const int a_row = (wm * TM * 32 + l31) * 64;
    const int w_row = (BM + wn * TN * 32 + l31) * 64;

#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
        if (t < ns) dma(t);
    wait_all_but(min(AHEAD, ns) - 1);
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ns; ++t) {
        if (t + AHEAD < ns) dma(t + AHEAD);             // its slot was last read in step t - 1, before the barrier
        const char* st = lbase + (t % NSTAGE) * STAGE_B;
        half8 a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            a0[i] = *reinterpret_cast<const half8*>(st + a_row + i * 32 * 64 + off0);
            a1[i] = *reinterpret_cast<const half8*>(st + a_row + i * 32 * 64 + off1);
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            b0[jn] = *reinterpret_cast<const half8*>(st + w_row + jn * 32 * 64 + off0);
            b1[jn] = *reinterpret_cast<const half8*>(st + w_row + jn * 32 * 64 + off1);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_hi<BF16>(a0[i], b0[jn], acc[i][jn]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_hi<BF16>(a1[i], b1[jn], acc[i][jn]);
        // step t + 1 must have landed (this thread's part); the later steps may stay in flight
        wait_all_but(min(AHEAD - 1, ns - t - 2));
        __builtin_amdgcn_s_barrier();
    }

    if (m0 + BM <= a.M && n0 + BN <= a.N) epilogue16s<EPI, BF16, true, TM, TN>(a, acc, m0, n0, wm, wn, lane);
    else epilogue16s<EPI, BF16, false, TM, TN>(a, acc, m0, n0, wm, wn, lane);
}

template <int EPI, bool BF16, int WM = 2, int TN = 4, int NSTAGE = 3, int MINB = 2>
hipError_t launch16s(const G16sArgs& a_in, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * TN;
    constexpr size_t LDS_BYTES = NSTAGE * (size_t)(BM + BN) * 64;
    static DeviceOnce attr_once;
    auto kern = &gemm16s_kernel<EPI, BF16, WM, TN, NSTAGE, MINB>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    G16sArgs a = a_in;
    const int m_tiles = (a.M + BM - 1) / BM;
    a.n_tiles = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)((m_tiles + 7) / 8 * 8 * a.n_tiles)), dim3(128 * WM), LDS_BYTES, s, a);
    return hipGetLastError();
}

// ---- fp32 -> 16-bit images ---------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void cast16_kernel(const float* __restrict__ x, uint32_t* __restrict__ y, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // eight elements per thread
    if (i >= n8) return;
    const f32x4 u = *reinterpret_cast<const f32x4*>(x + i * 8), v = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
    const u32x4 o = {pack16<BF16>(u[0], u[1]), pack16<BF16>(u[2], u[3]), pack16<BF16>(v[0], v[1]), pack16<BF16>(v[2], v[3])};
    *reinterpret_cast<u32x4*>(y + i * 4) = o;
}

// y16 = rn16(silu(x)): the conv module's activation (modules/conv/base_conv.py:68) written as the 16-bit operand of pointwise_conv2's GEMM
template <bool BF16>
__global__ __launch_bounds__(256) void silu16_kernel(const float* __restrict__ x, uint32_t* __restrict__ y, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // eight elements per thread
    if (i >= n8) return;
    const f32x4 u = *reinterpret_cast<const f32x4*>(x + i * 8), v = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
    float o[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[k] = u[k] * sigmoidf_(u[k]); o[4 + k] = v[k] * sigmoidf_(v[k]); }
    const u32x4 w = {pack16<BF16>(o[0], o[1]), pack16<BF16>(o[2], o[3]), pack16<BF16>(o[4], o[5]), pack16<BF16>(o[6], o[7])};
    *reinterpret_cast<u32x4*>(y + i * 4) = w;
}

// y16[m][n] = rn16(alpha * keep(m, n) * d[m][n]) with the G16S_RESDROP mask: the gradient of `alpha * dropout(y) + x` w.r.t. y, written
// directly as the 16-bit operand of the data-gradient GEMM.  A thread owns rows (2 q, 2 q + 1) of four adjacent columns: one word of
// dropout bits per column serves both rows.
template <bool BF16>
__global__ __launch_bounds__(256) void dropcast16_kernel(const float* __restrict__ d, uint16_t* __restrict__ y, int M, int N, float ak, uint32_t thr,
                                                          uint32_t k0, uint32_t k1) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nq = N >> 2;
    const int q = (int)(t / nq), n = (int)(t % nq) * 4;
    const int m = 2 * q;
    if (m >= M) return;
    uint32_t bits[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bits[i] = drop_bits((uint32_t)q * (uint32_t)N + (uint32_t)(n + i), k0, k1);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (m + e >= M) break;
        const f32x4 v = *reinterpret_cast<const f32x4*>(d + (size_t)(m + e) * N + n);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = ((e ? bits[i] >> 16 : bits[i] & 0xffffu) >= thr ? ak : 0.f) * v[i];
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w = {pack16<BF16>(o[0], o[1]), pack16<BF16>(o[2], o[3])};
        *reinterpret_cast<u32x2*>(y + (size_t)(m + e) * N + n) = w;
    }
}

// W [N, K] fp32 -> W16 [N, K] and W16T [K, N]: 64 x 64 tiles through LDS (both outputs leave as 128-byte row segments).
// `table` != null: blockIdx.z picks one of many weights from a descriptor table (five int64 per weight: source, W16, W16T addresses, N, K) -
// all the model's 16-bit weight images in ONE launch per optimiser step instead of one launch per weight.
template <bool BF16>
__global__ __launch_bounds__(256) void transpose16_kernel(const float* __restrict__ w, uint16_t* __restrict__ w16, uint16_t* __restrict__ w16t,
                                                           int N, int K, const int64_t* __restrict__ table) {
    __shared__ uint16_t tile[64][66];
    if (table != nullptr) {
        const int64_t* e = table + (size_t)blockIdx.z * 5;
        w = reinterpret_cast<const float*>(e[0]);
        w16 = reinterpret_cast<uint16_t*>(e[1]);
        w16t = reinterpret_cast<uint16_t*>(e[2]);
        N = (int)e[3];
        K = (int)e[4];
        if ((int)blockIdx.y * 64 >= N || (int)blockIdx.x * 64 >= K) return;     // the grid covers the largest weight of the table
    }
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = ty + 4 * q;
        const int n = n0 + r, k = k0 + tx;
        uint16_t v = 0;
        if (n < N && k < K) {
            v = (uint16_t)pack16<BF16>(w[(size_t)n * K + k], 0.f);
            if (w16 != nullptr) w16[(size_t)n * K + k] = v;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (w16t == nullptr) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = ty + 4 * q;
        const int k = k0 + r, n = n0 + tx;
        if (k < K && n < N) w16t[(size_t)k * N + n] = tile[tx][r];
    }
}

}  // namespace

// dropout threshold / scale / key words of a call site (shared by the GEMM epilogues and dropcast16_kernel)
static void drop_params(float p, uint64_t seed, uint32_t& thr, float& keep, uint32_t& k0, uint32_t& k1) {
    // 16 random bits per element: dropped when < thr; the scale is 1 / (1 - thr / 65536), the rate actually applied
    const uint32_t t = p > 0.f ? (uint32_t)(p * 65536.0f + 0.5f) : 0u;
    thr = t > 65535u ? 65535u : t;
    keep = 65536.0f / (65536.0f - (float)thr);
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;                     // splitmix64 of the call site's seed -> the two key words
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k0 = (uint32_t)z; k1 = (uint32_t)(z >> 32);
}

hipError_t launch_dropcast16(const float* d, void* y16, int M, int N, float alpha, float p, uint64_t seed, int bf16, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (N & 3) return hipErrorInvalidValue;
    uint32_t thr, k0, k1; float keep;
    drop_params(p, seed, thr, keep, k0, k1);
    const int64_t threads = (int64_t)((M + 1) / 2) * (N / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (bf16) hipLaunchKernelGGL(dropcast16_kernel<true>, grid, dim3(256), 0, s, d, static_cast<uint16_t*>(y16), M, N, alpha * keep, thr, k0, k1);
    else hipLaunchKernelGGL(dropcast16_kernel<false>, grid, dim3(256), 0, s, d, static_cast<uint16_t*>(y16), M, N, alpha * keep, thr, k0, k1);
    return hipGetLastError();
}

hipError_t launch_gemm16s(int epi, const void* A16, int lda, const void* B16, int ldb, const float* bias, void* C, int ldc, const void* H16,
                          int ldh, size_t plane_bytes, int M, int N, int K, int bf16, float p, uint64_t seed, float alpha, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return hipSuccess;
    if ((K & 31) || (lda & 7) || (ldb & 7) || (N & 1) || epi < G16S_F32 || epi > G16S_RESDROP) return hipErrorInvalidValue;
    G16sArgs a{};
    a.A = static_cast<const char*>(A16); a.B = static_cast<const char*>(B16); a.bias = bias; a.C = static_cast<char*>(C);
    a.H = static_cast<const char*>(H16);
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldh = ldh; a.plane = (uint32_t)plane_bytes;
    drop_params(p, seed, a.thr, a.keep, a.k0, a.k1);
    a.alpha = alpha;
    // Tile by grid size (round 5): the 128 x 256 tile gives an N = 512 GEMM over the reference's batch shape (8 x ~520 frames,
    // configs/base.yaml:55-56) 33 x 2 = 66 workgroups on 256 CUs - a quarter of the chip, 25 - 35 us of which 4 us are arithmetic.  Below
    // ~3/4 of a round of big tiles the 64 x 128 tile (2 wavefronts, 36 KB of LDS, up to 4 workgroups per CU) takes over: 4 x the
    // workgroups, the same k order per element - results bit-identical (tests/test_gpu_train_ffn16.py).
    const char* force_env = getenv("SOME_AMD_G16S_TILE");                                          // A/B and the bit-identity test: 0 big, 1 small
    const int force = force_env ? atoi(force_env) : -1;
    const long big_wgs = (long)((M + 127) / 128) * ((N + 255) / 256);
    const bool small = force >= 0 ? force == 1 : big_wgs < 192;
#define G16S_CASE(E) case E: return small ? (bf16 ? launch16s<E, true, 1, 2, 3, 4>(a, s) : launch16s<E, false, 1, 2, 3, 4>(a, s)) \
                                          : (bf16 ? launch16s<E, true>(a, s) : launch16s<E, false>(a, s));
    switch (epi) {
        G16S_CASE(G16S_F32) G16S_CASE(G16S_FFN1) G16S_CASE(G16S_DSILU) G16S_CASE(G16S_RESDROP)
    }
#undef G16S_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_silu16(const float* x, void* y16, int64_t n, int bf16, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n & 7) return hipErrorInvalidValue;
    const int64_t n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256));
    if (bf16) hipLaunchKernelGGL(silu16_kernel<true>, grid, dim3(256), 0, s, x, static_cast<uint32_t*>(y16), n8);
    else hipLaunchKernelGGL(silu16_kernel<false>, grid, dim3(256), 0, s, x, static_cast<uint32_t*>(y16), n8);
    return hipGetLastError();
}

hipError_t launch_cast16(const float* x, void* y16, int64_t n, int bf16, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n & 7) return hipErrorInvalidValue;
    const int64_t n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256));
    if (bf16) hipLaunchKernelGGL(cast16_kernel<true>, grid, dim3(256), 0, s, x, static_cast<uint32_t*>(y16), n8);
    else hipLaunchKernelGGL(cast16_kernel<false>, grid, dim3(256), 0, s, x, static_cast<uint32_t*>(y16), n8);
    return hipGetLastError();
}

hipError_t launch_transpose16(const float* w, void* w16, void* w16t, int N, int K, int bf16, hipStream_t s) {
    if (N <= 0 || K <= 0) return hipSuccess;
    const dim3 grid((unsigned)((K + 63) / 64), (unsigned)((N + 63) / 64));
    if (bf16) hipLaunchKernelGGL(transpose16_kernel<true>, grid, dim3(256), 0, s, w, static_cast<uint16_t*>(w16), static_cast<uint16_t*>(w16t), N, K, nullptr);
    else hipLaunchKernelGGL(transpose16_kernel<false>, grid, dim3(256), 0, s, w, static_cast<uint16_t*>(w16), static_cast<uint16_t*>(w16t), N, K, nullptr);
    return hipGetLastError();
}

hipError_t launch_transpose16_table(const int64_t* table_dev, int n, int max_n, int max_k, int bf16, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((max_k + 63) / 64), (unsigned)((max_n + 63) / 64), (unsigned)n);
    if (bf16) hipLaunchKernelGGL(transpose16_kernel<true>, grid, dim3(256), 0, s, nullptr, nullptr, nullptr, 0, 0, table_dev);
    else hipLaunchKernelGGL(transpose16_kernel<false>, grid, dim3(256), 0, s, nullptr, nullptr, nullptr, 0, 0, table_dev);
    return hipGetLastError();
}
