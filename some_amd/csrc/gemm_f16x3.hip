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
#include <atomic>
#include <type_traits>

#include "internal.h"
#include "split.h"

#ifdef GEMM_TIMELINE
__device__ unsigned long long* g_gemm_tl;      // [workgroup][wave][8] (tools/gemm_probe.hip)
__device__ int g_gemm_tl_cap;
__device__ unsigned long long* g_gemm_it;      // [wave][1024]: s_memtime behind every k-block barrier of the workgroups with blockIdx.x == 8
__device__ int g_gemm_it_n[8];
#define GEMM_IT_STAMP() do { if (g_gemm_it != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && lane == 0) { \
        const int i_ = g_gemm_it_n[wave]; if (i_ < 1024) { g_gemm_it[wave * 1024 + i_] = __builtin_amdgcn_s_memtime(); g_gemm_it_n[wave] = i_ + 1; } } } while (0)
#else
#define GEMM_IT_STAMP() do {} while (0)
#endif

namespace {

constexpr int LDT = 36;      // LDS row in dwords: 16 (32 hi halves) + 16 (32 lo halves) + 4 pad

// 1 / (1 + e^-x) with the hardware reciprocal (v_rcp_f32, 1 ulp): the IEEE divide costs ~10 VALU per element
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Epilogue memory traffic goes through buffer descriptors (T8): one 32-bit VGPR offset per MFMA tile + a scalar
// row offset per element instead of a 64-bit VGPR address each (the accumulators already hold 128 VGPRs), and
// hardware bounds checking instead of exec masking - rows >= M lie past num_records (loads return 0, stores are
// dropped), out-of-range columns are pushed there explicitly.
constexpr uint32_t kOob = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000);
}
__device__ __forceinline__ float ld32(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void st32(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, uint32_t bits) {
    __builtin_amdgcn_raw_buffer_store_b32(bits, r, voff, soff, 0);
}
__device__ __forceinline__ uint32_t pack_split_pair(float v, int lane) {
    // SPLIT32 output helper: lane pairs exchange halves so each lane stores one packed dword
    // (even lane -> two hi halves, odd lane -> two lo halves)
    half_t h, l;
    split_f16(v, h, l);
    const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
    // lane ^ 1 exchange as a DPP quad permute [1,0,3,2] (no LDS round trip, unlike __shfl_xor's ds_bpermute)
    const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);
    return (lane & 1) ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
}

// PFMAX: residual look-ahead in tiles of 16 values per lane (the persistent kernel, whose staging registers stay live across the
// epilogue, takes 2)
template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool OUT_SPLIT, bool FULL, int PFMAX = 3>
__device__ __forceinline__ void epilogue(const GemmArgs& a, const GemmGroup& g, f32x16 (&acc)[TM][TN], int m0, int n0,
                                         int wm, int wn, int lane, char* wave_lds = nullptr) {
    const int l31 = lane & 31, hi = lane >> 5;
    const uint32_t row_c = (uint32_t)a.ldc * 4u, row_r = (uint32_t)a.ldr * 4u;      // row pitches in bytes
    auto rk = [](int r) { return (uint32_t)((r & 3) + 8 * (r >> 2)); };             // row of element r within its tile
    auto row0 = [&](int i) { return m0 + (wm * TM + i) * 32 + 4 * hi; };
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(g.C, (size_t)a.M * row_c);
    if constexpr (EPI == EPI_GLU || EPI == EPI_GLU_RES) {
        static_assert(TN % 2 == 0, "GLU pairs adjacent 32-column tiles");
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(g.res, (size_t)a.M * row_r);
        // residual rows are fetched kPrefetch tiles ahead of the tile being finished (C may alias res, so the compiler
        // keeps every load behind the earlier tiles' stores: without the explicit look-ahead each tile pays a full
        // memory round trip)
        constexpr int NTILE = (TN / 2) * TM, PF = NTILE < PFMAX ? NTILE : PFMAX;
        auto tile_np = [&](int t) { return n0 + (wn * TN + 2 * (t / TM)) * 32 + l31; };
        auto load_res = [&](int t, float (&dst)[16]) {
            const int np = tile_np(t);
            const bool nv = FULL || np < g.N;
            const int oc = (np - l31) / 2 + l31;
            const uint32_t vr = nv ? (uint32_t)row0(t % TM) * row_r + (uint32_t)oc * 4u : kOob;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = ld32(rres, vr, rk(r) * row_r);
        };
        float rq[PF][16];
        if constexpr (EPI == EPI_GLU_RES) {
#pragma unroll
            for (int t = 0; t < PF; ++t) load_res(t, rq[t]);
        }
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            const int jp = t / TM, i = t % TM;
            const int np = tile_np(t);                                 // packed column of the `a` half
            const bool nv = FULL || np < g.N;
            const int oc = (np - l31) / 2 + l31;                       // packed 64-block -> 32 outputs
            const float ba = nv ? g.bias[np] : 0.f, bg = nv ? g.bias[np + 32] : 0.f;
            const uint32_t vc = nv ? (uint32_t)row0(i) * row_c + (uint32_t)oc * 4u : kOob;
            float rr[16];
            if constexpr (EPI == EPI_GLU_RES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rr[r] = rq[t % PF][r];
                if (t + PF < NTILE) load_res(t + PF, rq[t % PF]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (acc[i][2 * jp][r] + ba) * sigmoidf_(acc[i][2 * jp + 1][r] + bg);
                if constexpr (EPI == EPI_GLU_RES) {
                    v += rr[r];
                    if (g.mask != nullptr) {
                        const int m = row0(i) + (int)rk(r);
                        if (m < a.M && g.mask[m] == 0) v = 0.f;
                    }
                }
                st32(rc, vc, rk(r) * row_c, __builtin_bit_cast(uint32_t, v));
            }
        }
    } else if constexpr (EPI == EPI_QKV) {
        // N = 1536: columns [0,512) -> Q plane, [512,1024) -> K plane (both SPLIT32 rows of 512),
        // [1024,1536) -> V^T f16 planes with the frame index contiguous (what attention_f16x3.hip stages)
        const __amdgpu_buffer_rsrc_t rq = make_rsrc(g.C, (size_t)a.M * 2048);
        const __amdgpu_buffer_rsrc_t rkp = make_rsrc(g.C2, (size_t)a.M * 2048);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const int region = n >> 9;                                  // uniform per 32-column tile
            if (region < 2) {
                const int nn = n & 511;
                const uint32_t lane_off = (uint32_t)((nn - l31) >> 5) * 128u + ((lane & 1) ? 64u + (uint32_t)(l31 - 1) * 2u : (uint32_t)l31 * 2u);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const uint32_t vq = (uint32_t)row0(i) * 2048u + lane_off;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t word = pack_split_pair(acc[i][jn][r], lane);
                        st32(region == 0 ? rq : rkp, vq, rk(r) * 2048u, word);
                    }
                }
            } else {
                // V^T planes [d][frame] (f16 hi, then lo): a lane owns column d and 4-frame runs, so direct stores would be
                // 8-byte pieces of 32 different rows per instruction.  The 32 x (32 TM) tile goes through a wave-private
                // LDS patch instead ([plane][32 d][32 TM frames], rows padded to kVtRow bytes) and leaves as 16-byte
                // stores in which TM * 4 neighbouring lanes cover a whole row segment (64 TM contiguous bytes).
                constexpr int kVtRow = TM * 64 + 16;
                char* patch = wave_lds;                                  // [2][32][kVtRow]
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int rq4 = 0; rq4 < 4; ++rq4) {
                        half4 hh, ll;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { half_t h, l; split_f16(acc[i][jn][4 * rq4 + e], h, l); hh[e] = h; ll[e] = l; }
                        // frames are stored PERMUTED inside every aligned group of 16: quarters in the order 0, 2, 1, 3, so
                        // that the 8 keys an attention lane needs for one MFMA (frames 4 kg + {0..3} and 8 + 4 kg + {0..3})
                        // are 16 contiguous bytes - one conflict-free ds_read_b128 instead of two 2-way-conflicting b64
                        const int fq = 2 * rq4 + hi;                      // quarter index (4 frames) within the 32-frame tile
                        const int f = i * 32 + (fq & 4) * 4 + 4 * (((fq & 1) << 1) | ((fq >> 1) & 1));
                        *reinterpret_cast<half4*>(patch + l31 * kVtRow + f * 2) = hh;
                        *reinterpret_cast<half4*>(patch + 32 * kVtRow + l31 * kVtRow + f * 2) = ll;
                    }
                __builtin_amdgcn_wave_barrier();
                constexpr int LPR = TM * 4;                              // lanes per row (16 bytes = 8 frames each)
                constexpr int RPI = 64 / LPR;                            // rows per instruction
                const int d0 = n0 + (wn * TN + jn) * 32 - 1024;
                const int mrow = m0 + wm * TM * 32 + (lane % LPR) * 8;
#pragma unroll
                for (int pass = 0; pass < 32 / RPI; ++pass) {
                    const int dl = pass * RPI + lane / LPR;
                    const f32x4 vh4 = *reinterpret_cast<const f32x4*>(patch + dl * kVtRow + (lane % LPR) * 16);
                    const f32x4 vl4 = *reinterpret_cast<const f32x4*>(patch + 32 * kVtRow + dl * kVtRow + (lane % LPR) * 16);
                    if (mrow < g.ldv) {   // rows >= M carry exact zeros (zero-filled A rows, no bias): finite padding
                        char* vh = reinterpret_cast<char*>(g.C3) + ((size_t)(d0 + dl) * g.ldv + mrow) * 2;
                        *reinterpret_cast<f32x4*>(vh) = vh4;
                        *reinterpret_cast<f32x4*>(vh + (size_t)kDim * g.ldv * 2) = vl4;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(EPI == EPI_BIAS_RES ? (const void*)g.res : (const void*)g.C, (size_t)a.M * row_r);
        constexpr int NTILE = TN * TM, PF = NTILE < PFMAX ? NTILE : PFMAX;      // residual look-ahead, see the GLU path
        auto load_res = [&](int t, float (&dst)[16]) {
            const int n = n0 + (wn * TN + t / TM) * 32 + l31;
            const bool nv = FULL || n < g.N;
            const uint32_t vr = nv ? (uint32_t)row0(t % TM) * row_r + (uint32_t)n * 4u : kOob;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = ld32(rres, vr, rk(r) * row_r);
        };
        float rq[PF][16];
        if constexpr (EPI == EPI_BIAS_RES) {
#pragma unroll
            for (int t = 0; t < PF; ++t) load_res(t, rq[t]);
        }
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            const int jn = t / TM, i = t % TM;
            const int n = n0 + (wn * TN + jn) * 32 + l31;
            const bool nv = FULL || n < g.N;
            float bias = 0.f;
            if constexpr (EPI != EPI_NONE) bias = nv ? g.bias[n] : 0.f;
            const uint32_t split_off = (uint32_t)((n - l31) >> 5) * 128u + ((lane & 1) ? 64u + (uint32_t)(l31 - 1) * 2u : (uint32_t)l31 * 2u);
            const uint32_t vc = nv ? (uint32_t)row0(i) * row_c + (OUT_SPLIT ? split_off : (uint32_t)n * 4u) : kOob;
            float rr[16];
            if constexpr (EPI == EPI_BIAS_RES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rr[r] = rq[t % PF][r];
                if (t + PF < NTILE) load_res(t + PF, rq[t % PF]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][jn][r] + bias;
                if constexpr (EPI == EPI_BIAS) {
                    if (g.act == 1) v = sigmoidf_(v);
                    if (g.mask != nullptr) {
                        const int m = row0(i) + (int)rk(r);
                        if (m < a.M && g.mask[m] == 0) v = 0.f;
                    }
                } else if constexpr (EPI == EPI_BIAS_SILU) {
                    v = v * sigmoidf_(v);
                } else if constexpr (EPI == EPI_BIAS_RES) {
                    v = rr[r] + a.alpha * v;
                }
                if constexpr (OUT_SPLIT) st32(rc, vc, rk(r) * row_c, pack_split_pair(v, lane));
                else st32(rc, vc, rk(r) * row_c, __builtin_bit_cast(uint32_t, v));
            }
        }
    }
}

// ---- transposed accumulator layout (TR) ------------------------------------------------------------------------------
// acc = mfma(W fragment, A fragment): D rows <- W rows (n), D columns <- activation rows (m).  A lane then owns ONE
// output row m (= lane & 31) and, per 32-column MFMA tile, 16 CONSECUTIVE columns: the W fragment of D-row position
// p = 4h + 8q + e is read from W row pi(p) = 16h + 4q + e (conflict-free like the identity: the rows of one ds_read_b128
// lane group stay distinct mod 16), so accumulator register r of lane-half h is column 16h + r.  The epilogue therefore
// moves 64 contiguous bytes per lane and tile with 16-byte buffer instructions (a quarter of the memory instructions of
// the column-per-lane layout, no DPP exchange for SPLIT32 output, per-row quantities - residual row, mask, LayerNorm
// statistics - are per-lane scalars), cf. cdna guide T21.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int pi32(int p) { return ((p >> 2) & 1) * 16 + (p >> 3) * 4 + (p & 3); }
__device__ __forceinline__ void st128(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
__device__ __forceinline__ void st128h(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, half8 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// One row-tile x column-tile of results (16 values of one row) -> memory.  fp32: 4 x 16 B at byte 64 h of the tile's
// 128-byte row segment; SPLIT32: hi halves 32 B at 32 h, lo halves 32 B at 64 + 32 h.
template <bool OUT_SPLIT>
__device__ __forceinline__ void store16(__amdgpu_buffer_rsrc_t rc, uint32_t voff, uint32_t soff, const float (&v)[16]) {
    if constexpr (OUT_SPLIT) {
        half8 h0, h1, l0, l1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            half_t h, l;
            split_f16(v[e], h, l); h0[e] = h; l0[e] = l;
            split_f16(v[8 + e], h, l); h1[e] = h; l1[e] = l;
        }
        st128h(rc, voff, soff, h0);
        st128h(rc, voff, soff + 16, h1);
        st128h(rc, voff, soff + 64, l0);
        st128h(rc, voff, soff + 80, l1);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            st128(rc, voff, soff + 16 * q, o);
        }
    }
}

// Row-per-lane epilogue with WHOLE-LINE stores (round 6, persistent kernel only).  epilogue_tr stores 16 bytes per lane into 64 DIFFERENT
// 128-byte lines per instruction (a lane owns a row); the timeline (tools/gemm_probe.hip) shows the FFN1 epilogue running at the rate the
// CU's store path accepts such pieces (25 GB/s per CU).  Here a 32 x 32 tile of SPLIT32 results (32 rows x 128 bytes) goes through a
// wave-private 4 KiB LDS patch - written as the lane holds it (row l31, pieces 2 h, 2 h + 1, 4 + 2 h, 5 + 2 h of the row's eight
// 16-byte pieces; piece p of row r at slot p ^ (r & 7): conflict-free reads, two-way writes) - and read back eight lanes to a row, so
// that every store instruction writes 8 complete lines.  Same values: bit-identical output.
template <int WAVES_M, int WAVES_N, int TM, int TN, bool FULL>
__device__ __forceinline__ void epilogue_tr_lines(const GemmArgs& a, const GemmGroup& g, f32x16 (&acc)[TM][TN], int m0, int n0,
                                                  int wm, int wn, int lane, char* patch) {
    const int l31 = lane & 31, hi = lane >> 5;
    const uint32_t row_c = (uint32_t)a.ldc * 4u;
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(g.C, (size_t)a.M * row_c);
    const int rrow = lane >> 3, rslot = lane & 7;                    // read side: row rrow + 8 k, slot rslot
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int nb = n0 + (wn * TN + jn) * 32;
        if (!FULL && nb >= g.N) continue;
        float bias[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(g.bias + nb + 16 * hi + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * q + e] = t[e];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            half8 h0, h1, l0, l1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x0 = acc[i][jn][e] + bias[e], x1 = acc[i][jn][8 + e] + bias[8 + e];
                half_t h, l;
                split_f16(x0 * sigmoidf_(x0), h, l); h0[e] = h; l0[e] = l;
                split_f16(x1 * sigmoidf_(x1), h, l); h1[e] = h; l1[e] = l;
            }
            char* wrow = patch + l31 * 128;
            const int sw = l31 & 7;
            *reinterpret_cast<half8*>(wrow + (((2 * hi) ^ sw) << 4)) = h0;
            *reinterpret_cast<half8*>(wrow + (((2 * hi + 1) ^ sw) << 4)) = h1;
            *reinterpret_cast<half8*>(wrow + (((4 + 2 * hi) ^ sw) << 4)) = l0;
            *reinterpret_cast<half8*>(wrow + (((5 + 2 * hi) ^ sw) << 4)) = l1;
            __builtin_amdgcn_wave_barrier();
            const int mbase = m0 + (wm * TM + i) * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rrow + 8 * k;
                const half8 v = *reinterpret_cast<const half8*>(patch + r * 128 + (rslot << 4));
                const int piece = rslot ^ (r & 7);
                st128h(rc, (uint32_t)(mbase + r) * row_c + (uint32_t)piece * 16u, (uint32_t)nb * 4u, v);      // rows >= M: past num_records, dropped
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool OUT_SPLIT, bool FULL>
__device__ __forceinline__ void epilogue_tr(const GemmArgs& a, const GemmGroup& g, f32x16 (&acc)[TM][TN], int m0, int n0,
                                            int wm, int wn, int lane) {
    // Measured (32 x 30 s, round 2): row-per-lane 16-byte accesses win where the output is SPLIT32 (FFN1: -1.7 %, no DPP
    // exchange, 4 instead of 16 stores per tile) and LOSE 5-20 % for fp32 outputs / residual reads, whose 16-byte pieces
    // are partial 64-byte lines (the column-per-lane dword stores are whole 128-byte rows) - so only the SiLU epilogue
    // uses this layout.
    static_assert(EPI == EPI_BIAS_SILU, "the other epilogues keep the column-per-lane layout");
    const int l31 = lane & 31, hi = lane >> 5;
    const uint32_t row_c = (uint32_t)a.ldc * 4u;
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(g.C, (size_t)a.M * row_c);
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int nb = n0 + (wn * TN + jn) * 32;
        if (!FULL && nb >= g.N) continue;
        float bias[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(g.bias + nb + 16 * hi + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * q + e] = t[e];
        }
        const uint32_t soff = (uint32_t)nb * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 32 + l31;      // rows >= M lie past num_records: dropped by the hardware
#ifdef GEMM_TR_MUL24             // diagnostic: 24-bit multiply (v_mad_u32_u24: no SGPR carry-out) instead of v_mad_u64_u32
            const uint32_t vc = __umul24((uint32_t)m, row_c) + (uint32_t)hi * (OUT_SPLIT ? 32u : 64u);
#else
            const uint32_t vc = (uint32_t)m * row_c + (uint32_t)hi * (OUT_SPLIT ? 32u : 64u);
#endif
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = acc[i][jn][r] + bias[r];
                v[r] = x * sigmoidf_(x);
            }
            store16<OUT_SPLIT>(rc, vc, soff, v);
        }
    }
}

// TERMS = 3: x = hi + lo on both operands, three products (fp32-equivalent).  TERMS = 1: hi halves only - plain f16 operands
// with fp32 accumulation on the same SPLIT32 layout (mixed-precision training, some_train_gemm_f16).
// STAGES = 2: the double-buffered kernel described above (one workgroup per CU at 256 x 256).  STAGES = 1 (round 6, two workgroups per CU):
// ONE LDS stage + the register stage, so that a 128 x 256 tile (4 waves of 64 x 128, 55 KB) fits TWICE on a CU - two INDEPENDENT
// workgroups whose barriers, LDS-write phases and epilogues fall into each other's matrix phases (a VALU-only wavefront issues
// beside an MFMA wavefront of the same SIMD at no cost to it: profiles/r05_coissue_probe2.md table C; MI355X_MICROARCH.md
// "an MFMA wave and a VALU-only wave run concurrently").  Same wave tile, same fragment schedule, same product order per output
// element as the 256 x 256 kernel: bit-identical results.
template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool OUT_SPLIT, int TERMS = 3, bool TR = false, bool BF16 = false, int STAGES = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, STAGES == 1 ? 2 : 1) void hgemm3_kernel(GemmArgs a) {
    static_assert(!BF16 || (TERMS == 1 && !TR), "bf16 hi halves: one-product kernels only (split.h)");
    static_assert(STAGES == 2 || TERMS == 3, "the single-stage loop is the hand-scheduled three-product one");
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int STAGE = (BM + BN) * LDT;               // dwords
    constexpr int NLD = (BM + BN) * 8 / NT;              // 16-byte chunks per thread per k-block
    static_assert((BM + BN) * 8 % NT == 0, "staging must divide evenly");
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef GEMM_TIMELINE            // tools/gemm_probe.hip: per-workgroup timestamps (s_memtime) of the prologue / k-loop / epilogue
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memtime();
#endif

    GemmGroup g = a.g[blockIdx.y];
    const int n_tiles = a.n_tiles;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int m_tile = (j / n_tiles) * 8 + xcd;
    const int n_tile = j % n_tiles;
    const int m0 = a.m_begin + m_tile * BM, n0 = n_tile * BN;
    if (m0 >= a.M || n0 >= g.N) return;

    // The wavefront index stays lane-derived.  Forcing it into an SGPR (readfirstlane; removes the 64 one-trip waterfall loops hipcc
    // wraps around the row-per-lane epilogue's stores) measured no gain (1.0002 vs 1.0005 ms) AND made the results run-dependent.
    // Root cause (round 3, profiles/r03_sgpr_epilogue_hazard.md): the column offset then becomes the SGPR soffset of the 16-byte
    // stores, and hipcc schedules `buffer_store_dwordx4 v[158:161], ..., s13 offen` directly in front of a VALU instruction that
    // rewrites v158 - LLVM exempts SGPR-soffset stores from the ">64-bit store data" wait state, gfx950 does not honour the
    // exemption (the store sometimes writes the new value; `s_nop 1` behind the stores cures it).  tools/isa_hazard_scan.py keeps
    // that form out of the shipped library (CPU test suite).
#ifdef GEMM_WAVE_SGPR_EPI        // diagnostic builds only: SGPR wave index for the epilogues in the bit mask
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = ((GEMM_WAVE_SGPR_EPI >> EPI) & 1) ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
#else
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#endif
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, kg = lane >> 5;
    // split-K (weight gradients: small output, contraction over all frames): slice z covers k-blocks [kt0, kt0 + nk)
    // and writes its own partial plane; the launcher sizes the slices so that none is empty
    int nk = a.K >> 5, kt0 = 0;
    if (a.k_slices > 1) {
        const int per = (nk + a.k_slices - 1) / a.k_slices;
        kt0 = blockIdx.z * per;
        nk = min(nk, kt0 + per) - kt0;
        g.C += (size_t)blockIdx.z * a.slice_stride;
    }

    // ---- staging roles: thread t moves chunk (row t / 8 + (NT / 8) p, 16-byte column t % 8) of every k-block, p < NLD;
    // the first BM rows are A, the rest W.  Loads go through buffer descriptors: rows past the end of A (partial row
    // tile) or W (partial column tile) fall outside num_records and return zeros - no per-load branch, so the whole
    // k-iteration is ONE basic block and the scheduler interleaves the staging traffic with the MFMAs.
    constexpr int RPP = NT / 8;                          // rows per pass
    static_assert(BM % RPP == 0, "a pass must not straddle A and W");
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(g.A, (size_t)a.a_rows * a.lda * 4);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(g.W, (size_t)g.N * a.K * 4);
    const int srow = tid >> 3, scol = tid & 7;
    uint32_t voff[NLD];
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
        const int row = srow + p * RPP;
        voff[p] = row < BM ? (uint32_t)(m0 + row) * (uint32_t)a.lda * 4u + scol * 16u
                           : (uint32_t)(n0 + row - BM) * (uint32_t)a.K * 4u + scol * 16u;
    }
    if (a.row_map != nullptr) {      // row gather (QKV projection into clip-aligned rows): the offsets are fixed for the whole k-loop
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            const int row = srow + p * RPP;
            if (row < BM) {
                const int src = m0 + row < a.M ? a.row_map[m0 + row] : -1;
                voff[p] = src >= 0 ? (uint32_t)src * (uint32_t)a.lda * 4u + scol * 16u : kOob;
            }
        }
    }
    const int dst0 = srow * LDT + scol * 4;
    f32x4 stage[NLD];
    auto gload = [&](int kt) {
        const uint32_t koff = (uint32_t)(kt0 + kt) * 128u;
#pragma unroll
        for (int p = 0; p < NLD; ++p)
            stage[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(p * RPP < BM ? rsa : rsw, voff[p], koff, 0));
    };
    auto lstore = [&](int buf) {
        float* base = lds + buf * STAGE + dst0;
#pragma unroll
        for (int p = 0; p < NLD; ++p) *reinterpret_cast<f32x4*>(base + p * RPP * LDT) = stage[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    // Pipeline (one register set, cdna guide T14 "write after the barrier"): at the top of iteration kt the
    // registers hold k-block kt + 1 (loaded during iteration kt - 1, so its latency is already paid); they are
    // written to the other LDS buffer, immediately re-issued for k-block kt + 2, and the MFMAs of block kt run
    // while those loads fly.  One barrier per k-block; nothing waits on a just-issued load.
    gload(0);
    lstore(0);
    if (nk > 1) gload(1);
    __syncthreads();
#ifdef GEMM_TIMELINE
    const unsigned long long tl_t1 = __builtin_amdgcn_s_memtime();
#endif

    const int a_off = (wm * TM * 32 + l31) * LDT + kg * 4;
    const int w_off = (BM + wn * TN * 32 + (TR ? pi32(l31) : l31)) * LDT + kg * 4;
    // TR: the W fragment is the first MFMA operand (accumulator rows <- n, lane <- m), see epilogue_tr
    // GEMM_ABLATE_* (tools/build_variant.py builds only, wrong results): what each part of the k-loop costs
    auto mma = [](half8 x, half8 w, f32x16 c) {
#ifdef GEMM_ABLATE_NO_MFMA
        c[0] += (float)x[0] + (float)w[0];               // keeps the fragment reads alive
        return c;
#else
        return TR ? mfma_hi<BF16>(w, x, c) : mfma_hi<BF16>(x, w, c);
#endif
    };

    // the MFMAs of one k-block out of LDS buffer `buf`
    auto compute = [&](int buf) {
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
            // three sweeps over the accumulator tiles: consecutive MFMAs never share an accumulator
            if (TERMS == 3) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = mma(al[i], bh[jn], acc[i][jn]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = mma(ah[i], bl[jn], acc[i][jn]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = mma(ah[i], bh[jn], acc[i][jn]);
        }
    };
    // Steady state (TERMS = 3), hand-scheduled - hipcc left to itself puts all LDS writes at the head of the iteration
    // and sinks the global loads to its end (so the next head waits on them):
    //   * the first k = 16 slab's fragment reads go first: nothing sits in front of them in the LDS queue;
    //   * one LDS write (k-block kt + 1) + one global load (k-block kt + 2) of the staging traffic follows each of the
    //     first MFMAs - asynchronous instructions issued in the shadow of a 32-cycle MFMA;
    //   * the second slab's operands are read one product ahead: al / bh (first product) while the first slab's last
    //     product runs, ah / bl under the second slab's first product - into the registers the first slab has just
    //     retired (al dies after product 1, bl after product 2), so the fragment set stays at 48 VGPRs.
    // sched_barrier(0) pins that order; no condition inside the iteration, the last two k-blocks are peeled.
    auto frag = [&](const float* base, int tile, int s, int lo) {
        return *reinterpret_cast<const half8*>(base + tile * 32 * LDT + lo * 16 + s * 8);
    };
    // mode 0: steady state (store k-block kt + 1, load kt + 2); 1: second-to-last block (store only); 2: last block (neither)
    auto compute_staged = [&](int buf, int kt_load, int mode) {
        const float* As = lds + buf * STAGE + a_off;
        const float* Ws = lds + buf * STAGE + w_off;
        float* wbase = lds + (buf ^ 1) * STAGE + dst0;
        const uint32_t koff = (uint32_t)(kt0 + kt_load) * 128u;
        half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) al[i] = frag(As, i, 0, 1);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bh[jn] = frag(Ws, jn, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[i] = frag(As, i, 0, 0);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bl[jn] = frag(Ws, jn, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int SLOTS = 2 * TM * TN;                       // products 1 and 2 of the first slab carry the staging
        constexpr int OPS = (NLD + SLOTS - 1) / SLOTS;
        auto stage_ops = [&](int slot) {
#pragma unroll
            for (int o = 0; o < OPS; ++o) {
                const int q = slot * OPS + o;
                if (q < NLD && mode < 2) {
#ifndef GEMM_ABLATE_NO_LDSW
                    *reinterpret_cast<f32x4*>(wbase + q * RPP * LDT) = stage[q];
#endif
#ifndef GEMM_ABLATE_NO_LOAD
                    if (mode == 0)
                        stage[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q * RPP < BM ? rsa : rsw, voff[q], koff, 0));
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                acc[i][jn] = mma(al[i], bh[jn], acc[i][jn]);
                stage_ops(i * TN + jn);
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                acc[i][jn] = mma(ah[i], bl[jn], acc[i][jn]);
                stage_ops(TM * TN + i * TN + jn);
            }
        // second slab, first-product operands
        half8 al1[TM], bh1[TN], ah1[TM], bl1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) al1[i] = frag(As, i, 1, 1);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bh1[jn] = frag(Ws, jn, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mma(ah[i], bh[jn], acc[i][jn]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) ah1[i] = frag(As, i, 1, 0);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bl1[jn] = frag(Ws, jn, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mma(al1[i], bh1[jn], acc[i][jn]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mma(ah1[i], bl1[jn], acc[i][jn]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mma(ah1[i], bh1[jn], acc[i][jn]);
    };
    int kt = 0;
#ifdef GEMM_ABLATE_NO_LOOP
    nk = 1;                                              // one k-block: the epilogue alone (plus the prologue loads)
#endif
    if constexpr (STAGES == 1) {
        // one LDS stage: the products of k-block kt read it, a barrier retires the reads, the registers (k-block kt + 1, loaded
        // during iteration kt - 1) are written over it, the loads of k-block kt + 2 are issued, a second barrier publishes the
        // writes.  Nothing of THIS workgroup runs on the matrix pipe between the two barriers - the co-resident workgroup does.
        for (; kt + 1 < nk; ++kt) {
            compute_staged(0, 0, 2);
            __syncthreads();
            lstore(0);
            if (kt + 2 < nk) gload(kt + 2);
            __syncthreads();
        }
        compute_staged(0, 0, 2);
    } else {
    for (; kt + 2 < nk; ++kt) {
        if constexpr (TERMS == 3) {
            compute_staged(kt & 1, kt + 2, 0);
        } else {
            lstore((kt & 1) ^ 1);
            gload(kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute(kt & 1);
        }
        __syncthreads();
        GEMM_IT_STAMP();
    }
    if (kt + 1 < nk) {
        if constexpr (TERMS == 3) {
            compute_staged(kt & 1, 0, 1);
        } else {
            lstore((kt & 1) ^ 1);
            compute(kt & 1);
        }
        __syncthreads();
        ++kt;
    }
    if constexpr (TERMS == 3) compute_staged(kt & 1, 0, 2);
    else compute(kt & 1);
    }

#ifdef GEMM_TIMELINE
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long tl_t2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    // Interior workgroups take the unguarded path (no per-element exec masking).  Residual values are fetched one
    // 32 x 32 MFMA tile (16 per lane) at a time BEFORE that tile's stores: C may alias res (in-place residual
    // update), which otherwise forces the compiler into load -> wait -> store per element.
#ifdef GEMM_ABLATE_NO_EPI
    {
        float keep = 0.f;                                // every accumulator stays live: no product may be eliminated
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[i][jn][r];
        if (keep == 12345.678f) g.C[0] = keep;
        return;
    }
#endif
    if constexpr (TR) {
        if (n0 + BN <= g.N)      // rows beyond M fall outside the buffer descriptors (dropped by the hardware)
            epilogue_tr<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, true>(a, g, acc, m0, n0, wm, wn, lane);
        else
            epilogue_tr<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, false>(a, g, acc, m0, n0, wm, wn, lane);
    } else {
        char* wave_lds = nullptr;
        if constexpr (EPI == EPI_QKV) {       // V^T tiles leave through a wave-private LDS patch: the stages must be drained
            __syncthreads();
            wave_lds = reinterpret_cast<char*>(lds) + wave * (2 * 32 * (TM * 64 + 16));
        }
        if (m0 + BM <= a.M && n0 + BN <= g.N)
            epilogue<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, true>(a, g, acc, m0, n0, wm, wn, lane, wave_lds);
        else
            epilogue<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, false>(a, g, acc, m0, n0, wm, wn, lane, wave_lds);
    }
#ifdef GEMM_TIMELINE
    {
        const unsigned long long tl_t3 = __builtin_amdgcn_s_memtime();       // epilogue stores ISSUED
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tl_t4 = __builtin_amdgcn_s_memtime();       // ... and acknowledged
        const unsigned wg = (blockIdx.y * gridDim.x + blockIdx.x);
        if (g_gemm_tl != nullptr && lane == 0 && wg < (unsigned)g_gemm_tl_cap) {
            unsigned long long* o = g_gemm_tl + ((size_t)wg * (NT / 64) + wave) * 8;
            o[0] = tl_t0; o[1] = tl_t1; o[2] = tl_t2; o[3] = tl_t3; o[4] = tl_t4;
            o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);               // HW_ID: wave / simd / cu / sh / se
            o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);              // XCC_ID
            o[7] = ((unsigned long long)m_tile << 32) | (unsigned)n_tile;
        }
    }
#endif
}

// ---- persistent stream variant of the 256 x 256 kernel (round 6) ---------------------------------------------------------
// What the workgroup timeline of the kernel above shows (tools/gemm_probe.hip, profiles/r06_gemm_timeline.txt): of a workgroup's
// residency on its CU the FFN1 shape spends 8 % in the PROLOGUE (first k-block: HBM / L2 latency with nothing to overlap it - one
// workgroup per CU) and the CU then waits another 9 % of that time for the NEXT workgroup to be dispatched (147 KB of LDS, 8
// wavefronts); pw1 + GLU 10 % + 5 %, FFN2 2 % + 3 %.  Here one workgroup per CU stays resident and walks the tiles that the
// non-persistent grid would have handed to it (virtual block vb = blockIdx.x + r gridDim.x, the same vb -> (XCD, row tile,
// column tile) map), and its k-blocks form ONE stream across tile boundaries: the last two iterations of a tile already load /
// store the first two k-blocks of the next tile (second offset set), so the pipeline never drains - the epilogue runs with the
// next tile's first k-block in LDS and its second in flight into the staging registers.  Same products in the same order per
// output element: bit-identical to the kernel above (tests/test_gpu_kernels.py).  nk must be even (stream position parity = LDS
// buffer) and >= 4; no split-K.
template <int EPI, bool OUT_SPLIT, bool TR = false, bool LINES = false>
__global__ __launch_bounds__(512) void hgemm3p_kernel(GemmArgs a) {
    static_assert(!LINES || (TR && OUT_SPLIT && EPI == EPI_BIAS_SILU), "whole-line stores: the SPLIT32 row-per-lane epilogue");
    constexpr int WAVES_M = 4, WAVES_N = 2, TM = 2, TN = 4;
    constexpr int NT = 512, BM = 256, BN = 256;
    constexpr int STAGE = (BM + BN) * LDT;               // dwords
    constexpr int NLD = (BM + BN) * 8 / NT;              // 8 chunks of 16 bytes per thread and k-block
    constexpr int RPP = NT / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef GEMM_TIMELINE
    unsigned long long tl_t0 = __builtin_amdgcn_s_memtime();
#endif
    const GemmGroup g = a.g[blockIdx.y];
    // (scalars out of the argument block: lambdas that loop over `a.` fields make hipcc keep a copy of the whole block in scratch)
    const int n_tiles = a.n_tiles;
    const int vb_count = a.vb_count;                     // virtual blocks of one group: ((m_tiles + 7) / 8 * 8) n_tiles
    const int a_M = a.M, a_mbeg = a.m_begin, g_N = g.N, stride = (int)gridDim.x;
    const uint32_t row_a = (uint32_t)a.lda * 4u, row_w = (uint32_t)a.K * 4u;
    const int32_t* __restrict__ row_map = a.row_map;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, kg = lane >> 5;
    const int nk = a.K >> 5;

    // virtual block -> tile origin (the map of hgemm3_kernel); false: the block lies outside the matrix
    auto origin = [=](int vb, int& m0, int& n0) -> bool {
        const int xcd = vb & 7, j = vb >> 3;
        m0 = a_mbeg + ((j / n_tiles) * 8 + xcd) * BM;
        n0 = (j % n_tiles) * BN;
        return m0 < a_M && n0 < g_N;
    };
    auto next_valid = [=](int vb, int& m0, int& n0) -> int {      // first valid virtual block of this workgroup at or behind vb
        while (vb < vb_count && !origin(vb, m0, n0)) vb += stride;
        return vb;
    };

    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(g.A, (size_t)a.a_rows * a.lda * 4);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(g.W, (size_t)g.N * a.K * 4);
    const int srow = tid >> 3, scol = tid & 7;
    auto offsets = [=](int m0, int n0, uint32_t (&vo)[NLD]) {
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            const int row = srow + p * RPP;
            if (row < BM) {
                if (row_map != nullptr) {                // row gather (QKV projection into clip-aligned rows)
                    const int src = m0 + row < a_M ? row_map[m0 + row] : -1;
                    vo[p] = src >= 0 ? (uint32_t)src * row_a + scol * 16u : kOob;
                } else {
                    vo[p] = (uint32_t)(m0 + row) * row_a + scol * 16u;
                }
            } else {
                vo[p] = (uint32_t)(n0 + row - BM) * row_w + scol * 16u;
            }
        }
    };
    const int dst0 = srow * LDT + scol * 4;
    f32x4 stage[NLD];
    uint32_t voff[NLD];

    f32x16 acc[TM][TN];
    const int a_off = (wm * TM * 32 + l31) * LDT + kg * 4;
    const int w_off = (BM + wn * TN * 32 + (TR ? pi32(l31) : l31)) * LDT + kg * 4;
    auto mma = [](half8 x, half8 w, f32x16 c) { return TR ? mfma_hi<false>(w, x, c) : mfma_hi<false>(x, w, c); };
    auto frag = [&](const float* base, int tile, int s, int lo) {
        return *reinterpret_cast<const half8*>(base + tile * 32 * LDT + lo * 16 + s * 8);
    };
    // One k-block out of LDS buffer `buf`, hand-scheduled exactly as hgemm3_kernel's compute_staged.  mode 0: store the staged
    // k-block into the other buffer and load the k-block at byte offset `koff` of the rows `vo`; 1: store only; 2: neither.
    auto compute_staged = [&](int buf, const uint32_t (&vo)[NLD], uint32_t koff, int mode) {
        const float* As = lds + buf * STAGE + a_off;
        const float* Ws = lds + buf * STAGE + w_off;
        float* wbase = lds + (buf ^ 1) * STAGE + dst0;
        half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) al[i] = frag(As, i, 0, 1);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bh[jn] = frag(Ws, jn, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[i] = frag(As, i, 0, 0);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bl[jn] = frag(Ws, jn, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        auto stage_ops = [&](int q) {
            if (q < NLD && mode < 2) {
                *reinterpret_cast<f32x4*>(wbase + q * RPP * LDT) = stage[q];
                if (mode == 0)
                    stage[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q * RPP < BM ? rsa : rsw, vo[q], koff, 0));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                acc[i][jn] = mma(al[i], bh[jn], acc[i][jn]);
                stage_ops(i * TN + jn);
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                acc[i][jn] = mma(ah[i], bl[jn], acc[i][jn]);
                stage_ops(TM * TN + i * TN + jn);
            }
        half8 al1[TM], bh1[TN], ah1[TM], bl1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) al1[i] = frag(As, i, 1, 1);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bh1[jn] = frag(Ws, jn, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mma(ah[i], bh[jn], acc[i][jn]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) ah1[i] = frag(As, i, 1, 0);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bl1[jn] = frag(Ws, jn, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mma(al1[i], bh1[jn], acc[i][jn]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mma(ah1[i], bl1[jn], acc[i][jn]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mma(ah1[i], bh1[jn], acc[i][jn]);
    };

    int m0, n0;
    int vb = next_valid((int)blockIdx.x, m0, n0);
    if (vb >= vb_count) return;
    offsets(m0, n0, voff);
    // prologue of the FIRST tile only: k-block 0 -> LDS buffer 0, k-block 1 -> registers
#pragma unroll
    for (int q = 0; q < NLD; ++q)
        stage[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q * RPP < BM ? rsa : rsw, voff[q], 0u, 0));
#pragma unroll
    for (int q = 0; q < NLD; ++q) *reinterpret_cast<f32x4*>(lds + dst0 + q * RPP * LDT) = stage[q];
#pragma unroll
    for (int q = 0; q < NLD; ++q)
        stage[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q * RPP < BM ? rsa : rsw, voff[q], 128u, 0));
    __syncthreads();

    while (true) {
#ifdef GEMM_TIMELINE
        const unsigned long long tl_t1 = __builtin_amdgcn_s_memtime();
#endif
        int m0n, n0n;
        const int vbn = next_valid(vb + stride, m0n, n0n);
        const bool has_next = vbn < vb_count;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
        // ONE steady-state iteration body for the whole stream: iteration kt stores k-block kt + 1 and loads k-block kt + 2 - which for
        // the last two iterations of a tile are the first two k-blocks of the NEXT tile (nk is even: block nk - 2 sits in buffer 0, so
        // the next tile's block 0 lands in buffer 0 again).  Behind the last tile the offsets point outside the buffers: the loads
        // return zeros into registers nobody reads and the stores fill LDS buffers nobody reads - no second code path.
        for (int kt = 0; kt < nk; ++kt) {
            if (kt == nk - 2) {                          // this tile's loads have all been issued: the offsets turn to the next tile
                if (has_next) offsets(m0n, n0n, voff);
                else {
#pragma unroll
                    for (int q = 0; q < NLD; ++q) voff[q] = kOob;
                }
            }
            const int kl = kt + 2 < nk ? kt + 2 : kt + 2 - nk;
            compute_staged(kt & 1, voff, (uint32_t)kl * 128u, 0);
            __syncthreads();                             // (behind the last iteration: buffer 1 retired, the next tile's block 0 visible)
            GEMM_IT_STAMP();
        }
        char* wave_lds = nullptr;
        if constexpr (EPI == EPI_QKV)                    // the V^T patch lives in buffer 1 (buffer 0 already holds the next tile's first k-block)
            wave_lds = reinterpret_cast<char*>(lds + STAGE) + wave * (2 * 32 * (TM * 64 + 16));
#ifdef GEMM_TIMELINE
        asm volatile("s_nop 0" ::: "memory");
        const unsigned long long tl_t2 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (LINES) {
            char* patch = reinterpret_cast<char*>(lds + STAGE) + wave * 4096;      // buffer 1: retired by the barrier behind the last iteration
            if (n0 + BN <= g.N) epilogue_tr_lines<WAVES_M, WAVES_N, TM, TN, true>(a, g, acc, m0, n0, wm, wn, lane, patch);
            else epilogue_tr_lines<WAVES_M, WAVES_N, TM, TN, false>(a, g, acc, m0, n0, wm, wn, lane, patch);
        } else if constexpr (TR) {
            if (n0 + BN <= g.N) epilogue_tr<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, true>(a, g, acc, m0, n0, wm, wn, lane);
            else epilogue_tr<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, false>(a, g, acc, m0, n0, wm, wn, lane);
        } else {
            if (m0 + BM <= a.M && n0 + BN <= g.N) epilogue<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, true, 2>(a, g, acc, m0, n0, wm, wn, lane, wave_lds);
            else epilogue<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, false, 2>(a, g, acc, m0, n0, wm, wn, lane, wave_lds);
        }
#ifdef GEMM_TIMELINE
        {
            const unsigned long long tl_t3 = __builtin_amdgcn_s_memtime();
            const unsigned wgi = (unsigned)(blockIdx.y * vb_count + vb);
            if (g_gemm_tl != nullptr && lane == 0 && wgi < (unsigned)g_gemm_tl_cap) {
                unsigned long long* o = g_gemm_tl + ((size_t)wgi * 8 + wave) * 8;
                o[0] = tl_t0; o[1] = tl_t1; o[2] = tl_t2; o[3] = tl_t3; o[4] = tl_t3;
                o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                o[7] = ((unsigned long long)m0 << 32) | (unsigned)n0;
            }
            tl_t0 = tl_t3;
        }
#endif
        if (!has_next) break;
        if constexpr (EPI == EPI_QKV || LINES) __syncthreads();   // the patches in buffer 1 have been read: the next iteration stores a k-block there
        vb = vbn; m0 = m0n; n0 = n0n;
    }
}

template <int EPI, bool OUT_SPLIT, bool TR = false, bool LINES = false>
hipError_t launch_persist(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 256, BN = 256;
    constexpr size_t LDS_BYTES = 2 * (size_t)(BM + BN) * LDT * sizeof(float);
    static DeviceOnce attr_once;
    static std::atomic<int> cus{0};
    auto kern = &hgemm3p_kernel<EPI, OUT_SPLIT, TR, LINES>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0, n = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        cus.store(n > 0 ? n : 256);
        attr_once.mark();
    }
    int n_max = 0;
    for (int g = 0; g < a.groups; ++g) n_max = a.g[g].N > n_max ? a.g[g].N : n_max;
    const int m_tiles = (a.M - a.m_begin + BM - 1) / BM, n_tiles = (n_max + BN - 1) / BN;
    GemmArgs b = a;
    b.n_tiles = n_tiles;
    b.vb_count = (m_tiles + 7) / 8 * 8 * n_tiles;
    // one resident workgroup per CU, the groups side by side; a multiple of 8 per group keeps vb & 7 == blockIdx.x & 7 (the XCD)
    int per_group = cus.load() / (a.groups > 0 ? a.groups : 1) / 8 * 8;
    if (per_group < 8) per_group = 8;
    if (per_group > b.vb_count) per_group = b.vb_count;
    dim3 grid((unsigned)per_group, (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(512), LDS_BYTES, s, b);
    return hipGetLastError();
}

// ---- ring variant: 128 x 256 tile, 4 waves (2 x 2, each 64 x 128), TWO workgroups per CU -----------------------
// Ablation of the register-staged 256 x 256 kernel (FFN1 shape): 58 % of its time is the MFMA floor, 20 % exposed
// staging / fragment traffic, 22 % the epilogue - none of which overlaps anything with one workgroup per CU.
// Here every workgroup is half as big and two share a CU, so one's epilogue, DMA waits and LDS reads run under
// the other's MFMAs.  Operands go global -> LDS by DMA (global_load_lds, 16 B per lane: no VGPR round trip, no
// ds_write) in k-steps of 16 into a 3-stage ring (24 KiB per stage, 72 KiB per workgroup) that is filled two steps
// ahead behind a COUNTED vmcnt and a raw s_barrier (never drained to 0 in the loop).  LDS rows are 64 unpadded
// bytes = four 16-byte chunks [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15]; chunk slot p of row r holds logical chunk
// p ^ ((r >> 2) & 3) - the permutation is applied to the per-lane SOURCE address (the DMA destination is
// lane-linear) and again on the fragment reads, which makes every ds_read_b128 conflict-free.
template <int EPI, bool OUT_SPLIT>
__global__ __launch_bounds__(256, 2) void hgemm3_ring_kernel(GemmArgs a) {
    constexpr int WAVES_M = 2, WAVES_N = 2, TM = 2, TN = 4;
    constexpr int NT = 256, BM = 128, BN = 256;
    constexpr int ROWS = BM + BN;                       // 384 rows x 64 B per stage
    constexpr int STAGE_B = ROWS * 64;
    constexpr int NSTAGE = 3;
    constexpr int NDMA = ROWS * 4 / NT;                 // 6 chunks per thread per step
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* lbase = reinterpret_cast<char*>(lds);

    const GemmGroup g = a.g[blockIdx.y];
    const int n_tiles = a.n_tiles;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int m_tile = (j / n_tiles) * 8 + xcd;
    const int n_tile = j % n_tiles;
    const int m0 = a.m_begin + m_tile * BM, n0 = n_tile * BN;
    if (m0 >= a.M || n0 >= g.N) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, kg = lane >> 5;
    const int ns = a.K >> 4;                            // k-steps of 16

    // DMA roles: pass p moves linear chunk L = p * 256 + tid -> LDS byte 16 L = row (L >> 2), slot (L & 3).
    // Out-of-range rows are clamped to the last valid row (their products are never stored).
    const char* src[NDMA];
#pragma unroll
    for (int p = 0; p < NDMA; ++p) {
        const int L = p * NT + tid;
        const int row = L >> 2, slot = L & 3;
        const int c = slot ^ ((row >> 2) & 3);          // logical chunk: bit 1 = lo half, bit 0 = k 8-15
        const int coff = (c >> 1) * 64 + (c & 1) * 16;
        if (row < BM) {
            const int r = min(m0 + row, a.M - 1);
            src[p] = reinterpret_cast<const char*>(g.A) + (size_t)r * a.lda * 4 + coff;
        } else {
            const int r = min(n0 + row - BM, g.N - 1);
            src[p] = reinterpret_cast<const char*>(g.W) + (size_t)r * a.K * 4 + coff;
        }
    }
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma = [&](int t) {                             // k-step t -> ring slot t % 3
        const int koff = (t >> 1) * 128 + (t & 1) * 32;
        char* stage = lbase + (t % NSTAGE) * STAGE_B;
#pragma unroll
        for (int p = 0; p < NDMA; ++p)
            __builtin_amdgcn_global_load_lds(src[p] + koff, (lds_void*)(uintptr_t)(stage + (p * NT + wave * 64) * 16), 16, 0, 0);
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
    const int off_hi = (kg ^ swz) * 16, off_lo = ((2 + kg) ^ swz) * 16;
    const int a_row = (wm * TM * 32 + l31) * 64;
    const int w_row = (BM + wn * TN * 32 + l31) * 64;

    dma(0);
    if (ns > 1) dma(1);
    if (ns > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ns; ++t) {
        if (t + 2 < ns) dma(t + 2);                     // slot (t+2) % 3 was last read in step t - 1, before the barrier
        const char* st = lbase + (t % NSTAGE) * STAGE_B;
        half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            ah[i] = *reinterpret_cast<const half8*>(st + a_row + i * 32 * 64 + off_hi);
            al[i] = *reinterpret_cast<const half8*>(st + a_row + i * 32 * 64 + off_lo);
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            bh[jn] = *reinterpret_cast<const half8*>(st + w_row + jn * 32 * 64 + off_hi);
            bl[jn] = *reinterpret_cast<const half8*>(st + w_row + jn * 32 * 64 + off_lo);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[jn], acc[i][jn], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[jn], acc[i][jn], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jn], acc[i][jn], 0, 0, 0);
        // step t + 1 must have landed (this thread's part); step t + 2 (6 DMAs, just issued) may stay in flight
        if (t + 2 < ns) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    char* wave_lds = nullptr;
    if constexpr (EPI == EPI_QKV) {
        __syncthreads();
        wave_lds = lbase + wave * (2 * 32 * (TM * 64 + 16));
    }
    if (m0 + BM <= a.M && n0 + BN <= g.N)
        epilogue<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, true>(a, g, acc, m0, n0, wm, wn, lane, wave_lds);
    else
        epilogue<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, false>(a, g, acc, m0, n0, wm, wn, lane, wave_lds);
}

template <int EPI, bool OUT_SPLIT>
hipError_t launch_ring(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 128, BN = 256;
    constexpr size_t LDS_BYTES = 3 * (size_t)(BM + BN) * 64;
    static DeviceOnce attr_once;
    auto kern = &hgemm3_ring_kernel<EPI, OUT_SPLIT>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    int n_max = 0;
    for (int g = 0; g < a.groups; ++g) n_max = a.g[g].N > n_max ? a.g[g].N : n_max;
    const int m_tiles = (a.M - a.m_begin + BM - 1) / BM, n_tiles = (n_max + BN - 1) / BN;
    GemmArgs b = a;
    b.n_tiles = n_tiles;
    dim3 grid((unsigned)((m_tiles + 7) / 8 * 8 * n_tiles), (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, s, b);
    return hipGetLastError();
}

template <int EPI, bool OUT_SPLIT, bool TR = false>
hipError_t launch_single(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 128, BN = 256;
    constexpr size_t LDS_BYTES = (size_t)(BM + BN) * LDT * sizeof(float);
    static DeviceOnce attr_once;
    auto kern = &hgemm3_kernel<2, 2, 2, 4, EPI, OUT_SPLIT, 3, TR, false, 1>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    int n_max = 0;
    for (int g = 0; g < a.groups; ++g) n_max = a.g[g].N > n_max ? a.g[g].N : n_max;
    const int m_tiles = (a.M - a.m_begin + BM - 1) / BM, n_tiles = (n_max + BN - 1) / BN;
    GemmArgs b = a;
    b.n_tiles = n_tiles;
    dim3 grid((unsigned)((m_tiles + 7) / 8 * 8 * n_tiles), (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, s, b);
    return hipGetLastError();
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool OUT_SPLIT, int TERMS = 3, bool TR = false, bool BF16 = false>
hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t LDS_BYTES = 2 * (size_t)(BM + BN) * LDT * sizeof(float);
    static DeviceOnce attr_once;
    auto kern = &hgemm3_kernel<WAVES_M, WAVES_N, TM, TN, EPI, OUT_SPLIT, TERMS, TR, BF16>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    int n_max = 0;
    for (int g = 0; g < a.groups; ++g) n_max = a.g[g].N > n_max ? a.g[g].N : n_max;
    const int m_tiles = (a.M - a.m_begin + BM - 1) / BM, n_tiles = (n_max + BN - 1) / BN;
    GemmArgs b = a;
    b.n_tiles = n_tiles;
    dim3 grid((unsigned)((m_tiles + 7) / 8 * 8 * n_tiles), (unsigned)a.groups, (unsigned)(a.k_slices > 1 ? a.k_slices : 1));
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), LDS_BYTES, s, b);
    return hipGetLastError();
}


// ---- GEMMs on fp32 operands, converted on the way into LDS (training) --------------------------------------------------
//   C [M, N] = Aop [M, K] * Bop [N, K]^T  (+ bias),  Aop / Bop = fp32 arrays stored EITHER way round:
//     TA = false: A is [M, lda], the contraction index contiguous (activations in the forward, dY in the data gradient)
//     TA = true : A is [K, lda], the OUTPUT-row index contiguous (dY in dW = dY^T X: the contraction runs over frames)
//     TB likewise for B ([N, ldb] / [K, ldb]: W in the forward; W in dX = dY W and X in dW).
// MODE 1 / 2: one product on the f16 / bf16 roundings (mixed-precision training); MODE 3: the fp32-equivalent 3-term split
// (x = hi + lo, ah bh + ah bl + al bh) of the kernels above.
// The SPLIT32 kernels above need their operands split (hi | lo planes) and - for a contraction over rows - transposed by
// separate passes: 229 split + 233 transpose launches, 14 - 18 % of a training step.  Here the staging path does both: fp32
// rows are loaded as they lie (coalesced along whichever index is contiguous), rounded / split in registers
// (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32) and written to LDS as [row][32 k] halves - an 8 (k) x 2 / 4 (rows) register block of
// a row-contiguous operand leaves as one 16-byte LDS write per row, which IS the transposition.
// 4 waves (2 x 2), k-blocks of 32, TWO workgroups per CU, so one's epilogue (fp32 output rows: the HBM-heavy part of these
// GEMMs) runs under the other's products: 128 x 256 tile with 80-byte LDS rows (60 KB) in the one-product modes, 128 x 128
// with the [32 hi | 32 lo] 144-byte rows (72 KB) in the split mode.
// TA && TB (weight gradient): split-K (slices mapped to XCDs) into partial planes (ordered reduction afterwards), and the column
// sums of A's stored array - the bias gradient - accumulated in fp32 from the staging registers into C column `sum_col`.
struct Gemm16Args {
    const float* A; const float* B; const float* bias; float* C;
    int M, N, K;
    int lda, ldb, ldc;
    int m_tiles, n_tiles;
    int k_slices; size_t slice_stride;
    int sum_col;               // TA && TB only; -1 = none
};
constexpr int LD16 = 20;       // one-product LDS row in dwords: 32 halves (16 dwords) + 4 pad - 16 consecutive rows cover all 64 banks once
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of 16-bit values (element 0 in the low half); MODE 3 also returns the dword of lo halves
template <int MODE>
__device__ __forceinline__ uint32_t cvt2(float a, float b, uint32_t& lo) {
    const f32x2 v = {a, b};
    if constexpr (MODE == 2) { lo = 0; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
    const half2_t h = __builtin_convertvector(v, half2_t);
    if constexpr (MODE == 3) lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - __builtin_convertvector(h, f32x2), half2_t));
    else lo = 0;
    return __builtin_bit_cast(uint32_t, h);
}

template <int MODE> struct Gemm16Cfg {
    static constexpr bool SPLIT = MODE == 3;
    static constexpr int BM = 128, BN = SPLIT ? 128 : 256, TN = SPLIT ? 2 : 4;
    static constexpr int LDR = SPLIT ? LDT : LD16;
    static constexpr size_t LDS_BYTES = 2 * (size_t)(BM + BN) * LDR * sizeof(float);
};

// S16 (weight-gradient layout, one-product modes): both operands are STORED as 16-bit values of the mode's format (the FFN's 16-bit
// activations of train_gemm16s.hip) - half the bytes per k-block, no conversion; the register transposition is a byte permute.
template <bool TA, bool TB, int MODE, bool S16 = false>
__global__ __launch_bounds__(256, 2) void gemm16_kernel(Gemm16Args a) {
    using Cfg = Gemm16Cfg<MODE>;
    constexpr bool SPLIT = Cfg::SPLIT, BF16 = MODE == 2;
    static_assert(!S16 || (TA && TB && !SPLIT), "16-bit stored operands: weight-gradient layout of the one-product modes");
    constexpr int WAVES_M = 2, WAVES_N = 2, TM = 2, TN = Cfg::TN;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, LDR = Cfg::LDR;
    constexpr int STAGE = (BM + BN) * LDR;               // dwords
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint32_t* L = reinterpret_cast<uint32_t*>(lds);

    const int n_tiles = a.n_tiles;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    int m_tile, n_tile, slice = 0;
    if (a.k_slices > 1) {
        // weight gradient: ALL tiles of one contraction slice run on ONE XCD (workgroups go round-robin over the XCDs), so the
        // slice's rows of both operands are fetched into that XCD's L2 once instead of once per XCD
        const int tiles = a.m_tiles * n_tiles;
        slice = (j / tiles) * 8 + xcd;
        if (slice >= a.k_slices) return;
        m_tile = (j % tiles) / n_tiles;
        n_tile = j % n_tiles;
    } else {
        m_tile = (j / n_tiles) * 8 + xcd;
        n_tile = j % n_tiles;
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (m0 >= a.M || n0 >= a.N) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, kg = lane >> 5;

    int nk = (a.K + 31) >> 5, kt0 = 0;
    float* Cout = a.C;
    if (a.k_slices > 1) {
        const int per = (nk + a.k_slices - 1) / a.k_slices;
        kt0 = slice * per;
        nk = min(nk, kt0 + per) - kt0;
        Cout += (size_t)slice * a.slice_stride;
    }

    // ---- staging roles
    // contraction-contiguous operand (R rows of the tile): thread -> row tid / 8 + 32 p, 16-byte chunk tid % 8 of the 128-byte k-block
    // row-contiguous operand ([32 k][R]): wavefront kq = tid / 64 owns k rows 8 kq .. 8 kq + 7, lane c owns R / 64 adjacent columns
    constexpr int NA = TA ? 8 : BM / 32, NB = TB ? 8 : BN / 32;     // loads per thread and k-block
    constexpr int WA = TA ? BM / 64 : 4, WB = TB ? BN / 64 : 4;     // floats per load
    constexpr uint32_t ES = S16 ? 2u : 4u;                          // bytes per stored element
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(a.A, (size_t)(TA ? a.K : a.M) * a.lda * ES);
    const __amdgpu_buffer_rsrc_t rsb = make_rsrc(a.B, (size_t)(TB ? a.K : a.N) * a.ldb * ES);
    const int srow = tid >> 3, scol = tid & 7;
    uint32_t va[NA], vb[NB];                                         // byte offsets of k-block 0 (row-contiguous: of k row 8 kq + r)
#pragma unroll
    for (int p = 0; p < NA; ++p) {
        if constexpr (TA) {
            const int col = m0 + WA * lane;
            va[p] = col < a.M ? (uint32_t)(kt0 * 32 + 8 * wave + p) * (uint32_t)a.lda * ES + (uint32_t)col * ES : kOob;
        } else {
            va[p] = (uint32_t)(m0 + srow + 32 * p) * (uint32_t)a.lda * 4u + (uint32_t)kt0 * 128u + scol * 16u;
        }
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        if constexpr (TB) {
            const int col = n0 + WB * lane;
            vb[p] = col < a.N ? (uint32_t)(kt0 * 32 + 8 * wave + p) * (uint32_t)a.ldb * ES + (uint32_t)col * ES : kOob;
        } else {
            vb[p] = (uint32_t)(n0 + srow + 32 * p) * (uint32_t)a.ldb * 4u + (uint32_t)kt0 * 128u + scol * 16u;
        }
    }
    const uint32_t stepa = TA ? 32u * (uint32_t)a.lda * ES : 128u, stepb = TB ? 32u * (uint32_t)a.ldb * ES : 128u;
    float ra[S16 ? 1 : NA][WA], rb[S16 ? 1 : NB][WB];
    uint32_t qa[S16 ? NA : 1][WA / 2], qb[S16 ? NB : 1][WB / 2];      // S16: dwords of two adjacent columns
    auto ldw = [](__amdgpu_buffer_rsrc_t r, uint32_t off, float* dst, auto width) {
        constexpr int W = decltype(width)::value;
        if constexpr (W == 4) {
            const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
            dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3];
        } else {
            const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
            dst[0] = t[0]; dst[1] = t[1];
        }
    };
    auto ldq = [](__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t* dst, auto width) {
        constexpr int W = decltype(width)::value;                   // dwords
        if constexpr (W == 2) {
            const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
            dst[0] = t[0]; dst[1] = t[1];
        } else {
            dst[0] = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
        }
    };
    auto gload = [&]() {                  // the next k-block (offsets advance with every call)
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            if constexpr (S16) ldq(rsa, va[p], qa[p], std::integral_constant<int, WA / 2>{});
            else ldw(rsa, va[p], ra[p], std::integral_constant<int, WA>{});
            if (va[p] < kOob) va[p] += stepa;
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            if constexpr (S16) ldq(rsb, vb[p], qb[p], std::integral_constant<int, WB / 2>{});
            else ldw(rsb, vb[p], rb[p], std::integral_constant<int, WB>{});
            if (vb[p] < kOob) vb[p] += stepb;
        }
    };
    float csum[TA ? WA : 1] = {};
    const bool want_sum = a.sum_col >= 0 && n_tile == 0;            // only the first column tile writes the bias gradient
    // S16: column c of eight k rows (dwords q[0..7] of the column pair holding c) -> the 16 bytes of one LDS row piece
    auto put8q = [&](uint32_t* dst, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t q4, uint32_t q5, uint32_t q6, uint32_t q7, int c) {
        const uint32_t sel = (c & 1) ? 0x07060302u : 0x05040100u;  // v_perm_b32: the chosen halves of (second, first)
        const u32x4 v = {__builtin_amdgcn_perm(q1, q0, sel), __builtin_amdgcn_perm(q3, q2, sel), __builtin_amdgcn_perm(q5, q4, sel),
                         __builtin_amdgcn_perm(q7, q6, sel)};
        *reinterpret_cast<u32x4*>(dst) = v;
    };
    auto f16of = [](uint32_t q, int c) -> float {
        const uint32_t b = (c & 1) ? q >> 16 : q & 0xffffu;
        if constexpr (BF16) return __builtin_bit_cast(float, b << 16);
        else return (float)__builtin_bit_cast(half_t, (uint16_t)b);
    };
    // eight k values of one row -> 16 bytes of hi halves (and, split mode, 16 bytes of lo halves 16 dwords further)
    auto put8 = [&](uint32_t* dst, float k0, float k1, float k2, float k3, float k4, float k5, float k6, float k7) {
        uint32_t l0, l1, l2, l3;
        const u32x4 hi = {cvt2<MODE>(k0, k1, l0), cvt2<MODE>(k2, k3, l1), cvt2<MODE>(k4, k5, l2), cvt2<MODE>(k6, k7, l3)};
        *reinterpret_cast<u32x4*>(dst) = hi;
        if constexpr (SPLIT) { const u32x4 lo = {l0, l1, l2, l3}; *reinterpret_cast<u32x4*>(dst + 16) = lo; }
    };
    auto put4 = [&](uint32_t* dst, const float (&v)[4]) {
        uint32_t l0, l1;
        const u32x2 hi = {cvt2<MODE>(v[0], v[1], l0), cvt2<MODE>(v[2], v[3], l1)};
        *reinterpret_cast<u32x2*>(dst) = hi;
        if constexpr (SPLIT) { const u32x2 lo = {l0, l1}; *reinterpret_cast<u32x2*>(dst + 16) = lo; }
    };
    auto lstore = [&](int buf) {
        uint32_t* base = L + buf * STAGE;
        if constexpr (S16) {
#pragma unroll
            for (int c = 0; c < WA; ++c) {
                const int d = c >> 1;
                if (want_sum)
                    csum[c] += ((f16of(qa[0][d], c) + f16of(qa[1][d], c)) + (f16of(qa[2][d], c) + f16of(qa[3][d], c))) +
                               ((f16of(qa[4][d], c) + f16of(qa[5][d], c)) + (f16of(qa[6][d], c) + f16of(qa[7][d], c)));
                put8q(base + (WA * lane + c) * LDR + wave * 4, qa[0][d], qa[1][d], qa[2][d], qa[3][d], qa[4][d], qa[5][d], qa[6][d], qa[7][d], c);
            }
#pragma unroll
            for (int c = 0; c < WB; ++c) {
                const int d = c >> 1;
                put8q(base + (BM + WB * lane + c) * LDR + wave * 4, qb[0][d], qb[1][d], qb[2][d], qb[3][d], qb[4][d], qb[5][d], qb[6][d], qb[7][d], c);
            }
        } else {
            if constexpr (TA) {
#pragma unroll
                for (int c = 0; c < WA; ++c) {
                    if constexpr (TB) csum[c] += ((ra[0][c] + ra[1][c]) + (ra[2][c] + ra[3][c])) + ((ra[4][c] + ra[5][c]) + (ra[6][c] + ra[7][c]));
                    put8(base + (WA * lane + c) * LDR + wave * 4, ra[0][c], ra[1][c], ra[2][c], ra[3][c], ra[4][c], ra[5][c], ra[6][c], ra[7][c]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < NA; ++p) put4(base + (srow + 32 * p) * LDR + scol * 2, ra[p]);
            }
            if constexpr (TB) {
#pragma unroll
                for (int c = 0; c < WB; ++c)
                    put8(base + (BM + WB * lane + c) * LDR + wave * 4, rb[0][c], rb[1][c], rb[2][c], rb[3][c], rb[4][c], rb[5][c], rb[6][c], rb[7][c]);
            } else {
#pragma unroll
                for (int p = 0; p < NB; ++p) put4(base + (BM + srow + 32 * p) * LDR + scol * 2, rb[p]);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    const int a_off = (wm * TM * 32 + l31) * LDR + kg * 4;
    const int w_off = (BM + wn * TN * 32 + l31) * LDR + kg * 4;
    auto compute = [&](int buf) {
        const uint32_t* As = L + buf * STAGE + a_off;
        const uint32_t* Ws = L + buf * STAGE + w_off;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                    // two k = 16 slabs per k-block
            half8 ah[TM], bh[TN], al[SPLIT ? TM : 1], bl[SPLIT ? TN : 1];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(As + i * 32 * LDR + s * 8);
                if constexpr (SPLIT) al[i] = *reinterpret_cast<const half8*>(As + i * 32 * LDR + 16 + s * 8);
            }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                bh[jn] = *reinterpret_cast<const half8*>(Ws + jn * 32 * LDR + s * 8);
                if constexpr (SPLIT) bl[jn] = *reinterpret_cast<const half8*>(Ws + jn * 32 * LDR + 16 + s * 8);
            }
            if constexpr (SPLIT) {                       // three sweeps: consecutive MFMAs never share an accumulator
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_hi<false>(al[i], bh[jn], acc[i][jn]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_hi<false>(ah[i], bl[jn], acc[i][jn]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_hi<BF16>(ah[i], bh[jn], acc[i][jn]);
        }
    };

    // same one-register-set pipeline as hgemm3_kernel: the registers hold k-block kt + 1 at the top of iteration kt
    gload();
    lstore(0);
    if (nk > 1) gload();
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
        lstore((kt & 1) ^ 1);
        gload();
        __builtin_amdgcn_sched_barrier(0);
        compute(kt & 1);
        __syncthreads();
    }
    if (kt + 1 < nk) {
        lstore((kt & 1) ^ 1);
        compute(kt & 1);
        __syncthreads();
        ++kt;
    }
    compute(kt & 1);

    // ---- epilogue: the column-per-lane fp32 stores of the split kernels (whole 128-byte row segments per instruction)
    GemmArgs ga{};
    ga.M = a.M; ga.ldc = a.ldc; ga.ldr = a.ldc; ga.alpha = 1.f;
    GemmGroup g{};
    g.C = Cout; g.bias = a.bias; g.N = a.N;
    const bool full = m0 + BM <= a.M && n0 + BN <= a.N;
    if (a.bias != nullptr) {
        if (full) epilogue<WAVES_M, WAVES_N, TM, TN, EPI_BIAS, false, true>(ga, g, acc, m0, n0, wm, wn, lane);
        else epilogue<WAVES_M, WAVES_N, TM, TN, EPI_BIAS, false, false>(ga, g, acc, m0, n0, wm, wn, lane);
    } else {
        if (full) epilogue<WAVES_M, WAVES_N, TM, TN, EPI_NONE, false, true>(ga, g, acc, m0, n0, wm, wn, lane);
        else epilogue<WAVES_M, WAVES_N, TM, TN, EPI_NONE, false, false>(ga, g, acc, m0, n0, wm, wn, lane);
    }
    if constexpr (TA && TB) {
        // bias gradient: sum over this slice's contraction rows of A's stored array, per output row; wavefront w holds the rows
        // 8 w .. 8 w + 7 of every k-block - combined in wavefront order (deterministic) by the first column tile
        if (want_sum) {
            __syncthreads();
            float* red = lds;                            // [4][BM]
#pragma unroll
            for (int c = 0; c < WA; ++c) red[wave * BM + WA * lane + c] = csum[c];
            __syncthreads();
            if (tid < BM && m0 + tid < a.M)
                Cout[(size_t)(m0 + tid) * a.ldc + a.sum_col] = ((red[tid] + red[BM + tid]) + red[2 * BM + tid]) + red[3 * BM + tid];
        }
    }
}

template <bool TA, bool TB, int MODE, bool S16 = false>
hipError_t launch_gemm16_cfg(const Gemm16Args& a_in, hipStream_t s) {
    using Cfg = Gemm16Cfg<MODE>;
    static DeviceOnce attr_once;
    auto kern = &gemm16_kernel<TA, TB, MODE, S16>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    Gemm16Args a = a_in;
    a.m_tiles = (a.M + Cfg::BM - 1) / Cfg::BM;
    a.n_tiles = (a.N + Cfg::BN - 1) / Cfg::BN;
    dim3 grid(a.k_slices > 1 ? (unsigned)((a.k_slices + 7) / 8 * 8 * a.m_tiles * a.n_tiles) : (unsigned)((a.m_tiles + 7) / 8 * 8 * a.n_tiles));
    hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
}

template <int EPI, bool OUT_SPLIT>
hipError_t launch_epi(const GemmArgs& a, int tile, hipStream_t s) {
    constexpr bool kCanTr = EPI == EPI_BIAS_SILU && OUT_SPLIT;
    // the persistent stream kernel takes the 256 x 256 launches whose k-blocks pair up (even count >= 4, no split-K)
    const bool persist = tile == 2 && (a.flags & GEMM_FLAG_PERSIST) != 0 && a.k_slices <= 1 && ((a.K >> 5) & 1) == 0 && (a.K >> 5) >= 4;
    if (persist) {
        if constexpr (kCanTr) {
            bool tr = (a.flags & GEMM_FLAG_TR) != 0;
            for (int g = 0; g < a.groups; ++g) tr = tr && (a.g[g].N % 64) == 0;
            if (tr && (a.flags & GEMM_FLAG_LINES)) return launch_persist<EPI, OUT_SPLIT, true, true>(a, s);
            if (tr) return launch_persist<EPI, OUT_SPLIT, true>(a, s);
        }
        return launch_persist<EPI, OUT_SPLIT>(a, s);
    }
    if constexpr (kCanTr) {
        bool tr = (a.flags & GEMM_FLAG_TR) != 0 && tile != 3;
        for (int g = 0; g < a.groups; ++g) tr = tr && (a.g[g].N % 64) == 0;
        if (tr) {
            switch (tile) {
                case 5: return launch_single<EPI, OUT_SPLIT, true>(a, s);
                case 4: return launch_cfg<2, 2, 1, 2, EPI, OUT_SPLIT, 3, true>(a, s);
                case 0: return launch_cfg<2, 2, 2, 2, EPI, OUT_SPLIT, 3, true>(a, s);
                case 1: return launch_cfg<4, 2, 2, 2, EPI, OUT_SPLIT, 3, true>(a, s);
                default: return launch_cfg<4, 2, 2, 4, EPI, OUT_SPLIT, 3, true>(a, s);
            }
        }
    }
    switch (tile) {
        case 4: return launch_cfg<2, 2, 1, 2, EPI, OUT_SPLIT>(a, s);    // 64 x 128, 4 waves (small M: more workgroups)
        case 0: return launch_cfg<2, 2, 2, 2, EPI, OUT_SPLIT>(a, s);    // 128 x 128, 4 waves
        case 1: return launch_cfg<4, 2, 2, 2, EPI, OUT_SPLIT>(a, s);    // 256 x 128, 8 waves
        case 3: return launch_ring<EPI, OUT_SPLIT>(a, s);               // 128 x 256, 4 waves, DMA ring, 2 workgroups / CU
        case 5: return launch_single<EPI, OUT_SPLIT>(a, s);             // 128 x 256, 4 waves, ONE LDS stage, 2 workgroups / CU (round 6)
        default: return launch_cfg<4, 2, 2, 4, EPI, OUT_SPLIT>(a, s);   // 256 x 256, 8 waves
    }
}

}  // namespace

static hipError_t launch_one(GemmEpi epi, const GemmArgs& a, bool out_split, int tile, hipStream_t s) {
    switch (epi) {
        case EPI_NONE: return launch_epi<EPI_NONE, false>(a, tile, s);
        case EPI_BIAS: return launch_epi<EPI_BIAS, false>(a, tile, s);
        case EPI_BIAS_SILU: return out_split ? launch_epi<EPI_BIAS_SILU, true>(a, tile, s) : launch_epi<EPI_BIAS_SILU, false>(a, tile, s);
        case EPI_BIAS_RES: return launch_epi<EPI_BIAS_RES, false>(a, tile, s);
        case EPI_GLU: return launch_epi<EPI_GLU, false>(a, tile, s);
        case EPI_GLU_RES: return launch_epi<EPI_GLU_RES, false>(a, tile, s);
        case EPI_QKV: return launch_epi<EPI_QKV, false>(a, tile, s);
    }
    return hipErrorInvalidValue;
}

// Plain f16 operands (hi halves of the SPLIT32 layout), fp32 accumulate: the mixed-precision training GEMM.  256 x 256 tile,
// EPI_NONE / EPI_BIAS, optional split-K.
template <bool BF16>
static hipError_t launch_x1(GemmEpi epi, const GemmArgs& a, bool small, hipStream_t s) {
    if (epi == EPI_NONE)
        return small ? launch_cfg<2, 2, 2, 2, EPI_NONE, false, 1, false, BF16>(a, s) : launch_cfg<4, 2, 2, 4, EPI_NONE, false, 1, false, BF16>(a, s);
    if (epi == EPI_BIAS && a.k_slices <= 1)
        return small ? launch_cfg<2, 2, 2, 2, EPI_BIAS, false, 1, false, BF16>(a, s) : launch_cfg<4, 2, 2, 4, EPI_BIAS, false, 1, false, BF16>(a, s);
    return hipErrorInvalidValue;
}

hipError_t launch_gemm_f16x1(GemmEpi epi, const GemmArgs& a_in, int tile, hipStream_t s, int bf16) {
    if (a_in.M <= 0) return hipSuccess;
    if ((a_in.K & 31) || (a_in.lda & 31)) return hipErrorInvalidValue;
    GemmArgs a = a_in;
    a.m_begin = 0;
    if (a.a_rows <= 0) a.a_rows = a.M;
    const bool small = tile == 0;                     // 128 x 128 (two workgroups per CU) for small grids, else 256 x 256
    return bf16 ? launch_x1<true>(epi, a, small, s) : launch_x1<false>(epi, a, small, s);
}

// fp32 operands converted on the fly (gemm16_kernel): C [M, N] = Aop [M, K] Bop [N, K]^T (+ bias); ta / tb: the operand is stored
// with the contraction index as its ROW index ([K, ld]); mode 1 f16, 2 bf16 (one product), 3 split-f16 (three products).
// Contraction-contiguous operands need K % 32 == 0 and ld % 4 == 0, row-contiguous ones ld % 4 == 0 and an even (A) /
// multiple-of-4 (B) count of valid columns.  slices > 1: partial planes at C + z * slice_stride (ta && tb only).
int gemm16_tile_n(int mode) { return mode == 3 ? 128 : 256; }

template <int MODE>
static hipError_t launch_gemm16_mode(const Gemm16Args& a, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch_gemm16_cfg<false, false, MODE>(a, s);
    if (!ta && tb) return launch_gemm16_cfg<false, true, MODE>(a, s);
    return launch_gemm16_cfg<true, true, MODE>(a, s);
}

hipError_t launch_gemm16(const float* A, int lda, int ta, const float* B, int ldb, int tb, const float* bias, float* C, int ldc,
                         int M, int N, int K, int mode, int slices, size_t slice_stride, int sum_col, hipStream_t s, int stored16) {
    if (M <= 0 || N <= 0 || K <= 0) return hipSuccess;
    if ((lda & 3) || (ldb & 3) || (!(ta && tb) && (K & 31)) || (ta && (M & 1)) || (tb && (N & 3))) return hipErrorInvalidValue;
    if ((!ta || !tb) && (slices > 1 || sum_col >= 0)) return hipErrorInvalidValue;
    if ((ta && !tb) || mode < 1 || mode > 3) return hipErrorInvalidValue;
    if (stored16 && (!ta || !tb || mode == 3)) return hipErrorInvalidValue;
    Gemm16Args a{};
    a.A = A; a.B = B; a.bias = bias; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.k_slices = slices; a.slice_stride = slice_stride; a.sum_col = sum_col;
    if (stored16) return mode == 1 ? launch_gemm16_cfg<true, true, 1, true>(a, s) : launch_gemm16_cfg<true, true, 2, true>(a, s);
    if (mode == 1) return launch_gemm16_mode<1>(a, ta, tb, s);
    if (mode == 2) return launch_gemm16_mode<2>(a, ta, tb, s);
    return launch_gemm16_mode<3>(a, ta, tb, s);
}

hipError_t launch_gemm_f16x3(GemmEpi epi, const GemmArgs& a_in, bool out_split, int tile, hipStream_t s) {
    if (a_in.M <= 0) return hipSuccess;
    if ((a_in.K & 31) || (a_in.lda & 31)) return hipErrorInvalidValue;
    if (out_split && epi != EPI_BIAS_SILU) return hipErrorInvalidValue;
    if (epi == EPI_QKV)
        for (int g = 0; g < a_in.groups; ++g)
            if (a_in.g[g].N != 3 * kDim || !a_in.g[g].C2 || !a_in.g[g].C3 || (a_in.g[g].ldv & 255) || a_in.g[g].ldv < a_in.M) return hipErrorInvalidValue;
    GemmArgs a = a_in;
    a.m_begin = 0;
    if (a.a_rows <= 0) a.a_rows = a.M;
    if (a.row_map != nullptr && tile == 3) {
        // the DMA-ring kernel has no row gather: a forced SOME_AMD_TILE=3 falls back to the register-staged tile its grid size would pick
        auto blocks = [&](int bm, int bn) { long n = 0; for (int g = 0; g < a.groups; ++g) n += (long)((a.M + bm - 1) / bm) * ((a.g[g].N + bn - 1) / bn); return n; };
        tile = blocks(256, 256) >= 512 ? 2 : blocks(256, 128) >= 512 ? 1 : blocks(128, 128) >= 256 ? 0 : 4;
    }
    if (a.k_slices > 1) {
        if (epi != EPI_NONE || out_split || tile == 3 || tile == 5) return hipErrorInvalidValue;
        return launch_one(epi, a, false, tile, s);
    }
    // Wave quantisation: every workgroup of the 256-row tiles takes the same time and one fits per CU, so a grid
    // of W workgroups costs ceil(W / 256) rounds (M = 82 688, N = 512: 1292 -> 6 rounds for 5.05 rounds of work).
    // Launch the largest row range whose workgroup count is a whole number of rounds with the big tile and give
    // the remaining rows to the 128 x 128 kernel (two workgroups per CU, a quarter of the time each).
    if (tile >= 2) {
        const int BM = (tile == 3 || tile == 5) ? 128 : 256, BN = 256, slots = (tile == 3 || tile == 5) ? 512 : 256;
        int n_max = 0;
        for (int g = 0; g < a.groups; ++g) n_max = a.g[g].N > n_max ? a.g[g].N : n_max;
        const int per_m = ((n_max + BN - 1) / BN) * a.groups;                 // workgroups per row block
        int q = slots, x = per_m;                                              // q = slots / gcd(slots, per_m)
        for (int y = slots; y; ) { const int t = x % y; x = y; y = t; }
        q = slots / x;
        const int m_tiles = (a.M + BM - 1) / BM;
        const int main_tiles = m_tiles / q * q;
        if (main_tiles > 0 && main_tiles < m_tiles && (long)(m_tiles - main_tiles) * per_m <= slots / 2) {
            GemmArgs head = a;
            head.M = main_tiles * BM;                   // rows [0, head.M): full tiles only
            hipError_t e = launch_one(epi, head, out_split, tile, s);
            if (e != hipSuccess) return e;
            a.m_begin = main_tiles * BM;                // rows [m_begin, M) with the small tile
            return launch_one(epi, a, out_split, 0, s);
        }
    }
    return launch_one(epi, a, out_split, tile, s);
}
