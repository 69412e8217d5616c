// Attention backward on the f16 matrix pipe with 3-term split operands (fp32-equivalent, like attention_f16x3.hip):
// the same flash-style recomputation and the same two deterministic kernels as train_attention.hip
//     P = exp2(s2 - lse2),  dV = P^T dO,  dP = dO V^T,  dS = P (dP - D),  dQ = dS K / 8,  dK = dS^T Q / 8,
// with every product evaluated as ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_f16 (16x the f32 MFMA rate for 3 products).
//
// Operand formats (made by existing kernels from the fp32 tensors, once per call):
//   R  = split_rows(qkv)            [M, 1536] SPLIT32 rows: a head's 64 dims of q / k / v are 256 contiguous bytes
//   Rt = transpose(qkv, split)      [1536, Mp] SPLIT32 over FRAMES: channel-major, 32-frame blocks [32 hi | 32 lo]
//   D  = split_rows(dout)           [M, 512],   Dt = transpose(dout, split)  [512, Mp]
// The row-major forms feed the products that contract over d (S, dP); the frame-major forms feed the products that
// contract over frames (dV, dK, dQ), whose B operand is the P / dS tile exactly as it leaves the MFMA (C/D layout:
// register r of lane (l31, kg) is row (r & 3) + 8 (r >> 2) + 4 kg) - the matching k-permutation of the A operand is two
// ds_read_b64 per fragment, as in the forward's P V product.
// Frame-major tiles are aligned to GLOBAL 32-frame blocks (that is how Rt / Dt are laid out); frames of the block that
// lie outside the clip are masked (lse2 = +inf for queries, P = 0 for keys).
#include "internal.h"
#include "split.h"

namespace {

constexpr int QKV_LD = 3 * kDim;
constexpr int LDR = 68;                           // row-major tile row (dwords): 64 data (2 k-blocks of hi|lo) + 4 pad
constexpr int LDT = 36;                           // frame-major tile row (dwords): 32 data (1 k-block: 32 hi | 32 lo) + 4 pad
constexpr int FT = 32;                            // frames per streamed tile
constexpr float kScale = 0.125f;
constexpr float kC2 = 0.125f * 1.4426950408889634f;
constexpr float kPShift = 14.f, kPUnshift = 1.0f / 16384.0f;

__device__ __forceinline__ float exp2_(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int krow(int r, int kg) { return (r & 3) + 8 * (r >> 2) + 4 * kg; }

// x = hi + lo (packed round-toward-zero converts; hi = x with 13 low mantissa bits cleared): registers base..base+7
__device__ __forceinline__ void split8(const f32x16& p, int base, half8& h, half8& l) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const float p0 = p[base + i], p1 = p[base + i + 1];
        const float h0 = __uint_as_float(__float_as_uint(p0) & 0xFFFFE000u), h1 = __uint_as_float(__float_as_uint(p1) & 0xFFFFE000u);
        const half2_t hh = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(h0, h1));
        const half2_t ll = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(p0 - h0, p1 - h1));
        h[i] = hh[0]; h[i + 1] = hh[1];
        l[i] = ll[0]; l[i + 1] = ll[1];
    }
}

// product accumulate: acc += a * b with a = ah + al, b = bh + bl.  TERMS = 3: ah bh + ah bl + al bh (fp32-equivalent);
// TERMS = 1: ah bh only (plain f16 operands: the mixed-precision training mode)
// TERMS = 2: ONE product on bf16 operands (hi slots hold bf16, split.h) - the reference's pl_trainer_precision 'bf16'
template <int TERMS>
__device__ __forceinline__ void mma3(f32x16& acc, const half8& ah, const half8& al, const half8& bh, const half8& bl) {
    if (TERMS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    }
    acc = mfma_hi<TERMS == 2>(ah, bh, acc);
}
template <int TERMS>
__device__ __forceinline__ void split8t(const f32x16& p, int base, half8& h, half8& l) {
    if constexpr (TERMS == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { h[i] = bf16_as_half(p[base + i]); l[i] = (half_t)0.f; }
    } else {
        split8(p, base, h, l);
    }
}

// fragment of a ROW-MAJOR SPLIT32 row of 64 dims at `row` (dword pointer), k-step s (dims 16 s .. 16 s + 15): 8 dims per lane half
__device__ __forceinline__ void frag_row(const float* row, int s, int kg, half8& h, half8& l) {
    const int off = (s >> 1) * 32 + (s & 1) * 8 + kg * 4;
    h = *reinterpret_cast<const half8*>(row + off);
    l = *reinterpret_cast<const half8*>(row + off + 16);
}
// fragment of a FRAME-MAJOR row (32 frames: 32 hi | 32 lo halves) in the C/D-layout k-permutation of k-step sp (frames
// 16 sp + {0..3} + 4 kg and 16 sp + 8 + {0..3} + 4 kg)
__device__ __forceinline__ void frag_frames(const float* row, int sp, int kg, half8& h, half8& l) {
    const half_t* base = reinterpret_cast<const half_t*>(row);
    const int f0 = 16 * sp + 4 * kg;
    const half4 a0 = *reinterpret_cast<const half4*>(base + f0), a1 = *reinterpret_cast<const half4*>(base + f0 + 8);
    const half4 b0 = *reinterpret_cast<const half4*>(base + 32 + f0), b1 = *reinterpret_cast<const half4*>(base + 32 + f0 + 8);
    h = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    l = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct Bwd3Args {
    const float* R;        // [M, 1536] SPLIT32
    const float* Rt;       // [1536, Mp] SPLIT32 over frames
    const float* D;        // [M, 512] SPLIT32
    const float* Dt;       // [512, Mp]
    const float* lse;      // [8, M]
    const float* dsum;     // [8, M]
    float* dqkv;           // [M, 1536] fp32
    const int32_t* frame_offsets;
    int B, max_frames, M, Mp;
    // optional: dqkv leaves as 16-bit values (out16 = 1 f16, 2 bf16) scaled by *out_scale (a device scalar: the power of two the caller took
    // into dO) - the operand of the projection's data / weight gradient GEMMs without an fp32 round trip (csrc/train_gemm16s.hip)
    uint16_t* dqkv16;
    const float* out_scale;
    int out16;
};

typedef float bwd_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bwd_b2 __attribute__((ext_vector_type(2)));
typedef uint32_t bwd_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bwd_pack16(float a, float b, int out16) {
    const bwd_f2 v = {a, b};
    return out16 == 2 ? __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bwd_b2)) : __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2_t));
}
__device__ __forceinline__ void bwd_store4(float* p32, uint16_t* p16, const f32x4& v, float os, int out16) {
    if (out16) {
        const bwd_u2 w = {bwd_pack16(v[0] * os, v[1] * os, out16), bwd_pack16(v[2] * os, v[3] * os, out16)};
        *reinterpret_cast<bwd_u2*>(p16) = w;
    } else {
        *reinterpret_cast<f32x4*>(p32) = v;
    }
}

// ---- dK, dV: a lane owns a key -----------------------------------------------------------------------------------------------
constexpr int DKV_STAGE = 2 * FT * LDR + 2 * kHeadDim * LDT + 2 * FT;          // Qr, dOr, Qt, dOt, lse2[32], D[32]
constexpr size_t DKV_LDS = 2 * DKV_STAGE * sizeof(float);

template <int TERMS>
__global__ __launch_bounds__(256, 2) void attn3_bwd_dkv_kernel(Bwd3Args a, int nkb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int slot = jj / nkb, kb = jj % nkb;
    const int unit = slot * 8 + xcd;
    const int head = unit % kHeads, b = unit / kHeads;
    if (b >= a.B) return;
    const int f0 = a.frame_offsets[b];
    const int T = a.frame_offsets[b + 1] - f0;
    const int k0 = kb * 128;
    if (k0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg = lane >> 5;

    // this lane's key: K and V fragments (B operands of S = Q K^T and dP = dO V^T)
    const int key = k0 + wave * 32 + l31;
    const bool kv = key < T;
    half8 kh[4], kl[4], vh[4], vl[4];
    {
        const float* row = a.R + (size_t)(f0 + (kv ? key : 0)) * QKV_LD + head * kHeadDim;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            frag_row(row + kDim, s, kg, kh[s], kl[s]);
            frag_row(row + 2 * kDim, s, kg, vh[s], vl[s]);
            if (!kv) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { kh[s][i] = (half_t)0; kl[s][i] = (half_t)0; vh[s][i] = (half_t)0; vl[s][i] = (half_t)0; }
            }
        }
    }

    // staging: row-major tiles 32 rows x 16 chunks (16 B), frame-major tiles 64 rows x 8 chunks: 2 chunks per thread each
    const int g0 = f0 / FT, g1 = (f0 + T - 1) / FT;            // global 32-frame blocks touched by the clip
    const int nt = g1 - g0 + 1;
    const int rr = tid >> 4, rc = tid & 15;                    // row-major: rows rr, rr + 16; chunk rc
    const int tr = tid >> 3, tc = tid & 7;                     // frame-major: rows tr, tr + 32; chunk tc
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 sq[2], sd[2], sqt[2], sdt[2];
    float sl = 0.f;
    auto gload = [&](int i) {
        const int fr0 = (g0 + i) * FT;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int fr = fr0 + rr + 16 * p;
            const bool ok = fr < a.M;
            sq[p] = ok ? *reinterpret_cast<const f32x4*>(a.R + (size_t)fr * QKV_LD + head * kHeadDim + rc * 4) : zero4;
            sd[p] = ok ? *reinterpret_cast<const f32x4*>(a.D + (size_t)fr * kDim + head * kHeadDim + rc * 4) : zero4;
            const int d = tr + 32 * p;
            sqt[p] = *reinterpret_cast<const f32x4*>(a.Rt + (size_t)(head * kHeadDim + d) * a.Mp + fr0 + tc * 4);
            sdt[p] = *reinterpret_cast<const f32x4*>(a.Dt + (size_t)(head * kHeadDim + d) * a.Mp + fr0 + tc * 4);
        }
        if (tid < 64) {
            const int fr = fr0 + (tid & 31);
            const bool in = fr >= f0 && fr < f0 + T;           // frames of the block outside the clip: P = exp2(-inf) = 0
            sl = in ? (tid < 32 ? a.lse[(size_t)head * a.M + fr] : a.dsum[(size_t)head * a.M + fr]) : (tid < 32 ? INFINITY : 0.f);
        }
    };
    auto lstore = [&](int buf) {
        float* Qr = lds + buf * DKV_STAGE;
        float* Dr = Qr + FT * LDR;
        float* Qt = Dr + FT * LDR;
        float* Dt = Qt + kHeadDim * LDT;
        float* Ls = Dt + kHeadDim * LDT;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f32x4*>(Qr + (rr + 16 * p) * LDR + rc * 4) = sq[p];
            *reinterpret_cast<f32x4*>(Dr + (rr + 16 * p) * LDR + rc * 4) = sd[p];
            *reinterpret_cast<f32x4*>(Qt + (tr + 32 * p) * LDT + tc * 4) = sqt[p];
            *reinterpret_cast<f32x4*>(Dt + (tr + 32 * p) * LDT + tc * 4) = sdt[p];
        }
        if (tid < 64) Ls[tid] = sl;
    };

    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }

    gload(0);
    lstore(0);
    __syncthreads();
    for (int i = 0; i < nt; ++i) {
        const int buf = i & 1;
        if (i + 1 < nt) gload(i + 1);
        const float* Qr = lds + buf * DKV_STAGE;
        const float* Dr = Qr + FT * LDR;
        const float* Qt = Dr + FT * LDR;
        const float* Dt = Qt + kHeadDim * LDT;
        const float* Ls = Dt + kHeadDim * LDT;

        // S[q][key] = Q K^T (raw), dP[q][key] = dO V^T : A rows = the tile's 32 frames, B = this lane's key
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            half8 ah, al, bh, bl;
            frag_row(Qr + l31 * LDR, st, kg, ah, al);
            frag_row(Dr + l31 * LDR, st, kg, bh, bl);
            mma3<TERMS>(s, ah, al, kh[st], kl[st]);
            mma3<TERMS>(dp, bh, bl, vh[st], vl[st]);
        }
        // P = exp2(S c - lse2[q]), dS = P (dP - D[q]); register r <-> frame krow(r, kg) of the tile
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + 8 * r4 + 4 * kg);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ls + FT + 8 * r4 + 4 * kg);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r4 * 4 + j;
                // P goes into dV = P^T dO as 2^14 P (<= 16384, inside f16): probabilities far below 1 stay out of the
                // f16 subnormal range when split (as in the forward); the factor comes out again at the final store
                const float pp = exp2_(fmaf(s[r], kC2, kPShift - l4[j]));
                s[r] = pp;
                dp[r] = (pp * kPUnshift) * (dp[r] - d4[j]);
            }
        }
        // dV^T += dO^T P, dK^T += Q^T dS : contraction over the tile's 32 frames = 2 k-steps in the C/D permutation
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            half8 ph, pl, gh, gl;
            split8t<TERMS>(s, 8 * sp, ph, pl);
            split8t<TERMS>(dp, 8 * sp, gh, gl);
            half8 ah, al;
            frag_frames(Dt + l31 * LDT, sp, kg, ah, al);
            mma3<TERMS>(dv0, ah, al, ph, pl);
            frag_frames(Dt + (32 + l31) * LDT, sp, kg, ah, al);
            mma3<TERMS>(dv1, ah, al, ph, pl);
            frag_frames(Qt + l31 * LDT, sp, kg, ah, al);
            mma3<TERMS>(dk0, ah, al, gh, gl);
            frag_frames(Qt + (32 + l31) * LDT, sp, kg, ah, al);
            mma3<TERMS>(dk1, ah, al, gh, gl);
        }
        if (i + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }

    if (kv) {
        const size_t koff = (size_t)(f0 + key) * QKV_LD + kDim + head * kHeadDim;
        float* dKg = a.dqkv + koff;
        float* dVg = dKg + kDim;
        uint16_t* dKh = a.dqkv16 + koff;
        uint16_t* dVh = dKh + kDim;
        const float os = a.out16 ? *a.out_scale : 1.f;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = 8 * r4 + 4 * kg;
            f32x4 k_lo, k_hi, v_lo, v_hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                k_lo[j] = dk0[r4 * 4 + j] * kScale; k_hi[j] = dk1[r4 * 4 + j] * kScale;
                v_lo[j] = dv0[r4 * 4 + j] * kPUnshift; v_hi[j] = dv1[r4 * 4 + j] * kPUnshift;
            }
            bwd_store4(dKg + d, dKh + d, k_lo, os, a.out16);
            bwd_store4(dKg + 32 + d, dKh + 32 + d, k_hi, os, a.out16);
            bwd_store4(dVg + d, dVh + d, v_lo, os, a.out16);
            bwd_store4(dVg + 32 + d, dVh + 32 + d, v_hi, os, a.out16);
        }
    }
}

// ---- dQ: a lane owns a query --------------------------------------------------------------------------------------------------
constexpr int DQ_STAGE = 2 * FT * LDR + kHeadDim * LDT;                         // Kr, Vr, Kt
constexpr size_t DQ_LDS = 2 * DQ_STAGE * sizeof(float) > 4 * 32 * LDR * sizeof(float) ? 2 * DQ_STAGE * sizeof(float) : 4 * 32 * LDR * sizeof(float);

template <int TERMS>
__global__ __launch_bounds__(256, 2) void attn3_bwd_dq_kernel(Bwd3Args a, int nqb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int slot = jj / nqb, qb = jj % nqb;
    const int unit = slot * 8 + xcd;
    const int head = unit % kHeads, b = unit / kHeads;
    if (b >= a.B) return;
    const int f0 = a.frame_offsets[b];
    const int T = a.frame_offsets[b + 1] - f0;
    const int q0 = qb * 128;
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg = lane >> 5;

    // this lane's query: Q and dO fragments (B operands of S^T = K Q^T and dP^T = V dO^T), lse2, D
    const int q = q0 + wave * 32 + l31;
    const bool qv = q < T;
    half8 qh[4], ql[4], oh[4], ol[4];
    {
        const float* qrow = a.R + (size_t)(f0 + (qv ? q : 0)) * QKV_LD + head * kHeadDim;
        const float* drow = a.D + (size_t)(f0 + (qv ? q : 0)) * kDim + head * kHeadDim;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            frag_row(qrow, s, kg, qh[s], ql[s]);
            frag_row(drow, s, kg, oh[s], ol[s]);
            if (!qv) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { qh[s][i] = (half_t)0; ql[s][i] = (half_t)0; oh[s][i] = (half_t)0; ol[s][i] = (half_t)0; }
            }
        }
    }
    const float lse2 = qv ? a.lse[(size_t)head * a.M + f0 + q] : INFINITY;
    const float dsum = qv ? a.dsum[(size_t)head * a.M + f0 + q] : 0.f;

    const int g0 = f0 / FT, g1 = (f0 + T - 1) / FT;
    const int nt = g1 - g0 + 1;
    const int rr = tid >> 4, rc = tid & 15;
    const int tr = tid >> 3, tc = tid & 7;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 sk[2], sv[2], skt[2];
    auto gload = [&](int i) {
        const int fr0 = (g0 + i) * FT;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int fr = fr0 + rr + 16 * p;
            const bool ok = fr < a.M;
            const float* row = a.R + (size_t)fr * QKV_LD + head * kHeadDim + rc * 4;
            sk[p] = ok ? *reinterpret_cast<const f32x4*>(row + kDim) : zero4;
            sv[p] = ok ? *reinterpret_cast<const f32x4*>(row + 2 * kDim) : zero4;
            skt[p] = *reinterpret_cast<const f32x4*>(a.Rt + (size_t)(kDim + head * kHeadDim + tr + 32 * p) * a.Mp + fr0 + tc * 4);
        }
    };
    auto lstore = [&](int buf) {
        float* Kr = lds + buf * DQ_STAGE;
        float* Vr = Kr + FT * LDR;
        float* Kt = Vr + FT * LDR;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f32x4*>(Kr + (rr + 16 * p) * LDR + rc * 4) = sk[p];
            *reinterpret_cast<f32x4*>(Vr + (rr + 16 * p) * LDR + rc * 4) = sv[p];
            *reinterpret_cast<f32x4*>(Kt + (tr + 32 * p) * LDT + tc * 4) = skt[p];
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

    gload(0);
    lstore(0);
    __syncthreads();
    for (int i = 0; i < nt; ++i) {
        const int buf = i & 1;
        if (i + 1 < nt) gload(i + 1);
        const float* Kr = lds + buf * DQ_STAGE;
        const float* Vr = Kr + FT * LDR;
        const float* Kt = Vr + FT * LDR;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            half8 ah, al, bh, bl;
            frag_row(Kr + l31 * LDR, st, kg, ah, al);
            frag_row(Vr + l31 * LDR, st, kg, bh, bl);
            mma3<TERMS>(s, ah, al, qh[st], ql[st]);              // S^T[key][q]
            mma3<TERMS>(dp, bh, bl, oh[st], ol[st]);             // dP^T[key][q]
        }
        const int fr0 = (g0 + i) * FT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fr = fr0 + krow(r, kg);
            const float p = (fr >= f0 && fr < f0 + T) ? exp2_(fmaf(s[r], kC2, -lse2)) : 0.f;
            dp[r] = p * (dp[r] - dsum);                   // dS^T
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            half8 gh, gl, ah, al;
            split8t<TERMS>(dp, 8 * sp, gh, gl);
            frag_frames(Kt + l31 * LDT, sp, kg, ah, al);
            mma3<TERMS>(o0, ah, al, gh, gl);                     // dQ^T[d][q] += K^T dS^T
            frag_frames(Kt + (32 + l31) * LDT, sp, kg, ah, al);
            mma3<TERMS>(o1, ah, al, gh, gl);
        }
        if (i + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }

    float* patch = lds + wave * (32 * LDR);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = krow(r, kg);
        patch[l31 * LDR + d] = o0[r] * kScale;
        patch[l31 * LDR + 32 + d] = o1[r] * kScale;
    }
    __syncthreads();
    const size_t ooff = (size_t)f0 * QKV_LD + head * kHeadDim;
    const float os = a.out16 ? *a.out_scale : 1.f;
    const int orow = lane >> 4, ocol = (lane & 15) * 4;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int qlr = orow + 4 * p;
        const int qq = q0 + wave * 32 + qlr;
        if (qq < T)
            bwd_store4(a.dqkv + ooff + (size_t)qq * QKV_LD + ocol, a.dqkv16 + ooff + (size_t)qq * QKV_LD + ocol,
                       *reinterpret_cast<const f32x4*>(patch + qlr * LDR + ocol), os, a.out16);
    }
}

}  // namespace

template <int TERMS>
static hipError_t launch_bwd(const Bwd3Args& a, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_bwd_dkv_kernel<TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKV_LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_bwd_dq_kernel<TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DQ_LDS);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    const int units = a.B * kHeads, slots = (units + 7) / 8;
    const int nb = (a.max_frames + 127) / 128;
    hipLaunchKernelGGL(attn3_bwd_dkv_kernel<TERMS>, dim3((unsigned)(slots * nb * 8)), dim3(256), DKV_LDS, s, a, nb);
    hipLaunchKernelGGL(attn3_bwd_dq_kernel<TERMS>, dim3((unsigned)(slots * nb * 8)), dim3(256), DQ_LDS, s, a, nb);
    return hipGetLastError();
}

hipError_t launch_attention_bwd_f16x3(const float* R, const float* Rt, const float* D, const float* Dt, const float* lse, const float* dsum,
                                      const int32_t* frame_offsets, int B, int max_frames, int M, int Mp, float* dqkv, int hi_only, hipStream_t s,
                                      void* dqkv16, const float* out_scale, int out16) {
    if (B <= 0 || max_frames <= 0 || M <= 0) return hipSuccess;
    Bwd3Args a{R, Rt, D, Dt, lse, dsum, dqkv, frame_offsets, B, max_frames, M, Mp, static_cast<uint16_t*>(dqkv16), out_scale, out16};
    return hi_only == 2 ? launch_bwd<2>(a, s) : hi_only ? launch_bwd<1>(a, s) : launch_bwd<3>(a, s);
}
