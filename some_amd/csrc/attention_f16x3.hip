// fp32-equivalent flash attention on the f16 matrix pipe (3-term split, v_mfma_f32_32x32x16_f16).
//
// Same operator as attention.hip (F.scaled_dot_product_attention, 8 heads x 64, per clip, unmasked;
// modules/attention/base_attention.py:34-44) and the same register-only online softmax built on the transposed
// products  S^T = K Q^T  and  O^T = V^T P^T  (a lane owns one query; P^T in the MFMA C/D layout IS the B operand
// of the second product).  What changes is the arithmetic of the two products:
//     s = kh*qh + kh*ql + kl*qh,      o += vh*ph + vh*pl + vl*ph          (x = xh + xl, two f16 halves, fp32 acc)
// 48 MFMAs of 32 cycles per 64-key tile instead of 128 of 64 cycles.
//
// Operands arrive pre-split from the QKV projection's epilogue (gemm_f16x3.hip, EPI_QKV):
//   Q, K : [M, 512] SPLIT32 rows (split.h) - a head's 64 dims are 256 contiguous bytes (2 k-blocks of hi|lo);
//   V^T  : hi and lo f16 planes [512, ldv] with the FRAME index contiguous (every aligned group of 16 frames stored as
//          its quarters 0, 2, 1, 3), so a (d, 8-key) operand fragment is ONE ds_read_b128 and no transposition happens here.
// Inference: Q / K rows and V^T columns lie in CLIP-ALIGNED coordinates - clip b starts at pad_offsets[b], a multiple of 16
// (internal.h: launch_attn_plan; the QKV projection gathers its input rows accordingly) - and key tile i of a clip covers its
// frames 64 i .. 64 i + 63: every 16-byte V^T chunk and every permuted group of 16 stays aligned however the clips are packed,
// and the sequence of online-softmax steps a query sees depends on its clip alone, so a clip's result is bit-identical whatever
// it is batched with (the reference runs every chunk by itself, inference/base_infer.py:46-53).  Training forward
// (pad_offsets = nullptr): operands in packed coordinates, key tiles aligned in GLOBAL 64-frame blocks.
// Keys outside the clip get score -inf (their data is another clip's finite values or the zero rows the GEMM wrote, so
// 0 * v stays 0).
// The softmax scale 64^-0.5 * log2(e) is folded into the exp2 argument: p = exp2(fma(s, c, -m)).
#include <cstdlib>

#include "internal.h"
#include "split.h"

namespace {

constexpr int KT = 64;
constexpr int LDR = 68;                         // LDS row (dwords): 64 data + 4 pad
constexpr int K_DW = KT * LDR;                  // K tile: [64 keys][hi|lo hi|lo]
constexpr int V_DW = kHeadDim * LDR;            // V^T tile: [64 d][64 keys hi | 64 keys lo]
constexpr size_t LDS_BYTES = 2 * (K_DW + V_DW) * sizeof(float);
constexpr float kPShift = 14.f;                 // P is carried as 2^kPShift p <= 2^14 (see softmax)

__device__ __forceinline__ float exp2_(float x) { return __builtin_amdgcn_exp2f(x); }

// Round 4, measured and not kept (tools/patches/r04_attention_psplit_fma_mix.patch, profiles/r04_experiments.md): the P split as four
// v_fma_mixlo_f16 / v_fma_mixhi_f16 per two probabilities instead of and / and / sub / sub / cvt_pkrtz / cvt_pkrtz - 64 instead of 96 VALU
// instructions per key tile, and 2.55 vs 2.40 ms (the mix instructions write one half of their destination: a read-modify-write chain per
// dword, and they do not issue at the plain-VALU rate next to MFMAs).
// Measured and not kept (profiles/r02_experiments.md; the switches live in the git history, not in the shipped source): lazy rescale
// (2.59 vs 2.53 ms), packed-fp32 softmax (2.59), global loads pinned to the head of the step (noise), 64 queries per wavefront at
// one wavefront per SIMD (3.21 vs 2.62), priority over QK + softmax instead of PV (+1 %).  Kept: V^T tiles through buffer loads,
// s_setprio 1 over the PV product, scalar fp32 softmax arithmetic (-fno-slp-vectorize, build.py).

// v_max3_f32 without the v_max_f32 x, x, x canonicalisation clang puts in front of every fmaxf operand (scores are
// MFMA results or -inf, never signalling NaNs)
__device__ __forceinline__ float max2_(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float max3_(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// p = hi + lo.  hi = p with the 13 low mantissa bits cleared (the f16 round-toward-zero value of p, exact in f16 while
// p >= 2^-14), lo = rtz_f16(p - hi): |p - hi - lo| < 2^-21 p.  Packed fp32 subtract, packed converts: 5 VALU
// instructions per two probabilities.  hi need not be the NEAREST f16 - same error class as split_f16.
__device__ __forceinline__ void split8(const f32x16& p, int base, half8& h, half8& l) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const float p0 = p[base + i], p1 = p[base + i + 1];      // (scalars: bit_cast of a vector element miscompiles)
        const float hf[2] = {__uint_as_float(__float_as_uint(p0) & 0xFFFFE000u), __uint_as_float(__float_as_uint(p1) & 0xFFFFE000u)};
        const float lf[2] = {p0 - hf[0], p1 - hf[1]};
        const half2_t hh = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(hf[0], hf[1]));
        const half2_t ll = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(lf[0], lf[1]));
        h[i] = hh[0]; h[i + 1] = hh[1];
        l[i] = ll[0]; l[i + 1] = ll[1];
    }
}

// bf16 operand mode (TERMS = 1 only): hi = bf16(p) round-to-nearest in the hi slots, no lo
__device__ __forceinline__ void split8_bf16(const f32x16& p, int base, half8& h) {
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = bf16_as_half(p[base + i]);
}

// TRAIN = true: the training forward (train_api: some_train_attention_fwd_f16x3) - Q / K rows come straight from
// split_rows(qkv) (row stride 6144 B), V^T from transpose(qkv, split) (SPLIT32 over frames: 32-frame blocks [32 hi | 32 lo]),
// the output is fp32 and the base-2 log-sum-exp is stored for the backward.
// 128 queries per workgroup (32 per wavefront), two workgroups per CU: two wavefronts per SIMD cover each other's stalls.
template <bool TRAIN, int TERMS = 3, bool BF16 = false>      // BF16: bf16 hi halves, TERMS = 1 only (split.h)
__global__ __launch_bounds__(256, 2) void attention3_kernel(Attn3Args a, int nqb) {
    constexpr int QB = 128;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int slot = jj / nqb, qb = jj % nqb;
    const int unit = slot * 8 + xcd;
    const int hg = unit % (kHeads * a.groups), b = unit / (kHeads * a.groups);
    if (b >= a.B) return;
    const int head = hg % kHeads, g = hg / kHeads;
    const int fo = a.frame_offsets[b];                               // output rows (packed coordinates)
    const int T = a.frame_offsets[b + 1] - fo;
    const int f0 = TRAIN ? fo : a.pad_offsets[b];                    // operand rows / V^T columns
    const int q0 = qb * QB;
    if (q0 >= T) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    constexpr size_t ROW_B = TRAIN ? 6144 : 2048;       // bytes between consecutive frames of Q / K
    const char* __restrict__ Qp = reinterpret_cast<const char*>(a.q[g]) + head * 256;
    const char* __restrict__ Kp = reinterpret_cast<const char*>(a.k[g]) + head * 256;
    const char* __restrict__ Vh = reinterpret_cast<const char*>(a.vt[g]) + (size_t)head * kHeadDim * a.ldv * (TRAIN ? 4 : 2);
    const char* __restrict__ Vl = Vh + (size_t)kDim * a.ldv * 2;                          // (planes format only)

    // ---- Q fragments (B operand of S^T): slab s covers d = 16 s .. 16 s + 15; lane half kg holds 8 of them
    half8 qh[4], ql[4];
    {
        const int q = q0 + wave * 32 + l31;
        const bool qv = q < T;
        const char* row = Qp + (size_t)(f0 + (qv ? q : 0)) * ROW_B;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int off = (s >> 1) * 128 + (s & 1) * 32 + kg * 16;
            qh[s] = *reinterpret_cast<const half8*>(row + off);
            ql[s] = *reinterpret_cast<const half8*>(row + off + 64);
            if (!qv) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { qh[s][i] = (half_t)0; ql[s][i] = (half_t)0; }
            }
        }
    }

    // ---- staging roles: a K tile is 64 rows x 16 chunks, a V^T tile 64 rows x (8 hi + 8 lo) chunks; 4 + 4 per thread
    const int srow = tid >> 4, sc = tid & 15;       // rows srow + 16 p, chunk sc
    // first key of tile 0: the clip's own first frame (clip-aligned operands) or the global 64-frame block it lies in
    const int kt0 = TRAIN ? f0 / KT * KT : f0;
    const int n = (f0 + T - kt0 + KT - 1) / KT;     // key tiles of this clip
    f32x4 rk[4], rv[4];
    // K rows through a buffer descriptor: keys >= M (the tail of the last global tile) lie past num_records and read as
    // zeros - no exec-masked branch in the loop
    const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(Kp), 0, (int)((size_t)a.M * ROW_B - head * 256 > 0x7fffffffull ? 0x7fffffffull : (size_t)a.M * ROW_B - head * 256), 0x00020000);
    auto gload_k = [&](int i) {
        const int g0 = kt0 + i * KT;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            rk[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsk, (uint32_t)(g0 + srow + 16 * p) * (uint32_t)ROW_B + sc * 16u, 0, 0));
    };
    // V^T rows (inference layout) through a buffer descriptor as well: per-thread 32-bit offsets fixed for the whole kernel,
    // the tile position as the scalar offset - no 64-bit address arithmetic in the loop (every V^T read is in bounds: ldv pads M)
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(Vh), 0, (int)((size_t)(2 * kDim - head * kHeadDim) * a.ldv * 2), 0x00020000);
    uint32_t voff_v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
        voff_v[p] = (uint32_t)(srow + 16 * p) * (uint32_t)a.ldv * 2u + (uint32_t)(sc & 7) * 16u + (sc < 8 ? 0u : (uint32_t)kDim * (uint32_t)a.ldv * 2u);
    auto gload_v = [&](int i) {
        const int g0 = kt0 + i * KT;
        if constexpr (!TRAIN) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                rv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsv, voff_v[p], (uint32_t)g0 * 2u, 0));
            return;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const char* vsrc = TRAIN ? Vh + ((size_t)(srow + 16 * p) * a.ldv + g0) * 4 + sc * 16      // two 32-frame blocks of hi | lo
                                     : (sc < 8 ? Vh : Vl) + ((size_t)(srow + 16 * p) * a.ldv + g0) * 2 + (sc & 7) * 16;
            rv[p] = *reinterpret_cast<const f32x4*>(vsrc);
        }
    };
    // LDS: K ring [2] then V^T ring [2]; the K ring runs one tile ahead of the V^T ring
    auto kbuf = [&](int i) { return lds + (i & 1) * K_DW; };
    auto vbuf = [&](int i) { return lds + 2 * K_DW + (i & 1) * V_DW; };
    auto lstore_k = [&](int i) {
        float* Ks = kbuf(i);
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(Ks + (srow + 16 * p) * LDR + sc * 4) = rk[p];
    };
    auto lstore_v = [&](int i) {
        float* Vs = vbuf(i);
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(Vs + (srow + 16 * p) * LDR + sc * 4) = rv[p];
    };

    // S^T = K Q^T (raw, unscaled) for tile i, both 32-key sub-tiles, then -inf outside the clip
    auto qk = [&](int i, f32x16& s0, f32x16& s1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        const float* kp = kbuf(i) + l31 * LDR + kg * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int off = (s >> 1) * 32 + (s & 1) * 8;
            const half8 kh0 = *reinterpret_cast<const half8*>(kp + off);
            const half8 kl0 = *reinterpret_cast<const half8*>(kp + off + 16);
            const half8 kh1 = *reinterpret_cast<const half8*>(kp + 32 * LDR + off);
            const half8 kl1 = *reinterpret_cast<const half8*>(kp + 32 * LDR + off + 16);
            if (TERMS == 3) {
                s0 = mfma_hi<BF16>(kl0, qh[s], s0);
                s1 = mfma_hi<BF16>(kl1, qh[s], s1);
                s0 = mfma_hi<BF16>(kh0, ql[s], s0);
                s1 = mfma_hi<BF16>(kh1, ql[s], s1);
            }
            s0 = mfma_hi<BF16>(kh0, qh[s], s0);
            s1 = mfma_hi<BF16>(kh1, qh[s], s1);
        }
    };
    auto mask_tile = [&](int i, f32x16& s0, f32x16& s1) {
        const int gbase = kt0 + i * KT;
        if (gbase < f0 || gbase + KT > f0 + T) {          // last tile (and the first one of a clip inside a global block)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k0 = gbase + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (k0 < f0 || k0 >= f0 + T) s0[r] = -INFINITY;
                if (k0 + 32 < f0 || k0 + 32 >= f0 + T) s1[r] = -INFINITY;
            }
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float c = 0.125f * 1.4426950408889634f;       // head_dim^-0.5 * log2(e)

    // online softmax of one tile's scores (lane = query; scale folded into the exponent): s0/s1 become P
    auto softmax = [&](f32x16& s0, f32x16& s1) {
        float mxa = max2_(s0[0], s1[0]), mxb = max2_(s0[1], s1[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
            mxa = max3_(mxa, s0[r], s1[r]);
            mxb = max3_(mxb, s0[r + 1], s1[r + 1]);
        }
        float mx = max2_(mxa, mxb);
        mx = max2_(mx, __shfl_xor(mx, 32, 64));
        const float m_new = max2_(m_run, mx * c);      // finite: every tile holds at least one key of the clip
        const float alpha = exp2_(m_run - m_new);       // first tile: exp2(-inf) = 0
        m_run = m_new;
        // probabilities are kept scaled by 2^kPShift (<= 16384, inside f16): keys far below the running maximum
        // stay out of the f16 subnormal range when P is split; the scale cancels in O / l
        const float nm = kPShift - m_new;
        float ps[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = exp2_(fmaf(s0[r], c, nm));
            s1[r] = exp2_(fmaf(s1[r], c, nm));
            ps[r & 1] += s0[r] + s1[r];
        }
        l_run = l_run * alpha + (ps[0] + ps[1]);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    };
    // O^T += V^T P^T for tile i.  P slab (sub, s'): registers r = 8 s' .. 8 s' + 7 of the sub-tile hold keys
    // 16 s' + {0..3} + 4 kg and 16 s' + 8 + {0..3} + 4 kg  ->  two ds_read_b64 per V^T fragment
    auto pv = [&](int i, const f32x16& s0, const f32x16& s1) {
        const float* vp = vbuf(i) + l31 * LDR + (TRAIN ? 2 : 4) * kg;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                half8 ph, pl;
                if constexpr (BF16) split8_bf16(sub == 0 ? s0 : s1, 8 * sp, ph);
                else split8(sub == 0 ? s0 : s1, 8 * sp, ph, pl);
                // dword offset of key 32 sub + 16 s' and of its lo half: row = [64 hi | 64 lo] (planes) or
                // [32 hi | 32 lo][32 hi | 32 lo] (SPLIT32 over frames)
                constexpr int LO = TRAIN ? 16 : 32;
                const int kd = (TRAIN ? 32 : 16) * sub + 8 * sp;
                half8 vh0, vl0, vh1, vl1;
                if constexpr (TRAIN) {
                    const half4 a0 = *reinterpret_cast<const half4*>(vp + kd);
                    const half4 a1 = *reinterpret_cast<const half4*>(vp + kd + 4);
                    const half4 b0 = *reinterpret_cast<const half4*>(vp + LO + kd);
                    const half4 b1 = *reinterpret_cast<const half4*>(vp + LO + kd + 4);
                    const half4 c0 = *reinterpret_cast<const half4*>(vp + 32 * LDR + kd);
                    const half4 c1 = *reinterpret_cast<const half4*>(vp + 32 * LDR + kd + 4);
                    const half4 d0 = *reinterpret_cast<const half4*>(vp + 32 * LDR + LO + kd);
                    const half4 d1 = *reinterpret_cast<const half4*>(vp + 32 * LDR + LO + kd + 4);
                    vh0 = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    vl0 = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    vh1 = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7);
                    vl1 = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7);
                } else {
                    // the QKV epilogue stores every 16-frame group as quarters 0, 2, 1, 3: this lane half's 8 keys
                    // (4 kg + {0..3}, 8 + 4 kg + {0..3}) are the 16 bytes at 4 kg dwords - one conflict-free ds_read_b128
                    vh0 = *reinterpret_cast<const half8*>(vp + kd);
                    vl0 = *reinterpret_cast<const half8*>(vp + LO + kd);
                    vh1 = *reinterpret_cast<const half8*>(vp + 32 * LDR + kd);
                    vl1 = *reinterpret_cast<const half8*>(vp + 32 * LDR + LO + kd);
                    // the four reads of a slab together: left alone hipcc puts each read right in front of its first MFMA
                    // (register pressure) and every MFMA waits out an LDS round trip: 2.74 -> 2.51 ms with this fence
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (TERMS == 3) {
                    o0 = mfma_hi<BF16>(vl0, ph, o0);
                    o1 = mfma_hi<BF16>(vl1, ph, o1);
                    o0 = mfma_hi<BF16>(vh0, pl, o0);
                    o1 = mfma_hi<BF16>(vh1, pl, o1);
                }
                o0 = mfma_hi<BF16>(vh0, ph, o0);
                o1 = mfma_hi<BF16>(vh1, ph, o1);
            }
        }
    };

    // ---- software pipeline (cdna guide T15): in iteration i the matrix pipe computes the NEXT tile's scores
    // QK(i+1) while the VALU turns the CURRENT tile's scores into probabilities, then PV(i).  One barrier per
    // tile; registers carry K(i+2) / V(i+1) across it (write after the barrier, re-issue at once).
    gload_k(0);
    gload_v(0);
    lstore_k(0);
    lstore_v(0);
    if (n > 1) { gload_k(1); lstore_k(1); gload_v(1); }
    if (n > 2) gload_k(2);
    __syncthreads();
    f32x16 sa0, sa1, sb0, sb1;        // scores of even / odd tiles (ping-pong: no register copies)
    auto step = [&](int i, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
        if (i + 2 < n) lstore_k(i + 2);               // K ring slot i & 1: last read by QK(i) in the previous step
        lstore_v(i + 1);                              // V ring slot (i+1) & 1: last read by PV(i-1)
        if (i + 3 < n) gload_k(i + 3);
        if (i + 2 < n) gload_v(i + 2);
        qk(i + 1, n0, n1);                            // matrix pipe ...
        softmax(c0, c1);                              // ... overlapped with the VALU (independent of QK(i+1))
        pv(i, c0, c1);
        mask_tile(i + 1, n0, n1);
        __syncthreads();
    };
    // the same step for interior tiles (i + 3 < n): nothing conditional, tile i + 1 needs no masking - ONE basic block,
    // so the scheduler is free to spread the staging traffic and the fragment reads between the MFMAs
    auto step_full = [&](int i, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
        lstore_k(i + 2);
        lstore_v(i + 1);
        gload_k(i + 3);
        gload_v(i + 2);
        qk(i + 1, n0, n1);
        softmax(c0, c1);
        __builtin_amdgcn_s_setprio(1);                // raised priority over the PV product (pure MFMA + LDS reads): -3 % with
        pv(i, c0, c1);                                // -fno-slp-vectorize, which also frees the registers the flips would spill
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    };
    qk(0, sa0, sa1);
    mask_tile(0, sa0, sa1);
    // QK(0) read K ring slot 0, and the first step's lstore_k(2) overwrites that slot: without this barrier a fast
    // wavefront races ahead of a slow one still reading its K(0) fragments (seen as rare large errors, run-to-run
    // differences, on clips whose first key tile needs no masking - nothing else sat between the two)
    __syncthreads();
    int i = 0;
    if constexpr (!TRAIN) {       // (the training variants sit at the register limit: the extra loop body makes them spill)
        for (; i + 4 < n; i += 2) {
            step_full(i, sa0, sa1, sb0, sb1);
            step_full(i + 1, sb0, sb1, sa0, sa1);
        }
    }
    for (; i + 2 < n; i += 2) {
        step(i, sa0, sa1, sb0, sb1);
        step(i + 1, sb0, sb1, sa0, sa1);
    }
    if (i + 1 < n) {
        step(i, sa0, sa1, sb0, sb1);
        softmax(sb0, sb1);
        pv(n - 1, sb0, sb1);
    } else {
        softmax(sa0, sa1);
        pv(n - 1, sa0, sa1);
    }
    __syncthreads();

    // ---- normalise, transpose through LDS (wave-private 32 x 64 patch), SPLIT32 row stores
    float* patch = lds + wave * (32 * LDR);
    {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * kg;
            patch[l31 * LDR + d] = o0[r] * inv;
            patch[l31 * LDR + 32 + d] = o1[r] * inv;
        }
        __syncthreads();
        const int orow = lane >> 4, ocol = (lane & 15) * 4;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int ql_ = orow + 4 * p;
            const int q = q0 + wave * 32 + ql_;
            if (q < T) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(patch + ql_ * LDR + ocol);
                if (TRAIN) {
                    *reinterpret_cast<f32x4*>(a.out32[g] + (size_t)(fo + q) * kDim + head * kHeadDim + ocol) = v;
                } else {
                    half4 hh, ll;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { half_t h, l; split_f16(v[i], h, l); hh[i] = h; ll[i] = l; }
                    char* row = reinterpret_cast<char*>(a.out[g]) + ((size_t)(fo + q) * kDim + head * kHeadDim) * 4 +
                                (ocol >> 5) * 128 + (ocol & 31) * 2;
                    *reinterpret_cast<half4*>(row) = hh;
                    *reinterpret_cast<half4*>(row + 64) = ll;
                }
            }
        }
        if (TRAIN && kg == 0) {       // P was carried as 2^kPShift p: lse2 = max + log2(sum p)
            const int q = q0 + wave * 32 + l31;
            if (q < T) a.lse[g][(size_t)head * a.M + fo + q] = m_run + __log2f(l_tot) - kPShift;
        }
    }
}


// =====================================================================================================================
// Round 5: the inference kernel with the instruction stream PLACED.  profiles/r05_coissue_probe2.md: on a gfx950 SIMD up to five
// plain VALU instructions (v_exp_f32 counts double) and at most one ds_read_b128 hide in the 32-cycle shadow of every
// v_mfma_f32_32x32x16_f16 - 33.5 - 35.5 cycles per MFMA - but only where they are PUT there; the compiler's own order of the kernel
// above (7 VALU per gap in the QK product, a 32-instruction O-rescale blob with the matrix pipe idle, back-to-back MFMA pairs in the PV
// product) runs at 48.7.  Same operator, same operand layouts, same tiling (128 queries per workgroup, 32 per wavefront, 64-key tiles
// counted from the clip's first frame, K ring one tile ahead of the V^T ring, one barrier per tile); what changes:
//   * a tile step is 48 SLOTS = one MFMA + its share of everything else, fenced by sched_barrier(0): phase A = the 24 MFMAs of
//     S(i+1) = K(i+1) Q^T beside exp2 of tile i and the P split of its first two slabs; phase B = the 24 MFMAs of O += V(i)^T P(i)^T
//     beside the other two slabs' split, the row sums, the row maximum of S(i+1) and the K / V^T staging traffic.  About 4 VALU per
//     slot by construction, fragment reads issued one slab ahead into the registers the previous slab has just retired;
//   * the P split is 4 instructions per two probabilities instead of 6: hi = cvt_pkrtz(p0, p1), lo_j = p_j - f32(hi_j) as ONE
//     v_fma_mix_f32 each (f16 source, full f32 destination - not the half-register v_fma_mixlo_f16 that round 4 measured slower),
//     lo = cvt_pkrtz(lo0, lo1).  hi is the round-toward-zero f16 of p (what the mask produced), lo the rtz f16 of the exact rest;
//   * LAZY rescale: the running maximum moves only when a tile's row maximum exceeds it by more than kLazy (2^4.5 in p) for some
//     query of the wavefront - a wavefront-uniform branch between tiles; otherwise O and l are left alone (no 32 multiplies per tile)
//     and P is taken against the old maximum.  P is carried as 2^kPShiftI p <= 2^(11 + 4.5) < 65504, so the f16 halves cannot
//     overflow.  The decision depends on the wavefront's 32 queries and their clip's keys only: results stay independent of packing.
#ifndef SOME_ATTN_ABL
#define SOME_ATTN_ABL 0      // measurement builds only (tools/build_variant.py -DSOME_ATTN_ABL=mask): 1 no softmax VALU, 2 no fragment reads,
#endif                       // 4 no staging traffic, 8 no barrier in the loop - results are garbage, only the time means something
constexpr int kAbl = SOME_ATTN_ABL;
#ifndef SOME_ATTN_DBG_WG
#define SOME_ATTN_DBG_WG 2048      // (timeline builds, tools/attn_probe.hip) a workgroup of the second round
#endif
constexpr int kStageSlot = 8;      // slots 8 - 15 of phase A carry the tile DMA (0 / 4 / 8 / 16 measured the same: profiles/r05m_attn_env.txt)
constexpr float kPShiftI = 11.f;
constexpr float kLazy = 4.5f;

// p - f32(half of hi_pk) as ONE v_fma_mix_f32: hipcc selects it for fma(fpext(f16), f32, f32) when the multiplier is not a constant it
// can fold (`mone` = -1.0f built from a kernel argument), and - unlike inline asm - then knows the one-wait-state hazard between a
// v_fma_mix write and a dependent read, so it puts an independent instruction there instead of an s_nop.
__device__ __forceinline__ float mix_sub_(half_t h, float mone, float p) { return __builtin_fmaf((float)h, mone, p); }

template <bool V>
struct Flag { static constexpr bool value = V; };

// DMA = true: the K / V^T tiles go from L2 straight into the LDS rings (buffer_load_dwordx4 ... lds: no staging registers, no
// ds_write_b128 - whose VGPR -> LDS transfer blocks the SIMD pair's LDS path for 13+ cycles each, MI355X_MICROARCH.md LDS section).
// A DMA piece is 1 KiB of CONTIGUOUS LDS (lane t -> 16 bytes at 16 t), so the tiles are unpadded 256-byte rows and bank conflicts are
// avoided by an XOR swizzle instead of padding: chunk c (16 bytes) of row r sits at chunk position c ^ (r & 15) - the 16 lanes of a
// ds_read_b128 lane group hold 16 rows with distinct r & 15, i.e. 16 distinct positions = all 64 banks.  The swizzle costs nothing
// on the write side (a lane fetches the global chunk that belongs at its position) and nothing on the read side (the eight chunk
// offsets a lane ever needs - kg ^ (r & 15) ^ {0, 2, .. 14} - are kept in eight registers; ring slot and row + 32 are immediates).
// FAST (SOME_PRECISION_F16X3_FAST, opt-in - never the default): products the softmax average is least sensitive to are left out.
//   FAST >= 1: P V without its `vh * pl` term - P enters as ph = rn_f16(2^11 p) alone (round to NEAREST: no bias; the shipped split
//              truncates ph because pl carries the remainder), so 16 instead of 24 MFMAs in phase B and one conversion instead of the
//              four-instruction split per pair.  O = sum ph v / sum p: relative error <= 2^-12 per term, unbiased.
//   FAST == 2: additionally Q K^T without `kh * ql` (16 MFMAs in phase A): the scores lose ~2^-12 |s| - measured, not shipped as a mode.
template <bool DMA, int FAST = 0>
__global__ __launch_bounds__(256, 2) void attention3i_kernel(Attn3Args a, int nqb) {
    constexpr int QB = 128;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int slot = jj / nqb, qb = jj % nqb;
    const int unit = slot * 8 + xcd;
    const int hg = unit % (kHeads * a.groups), b = unit / (kHeads * a.groups);
    if (b >= a.B) return;
    const int head = hg % kHeads, g = hg / kHeads;
    const int fo = a.frame_offsets[b];                               // output rows (packed coordinates)
    const int T = a.frame_offsets[b + 1] - fo;
    const int f0 = a.pad_offsets[b];                                 // operand rows / V^T columns (clip-aligned)
    const int q0 = qb * QB;
    if (q0 >= T) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    constexpr size_t ROW_B = 2048;
    const char* __restrict__ Qp = reinterpret_cast<const char*>(a.q[g]) + head * 256;
    const char* __restrict__ Kp = reinterpret_cast<const char*>(a.k[g]) + head * 256;
    const char* __restrict__ Vh = reinterpret_cast<const char*>(a.vt[g]) + (size_t)head * kHeadDim * a.ldv * 2;

    half8 qh[4], ql[4];
    {
        const int q = q0 + wave * 32 + l31;
        const bool qv = q < T;
        const char* row = Qp + (size_t)(f0 + (qv ? q : 0)) * ROW_B;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int off = (s >> 1) * 128 + (s & 1) * 32 + kg * 16;
            qh[s] = *reinterpret_cast<const half8*>(row + off);
            ql[s] = *reinterpret_cast<const half8*>(row + off + 64);
            if (!qv) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { qh[s][i] = (half_t)0; ql[s][i] = (half_t)0; }
            }
        }
    }

    const int srow = tid >> 4, sc = tid & 15;
    const int kt0 = f0;
    const int n = (T + KT - 1) / KT;
    f32x4 rk[4], rv[4];
    const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(Kp), 0, (int)((size_t)a.M * ROW_B - head * 256 > 0x7fffffffull ? 0x7fffffffull : (size_t)a.M * ROW_B - head * 256), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(Vh), 0, (int)((size_t)(2 * kDim - head * kHeadDim) * a.ldv * 2), 0x00020000);
    uint32_t voff_v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
        voff_v[p] = (uint32_t)(srow + 16 * p) * (uint32_t)a.ldv * 2u + (uint32_t)(sc & 7) * 16u + (sc < 8 ? 0u : (uint32_t)kDim * (uint32_t)a.ldv * 2u);
    // DMA roles: piece j = 4 p + wave (p < 4) of a tile = its rows 4 j .. 4 j + 3; lane t fills chunk position t & 15 of row 4 j + (t >> 4)
    constexpr int KD = DMA ? KT * 64 : K_DW;               // ring tile (dwords): unpadded rows when DMA
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int drow = 4 * wave + (lane >> 4);               // row within a 16-row group (the group index p goes into the scalar offset)
    const int dgc = (lane & 15) ^ (drow & 15);             // the global chunk that belongs at this lane's position
    const uint32_t dma_off_k = (uint32_t)drow * (uint32_t)ROW_B + (uint32_t)dgc * 16u;
    const uint32_t dma_off_v = (uint32_t)drow * (uint32_t)a.ldv * 2u + (uint32_t)(dgc & 7) * 16u + (dgc < 8 ? 0u : (uint32_t)kDim * (uint32_t)a.ldv * 2u);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto dma_k = [&](int i, int p) {
        float* dst = lds + (i & 1) * KD + (4 * p + wave_u) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk, (lds_ptr_t)dst, 16, dma_off_k, (uint32_t)(kt0 + i * KT + 16 * p) * (uint32_t)ROW_B, 0, 0);
    };
    auto dma_v = [&](int i, int p) {
        float* dst = lds + 2 * KD + (i & 1) * KD + (4 * p + wave_u) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (lds_ptr_t)dst, 16, dma_off_v, (uint32_t)(kt0 + i * KT) * 2u + (uint32_t)(16 * p) * (uint32_t)a.ldv * 2u, 0, 0);
    };
    // swizzled fragment offsets (dwords): row l31, chunk (kg | C) ^ (l31 & 15) for the even C = 2 c
    int xoff[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) xoff[cc] = l31 * 64 + (((kg ^ (l31 & 15)) ^ (2 * cc)) * 4);
    auto gload_k1 = [&](int i, int p) {
        rk[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsk, (uint32_t)(kt0 + i * KT + srow + 16 * p) * (uint32_t)ROW_B + sc * 16u, 0, 0));
    };
    auto gload_v1 = [&](int i, int p) {
        rv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsv, voff_v[p], (uint32_t)(kt0 + i * KT) * 2u, 0));
    };
    auto kbuf = [&](int i) { return lds + (i & 1) * K_DW; };
    auto vbuf = [&](int i) { return lds + 2 * K_DW + (i & 1) * V_DW; };
    auto lstore_k1 = [&](int i, int p) { *reinterpret_cast<f32x4*>(kbuf(i) + (srow + 16 * p) * LDR + sc * 4) = rk[p]; };
    auto lstore_v1 = [&](int i, int p) { *reinterpret_cast<f32x4*>(vbuf(i) + (srow + 16 * p) * LDR + sc * 4) = rv[p]; };

    // fragments: kf = {kh0, kl0, kh1, kl1} of a k = 16 slab of the K tile (two 32-key sub-tiles), vf = {vh0, vl0, vh1, vl1} of a
    // 16-key slab of the V^T tile (two 32-row d halves).  Every read lands in the register its predecessor has just left.
    half8 kf[4], vf[4];
    auto read_k1 = [&](int i, int s, int j) {        // j: 0 kh0, 1 kl0, 2 kh1, 3 kl1
        if ((kAbl & 2) && i > 1) return;
        if constexpr (DMA) {
            // chunk = kg + 8 (s >> 1) + 2 (s & 1) + 4 lo  ->  C / 2 = 4 (s >> 1) + (s & 1) + 2 lo
            kf[j] = *reinterpret_cast<const half8*>(lds + (i & 1) * KD + (j >> 1) * 32 * 64 + xoff[4 * (s >> 1) + (s & 1) + 2 * (j & 1)]);
            return;
        }
        const float* kp = kbuf(i) + l31 * LDR + kg * 4 + (s >> 1) * 32 + (s & 1) * 8;
        kf[j] = *reinterpret_cast<const half8*>(kp + (j >> 1) * 32 * LDR + (j & 1) * 16);
    };
    auto read_v1 = [&](int i, int q, int j) {        // slab q = 2 sub + sp; j: 0 vh0, 1 vl0, 2 vh1, 3 vl1
        if ((kAbl & 2) && i > 0) return;
        if constexpr (DMA) {
            // chunk = kg + 4 sub + 2 sp + 8 lo  ->  C / 2 = 2 sub + sp + 4 lo = q + 4 lo
            vf[j] = *reinterpret_cast<const half8*>(lds + 2 * KD + (i & 1) * KD + (j >> 1) * 32 * 64 + xoff[q + 4 * (j & 1)]);
            return;
        }
        const float* vp = vbuf(i) + l31 * LDR + 4 * kg + 16 * (q >> 1) + 8 * (q & 1);
        vf[j] = *reinterpret_cast<const half8*>(vp + (j >> 1) * 32 * LDR + (j & 1) * 32);
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f, nm = 0.f;
    const float c = 0.125f * 1.4426950408889634f;

    // Row maximum of a tile's raw scores (both lane halves), scaled; then the lazy update of the running maximum.
    auto rowmax = [&](const f32x16& s0, const f32x16& s1) {
        float mxa = max2_(s0[0], s1[0]), mxb = max2_(s0[1], s1[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
            mxa = max3_(mxa, s0[r], s1[r]);
            mxb = max3_(mxb, s0[r + 1], s1[r + 1]);
        }
        float mx = max2_(mxa, mxb);
        mx = max2_(mx, __shfl_xor(mx, 32, 64));
        return mx * c;
    };
    auto lazy_rescale = [&](float mx) {
        if (__builtin_amdgcn_ballot_w64(mx > m_run + kLazy) != 0) {      // wavefront-uniform (first tile: m_run = -inf)
            const float m_new = max2_(m_run, mx);                        // finite: every tile holds at least one key of the clip
            const float alpha = exp2_(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        nm = kPShiftI - m_run;
    };
    auto mask_tile = [&](int i, f32x16& s0, f32x16& s1) {
        const int gbase = i * KT;                                        // clip-local key index of the tile's first key
        if (gbase + KT > T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k0 = gbase + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (k0 >= T) s0[r] = -INFINITY;
                if (k0 + 32 >= T) s1[r] = -INFINITY;
            }
        }
    };

    const float mone = __builtin_bit_cast(float, 0xBF800000u | ((uint32_t)a.B >> 30));      // -1.0f the compiler cannot fold (B < 2^30)
    // P split state: pair j of slab q covers P elements 8 q + 2 j, + 1 (element e: e < 16 ? c0[e] : c1[e - 16])
    uint32_t hpk[4][4];            // [slab][pair] packed hi halves
    uint32_t lpk[4][4];            // [slab][pair] packed lo halves
    float lo0_[4][4];              // first half-step's leftover
    auto pel = [&](const f32x16& c0, const f32x16& c1, int e) -> float { return e < 16 ? c0[e] : c1[e - 16]; };
    float ps0 = 0.f, ps1 = 0.f;                       // row sums of the tile in flight (two chains)
    auto split_half = [&](f32x16& c0, f32x16& c1, int q, int j, int half) {
        const float p0 = pel(c0, c1, 8 * q + 2 * j), p1 = pel(c0, c1, 8 * q + 2 * j + 1);
        if constexpr (FAST >= 1) {
            if (half == 0) {
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
                const f32x2_ pp = {p0, p1};
                const half2_t hh = __builtin_convertvector(pp, half2_t);                             // v_cvt_pk_f16_f32: round to nearest even
                hpk[q][j] = __builtin_bit_cast(uint32_t, hh);
                // the row sum counts what the matrix pipe will see: the weights ph / sum ph add up to exactly 1, so a query whose
                // softmax is one key (the worst case for a rounded P) gets that key's V row exactly
                ps0 += (float)hh[0];
                ps1 += (float)hh[1];
            }
        } else if (half == 0) {
            const half2_t hh = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(p0, p1));
            hpk[q][j] = __builtin_bit_cast(uint32_t, hh);
            lo0_[q][j] = mix_sub_(hh[0], mone, p0);
        } else {
            const half2_t hh = __builtin_bit_cast(half2_t, hpk[q][j]);
            const float l1 = mix_sub_(hh[1], mone, p1);
            lpk[q][j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lo0_[q][j], l1));
        }
    };
    auto frag_of = [&](const uint32_t (&w)[4]) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {w[0], w[1], w[2], w[3]};
        return __builtin_bit_cast(half8, v);
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // One tile step.  c0 / c1: raw scores of tile i on entry, its probabilities afterwards.  n0 / n1: receive S(i+1).
    //   FULL: tiles i + 1 .. i + 3 all exist and tile i + 1 needs no masking (interior of the clip).  QK = false: last tile (no S(i+1)).
    auto step = [&](int i, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1, auto full_, auto qk_) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_)::value, QK = decltype(qk_)::value;
        // ---- phase A: S(i+1) = K(i+1) Q^T  |  P(i) = exp2(S(i) c + nm), split of slabs 0 and 1, staging
        __builtin_amdgcn_s_setprio(1);                 // raised priority over the QK phase: 2.338 -> 2.310 ms (over the PV phase: 2.333; profiles/r05j_attn_variants.txt)
        if constexpr (QK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) read_k1(i + 1, 0, j);
        }
#pragma unroll
        for (int t = 0; t < 24; ++t) {
            const int s = t / 6, u = t % 6;
            if constexpr (QK) {
                // kl0 qh, kl1 qh, kh0 ql, kh1 ql, kh0 qh, kh1 qh
                const half8& ka = kf[u == 0 ? 1 : u == 1 ? 3 : (u & 1) ? 2 : 0];
                const half8& qb_ = (u == 2 || u == 3) ? ql[s] : qh[s];
                f32x16& acc = (u & 1) ? n1 : n0;
                if (!(FAST == 2 && (u == 2 || u == 3))) acc = mfma_hi<false>(ka, qb_, (t < 2) ? zero16 : acc);
                // the next slab's fragments into the registers that have just been read for the last time
                if (s < 3) {
                    if (u == 0) read_k1(i + 1, s + 1, 1);
                    if (u == 1) read_k1(i + 1, s + 1, 3);
                    if (u == 4) read_k1(i + 1, s + 1, 0);
                    if (u == 5) read_k1(i + 1, s + 1, 2);
                }
            }
            // staging, EARLY in the step (slots 8 - 15): K tile i + 2 and V^T tile i + 1 out of the registers into the rings, the next loads
            // behind them.  The ring stores then have 32 slots to drain before the barrier's lgkmcnt(0); at the END of the step they cost
            // 0.6 ms per launch (profiles/r05_experiments.md: ablations 16 / 32 / 8 - it is the store -> barrier wait, not the loads)
            if (t >= kStageSlot && t < kStageSlot + 8 && !(kAbl & 4)) {
                const int p = (t - kStageSlot) & 3;
                if constexpr (DMA) {
                    if (t < kStageSlot + 4) { if (FULL || i + 2 < n) dma_k(i + 2, p); }
                    else { if (FULL || i + 1 < n) dma_v(i + 1, p); }
                } else if (t < kStageSlot + 4) {
                    if (!(kAbl & 32)) { if (FULL || i + 2 < n) lstore_k1(i + 2, p); } else asm volatile("" :: "v"(rk[p]));
                    if (!(kAbl & 16)) { if (FULL || i + 3 < n) gload_k1(i + 3, p); }
                } else {
                    if (!(kAbl & 32)) { if (FULL || i + 1 < n) lstore_v1(i + 1, p); } else asm volatile("" :: "v"(rv[p]));
                    if (!(kAbl & 16)) { if (FULL || i + 2 < n) gload_v1(i + 2, p); }
                }
            }
            if (t >= 20) read_v1(i, 0, t - 20);                   // first slab of the PV product (V^T tile i: visible since the last barrier)
            if ((kAbl & 1) && i > 0) {
            } else if (t < 8) {
#pragma unroll
                for (int e = 2 * t; e < 2 * t + 2; ++e) c0[e] = exp2_(fmaf(c0[e], c, nm));
            } else {
                const int j = t - 8;                              // 0 .. 15
                c1[j] = exp2_(fmaf(c1[j], c, nm));
                split_half(c0, c1, j >> 3, (j >> 1) & 3, j & 1);  // slab 0 in slots 8 - 15, slab 1 in 16 - 23
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (QK && !FULL) mask_tile(i + 1, n0, n1);
        // ---- phase B: O += V(i)^T P(i)^T  |  split of slabs 2 and 3, row sums, row maximum of S(i+1)
        float mxa = 0.f, mxb = 0.f;
#pragma unroll
        for (int t = 0; t < 24; ++t) {
            const int q = t / 6, u = t % 6;
            {
                // vl0 ph, vl1 ph, vh0 pl, vh1 pl, vh0 ph, vh1 ph
                const half8& va = vf[u == 0 ? 1 : u == 1 ? 3 : (u & 1) ? 2 : 0];
                if (!(FAST >= 1 && (u == 2 || u == 3))) {
                    const half8 pb = (u == 2 || u == 3) ? frag_of(lpk[q]) : frag_of(hpk[q]);
                    if (u & 1) o1 = mfma_hi<false>(va, pb, o1);
                    else o0 = mfma_hi<false>(va, pb, o0);
                }
                if (q < 3) {
                    if (u == 0) read_v1(i, q + 1, 1);
                    if (u == 1) read_v1(i, q + 1, 3);
                    if (u == 4) read_v1(i, q + 1, 0);
                    if (u == 5) read_v1(i, q + 1, 2);
                }
            }
            // split: slab 2 in slots 2 - 9, slab 3 in slots 8 - 15 (half a pair per slot)
            if (!((kAbl & 1) && i > 0)) {
            if (t >= 2 && t < 10) split_half(c0, c1, 2, (t - 2) >> 1, (t - 2) & 1);
            if (t >= 8 && t < 16) split_half(c0, c1, 3, (t - 8) >> 1, (t - 8) & 1);
            // row sums: 32 probabilities over slots 0 - 15
            if (FAST == 0 && t < 16) { ps0 += pel(c0, c1, 2 * t); ps1 += pel(c0, c1, 2 * t + 1); }
            }
            // staging: K tile i + 2 and V^T tile i + 1 out of the registers into the rings, the next loads behind them
            // row maximum of S(i+1): two chains over slots 16 - 23
            if constexpr (QK && !(kAbl & 1)) {
                if (t == 16) { mxa = max2_(n0[0], n1[0]); mxb = max2_(n0[1], n1[1]); }
                if (t > 16) {
                    const int r = 2 * (t - 16);
                    mxa = max3_(mxa, n0[r], n1[r]);
                    mxb = max3_(mxb, n0[r + 1], n1[r + 1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run += ps0 + ps1;
        ps0 = 0.f; ps1 = 0.f;
        if constexpr (QK && !(kAbl & 1)) {
            float mx = max2_(mxa, mxb);
            mx = max2_(mx, __shfl_xor(mx, 32, 64)) * c;
            lazy_rescale(mx);
        }
#ifdef SOME_ATTN_DBG
        const bool dbg_wg = blockIdx.x == SOME_ATTN_DBG_WG;
        unsigned long long* dbg_l = reinterpret_cast<unsigned long long*>(lds + LDS_BYTES / 4);      // (the probe launches with 4 KiB more LDS)
        const unsigned long long t_arr = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (timeline builds: how long the tile DMA is waited for)
        const unsigned long long t_land = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA completion is tracked by vmcnt: a tile must have LANDED before the barrier publishes it (explicit - not left to what hipcc emits around __syncthreads)
        if (!(kAbl & 8)) __syncthreads();
#ifdef SOME_ATTN_DBG
        if (dbg_wg && lane == 0) {
            const unsigned long long t_rel = __builtin_amdgcn_s_memtime();
            dbg_l[i * 8 + wave * 2] = t_arr;
            dbg_l[i * 8 + wave * 2 + 1] = ((t_land - t_arr) << 40) | (t_rel & 0xFFFFFFFFFFull);      // DMA wait (cycles) in the top bits
        }
#endif
    };

    // ---- prologue: tiles 0 (both rings), 1 (K ring; V^T in registers), 2 (K in registers); S(0) unscheduled
    if constexpr (DMA) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { dma_k(0, p); dma_v(0, p); }
        if (n > 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) dma_k(1, p);
        }
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) { gload_k1(0, p); gload_v1(0, p); }
#pragma unroll
        for (int p = 0; p < 4; ++p) { lstore_k1(0, p); lstore_v1(0, p); }
        if (n > 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) gload_k1(1, p);
#pragma unroll
            for (int p = 0; p < 4; ++p) { lstore_k1(1, p); gload_v1(1, p); }
        }
        if (n > 2) {
#pragma unroll
            for (int p = 0; p < 4; ++p) gload_k1(2, p);
        }
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA completion is tracked by vmcnt: a tile must have LANDED before the barrier publishes it (explicit - not left to what hipcc emits around __syncthreads)
    __syncthreads();
    f32x16 sa0, sa1, sb0, sb1;
    // A wavefront whose 32 queries all lie behind the clip's end (the last 128-query block of a clip: T = 2584 leaves 24 queries for
    // wavefront 0 and none for the other three) only keeps its staging role and the barriers: its matrix-pipe and VALU slots go to the
    // other workgroup of the CU (3 of 84 wavefront-blocks per clip and head at 30 s)
    const bool active = q0 + wave * 32 < T;
    if (active) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) read_k1(0, s, j);
            sa0 = mfma_hi<false>(kf[1], qh[s], s == 0 ? zero16 : sa0);
            sa1 = mfma_hi<false>(kf[3], qh[s], s == 0 ? zero16 : sa1);
            sa0 = mfma_hi<false>(kf[0], ql[s], sa0);
            sa1 = mfma_hi<false>(kf[2], ql[s], sa1);
            sa0 = mfma_hi<false>(kf[0], qh[s], sa0);
            sa1 = mfma_hi<false>(kf[2], qh[s], sa1);
        }
        mask_tile(0, sa0, sa1);
        lazy_rescale(rowmax(sa0, sa1));
    }
    // S(0) read K ring slot 0 and the first step stores K(2) into it (the round-1 race, see attention3_kernel)
    __syncthreads();
    int i = 0;
    if (active) {
        for (; i + 4 < n; i += 2) {
            step(i, sa0, sa1, sb0, sb1, Flag<true>{}, Flag<true>{});
            step(i + 1, sb0, sb1, sa0, sa1, Flag<true>{}, Flag<true>{});
        }
        for (; i + 2 < n; i += 2) {
            step(i, sa0, sa1, sb0, sb1, Flag<false>{}, Flag<true>{});
            step(i + 1, sb0, sb1, sa0, sa1, Flag<false>{}, Flag<true>{});
        }
        if (i + 1 < n) {
            step(i, sa0, sa1, sb0, sb1, Flag<false>{}, Flag<true>{});
            step(i + 1, sb0, sb1, sa0, sa1, Flag<false>{}, Flag<false>{});
        } else {
            step(i, sa0, sa1, sb0, sb1, Flag<false>{}, Flag<false>{});
        }
    } else {
        for (; i < n; ++i) {                       // the same ring traffic and barriers as step(i), nothing else
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if constexpr (DMA) { if (i + 2 < n) dma_k(i + 2, p); }
                else {
                    if (i + 2 < n) lstore_k1(i + 2, p);
                    if (i + 3 < n) gload_k1(i + 3, p);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if constexpr (DMA) { if (i + 1 < n) dma_v(i + 1, p); }
                else {
                    if (i + 1 < n) lstore_v1(i + 1, p);
                    if (i + 2 < n) gload_v1(i + 2, p);
                }
            }
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA completion is tracked by vmcnt: a tile must have LANDED before the barrier publishes it (explicit - not left to what hipcc emits around __syncthreads)
            __syncthreads();
        }
    }

#ifdef SOME_ATTN_DBG
    if (blockIdx.x == SOME_ATTN_DBG_WG) {
        __syncthreads();
        const unsigned long long* dbg_l = reinterpret_cast<const unsigned long long*>(lds + LDS_BYTES / 4);
        for (int j = tid; j < n * 8; j += 256) g_attn_dbg[j] = dbg_l[j];
    }
#endif
    // ---- normalise, transpose through LDS (wave-private 32 x 64 patch), SPLIT32 row stores (the last step ended with a barrier)
    float* patch = lds + wave * (32 * LDR);
    {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * kg;
            patch[l31 * LDR + d] = o0[r] * inv;
            patch[l31 * LDR + 32 + d] = o1[r] * inv;
        }
        __syncthreads();
        const int orow = lane >> 4, ocol = (lane & 15) * 4;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int ql_ = orow + 4 * p;
            const int q = q0 + wave * 32 + ql_;
            if (q < T) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(patch + ql_ * LDR + ocol);
                half4 hh, ll;
#pragma unroll
                for (int e = 0; e < 4; ++e) { half_t h, l; split_f16(v[e], h, l); hh[e] = h; ll[e] = l; }
                char* row = reinterpret_cast<char*>(a.out[g]) + ((size_t)(fo + q) * kDim + head * kHeadDim) * 4 + (ocol >> 5) * 128 + (ocol & 31) * 2;
                *reinterpret_cast<half4*>(row) = hh;
                *reinterpret_cast<half4*>(row + 64) = ll;
            }
        }
    }
}

}  // namespace

hipError_t launch_attention_f16x3(const Attn3Args& a, hipStream_t s) {
    if (a.B <= 0 || a.max_frames <= 0) return hipSuccess;
    static DeviceOnce attr_once;
    static const bool placed = !(getenv("SOME_AMD_ATTN_V1") && getenv("SOME_AMD_ATTN_V1")[0] == '1');       // A/B switch: the round-1..4 kernel
    static const bool dma = !(getenv("SOME_AMD_ATTN_DMA") && getenv("SOME_AMD_ATTN_DMA")[0] == '0');        // A/B switch: register-staged rings
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3i_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3i_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3i_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3i_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3_kernel<true, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    const bool inference = a.out32[0] == nullptr;
    if (inference && a.pad_offsets == nullptr) return hipErrorInvalidValue;
    constexpr int QB = 128;
    const int nqb = (a.max_frames + QB - 1) / QB;
    const int units = a.B * kHeads * a.groups;
    const int slots = (units + 7) / 8;
    if (!inference && a.hi_only == 2) hipLaunchKernelGGL((attention3_kernel<true, 1, true>), dim3((unsigned)(slots * nqb * 8)), dim3(256), LDS_BYTES, s, a, nqb);
    else if (!inference && a.hi_only) hipLaunchKernelGGL((attention3_kernel<true, 1>), dim3((unsigned)(slots * nqb * 8)), dim3(256), LDS_BYTES, s, a, nqb);
    else if (!inference) hipLaunchKernelGGL(attention3_kernel<true>, dim3((unsigned)(slots * nqb * 8)), dim3(256), LDS_BYTES, s, a, nqb);
    else if (placed && dma && a.fast == 1) hipLaunchKernelGGL((attention3i_kernel<true, 1>), dim3((unsigned)(slots * nqb * 8)), dim3(256), LDS_BYTES, s, a, nqb);
    else if (placed && dma && a.fast == 2) hipLaunchKernelGGL((attention3i_kernel<true, 2>), dim3((unsigned)(slots * nqb * 8)), dim3(256), LDS_BYTES, s, a, nqb);
    else if (placed && dma) hipLaunchKernelGGL(attention3i_kernel<true>, dim3((unsigned)(slots * nqb * 8)), dim3(256), LDS_BYTES, s, a, nqb);
    else if (placed) hipLaunchKernelGGL(attention3i_kernel<false>, dim3((unsigned)(slots * nqb * 8)), dim3(256), LDS_BYTES, s, a, nqb);
    else hipLaunchKernelGGL(attention3_kernel<false>, dim3((unsigned)(slots * nqb * 8)), dim3(256), LDS_BYTES, s, a, nqb);
    return hipGetLastError();
}
