// Exact-fp32 flash attention for gfx950 (v_mfma_f32_32x32x2_f32), per clip, non-causal, unmasked.
//
// Replaces F.scaled_dot_product_attention(q, k, v) with 8 heads x 64 and scale 64^-0.5
// (modules/attention/base_attention.py:34-44; mask is never passed - modules/conform/Gconform.py:60) plus the
// two einops rearranges around it: reads q|k|v straight out of the fused projection output [M, 1536] and
// writes the merged-head layout [M, 512].  The T x T score matrix is never materialised.
//
// Work split: one workgroup = 128 queries of one (clip, head, stream); 4 waves x 32 queries.  Keys/values
// stream through LDS in tiles of 64 (double buffered, next tile's global loads in flight during the MFMAs).
//
// Register-only softmax via the transposed product: each wave computes S^T = K * Q^T, so in the MFMA C/D
// layout a LANE owns one query (column = lane & 31) and 16 of the 32 keys of a sub-tile
// (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  Row max / row sum are therefore 16-element in-register
// reductions plus ONE exchange with lane ^ 32.  P^T in that same register layout IS the B operand of the
// second product O^T = V^T * P^T (B[k][j]: j = lane & 31 = query, k-slot = lane >> 5), so P never moves
// through LDS or across lanes; the A operand V^T[d][key] is one ds_read_b32 per MFMA step.  O^T again has
// the query in the lane, so the online-softmax rescale is a per-lane scalar.
//
// Cost per 64-key tile and wave: 64 + 64 MFMAs = 8192 matrix-pipe cycles vs ~600 VALU cycles of softmax
// (hidden by the second workgroup on the CU), 16 ds_read_b128 (K) + 64 ds_read_b32 (V).
//
// LDS: K tile [64][68] (row pad 4 floats -> conflict-free ds_read_b128 down a column of keys),
//      V tile [64][64]; 2 stages = 67,584 B -> 2 workgroups / CU.
#include "internal.h"
#include "split.h"

namespace {

constexpr int QB = 128;           // queries per workgroup
constexpr int KT = 64;            // keys per LDS tile
constexpr int LDK = kHeadDim + 4; // padded K row
constexpr int K_FLOATS = KT * LDK;
constexpr int V_FLOATS = KT * kHeadDim;
constexpr int STAGE = K_FLOATS + V_FLOATS;
constexpr size_t ATT_LDS_BYTES = 2 * STAGE * sizeof(float);
constexpr int QKV_LD = 3 * kDim;  // 1536

__device__ __forceinline__ float exp2_(float x) { return __builtin_amdgcn_exp2f(x); }

__global__ __launch_bounds__(256, 2) void attention_kernel(AttnArgs a, int nqb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // unit = (clip, head, stream); all q-blocks of a unit run on one XCD (blockIdx.x % 8) so the unit's
    // K/V (T x 64 x 2 fp32, 1.3 MB at T = 2584) is fetched into one L2 only.
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int slot = jj / nqb, qb = jj % nqb;
    const int unit = slot * 8 + xcd;
    const int hg = unit % (kHeads * a.groups), b = unit / (kHeads * a.groups);
    if (b >= a.B) return;
    const int head = hg % kHeads, g = hg / kHeads;
    const int f0 = a.frame_offsets[b];
    const int T = a.frame_offsets[b + 1] - f0;
    const int q0 = qb * QB;
    if (q0 >= T) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const float* __restrict__ base = a.qkv[g] + (size_t)f0 * QKV_LD + head * kHeadDim;
    const float* __restrict__ Kg = base + kDim;
    const float* __restrict__ Vg = base + 2 * kDim;

    // ---- Q^T fragment (B operand of S^T = K Q^T): lane (q, hi) holds Q[q][8j + 4hi + s], pre-scaled
    const float qscale = 0.125f * 1.4426950408889634f;   // head_dim^-0.5 * log2(e): softmax in base 2
    f32x4 qf[8];
    {
        const int q = q0 + wave * 32 + l31;
        const bool qv = q < T;
        const float* qrow = base + (size_t)(qv ? q : 0) * QKV_LD + hi * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qrow + j * 8);
            qf[j] = qv ? v * qscale : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- staging roles: a tile is 64 rows x 64 floats = 1024 float4 per operand; 4 per thread
    const int srow = tid >> 4, scol = (tid & 15) * 4;   // rows srow + 16 p
    f32x4 rk[4], rv[4];
    auto gload = [&](int kt) {
        const int k0 = kt * KT;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int key = k0 + srow + 16 * p;
            if (key < T) {
                rk[p] = *reinterpret_cast<const f32x4*>(Kg + (size_t)key * QKV_LD + scol);
                rv[p] = *reinterpret_cast<const f32x4*>(Vg + (size_t)key * QKV_LD + scol);
            } else {   // zero rows: 0 * p keeps O finite, scores of these keys are forced to -inf below
                rk[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
                rv[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto lstore = [&](int buf) {
        float* Ks = lds + buf * STAGE;
        float* Vs = Ks + K_FLOATS;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f32x4*>(Ks + (srow + 16 * p) * LDK + scol) = rk[p];
            *reinterpret_cast<f32x4*>(Vs + (srow + 16 * p) * kHeadDim + scol) = rv[p];
        }
    };

    f32x16 o0, o1;            // O^T tiles: d in [0,32) and [32,64); column = query (lane & 31)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY;  // running max (base-2 logits), identical in lanes q and q + 32
    float l_run = 0.f;        // this lane's PARTIAL row sum (its 16-of-32 keys per sub-tile)

    const int nkt = (T + KT - 1) / KT;
    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const float* Ks = lds + buf * STAGE;
        const float* Vs = Ks + K_FLOATS;

        // ---- S^T = K Q^T for the two 32-key sub-tiles
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        const float* kp = Ks + l31 * LDK + hi * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 k0v = *reinterpret_cast<const f32x4*>(kp + j * 8);
            const f32x4 k1v = *reinterpret_cast<const f32x4*>(kp + 32 * LDK + j * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0v[s], qf[j][s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1v[s], qf[j][s], s1, 0, 0, 0);
            }
        }
        // keys past the end of the clip (last tile only)
        const int kbase = kt * KT;
        if (kbase + KT > T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (kbase + kl >= T) s0[r] = -INFINITY;
                if (kbase + 32 + kl >= T) s1[r] = -INFINITY;
            }
        }
        // ---- online softmax (per lane = per query)
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);          // finite: key kbase < T is valid in every tile
        const float alpha = exp2_(m_run - m_new);       // first tile: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = exp2_(s0[r] - m_new);
            s1[r] = exp2_(s1[r] - m_new);
            psum += s0[r] + s1[r];
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- O^T += V^T P^T : A = V^T[d][key] (ds_read_b32), B = P^T (registers as they are)
        const float* vp = Vs + (4 * hi) * kHeadDim + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2);
            const float va0 = vp[kl * kHeadDim];
            const float va1 = vp[kl * kHeadDim + 32];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, s0[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, s0[r], o1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = 32 + (r & 3) + 8 * (r >> 2);
            const float va0 = vp[kl * kHeadDim];
            const float va1 = vp[kl * kHeadDim + 32];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, s1[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, s1[r], o1, 0, 0, 0);
        }

        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- normalise, transpose through LDS (wave-private 32 x 64 patch, row pad 4), coalesced row stores
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (a.lse[g] != nullptr && hi == 0) {            // training: P = exp2(s2 - lse2) is recomputed by the backward
        const int q = q0 + wave * 32 + l31;
        if (q < T) a.lse[g][(size_t)head * a.M + f0 + q] = m_run + __log2f(l_tot);
    }
    float* patch = lds + wave * (32 * LDK);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * hi;
        patch[l31 * LDK + d] = o0[r] * inv;
        patch[l31 * LDK + 32 + d] = o1[r] * inv;
    }
    __syncthreads();
    float* __restrict__ og = a.out[g] + (size_t)f0 * kDim + head * kHeadDim;
    const int orow = lane >> 4, ocol = (lane & 15) * 4;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int ql = orow + 4 * p;
        const int q = q0 + wave * 32 + ql;
        if (q < T) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(patch + ql * LDK + ocol);
            if (a.out_split) {   // SPLIT32 row of 512: head h covers k-blocks 2h, 2h+1; 8 lanes per k-block
                half4 hh, ll;
#pragma unroll
                for (int i = 0; i < 4; ++i) { half_t h, l; split_f16(v[i], h, l); hh[i] = h; ll[i] = l; }
                char* row = reinterpret_cast<char*>(a.out[g]) + ((size_t)(f0 + q) * kDim + head * kHeadDim) * 4 +
                            (ocol >> 5) * 128 + (ocol & 31) * 2;
                *reinterpret_cast<half4*>(row) = hh;
                *reinterpret_cast<half4*>(row + 64) = ll;
            } else {
                *reinterpret_cast<f32x4*>(og + (size_t)q * kDim + ocol) = v;
            }
        }
    }
}

}  // namespace

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.max_frames <= 0) return hipSuccess;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    const int nqb = (a.max_frames + QB - 1) / QB;
    const int units = a.B * kHeads * a.groups;
    const int slots = (units + 7) / 8;
    dim3 grid((unsigned)(slots * nqb * 8));
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), ATT_LDS_BYTES, s, a, nqb);
    return hipGetLastError();
}
