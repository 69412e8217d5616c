// Backward of the exact-fp32 flash attention (attention.hip) for the training path:
// the gradient of F.scaled_dot_product_attention(q, k, v) (modules/attention/base_attention.py:42-44, 8 heads x 64,
// scale 1/8, per clip, unmasked) with respect to the fused projection output qkv [M, 1536], given dout [M, 512].
//
// Flash-style: the T x T matrices are recomputed tile by tile from the saved base-2 log-sum-exp (attention.hip writes
// lse2 = max + log2(sum)), never stored.  With s2 = q.k / 8 * log2(e):  P = exp2(s2 - lse2),  D[q] = sum_d dO O,
//     dV = P^T dO,     dP = dO V^T,     dS = P (dP - D),     dQ = dS K / 8,     dK = dS^T Q / 8.
// Two deterministic kernels instead of atomics on dQ:
//   attn_bwd_dkv_kernel  one wave owns 32 keys (a lane = one key), loops over 32-query tiles staged in LDS;
//   attn_bwd_dq_kernel   one wave owns 32 queries (a lane = one query), loops over 64-key tiles like the forward.
// In both, the recomputed P / dS tile comes out of v_mfma_f32_32x32x2_f32 in the C/D layout whose column is the lane's
// key (query), which is exactly the B-operand layout of the next product - no LDS round trip, as in the forward.
#include "internal.h"

namespace {

constexpr int QKV_LD = 3 * kDim;
constexpr int LD = kHeadDim + 4;                 // padded LDS row (floats): conflict-free ds_read_b128 down a column
constexpr float kScale = 0.125f;                 // head_dim^-0.5
constexpr float kScale2 = 0.125f * 1.4426950408889634f;

__device__ __forceinline__ float exp2_(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int krow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }   // C/D row of register r

// D[h][m] = sum_d dO[m, 64 h + d] O[m, 64 h + d]: one wave per row, 8 lanes per head.  factor (optional, a power of two on the device: the
// one the split dO operands were multiplied by, train_ops.hip split_transpose_kernel) scales the sums - exactly what summing factor * dO gives.
__global__ __launch_bounds__(256) void attn_dsum_kernel(const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ dsum, int M,
                                                        const float* __restrict__ factor) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(out + (size_t)row * kDim + lane * 8);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(out + (size_t)row * kDim + lane * 8 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(dout + (size_t)row * kDim + lane * 8);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(dout + (size_t)row * kDim + lane * 8 + 4);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += a0[i] * b0[i] + a1[i] * b1[i];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (factor) s *= factor[0];
    if ((lane & 7) == 0) dsum[(size_t)(lane >> 3) * M + row] = s;
}

// ---- dK, dV: workgroup = 128 keys of one (clip, head); wave = 32 keys; loop over 32-query tiles ------------------------
constexpr int QT = 32;
constexpr int KV_STAGE = 2 * QT * LD + 2 * QT;          // Q tile, dO tile, lse2[32], D[32]
constexpr size_t DKV_LDS_BYTES = 2 * KV_STAGE * sizeof(float);

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnBwdArgs a, int nkb) {
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
    const int l31 = lane & 31, hi = lane >> 5;
    const float* __restrict__ base = a.qkv + (size_t)f0 * QKV_LD + head * kHeadDim;
    const float* __restrict__ dOg = a.dout + (size_t)f0 * kDim + head * kHeadDim;
    const float* __restrict__ lseg = a.lse + (size_t)head * a.M + f0;
    const float* __restrict__ dsg = a.dsum + (size_t)head * a.M + f0;

    // this lane's key: K (pre-scaled to base-2 logits) and V fragments, element d = 8 j + 4 hi + s
    const int key = k0 + wave * 32 + l31;
    const bool kv = key < T;
    f32x4 kf[8], vf[8];
    {
        const float* krow_ = base + kDim + (size_t)(kv ? key : 0) * QKV_LD + hi * 4;
        const float* vrow_ = base + 2 * kDim + (size_t)(kv ? key : 0) * QKV_LD + hi * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 kk = *reinterpret_cast<const f32x4*>(krow_ + j * 8);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(vrow_ + j * 8);
            kf[j] = kv ? kk * kScale2 : (f32x4){0.f, 0.f, 0.f, 0.f};
            vf[j] = kv ? vv : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    // staging: Q and dO tiles are 32 rows x 16 float4 each = 512 chunks -> 2 + 2 per thread; lse / D by the first wave
    const int srow = tid >> 4, scol = (tid & 15) * 4;       // rows srow, srow + 16
    f32x4 rq[2], rd[2];
    float rl = 0.f;
    auto gload = [&](int qt) {
        const int q0 = qt * QT;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int q = q0 + srow + 16 * p;
            if (q < T) {
                rq[p] = *reinterpret_cast<const f32x4*>(base + (size_t)q * QKV_LD + scol);
                rd[p] = *reinterpret_cast<const f32x4*>(dOg + (size_t)q * kDim + scol);
            } else {
                rq[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
                rd[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (tid < 64) {
            const int q = q0 + (tid & 31);
            // queries past the end of the clip: lse2 = +inf makes P = exp2(-inf) = 0
            rl = q < T ? (tid < 32 ? lseg[q] : dsg[q]) : (tid < 32 ? INFINITY : 0.f);
        }
    };
    auto lstore = [&](int buf) {
        float* Qs = lds + buf * KV_STAGE;
        float* Ds = Qs + QT * LD;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f32x4*>(Qs + (srow + 16 * p) * LD + scol) = rq[p];
            *reinterpret_cast<f32x4*>(Ds + (srow + 16 * p) * LD + scol) = rd[p];
        }
        if (tid < 64) (Ds + QT * LD)[tid] = rl;              // lse2[0..31] then D[0..31]
    };

    f32x16 dk0, dk1, dv0, dv1;                               // dK^T / dV^T: rows d (two halves), column = this lane's key
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }

    const int nqt = (T + QT - 1) / QT;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int qt = 0; qt < nqt; ++qt) {
        const int buf = qt & 1;
        if (qt + 1 < nqt) gload(qt + 1);
        const float* Qs = lds + buf * KV_STAGE;
        const float* Ds = Qs + QT * LD;
        const float* Ls = Ds + QT * LD;

        // S2[q][key] = Q K^T (base-2 logits) and dP[q][key] = dO V^T : A rows = queries (lane & 31), B = this lane's key
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const float* qp = Qs + l31 * LD + hi * 4;
        const float* dpp = Ds + l31 * LD + hi * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + j * 8);
            const f32x4 dv = *reinterpret_cast<const f32x4*>(dpp + j * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[t], kf[j][t], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[t], vf[j][t], dp, 0, 0, 0);
            }
        }
        // P = exp2(S2 - lse2[q]),  dS = P (dP - D[q]);  register r <-> query row krow(r, hi)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + 8 * r4 + 4 * hi);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ls + QT + 8 * r4 + 4 * hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r4 * 4 + i;
                const float p = exp2_(s[r] - l4[i]);
                s[r] = p;
                dp[r] = p * (dp[r] - d4[i]);
            }
        }
        // dV^T += dO^T P,  dK^T += Q^T dS : A[i = d][k = query] read column-wise from the staged tiles
        const float* qa = Qs + (4 * hi) * LD + l31;
        const float* da = Ds + (4 * hi) * LD + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = (r & 3) + 8 * (r >> 2);
            dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(da[ql * LD], s[r], dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(da[ql * LD + 32], s[r], dv1, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[ql * LD], dp[r], dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[ql * LD + 32], dp[r], dk1, 0, 0, 0);
        }
        if (qt + 1 < nqt) lstore(buf ^ 1);
        __syncthreads();
    }

    if (kv) {
        float* dKg = a.dqkv + (size_t)(f0 + key) * QKV_LD + kDim + head * kHeadDim;
        float* dVg = dKg + kDim;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = 8 * r4 + 4 * hi;
            f32x4 k_lo, k_hi, v_lo, v_hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                k_lo[i] = dk0[r4 * 4 + i] * kScale; k_hi[i] = dk1[r4 * 4 + i] * kScale;
                v_lo[i] = dv0[r4 * 4 + i]; v_hi[i] = dv1[r4 * 4 + i];
            }
            *reinterpret_cast<f32x4*>(dKg + d) = k_lo;
            *reinterpret_cast<f32x4*>(dKg + 32 + d) = k_hi;
            *reinterpret_cast<f32x4*>(dVg + d) = v_lo;
            *reinterpret_cast<f32x4*>(dVg + 32 + d) = v_hi;
        }
    }
}

// ---- dQ: workgroup = 128 queries of one (clip, head); wave = 32 queries; loop over 64-key tiles ----------------------------
constexpr int QB = 128, KT = 64;
constexpr int DQ_STAGE = 2 * KT * LD;                     // K tile, V tile (both padded rows)
constexpr size_t DQ_LDS_BYTES = 2 * DQ_STAGE * sizeof(float);

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnBwdArgs a, int nqb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int slot = jj / nqb, qb = jj % nqb;
    const int unit = slot * 8 + xcd;
    const int head = unit % kHeads, b = unit / kHeads;
    if (b >= a.B) return;
    const int f0 = a.frame_offsets[b];
    const int T = a.frame_offsets[b + 1] - f0;
    const int q0 = qb * QB;
    if (q0 >= T) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const float* __restrict__ base = a.qkv + (size_t)f0 * QKV_LD + head * kHeadDim;
    const float* __restrict__ Kg = base + kDim;
    const float* __restrict__ Vg = base + 2 * kDim;

    // this lane's query: Q^T (pre-scaled) and dO^T fragments, lse2 and D
    const int q = q0 + wave * 32 + l31;
    const bool qv = q < T;
    f32x4 qf[8], dof[8];
    {
        const float* qrow = base + (size_t)(qv ? q : 0) * QKV_LD + hi * 4;
        const float* drow = a.dout + (size_t)(f0 + (qv ? q : 0)) * kDim + head * kHeadDim + hi * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(qrow + j * 8);
            const f32x4 y = *reinterpret_cast<const f32x4*>(drow + j * 8);
            qf[j] = qv ? x * kScale2 : (f32x4){0.f, 0.f, 0.f, 0.f};
            dof[j] = qv ? y : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float lse2 = qv ? a.lse[(size_t)head * a.M + f0 + q] : INFINITY;      // invalid query: P = 0
    const float dsum = qv ? a.dsum[(size_t)head * a.M + f0 + q] : 0.f;

    const int srow = tid >> 4, scol = (tid & 15) * 4;
    f32x4 rk[4], rv[4];
    auto gload = [&](int kt) {
        const int kk0 = kt * KT;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int key = kk0 + srow + 16 * p;
            if (key < T) {
                rk[p] = *reinterpret_cast<const f32x4*>(Kg + (size_t)key * QKV_LD + scol);
                rv[p] = *reinterpret_cast<const f32x4*>(Vg + (size_t)key * QKV_LD + scol);
            } else {
                rk[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
                rv[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto lstore = [&](int buf) {
        float* Ks = lds + buf * DQ_STAGE;
        float* Vs = Ks + KT * LD;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f32x4*>(Ks + (srow + 16 * p) * LD + scol) = rk[p];
            *reinterpret_cast<f32x4*>(Vs + (srow + 16 * p) * LD + scol) = rv[p];
        }
    };

    f32x16 o0, o1;                                          // dQ^T: rows d (two halves), column = this lane's query
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

    const int nkt = (T + KT - 1) / KT;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const float* Ks = lds + buf * DQ_STAGE;
        const float* Vs = Ks + KT * LD;
        const int kbase = kt * KT;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {                 // two 32-key sub-tiles
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            const float* kp = Ks + (32 * sub + l31) * LD + hi * 4;
            const float* vp = Vs + (32 * sub + l31) * LD + hi * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 kx = *reinterpret_cast<const f32x4*>(kp + j * 8);
                const f32x4 vx = *reinterpret_cast<const f32x4*>(vp + j * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx[t], qf[j][t], s, 0, 0, 0);       // S2^T[key][q]
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vx[t], dof[j][t], dp, 0, 0, 0);    // dP^T[key][q]
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool valid = kbase + 32 * sub + krow(r, hi) < T;
                const float p = valid ? exp2_(s[r] - lse2) : 0.f;
                dp[r] = p * (dp[r] - dsum);                                                      // dS^T
            }
            // dQ^T += K^T dS^T : A[i = d][k = key] = K[key][d]
            const float* ka = Ks + (32 * sub + 4 * hi) * LD + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = (r & 3) + 8 * (r >> 2);
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kl * LD], dp[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kl * LD + 32], dp[r], o1, 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }

    // transpose through a wave-private LDS patch, coalesced row stores into the q third of dqkv
    float* patch = lds + wave * (32 * LD);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = krow(r, hi);
        patch[l31 * LD + d] = o0[r] * kScale;
        patch[l31 * LD + 32 + d] = o1[r] * kScale;
    }
    __syncthreads();
    float* __restrict__ og = a.dqkv + (size_t)f0 * QKV_LD + head * kHeadDim;
    const int orow = lane >> 4, ocol = (lane & 15) * 4;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int ql = orow + 4 * p;
        const int qq = q0 + wave * 32 + ql;
        if (qq < T) *reinterpret_cast<f32x4*>(og + (size_t)qq * QKV_LD + ocol) = *reinterpret_cast<const f32x4*>(patch + ql * LD + ocol);
    }
}

}  // namespace

hipError_t launch_attention_dsum(const float* out, const float* dout, float* dsum, int M, hipStream_t s, const float* factor) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(attn_dsum_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, out, dout, dsum, M, factor);
    return hipGetLastError();
}

hipError_t launch_attention_bwd(const AttnBwdArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.max_frames <= 0 || a.M <= 0) return hipSuccess;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)DQ_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKV_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    hipLaunchKernelGGL(attn_dsum_kernel, dim3((unsigned)((a.M + 3) / 4)), dim3(256), 0, s, a.out, a.dout, a.dsum, a.M, nullptr);
    const int units = a.B * kHeads, slots = (units + 7) / 8;
    const int nkb = (a.max_frames + 127) / 128, nqb = (a.max_frames + QB - 1) / QB;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(slots * nkb * 8)), dim3(256), DKV_LDS_BYTES, s, a, nkb);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(slots * nqb * 8)), dim3(256), DQ_LDS_BYTES, s, a, nqb);
    return hipGetLastError();
}
