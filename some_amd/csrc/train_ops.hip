// Training-side operators (train.py / training/me_task.py:79-111): everything of the
// forward + backward pass of midi_conforms that is not a GEMM or attention.  All HBM-bound row / column movers over
// packed [M, C] fp32 activations (M = B * T_max frames of the padded training batch, clip b = rows [bT, (b+1)T)):
//
//   transpose            dY^T / X^T operands of the weight-gradient GEMMs (the GEMM kernels contract the contiguous axis)
//   column reductions    bias gradients, LayerNorm / BatchNorm gamma-beta gradients, BatchNorm batch statistics,
//                        depthwise-conv tap gradients: deterministic two-pass (per-chunk partials, ordered final sum)
//   layernorm fwd / bwd  nn.LayerNorm(512) with saved mean / rstd            (modules/conform/Gconform.py:48-52)
//   batchnorm fwd / bwd  nn.BatchNorm1d(512) in train mode, running stats    (modules/conv/base_conv.py:56)
//   silu, glu, sigmoid   fwd / bwd                                           (Gconform.py:11-17,27; base_conv.py:22,64-67)
//   axpy, mask_rows, dropout                                                 (Gconform.py:57-61,128-133)
//   depthwise conv fwd / bwd-data / bwd-taps, k = 31, zero padding per clip  (base_conv.py:48-53)
//   BCEWithLogits(mean) and BinaryEMDLoss with their gradients               (training/me_task.py:74-75, modules/losses/bound_loss.py:6-19)
//   AdamW step on flat parameter / gradient / moment arrays                  (configs/two_head_model.yaml:42-47)
#include "internal.h"
#include "split.h"

namespace {

constexpr int kLnDim = kDim;                    // LayerNorm / BatchNorm width of the model (512)

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- transpose ---------------------------------------------------------------------------------------------
template <int SPLIT>      // 0: fp32, 1: SPLIT32 (f16 hi + lo), 2: SPLIT32 slots with bf16 hi, zero lo (split.h)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int M, int N, int ld_in,
                                                         float* __restrict__ out, int ld_out) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 8 * i, n = n0 + tx;
        tile[ty + 8 * i][tx] = (m < M && n < N) ? in[(size_t)m * ld_in + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, m = m0 + tx;
        if (n < N && m < ld_out) {
            const float v = tile[tx][ty + 8 * i];                                        // zeros beyond M
            if (SPLIT) {   // SPLIT32 row (split.h): the 32 columns of this tile are exactly one k-block [32 hi | 32 lo]
                half_t h, l;
                if (SPLIT == 2) { h = bf16_as_half(v); l = (half_t)0.f; }
                else split_f16(v, h, l);
                half_t* blk = reinterpret_cast<half_t*>(out + (size_t)n * ld_out + m0);
                blk[tx] = h;
                blk[32 + tx] = l;
            } else {
                out[(size_t)n * ld_out + m] = v;
            }
        }
    }
}

// ---- attention operand preparation: row-split AND transposed-split copies of one fp32 matrix in one pass -------------------------
// The split-f16 attention kernels read every operand twice - row-major SPLIT32 (split.h) and transposed SPLIT32 - and, in the
// backward pass, dO first multiplied by the power of two that brings its largest element to [2^9, 2^10] (f16 halves have an absolute
// floor of 2^-25).  absmax_partial_kernel leaves 256 per-workgroup maxima of |x| (as bit patterns: order-independent, deterministic);
// split_transpose_kernel derives the factor from them in every workgroup (256 values, one per thread), writes both layouts from
// one 32 x 32 LDS tile and leaves {factor, 1 / factor} for the kernels behind it.
constexpr int kAbsmaxBlocks = 256;

__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ x, int64_t n4, uint32_t* __restrict__ partial) {
    __shared__ uint32_t red[4];
    uint32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)kAbsmaxBlocks * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) m = max(m, __float_as_uint(v[j]) & 0x7fffffffu);      // NaN / inf order above every finite value
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}

// 2^floor(10 - log2(max(a, 1e-30))) for a = |x|max given as its bit pattern, in integer arithmetic (exact); NaN when x held inf / NaN
__device__ __forceinline__ float pow2_factor(uint32_t bits, float& inv) {
    if (bits >= 0x7f800000u) { inv = __uint_as_float(0x7fc00000u); return inv; }
    if (__uint_as_float(bits) < 1e-30f) bits = __float_as_uint(1e-30f);
    const int e = (int)(bits >> 23) - 127;
    const int k = (bits & 0x7fffffu) ? 9 - e : 10 - e;                 // e in [-100, 127]: 2^k and 2^-k are normal numbers
    inv = __uint_as_float((uint32_t)(127 - k) << 23);
    return __uint_as_float((uint32_t)(127 + k) << 23);
}

template <int SPLIT>      // 1: f16 hi + lo, 2: bf16 hi, zero lo (as transpose_kernel)
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ in, int M, int N, float* __restrict__ out_rows,
                                                               float* __restrict__ out_t, int ld_t, const uint32_t* __restrict__ absmax,
                                                               float* __restrict__ factor_out) {
    __shared__ float tile[32][33];
    __shared__ uint32_t red[4];
    float factor = 1.f;
    if (absmax) {
        uint32_t m = absmax[threadIdx.x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        float inv;
        factor = pow2_factor(max(max(red[0], red[1]), max(red[2], red[3])), inv);
        if (factor_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { factor_out[0] = factor; factor_out[1] = inv; }
    }
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 8 * i, n = n0 + tx;                  // N % 32 == 0: n < N
        float v = 0.f;
        if (m < M) {
            v = in[(size_t)m * N + n] * factor;
            half_t h, l;
            if (SPLIT == 2) { h = bf16_as_half(v); l = (half_t)0.f; }
            else split_f16(v, h, l);
            half_t* blk = reinterpret_cast<half_t*>(out_rows + (size_t)m * N + n0);      // k-block n0 / 32 of row m: [32 hi | 32 lo]
            blk[tx] = h;
            blk[32 + tx] = l;
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i;
        const float v = tile[tx][ty + 8 * i];                        // zeros beyond M
        half_t h, l;
        if (SPLIT == 2) { h = bf16_as_half(v); l = (half_t)0.f; }
        else split_f16(v, h, l);
        half_t* blk = reinterpret_cast<half_t*>(out_t + (size_t)n * ld_t + m0);
        blk[tx] = h;
        blk[32 + tx] = l;
    }
}

// ---- column reductions: partial[p][2][N] over row chunk p, then an ordered final sum -----------------------------
constexpr int kChunkRows = 128;      // 162 row chunks x N / 64 column groups at 8 x 2584 frames: enough workgroups to keep HBM busy (512: 1.9 TB/s)

template <class F>
__global__ __launch_bounds__(256) void col_partial_kernel(int M, int N, float* __restrict__ partial, F f) {
    __shared__ float red[2][4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    const int r0 = blockIdx.y * kChunkRows, r1 = min(M, r0 + kChunkRows);
    float s1 = 0.f, s2 = 0.f;
    if (n < N)
        for (int m = r0 + ty; m < r1; m += 4) f(m, n, s1, s2);
    red[0][ty][threadIdx.x & 63] = s1;
    red[1][ty][threadIdx.x & 63] = s2;
    __syncthreads();
    if (ty == 0 && n < N) {
        const int c = threadIdx.x;
        float* p = partial + (size_t)blockIdx.y * 2 * N;
        p[n] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        p[N + n] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// Sums of the chunk partials in double: a workgroup owns 64 columns, its wavefront g (of kColWaves = 16) adds the chunks p = g, g + 16, ...
// in ascending order and the sixteen sub-sums are combined in a fixed binary tree - deterministic for a given chunk count.  (Round 3: four
// wavefronts, 162 chunks at 8 x 2584 frames: 12 - 15 us per call x 57 calls per training step; round 4's 16-row LayerNorm chunks at small
// batches make it 260 chunks - a dependent chain of 65 loads per wavefront was 19 us per call.)
constexpr int kColWaves = 16;
__device__ __forceinline__ bool col_total(const float* __restrict__ partial, int P, int N, int& n, double& a, double& b) {
    __shared__ double red[2][kColWaves][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    n = blockIdx.x * 64 + c;
    a = 0.0; b = 0.0;
    if (n < N) {
#pragma unroll 4
        for (int p = g; p < P; p += kColWaves) { a += partial[(size_t)p * 2 * N + n]; b += partial[(size_t)p * 2 * N + N + n]; }
    }
    red[0][g][c] = a;
    red[1][g][c] = b;
    __syncthreads();
    if (g != 0 || n >= N) return false;
    double t[2][kColWaves];
#pragma unroll
    for (int k = 0; k < kColWaves; ++k) { t[0][k] = red[0][k][c]; t[1][k] = red[1][k][c]; }
#pragma unroll
    for (int w = 1; w < kColWaves; w *= 2)
#pragma unroll
        for (int k = 0; k < kColWaves; k += 2 * w) { t[0][k] += t[0][k + w]; t[1][k] += t[1][k + w]; }
    a = t[0][0];
    b = t[1][0];
    return true;
}

// out1[n] (+)= sum_p partial[p][0][n], out2[n] (+)= sum_p partial[p][1][n]
__global__ __launch_bounds__(64 * kColWaves) void col_final_kernel(const float* __restrict__ partial, int P, int N, float* out1, float* out2,
                                                         int accumulate) {
    int n;
    double a, b;
    if (!col_total(partial, P, N, n, a, b)) return;
    if (out1) out1[n] = (accumulate ? out1[n] : 0.f) + (float)a;
    if (out2) out2[n] = (accumulate ? out2[n] : 0.f) + (float)b;
}

struct ColsumF {
    const float* x; int ld;
    __device__ void operator()(int m, int n, float& s1, float& s2) const { s1 += x[(size_t)m * ld + n]; (void)s2; }
};
struct RowWeightedF {   // s1 = sum_m w[m] x[m, n] (weight gradient of a one-output nn.Linear), s2 = sum_m w[m] (its bias gradient)
    const float* w; int ldw; const float* x; int ld;
    __device__ void operator()(int m, int n, float& s1, float& s2) const { const float v = w[(size_t)m * ldw]; s1 += v * x[(size_t)m * ld + n]; s2 += v; }
};
struct ColStatsF {
    const float* x; int ld;
    __device__ void operator()(int m, int n, float& s1, float& s2) const { const float v = x[(size_t)m * ld + n]; s1 += v; s2 += v * v; }
};
struct BnGradF {         // per-COLUMN statistics
    const float* dy; const float* x; const float* mean; const float* rstd; int ld;
    __device__ void operator()(int m, int n, float& s1, float& s2) const {
        const float d = dy[(size_t)m * ld + n];
        s1 += d;
        s2 += d * (x[(size_t)m * ld + n] - mean[n]) * rstd[n];
    }
};

// ---- LayerNorm (width 512): one wave per row, lane owns columns 4 l .. 4 l + 3 and 256 + 4 l .. -------------------
// OUT16: 0 fp32 output; 1 / 2: the output is written as f16 / bf16 (the 16-bit operand of the FFN's first GEMM, train_gemm16s.hip) - same
// arithmetic, one rounding at the store
template <int OUT16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                      void* __restrict__ y_out, float* __restrict__ mean, float* __restrict__ rstd, int M) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (size_t)row * kLnDim + lane * 4);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(x + (size_t)row * kLnDim + 256 + lane * 4);
    float s = (a0[0] + a0[1]) + (a0[2] + a0[3]) + (a1[0] + a1[1]) + (a1[2] + a1[3]);
    const float mu = wave_sum(s) * (1.0f / kLnDim);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { q += (a0[i] - mu) * (a0[i] - mu); q += (a1[i] - mu) * (a1[i] - mu); }
    const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / kLnDim) + 1e-5f);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(g + lane * 4), g1 = *reinterpret_cast<const f32x4*>(g + 256 + lane * 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + lane * 4), b1 = *reinterpret_cast<const f32x4*>(b + 256 + lane * 4);
    f32x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o0[i] = (a0[i] - mu) * rs * g0[i] + b0[i]; o1[i] = (a1[i] - mu) * rs * g1[i] + b1[i]; }
    if constexpr (OUT16 == 0) {
        float* y = static_cast<float*>(y_out);
        *reinterpret_cast<f32x4*>(y + (size_t)row * kLnDim + lane * 4) = o0;
        *reinterpret_cast<f32x4*>(y + (size_t)row * kLnDim + 256 + lane * 4) = o1;
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        auto pk = [](float a, float c) -> uint32_t {
            const f2 v = {a, c};
            if constexpr (OUT16 == 2) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
            else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
        };
        uint16_t* y = static_cast<uint16_t*>(y_out);
        const u2 w0 = {pk(o0[0], o0[1]), pk(o0[2], o0[3])}, w1 = {pk(o1[0], o1[1]), pk(o1[2], o1[3])};
        *reinterpret_cast<u2*>(y + (size_t)row * kLnDim + lane * 4) = w0;
        *reinterpret_cast<u2*>(y + (size_t)row * kLnDim + 256 + lane * 4) = w1;
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)), and the column partials of dbeta = sum dy, dgamma = sum dy xhat, in ONE pass over
// dy and x (a stand-alone column reduction would read both again: 210 -> 126 MB per call at 8 x 2584 frames, one launch less).  A workgroup
// owns kLnChunk rows, its
// wavefront w the rows w, w + 8, ...; every lane keeps the running sums of its 8 columns, the eight wavefronts' sums are combined in a fixed
// tree through LDS and written as partial[chunk][2][512] for col_final_kernel (fixed order: deterministic).
// Rows per workgroup (round 4): 128 filled 13 % of the CUs at the reference's batch shape (4 160 frames -> 33 workgroups, 30 us per call for
// 25 MB of traffic), so the chunk shrinks with M - 16 rows up to 8 192 frames, 64 up to 16 384, else 128; the partial
// planes are summed in chunk order whatever their number (deterministic for a given M).  Measured (interleaved, two lanes): 8 x 520 frames
// 9.7 -> 9.2 ms per step with 16-row chunks; at 8 x 2584 frames 128 rows stay best (26.5 - 27.0 ms vs 27.5 - 28.0 with 32, 27.0 - 27.4 with 64:
// the two lanes' kernels fill the CUs together and more partial planes only add work).
constexpr int kLnWaves = 8;
#ifdef LN_CHUNK_FIXED                 // (tools/build_variant.py A/B builds)
inline int ln_chunk_rows(int) { return LN_CHUNK_FIXED; }
#else
inline int ln_chunk_rows(int M) { return M <= 8192 ? 16 : M <= 16384 ? 64 : 128; }
#endif
__global__ __launch_bounds__(64 * kLnWaves) void ln_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ add, float* __restrict__ dx, float* __restrict__ partial, int M,
                                                            int kLnChunk) {
    // add (may be null): a second gradient of x - the residual branch around the LayerNorm - summed into dx here instead of by a
    // separate pass; (ln gradient) + add as its own rounding step (no FMA), i.e. the bits a separate addition would give
    __shared__ float red[2][kLnWaves][kLnDim];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * kLnChunk;
    float gv[8], sb[8], sg[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(g + h * 256 + lane * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { gv[h * 4 + i] = t[i]; sb[h * 4 + i] = 0.f; sg[h * 4 + i] = 0.f; }
    }
    for (int j = 0; j < kLnChunk / kLnWaves; ++j) {
        const int row = r0 + wave + kLnWaves * j;
        if (row >= M) break;
        const float mu = mean[row], rs = rstd[row];
        float gd[8], xh[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = h * 256 + lane * 4;
            const f32x4 d = *reinterpret_cast<const f32x4*>(dy + (size_t)row * kLnDim + c);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)row * kLnDim + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = h * 4 + i;
                gd[k] = d[i] * gv[k];
                xh[k] = (xv[i] - mu) * rs;
                s1 += gd[k];
                s2 += gd[k] * xh[k];
                sb[k] += d[i];
                sg[k] += d[i] * (xv[i] - mu) * rs;
            }
        }
        const float c1 = wave_sum(s1) * (1.0f / kLnDim), c2 = wave_sum(s2) * (1.0f / kLnDim);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = rs * (gd[h * 4 + i] - c1 - xh[h * 4 + i] * c2);
            if (add != nullptr) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(add + (size_t)row * kLnDim + h * 256 + lane * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = __fadd_rn(a[i], o[i]);
            }
            *reinterpret_cast<f32x4*>(dx + (size_t)row * kLnDim + h * 256 + lane * 4) = o;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[0][wave][h * 256 + lane * 4 + i] = sb[h * 4 + i];
            red[1][wave][h * 256 + lane * 4 + i] = sg[h * 4 + i];
        }
    __syncthreads();
    float* p = partial + (size_t)blockIdx.x * 2 * kLnDim;
    for (int c = threadIdx.x; c < kLnDim; c += 64 * kLnWaves) {
        p[c] = ((red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c])) + ((red[0][4][c] + red[0][5][c]) + (red[0][6][c] + red[0][7][c]));
        p[kLnDim + c] = ((red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c])) + ((red[1][4][c] + red[1][5][c]) + (red[1][6][c] + red[1][7][c]));
    }
}

// ---- BatchNorm1d(train) over the M rows of [M, C] -------------------------------------------------------------------
__global__ __launch_bounds__(64 * kColWaves) void bn_finalize_kernel(const float* __restrict__ partial, int P, int C, int M, float eps, float momentum,
                                                           float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                           float* running_mean, float* running_var) {
    int n;
    double a, b;
    if (!col_total(partial, P, C, n, a, b)) return;
    const double mu = a / M;
    double var = b / M - mu * mu;                         // biased (normalisation); double: no cancellation issue
    if (var < 0.0) var = 0.0;
    save_mean[n] = (float)mu;
    save_rstd[n] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * M / (M - 1) : var;
        running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (float)mu;
        running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)unbiased;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ y,
                                                        int64_t n4, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (v[k] - mean[c + k]) * rstd[c + k] * g[c + k] + b[c + k];
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
}

// dx = g rstd (dy - dbeta / M - xhat dgamma / M)
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ dgamma, const float* __restrict__ dbeta, float* __restrict__ dx,
                                                         int64_t n4, int C, float inv_m) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + i * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float xh = (v[k] - mean[c + k]) * rstd[c + k];
        o[k] = g[c + k] * rstd[c + k] * (d[k] - dbeta[c + k] * inv_m - xh * dgamma[c + k] * inv_m);
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
}

// ---- element-wise ----------------------------------------------------------------------------------------------------
enum { ELT_SILU_FWD, ELT_SILU_BWD, ELT_SIGMOID_FWD, ELT_SIGMOID_BWD, ELT_AXPY, ELT_DROPOUT, ELT_SILU_DROP_FWD, ELT_SILU_DROP_BWD, ELT_AXPY_DROP };

__device__ __forceinline__ uint32_t mix32(uint64_t z) {          // splitmix64 finaliser -> 32 random bits per element
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

template <int OP>
__device__ __forceinline__ float eltwise_one(float x, float bv, int64_t i, float alpha, float p, uint64_t seed) {
    // dropout factor of element i: 0 with probability p, else 1 / (1 - p); a pure function of (seed, i), so the backward
    // regenerates the forward's mask
    float keep = 1.f;
    if (OP == ELT_DROPOUT || OP == ELT_SILU_DROP_FWD || OP == ELT_SILU_DROP_BWD || OP == ELT_AXPY_DROP) {
        const float u = (float)(mix32(seed + (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
        keep = u >= p ? 1.0f / (1.0f - p) : 0.f;
    }
    if (OP == ELT_SILU_FWD) return x * sigmoid_(x);
    if (OP == ELT_SILU_BWD) { const float s = sigmoid_(bv); return x * s * (1.f + bv * (1.f - s)); }       // a = dy, b = pre-activation
    if (OP == ELT_SIGMOID_FWD) return sigmoid_(x);
    if (OP == ELT_SIGMOID_BWD) return x * bv * (1.f - bv);                                                // a = dy, b = sigmoid output
    if (OP == ELT_AXPY) return alpha * x + bv;                                                            // alpha a (+ b)
    if (OP == ELT_DROPOUT) return x * keep;
    if (OP == ELT_SILU_DROP_FWD) return x * sigmoid_(x) * keep;                                           // dropout(silu(a))
    if (OP == ELT_SILU_DROP_BWD) { const float s = sigmoid_(bv); return x * keep * s * (1.f + bv * (1.f - s)); }
    return alpha * (x * keep) + bv;                                                                       // alpha dropout(a) (+ b)
}

// four consecutive elements per thread: 16-byte accesses when all three pointers allow them, scalar otherwise and for the tail
template <int OP>
__global__ __launch_bounds__(256) void eltwise_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                       int64_t n, float alpha, float p, uint64_t seed) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const bool aligned = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned && i + 4 <= n) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + i);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (b) bv = *reinterpret_cast<const f32x4*>(b + i);
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = eltwise_one<OP>(x[k], bv[k], i + k, alpha, p, seed);
        *reinterpret_cast<f32x4*>(out + i) = r;
    } else {
        for (int64_t k = i; k < n && k < i + 4; ++k) out[k] = eltwise_one<OP>(a[k], b ? b[k] : 0.f, k, alpha, p, seed);
    }
}

// GLU over the channel halves of [M, 2C]: y = x[:, :C] * sigmoid(x[:, C:]); four adjacent channels per thread (C % 4 == 0)
__global__ __launch_bounds__(256) void glu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t M, int C) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= M * C) return;
    const int64_t m = i / C;
    const int c = (int)(i % C);
    const f32x4 o = *reinterpret_cast<const f32x4*>(x + m * 2 * C + c), gt = *reinterpret_cast<const f32x4*>(x + m * 2 * C + C + c);
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = o[k] * sigmoid_(gt[k]);
    *reinterpret_cast<f32x4*>(y + i) = r;
}
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t M, int C) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= M * C) return;
    const int64_t m = i / C;
    const int c = (int)(i % C);
    const f32x4 o = *reinterpret_cast<const f32x4*>(x + m * 2 * C + c), gt = *reinterpret_cast<const f32x4*>(x + m * 2 * C + C + c);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + i);
    f32x4 da, dg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sg = sigmoid_(gt[k]);
        da[k] = d[k] * sg;
        dg[k] = d[k] * o[k] * sg * (1.f - sg);
    }
    *reinterpret_cast<f32x4*>(dx + m * 2 * C + c) = da;
    *reinterpret_cast<f32x4*>(dx + m * 2 * C + C + c) = dg;
}
// rows with mask == 0 become 0 (masked_fill(~mask, 0)); the same op maps dy -> dx
__global__ __launch_bounds__(256) void mask_rows_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, float* __restrict__ y, int64_t M, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * C) return;
    y[i] = mask[i / C] ? x[i] : 0.f;
}

// ---- depthwise conv k = 31 over time, zero padding at clip edges; taps [31][C] -------------------------------------
constexpr int kTaps = kConvK;

// a thread owns two adjacent channels and kDwRows consecutive frames: the 8 + 30 input rows they need are loaded once into a
// register window (4.75 row loads + 3.9 tap loads per output instead of 31 + 31)
constexpr int kDwRows = 8;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void dwconv_train_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const int32_t* __restrict__ frame_offsets, int B, float* __restrict__ y, int C, int flip) {
    constexpr int H = kTaps / 2, WIN = kDwRows + 2 * H;
    const int b = blockIdx.z;
    const int f0 = frame_offsets[b], T = frame_offsets[b + 1] - f0;
    const int t0 = blockIdx.y * kDwRows;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (t0 >= T || c >= C) return;
    f32x2_t xw[WIN];
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
        const int u = t0 + j - H;
        xw[j] = (u >= 0 && u < T) ? *reinterpret_cast<const f32x2_t*>(x + (size_t)(f0 + u) * C + c) : f32x2_t{0.f, 0.f};
    }
    f32x2_t acc[kDwRows];
    const f32x2_t bv = bias ? *reinterpret_cast<const f32x2_t*>(bias + c) : f32x2_t{0.f, 0.f};
#pragma unroll
    for (int r = 0; r < kDwRows; ++r) acc[r] = bv;
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
        const f32x2_t wv = *reinterpret_cast<const f32x2_t*>(w + (size_t)(flip ? kTaps - 1 - k : k) * C + c);
#pragma unroll
        for (int r = 0; r < kDwRows; ++r) acc[r] += xw[r + k] * wv;
    }
#pragma unroll
    for (int r = 0; r < kDwRows; ++r)
        if (t0 + r < T) *reinterpret_cast<f32x2_t*>(y + (size_t)(f0 + t0 + r) * C + c) = acc[r];
}

// tap gradients: partial[p][k][c] = sum over the chunk's rows t of dy[t, c] x[t + k - 15, c] (inside the row's clip).
// Short chunks (64 rows): the per-thread work is a chain of dependent row iterations, so the grid has to be wide.
constexpr int kDwChunk = 64;
__global__ __launch_bounds__(256) void dwconv_bwd_w_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    const int32_t* __restrict__ clip_of_row, const int32_t* __restrict__ frame_offsets,
                                                                    int M, int C, float* __restrict__ partial) {
    __shared__ float red[4][kTaps][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    const int r0 = blockIdx.y * kDwChunk, r1 = min(M, r0 + kDwChunk);
    float acc[kTaps];
#pragma unroll
    for (int k = 0; k < kTaps; ++k) acc[k] = 0.f;
    // wavefront ty owns the kDwChunk / 4 CONSECUTIVE rows [a, e]: away from clip edges (all but ~1 % of the chunks) the x values
    // its rows need are one sliding window of 16 + 30 rows, loaded once (46 loads instead of 16 x 31) and used from registers
    constexpr int R = kDwChunk / 4, H = kTaps / 2;
    const int a = r0 + ty * R, e = min(r1, a + R) - 1;
    if (c < C && a <= e) {
        const int b0 = clip_of_row[a], b1 = clip_of_row[e];
        const int lo0 = frame_offsets[b0], hi0 = frame_offsets[b0 + 1];
        if (b0 == b1 && a - H >= lo0 && e + H < hi0) {
            float xw[R + 2 * H];
#pragma unroll
            for (int j = 0; j < R + 2 * H; ++j) xw[j] = a + j - H <= e + H ? x[(size_t)(a + j - H) * C + c] : 0.f;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float d = a + i <= e ? dy[(size_t)(a + i) * C + c] : 0.f;
#pragma unroll
                for (int k = 0; k < kTaps; ++k) acc[k] += d * xw[i + k];
            }
        } else {
            for (int m = a; m <= e; ++m) {
                const int b = clip_of_row[m];
                const int lo = frame_offsets[b], hi = frame_offsets[b + 1];
                const float d = dy[(size_t)m * C + c];
#pragma unroll
                for (int k = 0; k < kTaps; ++k) {
                    const int u = m + k - H;
                    if (u >= lo && u < hi) acc[k] += d * x[(size_t)u * C + c];
                }
            }
        }
    }
    // the four wavefronts' sums of all 31 taps through LDS in ONE exchange (round 4: a barrier pair per tap - 62 barriers - was most of the
    // kernel's 72 us); wavefront ty then combines the taps k = ty, ty + 4, ... in the same fixed tree (0 + 1) + (2 + 3)
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < kTaps; ++k) red[ty][k][l] = acc[k];
    __syncthreads();
    if (c < C)
        for (int k = ty; k < kTaps; k += 4)
            partial[((size_t)blockIdx.y * kTaps + k) * C + c] = (red[0][k][l] + red[1][k][l]) + (red[2][k][l] + red[3][k][l]);
}
// weight_c > 0: dw is the Conv1d weight's own layout [C][taps] (element (tap k, channel c) of the tap-major sums goes to c * taps + k) -
// the parameter's gradient array itself; 0: tap-major [taps][C] as the sums lie
__global__ __launch_bounds__(256) void dwconv_bwd_w_final_kernel(const float* __restrict__ partial, int P, int n, float* __restrict__ dw, int accumulate,
                                                                 int weight_c) {
    // 64 elements per workgroup, wavefront g sums the chunks p = g, g + 4, ... in ascending order, combined as (0 + 1) + (2 + 3)
    __shared__ double red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    double a = 0.0;
    if (i < n)
        for (int p = g; p < P; p += 4) a += partial[(size_t)p * n + i];
    red[g][c] = a;
    __syncthreads();
    if (g == 0 && i < n) {
        const int o = weight_c > 0 ? (i % weight_c) * (n / weight_c) + i / weight_c : i;
        dw[o] = (accumulate ? dw[o] : 0.f) + (float)((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    }
}

// ---- losses ------------------------------------------------------------------------------------------------------------
// BCEWithLogitsLoss(reduction='mean'): partial sums per block + gradient (sigmoid(x) - t) / n
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t n, float inv_n,
                                                   float* __restrict__ dx, double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i], tv = t[i];
        s += (double)(fmaxf(v, 0.f) - v * tv + log1pf(__expf(-fabsf(v))));
        if (dx) dx[i] = (sigmoid_(v) - tv) * inv_n;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sum_partials_kernel(const double* __restrict__ partial, int P, double scale, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) s += partial[p];
        out[0] = (float)(s * scale);
    }
}

// nn.CrossEntropyLoss(ignore_index) over rows of logits [M, N] (training/me_quant_task.py:42,77: 129 classes, target -1 = padding):
//   loss = mean over the rows with target != ignore of (logsumexp(x) - x[target]);  d loss / dx = (softmax(x) - onehot) / count on those rows, 0 elsewhere.
// ce_count_kernel: one workgroup counts the valid rows (device scalar: the mean's denominator is needed by every gradient element);
// ce_kernel: a wavefront per row (max, sum of exp, gradient), the workgroup's four row losses -> one partial.
__global__ __launch_bounds__(256) void ce_count_kernel(const int64_t* __restrict__ target, int M, int64_t ignore, float* __restrict__ count) {
    __shared__ int red[4];
    int c = 0;
    for (int i = threadIdx.x; i < M; i += 256) c += target[i] != ignore;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) count[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ x, const int64_t* __restrict__ target, int M, int N, int64_t ignore,
                                                  const float* __restrict__ count, float* __restrict__ dx, double* __restrict__ partial) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.0f / count[0];                   // count = 0: inf, and the mean over no rows is NaN as in torch
    double acc = 0.0;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * N;
        const int64_t t = target[row];
        if (t == ignore) {
            if (dx) for (int j = lane; j < N; j += 64) dx[(size_t)row * N + j] = 0.f;
            continue;
        }
        if (t < 0 || t >= N) {
            // a class outside [0, N) that is not ignore_index (torch device-asserts here: nn.CrossEntropyLoss, me_quant_task.py:13-78): never
            // an out-of-bounds read - the loss AND this row's gradient become NaN, which the trainer's gradient-norm check reports as a
            // non-finite update (e.g. a quantised dataset with class 128 under a config whose midi_num_bins is not 129)
            if (lane == 0) acc += (double)NAN;
            if (dx) for (int j = lane; j < N; j += 64) dx[(size_t)row * N + j] = NAN;
            continue;
        }
        float mx = -INFINITY;
        for (int j = lane; j < N; j += 64) mx = fmaxf(mx, xr[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float se = 0.f;
        for (int j = lane; j < N; j += 64) se += __expf(xr[j] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
        const float lse = mx + __logf(se);
        if (lane == 0) acc += (double)(lse - xr[t]);
        if (dx) {
            const float rs = inv / se;
            for (int j = lane; j < N; j += 64) dx[(size_t)row * N + j] = __expf(xr[j] - mx) * rs - (j == (int)t ? inv : 0.f);
        }
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void ce_finish_kernel(const double* __restrict__ partial, int P, const float* __restrict__ count, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) s += partial[p];
        out[0] = (float)(s / (double)count[0]);
    }
}

// sum of squares of a flat array (global gradient norm for clip_grad_norm); NaN / inf propagate into the result
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double v = x[i]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sum_partials_f64_kernel(const double* __restrict__ partial, int P, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) s += partial[p];
        out[0] = s;
    }
}

// BinaryEMDLoss (bound_loss.py:12-19, bidirectional=False): mean |cumsum(pred) - cumsum(gt)| / sqrt(T) over [B, T];
// grad pred[b, t] = sum_{t' >= t} sign(cp - cg)[t'] / (sqrt(T) B T).  One workgroup per row, one contiguous segment per
// thread: segment sums -> exclusive scan -> per-element cumsum, |.| and sign -> suffix sums of the signs.
__global__ __launch_bounds__(256) void emd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int T, float inv_scale, float inv_n,
                                                   float* __restrict__ dpred, double* __restrict__ partial) {
    __shared__ double seg[256];
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = pred + (size_t)b * T;
    const float* g = gt + (size_t)b * T;
    float* dp = dpred ? dpred + (size_t)b * T : nullptr;
    const int per = (T + 255) / 256;
    const int t0 = min(T, tid * per), t1 = min(T, t0 + per);
    double s = 0.0;
    for (int t = t0; t < t1; ++t) s += (double)p[t] - (double)g[t];
    seg[tid] = s;
    __syncthreads();
    if (tid == 0) { double run = 0.0; for (int i = 0; i < 256; ++i) { const double v = seg[i]; seg[i] = run; run += v; } }   // exclusive scan
    __syncthreads();
    double run = seg[tid], loss = 0.0;
    int cnt = 0;                                      // sum of signs inside the segment
    for (int t = t0; t < t1; ++t) {
        run += (double)p[t] - (double)g[t];
        const float d = (float)run * inv_scale;       // (cp - cg) / scale
        loss += (double)fabsf(d);
        const int sg = (d > 0.f) - (d < 0.f);
        cnt += sg;
        if (dp) dp[t] = (float)sg;                    // parked; turned into the suffix sum below
    }
    __syncthreads();
    seg[tid] = (double)cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = loss;
    __syncthreads();
    if (tid == 0) {
        partial[b] = (red[0] + red[1]) + (red[2] + red[3]);
        double r = 0.0;                                // signs of all LATER segments
        for (int i = 255; i >= 0; --i) { const double v = seg[i]; seg[i] = r; r += v; }
    }
    __syncthreads();
    if (dp) {
        float acc = (float)seg[tid];
        const float k = inv_scale * inv_n;
        for (int t = t1 - 1; t >= t0; --t) { acc += dp[t]; dp[t] = acc * k; }
    }
}

// AdamW (torch.optim.AdamW semantics, amsgrad off): decoupled decay, bias-corrected moments
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                     int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                                     float bc1, float bc2_sqrt, float grad_scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.f - lr * weight_decay);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

// The same update with the gradient scale taken from DEVICE memory: grad_scale = min(1, clip / (sqrt(sumsq) / denom + 1e-6)) / denom - the
// trainer's clip_grad_norm arithmetic (torch.nn.utils.clip_grad_norm_; task.py) in the same IEEE double operations, so the step needs no
// host readback of the gradient norm.  A non-finite sumsq leaves everything untouched (the host finds out one step later).
__global__ __launch_bounds__(256) void adamw_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                                          float bc1, float bc2_sqrt, const double* __restrict__ sumsq, double clip, double denom) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double ss = *sumsq;
    if (!(ss == ss) || ss == INFINITY) return;
    const double norm = sqrt(ss) / denom;
    const double coef = clip > 0.0 ? fmin(1.0, clip / (norm + 1e-6)) : 1.0;
    const float grad_scale = (float)(coef / denom);
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.f - lr * weight_decay);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float den = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / den);
}

}  // namespace

// out[i] = sum over slices of partial[s][i] (split-K partial planes), slice order
static __global__ __launch_bounds__(256) void reduce_slices_kernel(const float* __restrict__ partial, int slices, size_t n4, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = *reinterpret_cast<const f32x4*>(partial + i * 4);
    for (int s = 1; s < slices; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(partial + ((size_t)s * n4 + i) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += v[k];
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = acc;
}

hipError_t launch_reduce_slices(const float* partial, int slices, size_t n, float* out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, partial, slices, n4, out);
    return hipGetLastError();
}

// Weight-gradient planes [slice][M][ldc] (ldc = N + 4: the GEMM's N columns, then the column-sum column) -> the parameter-gradient
// arrays themselves: dW [M, N] contiguous and db [M], summed in slice order, added to what is there when `accumulate` (the flat
// gradient buffer across micro-batches) - no intermediate tensor, no separate add.
static __device__ __forceinline__ void reduce_wgrad_body(const float* __restrict__ partial, int slices, size_t stride, int M, int N, int ldc,
                                                         float* __restrict__ dw, float* __restrict__ db, int accumulate, size_t i) {
    const int n4 = N / 4;
    if (i >= (size_t)M * n4) return;
    const int row = (int)(i / n4), c = (int)(i % n4) * 4;
    const float* src = partial + (size_t)row * ldc + c;
    f32x4 acc = *reinterpret_cast<const f32x4*>(src);
    for (int s = 1; s < slices; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)s * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += v[k];
    }
    f32x4* out = reinterpret_cast<f32x4*>(dw + (size_t)row * N + c);
    if (accumulate) {
        const f32x4 o = *out;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += o[k];
    }
    *out = acc;
    if (db != nullptr && c == 0) {
        float b = partial[(size_t)row * ldc + N];
        for (int s = 1; s < slices; ++s) b += partial[(size_t)s * stride + (size_t)row * ldc + N];
        db[row] = (accumulate ? db[row] : 0.f) + b;
    }
}

static __global__ __launch_bounds__(256) void reduce_wgrad_kernel(const float* __restrict__ partial, int slices, size_t stride, int M, int N, int ldc,
                                                                  float* __restrict__ dw, float* __restrict__ db, int accumulate) {
    reduce_wgrad_body(partial, slices, stride, M, N, ldc, dw, db, accumulate, (size_t)blockIdx.x * 256 + threadIdx.x);
}

// Up to kWgradTable reductions in one launch (the weight-gradient lanes' deferred reductions): workgroup -> entry by the running block
// counts (a scalar scan of <= 32 integers), then the body of reduce_wgrad_kernel on that entry - the same additions in the same order.
struct WgradTableArgs { WgradReduce e[kWgradTable]; int begin[kWgradTable + 1]; int n; };
static __global__ __launch_bounds__(256) void reduce_wgrad_table_kernel(WgradTableArgs t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.begin[k + 1]) ++k;
    const WgradReduce& r = t.e[k];
    reduce_wgrad_body(r.partial, r.slices, r.stride, r.M, r.N, r.ldc, r.dw, r.db, r.accumulate, (size_t)((int)blockIdx.x - t.begin[k]) * 256 + threadIdx.x);
}

hipError_t launch_reduce_wgrad_table(const WgradReduce* entries, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > kWgradTable) return hipErrorInvalidValue;
    WgradTableArgs t{};
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        const WgradReduce& r = entries[k];
        if (r.M <= 0 || r.N <= 0 || (r.N & 3) || (r.ldc & 3) || (reinterpret_cast<uintptr_t>(r.dw) & 15)) return hipErrorInvalidValue;
        t.e[k] = r;
        t.begin[k] = blocks;
        blocks += (int)(((size_t)r.M * (r.N / 4) + 255) / 256);
    }
    t.begin[n] = blocks;
    t.n = n;
    hipLaunchKernelGGL(reduce_wgrad_table_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t);
    return hipGetLastError();
}

hipError_t launch_reduce_wgrad(const float* partial, int slices, size_t stride, int M, int N, int ldc, float* dw, float* db, int accumulate,
                               hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if ((N & 3) || (ldc & 3) || (reinterpret_cast<uintptr_t>(dw) & 15)) return hipErrorInvalidValue;
    const size_t n = (size_t)M * (N / 4);
    hipLaunchKernelGGL(reduce_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, partial, slices, stride, M, N, ldc, dw, db, accumulate);
    return hipGetLastError();
}

// ---- launchers -----------------------------------------------------------------------------------------------------------
static inline int n_chunks(int M) { return (M + kChunkRows - 1) / kChunkRows; }
size_t train_col_scratch_bytes(int M, int N) { return (size_t)n_chunks(M) * 2 * N * sizeof(float); }
size_t train_ln_scratch_bytes(int M) {      // ln_bwd_fused_kernel: one [2][512] partial plane per chunk of ln_chunk_rows(M) rows
    const int chunk = ln_chunk_rows(M);
    return (size_t)((M + chunk - 1) / chunk) * 2 * 512 * sizeof(float);
}
size_t train_dwconv_w_scratch_bytes(int M, int C) { return (size_t)((M + kDwChunk - 1) / kDwChunk) * kTaps * C * sizeof(float); }

hipError_t launch_transpose(const float* in, int M, int N, int ld_in, float* out, int ld_out, int split_out, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    dim3 grid((unsigned)((ld_out + 31) / 32), (unsigned)((N + 31) / 32));
    if (split_out == 2) hipLaunchKernelGGL(transpose_kernel<2>, grid, dim3(256), 0, s, in, M, N, ld_in, out, ld_out);
    else if (split_out) hipLaunchKernelGGL(transpose_kernel<1>, grid, dim3(256), 0, s, in, M, N, ld_in, out, ld_out);
    else hipLaunchKernelGGL(transpose_kernel<0>, grid, dim3(256), 0, s, in, M, N, ld_in, out, ld_out);
    return hipGetLastError();
}

hipError_t launch_split_transpose(const float* in, int M, int N, float* out_rows, float* out_t, int ld_t, int bf16, int prescale,
                                  uint32_t* absmax, float* factor_out, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if ((N & 31) || (ld_t & 31) || ld_t < M || (prescale && !absmax)) return hipErrorInvalidValue;
    if (prescale) hipLaunchKernelGGL(absmax_partial_kernel, dim3(kAbsmaxBlocks), dim3(256), 0, s, in, (int64_t)M * N / 4, absmax);
    const uint32_t* am = prescale ? absmax : nullptr;
    dim3 grid((unsigned)(ld_t / 32), (unsigned)(N / 32));
    if (bf16) hipLaunchKernelGGL(split_transpose_kernel<2>, grid, dim3(256), 0, s, in, M, N, out_rows, out_t, ld_t, am, factor_out);
    else hipLaunchKernelGGL(split_transpose_kernel<1>, grid, dim3(256), 0, s, in, M, N, out_rows, out_t, ld_t, am, factor_out);
    return hipGetLastError();
}

template <class F>
static hipError_t col_reduce(int M, int N, F f, float* out1, float* out2, int accumulate, float* scratch, hipStream_t s) {
    const int P = n_chunks(M);
    hipLaunchKernelGGL(col_partial_kernel<F>, dim3((unsigned)((N + 63) / 64), (unsigned)P), dim3(256), 0, s, M, N, scratch, f);
    hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64 * kColWaves), 0, s, scratch, P, N, out1, out2, accumulate);
    return hipGetLastError();
}

hipError_t launch_colsum(const float* x, int M, int N, int ld, float* out, int accumulate, float* scratch, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    return col_reduce(M, N, ColsumF{x, ld}, out, nullptr, accumulate, scratch, s);
}

hipError_t launch_weighted_colsum(const float* w, int ldw, const float* x, int M, int N, int ld, float* out, float* wsum, float* scratch, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    return col_reduce(M, N, RowWeightedF{w, ldw, x, ld}, out, wsum, 0, scratch, s);
}

hipError_t launch_ln_fwd(const float* x, const float* g, const float* b, void* y, float* mean, float* rstd, int M, int out16, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    const dim3 grid((unsigned)((M + 3) / 4));
    if (out16 == 0) hipLaunchKernelGGL(ln_fwd_kernel<0>, grid, dim3(256), 0, s, x, g, b, y, mean, rstd, M);
    else if (out16 == 1) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, dim3(256), 0, s, x, g, b, y, mean, rstd, M);
    else hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, dim3(256), 0, s, x, g, b, y, mean, rstd, M);
    return hipGetLastError();
}

hipError_t launch_ln_bwd(const float* dy, const float* x, const float* g, const float* mean, const float* rstd, const float* add, float* dx,
                         float* dgamma, float* dbeta, int accumulate, int M, float* scratch, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    const int chunk = ln_chunk_rows(M);
    const int P = (M + chunk - 1) / chunk;
    hipLaunchKernelGGL(ln_bwd_fused_kernel, dim3((unsigned)P), dim3(64 * kLnWaves), 0, s, dy, x, g, mean, rstd, add, dx, scratch, M, chunk);
    hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)(kLnDim / 64)), dim3(64 * kColWaves), 0, s, scratch, P, kLnDim, dbeta, dgamma, accumulate);
    return hipGetLastError();
}

hipError_t launch_bn_fwd(const float* x, const float* g, const float* b, int M, int C, float eps, float momentum, float* running_mean,
                         float* running_var, float* y, float* save_mean, float* save_rstd, float* scratch, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    const int P = n_chunks(M);
    hipLaunchKernelGGL(col_partial_kernel<ColStatsF>, dim3((unsigned)((C + 63) / 64), (unsigned)P), dim3(256), 0, s, M, C, scratch, ColStatsF{x, C});
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64 * kColWaves), 0, s, scratch, P, C, M, eps, momentum, save_mean, save_rstd,
                       running_mean, running_var);
    const int64_t n4 = (int64_t)M * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, g, b, save_mean, save_rstd, y, n4, C);
    return hipGetLastError();
}

hipError_t launch_bn_bwd(const float* dy, const float* x, const float* g, const float* save_mean, const float* save_rstd, int M, int C,
                         float* dx, float* dgamma, float* dbeta, float* scratch, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipError_t e = col_reduce(M, C, BnGradF{dy, x, save_mean, save_rstd, C}, dbeta, dgamma, 0, scratch, s);
    if (e != hipSuccess) return e;
    const int64_t n4 = (int64_t)M * C / 4;
    hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dy, x, g, save_mean, save_rstd, dgamma, dbeta, dx, n4, C,
                       1.0f / (float)M);
    return hipGetLastError();
}

hipError_t launch_eltwise(int op, const float* a, const float* b, float* out, int64_t n, float alpha, float p, uint64_t seed, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 1023) / 1024));
#define ELT_CASE(OP) case OP: hipLaunchKernelGGL(eltwise_kernel<OP>, grid, dim3(256), 0, s, a, b, out, n, alpha, p, seed); break;
    switch (op) {
        ELT_CASE(ELT_SILU_FWD) ELT_CASE(ELT_SILU_BWD) ELT_CASE(ELT_SIGMOID_FWD) ELT_CASE(ELT_SIGMOID_BWD) ELT_CASE(ELT_AXPY)
        ELT_CASE(ELT_DROPOUT) ELT_CASE(ELT_SILU_DROP_FWD) ELT_CASE(ELT_SILU_DROP_BWD) ELT_CASE(ELT_AXPY_DROP)
        default: return hipErrorInvalidValue;
    }
#undef ELT_CASE
    return hipGetLastError();
}

hipError_t launch_glu(const float* dy, const float* x, float* out, int64_t M, int C, int backward, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    if ((C & 3) || ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((M * C / 4 + 255) / 256));
    if (backward) hipLaunchKernelGGL(glu_bwd_kernel, grid, dim3(256), 0, s, dy, x, out, M, C);
    else hipLaunchKernelGGL(glu_fwd_kernel, grid, dim3(256), 0, s, x, out, M, C);
    return hipGetLastError();
}

hipError_t launch_mask_rows(const float* x, const uint8_t* mask, float* y, int64_t M, int C, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, s, x, mask, y, M, C);
    return hipGetLastError();
}

hipError_t launch_dwconv_train(const float* x, const float* w, const float* bias, const int32_t* frame_offsets, int B, int max_frames, float* y,
                               int C, int flip, hipStream_t s) {
    if (B <= 0 || max_frames <= 0) return hipSuccess;
    dim3 grid((unsigned)((C / 2 + 255) / 256), (unsigned)((max_frames + kDwRows - 1) / kDwRows), (unsigned)B);
    hipLaunchKernelGGL(dwconv_train_kernel, grid, dim3(256), 0, s, x, w, bias, frame_offsets, B, y, C, flip);
    return hipGetLastError();
}

hipError_t launch_dwconv_bwd_w(const float* dy, const float* x, const int32_t* clip_of_row, const int32_t* frame_offsets, int M, int C, float* dw,
                               int accumulate, float* scratch, hipStream_t s, int weight_layout) {
    if (M <= 0) return hipSuccess;
    const int P = (M + kDwChunk - 1) / kDwChunk;
    hipLaunchKernelGGL(dwconv_bwd_w_partial_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)P), dim3(256), 0, s, dy, x, clip_of_row, frame_offsets, M, C,
                       scratch);
    hipLaunchKernelGGL(dwconv_bwd_w_final_kernel, dim3((unsigned)((kTaps * C + 63) / 64)), dim3(256), 0, s, scratch, P, kTaps * C, dw, accumulate,
                       weight_layout ? C : 0);
    return hipGetLastError();
}

hipError_t launch_bce(const float* x, const float* t, int64_t n, float* dx, float* loss, double* scratch, hipStream_t s) {
    const int blocks = (int)(n + 255) / 256 < 1024 ? (int)((n + 255) / 256) : 1024;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, t, n, 1.0f / (float)n, dx, scratch);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, scratch, blocks, 1.0 / (double)n, loss);
    return hipGetLastError();
}

hipError_t launch_cross_entropy(const float* x, const int64_t* target, int M, int N, int64_t ignore, float* dx, float* loss, double* scratch, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int blocks = (M + 3) / 4 < 1024 ? (M + 3) / 4 : 1024;
    float* count = reinterpret_cast<float*>(scratch + 1024);              // scratch: 1024 partial doubles, then the row count
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, s, target, M, ignore, count);
    hipLaunchKernelGGL(ce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, target, M, N, ignore, count, dx, scratch);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, s, scratch, blocks, count, loss);
    return hipGetLastError();
}

hipError_t launch_emd(const float* pred, const float* gt, int B, int T, float* dpred, float* loss, double* scratch, hipStream_t s) {
    if (B <= 0 || T <= 0) return hipSuccess;
    const float inv_scale = 1.0f / sqrtf((float)T), inv_n = 1.0f / ((float)B * (float)T);
    hipLaunchKernelGGL(emd_kernel, dim3((unsigned)B), dim3(256), 0, s, pred, gt, T, inv_scale, inv_n, dpred, scratch);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, scratch, B, (double)inv_n, loss);
    return hipGetLastError();
}

hipError_t launch_sumsq(const float* x, int64_t n, double* out, double* scratch, hipStream_t s) {
    const int blocks = n <= 0 ? 1 : (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, scratch);
    hipLaunchKernelGGL(sum_partials_f64_kernel, dim3(1), dim3(64), 0, s, scratch, blocks, out);
    return hipGetLastError();
}

hipError_t launch_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, const double* sumsq, double clip, double denom, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_clip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1,
                       sqrtf(bc2), sumsq, clip, denom);
    return hipGetLastError();
}

hipError_t launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, float grad_scale, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1,
                       sqrtf(bc2), grad_scale);
    return hipGetLastError();
}
