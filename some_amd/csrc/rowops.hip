// Row-wise HBM-bound kernels: LayerNorm(512) and the head softmax.
//
// LayerNorm replaces nn.LayerNorm(dim) x5 per conformer block (modules/conform/Gconform.py:49-53, 56-63;
// eps 1e-5, biased variance).  One 64-lane wave owns one 512-float row: two coalesced 16-byte loads per
// lane, two wavefront butterfly reductions (mean, then centred second moment - the two-pass form keeps
// fp32 cancellation out), one coalesced store.  Algorithmic traffic: 4 KiB per row (read + write).
//
// Softmax replaces F.softmax(midi, dim=2) (modules/model/Gmidi_conform.py:36-37) on the [M, outdim] head.
#include "internal.h"
#include "split.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

#ifndef LN_ROWS
#define LN_ROWS 2               // rows per wavefront and iteration (independent load -> reduce -> store chains in flight)
#endif
#ifndef LN_STORE16
#define LN_STORE16 1            // 1 (round 6: LayerNorm 0.1326 -> 0.1251 ms, profiles/r06k_step_ab_ln16.txt): SPLIT32 output as 16-byte stores (lane pairs exchange halves by DPP: the even lane stores the pair's 8 hi
                                // halves, the odd lane its 8 lo halves) instead of four 8-byte stores per lane - same values, same arithmetic
#endif
#ifndef LN_NT
#define LN_NT 0                 // 1: non-temporal stores (the output is re-read by a GEMM after 0.7 GB of other traffic)
#endif
template <typename T>
__device__ __forceinline__ void ln_store(T* p, T v) {
#if LN_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave_in_grid = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * 256) >> 6;
    const int g = blockIdx.y;
    const float* __restrict__ x = a.x[g];
    float* __restrict__ y = a.y[g];
    char* __restrict__ ys = reinterpret_cast<char*>(a.ys[g]);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma[g] + lane * 4);
    const f32x4 g1 = *reinterpret_cast<const f32x4*>(a.gamma[g] + 256 + lane * 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.beta[g] + lane * 4);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.beta[g] + 256 + lane * 4);
    for (int mb = wave_in_grid * LN_ROWS; mb < a.M; mb += n_waves * LN_ROWS) {
        f32x4 v0[LN_ROWS], v1[LN_ROWS];
#pragma unroll
        for (int u = 0; u < LN_ROWS; ++u) {
            const float* row = x + (size_t)min(mb + u, a.M - 1) * kDim;
            v0[u] = *reinterpret_cast<const f32x4*>(row + lane * 4);
            v1[u] = *reinterpret_cast<const f32x4*>(row + 256 + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < LN_ROWS; ++u) {
            const int m = mb + u;
            float s = (v0[u][0] + v0[u][1]) + (v0[u][2] + v0[u][3]) + (v1[u][0] + v1[u][1]) + (v1[u][2] + v1[u][3]);
            const float mean = wave_sum(s) * (1.0f / kDim);
            v0[u] -= mean;
            v1[u] -= mean;
            float q = (v0[u][0] * v0[u][0] + v0[u][1] * v0[u][1]) + (v0[u][2] * v0[u][2] + v0[u][3] * v0[u][3]) +
                      (v1[u][0] * v1[u][0] + v1[u][1] * v1[u][1]) + (v1[u][2] * v1[u][2] + v1[u][3] * v1[u][3]);
            const float var = wave_sum(q) * (1.0f / kDim);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            const f32x4 o0 = v0[u] * rstd * g0 + b0;
            const f32x4 o1 = v1[u] * rstd * g1 + b1;
            if (LN_ROWS > 1 && m >= a.M) break;      // wave-uniform
            if (y != nullptr) {
                float* out = y + (size_t)m * kDim;
                ln_store(reinterpret_cast<f32x4*>(out + lane * 4), o0);
                ln_store(reinterpret_cast<f32x4*>(out + 256 + lane * 4), o1);
            }
            if (ys != nullptr) {     // SPLIT32: k-block = 8 lanes; 4 hi halves (8 B) + 4 lo halves (8 B) per lane
                half4 h0, l0, h1, l1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    half_t h, l;
                    split_f16(o0[i], h, l); h0[i] = h; l0[i] = l;
                    split_f16(o1[i], h, l); h1[i] = h; l1[i] = l;
                }
#if LN_STORE16
                // lanes 2 j, 2 j + 1 hold elements 8 j .. 8 j + 7 of a 32-element k-block (4 each): after the exchange the even lane owns
                // the 16 bytes of hi halves, the odd lane the 16 bytes of lo halves of those 8 elements
                typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
                typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
                const bool odd = lane & 1;
                auto xchg = [&](half4 mine_keep, half4 mine_give) -> u32x4_ {
                    const u32x2_ keep = __builtin_bit_cast(u32x2_, mine_keep), give = __builtin_bit_cast(u32x2_, mine_give);
                    // quad permute [1, 0, 3, 2]: every lane reads its pair partner's `give`
                    const uint32_t o0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)give[0], 0xB1, 0xF, 0xF, true);
                    const uint32_t o1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)give[1], 0xB1, 0xF, 0xF, true);
                    // even lane: [own hi (elements 0-3) | partner's hi (4-7)]; odd lane: [partner's lo (0-3) | own lo (4-7)]
                    return odd ? u32x4_{o0, o1, keep[0], keep[1]} : u32x4_{keep[0], keep[1], o0, o1};
                };
                // even lane keeps hi and gives lo; odd lane keeps lo and gives hi
                const u32x4_ w0 = xchg(odd ? l0 : h0, odd ? h0 : l0);
                const u32x4_ w1 = xchg(odd ? l1 : h1, odd ? h1 : l1);
                char* row = ys + (size_t)m * kDim * 4 + (lane >> 3) * 128 + ((lane & 7) >> 1) * 16 + (odd ? 64 : 0);
                ln_store(reinterpret_cast<u32x4_*>(row), w0);
                ln_store(reinterpret_cast<u32x4_*>(row + 1024), w1);
#else
                char* row = ys + (size_t)m * kDim * 4 + (lane >> 3) * 128 + (lane & 7) * 8;
                ln_store(reinterpret_cast<half4*>(row), h0);
                ln_store(reinterpret_cast<half4*>(row + 64), l0);
                ln_store(reinterpret_cast<half4*>(row + 1024), h1);
                ln_store(reinterpret_cast<half4*>(row + 1024 + 64), l1);
#endif
            }
        }
    }
}

// one wave per row, n <= 256 (129 in the reference's quantised config)
__global__ __launch_bounds__(256) void row_softmax_kernel(float* __restrict__ x, int64_t rows, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_in_grid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t m = wave_in_grid; m < rows; m += n_waves) {
        float* row = x + m * n;
        float v[4];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < n ? row[c] : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = expf(v[i] - mx);
            s += v[i];
        }
        s = wave_sum(s);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < n) row[c] = v[i] / s;
        }
    }
}

// fp32 rows -> SPLIT32 (split.h): one thread per 4 consecutive elements.  BF16: hi = bf16(x), lo = 0 (split.h)
template <bool BF16>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, char* __restrict__ out, int64_t n4, int k4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / k4;
        const int c = (int)(i % k4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        half4 hh, ll;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            half_t h, l;
            if (BF16) { h = bf16_as_half(v[j]); l = (half_t)0.f; }
            else split_f16(v[j], h, l);
            hh[j] = h; ll[j] = l;
        }
        char* p = out + row * (int64_t)k4 * 16 + (c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<half4*>(p) = hh;
        *reinterpret_cast<half4*>(p + 64) = ll;
    }
}

// ---- clip-aligned attention coordinates (internal.h: launch_attn_plan) -------------------------------------------------
// pad_offsets[b] = sum_{c < b} roundup(T_c, 16): one workgroup, each thread sums a run of clips, block scan over the run totals.
__global__ __launch_bounds__(256) void attn_plan_scan_kernel(const int32_t* __restrict__ fo, int B, int32_t* __restrict__ pad) {
    __shared__ int32_t part[256];
    const int tid = threadIdx.x;
    const int per = (B + 255) / 256;
    const int b0 = min(B, tid * per), b1 = min(B, b0 + per);
    int32_t sum = 0;
    for (int b = b0; b < b1; ++b) sum += (fo[b + 1] - fo[b] + kClipAlign - 1) / kClipAlign * kClipAlign;
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {           // Hillis-Steele inclusive scan
        const int32_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int32_t run = part[tid] - sum;                 // exclusive prefix of this thread's run
    for (int b = b0; b < b1; ++b) {
        pad[b] = run;
        run += (fo[b + 1] - fo[b] + kClipAlign - 1) / kClipAlign * kClipAlign;
    }
    if (tid == 255) pad[B] = part[255];
}
// row_map[p] = the packed frame that padded row p holds, -1 for the alignment rows behind a clip and the tail
__global__ __launch_bounds__(256) void attn_plan_map_kernel(const int32_t* __restrict__ fo, const int32_t* __restrict__ pad, int B, int rows,
                                                            int32_t* __restrict__ row_map) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= rows) return;
    int lo = 0, hi = B;                            // last b in [0, B] with pad[b] <= p (pad[0] = 0)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pad[mid] <= p) lo = mid; else hi = mid - 1;
    }
    int32_t src = -1;
    if (lo < B) {
        const int t = p - pad[lo];
        if (t < fo[lo + 1] - fo[lo]) src = fo[lo] + t;
    }
    row_map[p] = src;
}

}  // namespace

hipError_t launch_attn_plan(const int32_t* frame_offsets, int B, int rows_cover, int32_t* pad_offsets, int32_t* row_map, hipStream_t s) {
    if (B <= 0 || rows_cover <= 0) return hipSuccess;
    hipLaunchKernelGGL(attn_plan_scan_kernel, dim3(1), dim3(256), 0, s, frame_offsets, B, pad_offsets);
    hipLaunchKernelGGL(attn_plan_map_kernel, dim3((unsigned)((rows_cover + 255) / 256)), dim3(256), 0, s, frame_offsets, pad_offsets, B, rows_cover, row_map);
    return hipGetLastError();
}

hipError_t launch_split_rows(const float* x, float* out, int64_t rows, int K, hipStream_t s, int bf16) {
    if (rows <= 0) return hipSuccess;
    if (K & 31) return hipErrorInvalidValue;
    const int64_t n4 = rows * (K / 4);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (bf16) hipLaunchKernelGGL(split_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, x, reinterpret_cast<char*>(out), n4, K / 4);
    else hipLaunchKernelGGL(split_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, x, reinterpret_cast<char*>(out), n4, K / 4);
    return hipGetLastError();
}

hipError_t launch_layernorm(const LnArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    int blocks = (a.M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(layernorm_kernel, dim3(blocks, a.groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_row_softmax(float* x, int64_t rows, int n, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (n > 256) return hipErrorInvalidValue;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, rows, n);
    return hipGetLastError();
}
