// Depthwise Conv1d(k = 31, pad 15, groups = 512) + BatchNorm1d(eval) + SiLU on [M, 512] time-major rows.
//
// Replaces depthwise_conv -> norm -> act2 of the reference's conv module (modules/conv/base_conv.py:66-68)
// and the two transposes around it (:64,:70): the data stays [time, channel], so the 512 channels are the
// coalesced axis and the 31 taps walk down rows.  BatchNorm is folded into the taps at load time
// (some_pack_weights): w' = w * gamma / sqrt(var + eps), b' = (b - mean) * gamma / sqrt(var + eps) + beta.
// Zero padding is per clip (frame_offsets), exactly like the reference's per-chunk B = 1 call.
//
// HBM-bound: 4 KiB per frame (read + write); each thread owns one channel and slides a 38-deep register
// window down 128 frames, 8 outputs per step (31 FMAs each), so every input is loaded from memory once per
// block (+ a 30-frame halo shared with the neighbouring block through L2).
#include "internal.h"
#include "split.h"

namespace {

constexpr int TC = 128;   // output frames per block (large batches); small launches use TC_SMALL for more blocks
constexpr int TC_SMALL = 32;
constexpr int G = 8;      // outputs per window step
constexpr int HALO = (kConvK - 1) / 2;

template <int TCV>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a) {
    const int b = blockIdx.z;
    const int g = blockIdx.y >> 1;
    const int c = (blockIdx.y & 1) * 256 + threadIdx.x;
    const int f0 = a.frame_offsets[b];
    const int T = a.frame_offsets[b + 1] - f0;
    const int t_begin = blockIdx.x * TCV;
    if (t_begin >= T) return;
    const int t_end = min(t_begin + TCV, T);
    const float* __restrict__ x = a.x[g] + (size_t)f0 * kDim + c;
    float* __restrict__ y = a.y[g] + (size_t)f0 * kDim + c;

    float w[kConvK];
#pragma unroll
    for (int j = 0; j < kConvK; ++j) w[j] = a.w[g][j * kDim + c];
    const float bias = a.b[g][c];

    float win[kConvK - 1 + G];
    // window slot i holds x[t0 - HALO + i]
#pragma unroll
    for (int i = 0; i < kConvK - 1; ++i) {
        const int t = t_begin - HALO + i;
        win[i] = (t >= 0 && t < T) ? x[(size_t)t * kDim] : 0.f;
    }
    for (int t0 = t_begin; t0 < t_end; t0 += G) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int t = t0 + HALO + i;
            win[kConvK - 1 + i] = (t < T) ? x[(size_t)t * kDim] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < G; ++o) {
            float acc = bias;
#pragma unroll
            for (int j = 0; j < kConvK; ++j) acc = fmaf(w[j], win[o + j], acc);
            const float v = acc / (1.0f + __expf(-acc));
            if (a.out_split) {   // SPLIT32: lane pairs pack two hi (even lane) / two lo (odd lane) halves per dword
                half_t h, l;
                split_f16(v, h, l);
                const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
                const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);   // lane ^ 1
                const int odd = threadIdx.x & 1;
                const uint32_t word = odd ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
                if (t0 + o < t_end) {
                    char* row = reinterpret_cast<char*>(a.y[g]) + (size_t)(f0 + t0 + o) * kDim * 4 + (c >> 5) * 128;
                    *reinterpret_cast<uint32_t*>(row + (odd ? 64 + ((c & 31) - 1) * 2 : (c & 31) * 2)) = word;
                }
            } else if (t0 + o < t_end) {
                y[(size_t)(t0 + o) * kDim] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < kConvK - 1; ++i) win[i] = win[i + G];
    }
}

}  // namespace

hipError_t launch_dwconv(const DwArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.max_frames <= 0) return hipSuccess;
    const long blocks_big = (long)((a.max_frames + TC - 1) / TC) * 2 * a.groups * a.B;
    if (blocks_big >= 1024) {
        dim3 grid((unsigned)((a.max_frames + TC - 1) / TC), (unsigned)(2 * a.groups), (unsigned)a.B);
        hipLaunchKernelGGL(dwconv_kernel<TC>, grid, dim3(256), 0, s, a);
    } else {    // few clips: shorter time chunks keep all CUs busy (the 30-frame halo is re-read from L2)
        dim3 grid((unsigned)((a.max_frames + TC_SMALL - 1) / TC_SMALL), (unsigned)(2 * a.groups), (unsigned)a.B);
        hipLaunchKernelGGL(dwconv_kernel<TC_SMALL>, grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
