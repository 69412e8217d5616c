// Host-ingest kernels either side of the silence slicer: the WAV payload is uploaded once
// as it sits in the file (int16 PCM, 2 bytes per sample), and
//   slicer_rms_kernel   computes get_rms(y, frame_length, hop_length) (utils/slicer2.py:5-38) bit-for-bit, so the host
//                       only runs the silence state machine on the tiny RMS vector (slicer2.py:84-134);
//   pcm_gather_kernel   cuts the chunks the state machine chose (slicer2.py:73-82) out of the uploaded clips and
//                       converts them to the packed fp32 layout some_logmel reads: librosa.load's int16 -> float32
//                       scaling (infer.py:34, batch_infer.py:51) fused with the cut.
// Both are HBM-bound byte movers: RMS reads each sample 4 times (frames overlap 4x; the re-reads hit L2) and writes
// 4 bytes per 882 samples; the gather reads 2 and writes 4 bytes per kept sample.
#include "internal.h"
#include "rms_core.h"

namespace {

template <class T> struct Sample;
template <> struct Sample<float> { static __device__ __forceinline__ float get(const float* p, int64_t i) { return p[i]; } };
// int16 / 32768 is exact in float32 (as is numpy's float32(x) / float32(32768))
template <> struct Sample<int16_t> { static __device__ __forceinline__ float get(const int16_t* p, int64_t i) { return (float)p[i] * (1.0f / 32768.0f); } };

// One thread per RMS frame.  Frame j of a clip covers samples [j hop - fl/2, j hop + fl - fl/2) of the clip, zeros
// outside (np.pad(..., mode='constant')).
template <class T>
__global__ __launch_bounds__(128) void slicer_rms_kernel(const T* __restrict__ audio, const int64_t* __restrict__ sample_offsets,
                                                          const int64_t* __restrict__ rms_offsets, int frame_length, int hop,
                                                          float* __restrict__ rms) {
    const int b = blockIdx.y;
    const int64_t r0 = rms_offsets[b];
    const int64_t nfr = rms_offsets[b + 1] - r0;
    const int64_t j = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (j >= nfr) return;
    const int64_t s0 = sample_offsets[b];
    const int64_t n = sample_offsets[b + 1] - s0;
    const T* __restrict__ x = audio + s0;
    const int64_t first = j * hop - frame_length / 2;
    float sum;
    if (first >= 0 && first + frame_length <= n) {            // interior frame: no bounds checks
        const T* __restrict__ xf = x + first;
        sum = rms_pairwise_sumsq([xf](int i) { return Sample<T>::get(xf, i); }, frame_length);
    } else {
        sum = rms_pairwise_sumsq([x, first, n](int i) {
            const int64_t q = first + i;
            return (q >= 0 && q < n) ? Sample<T>::get(x, q) : 0.f;
        }, frame_length);
    }
    // sqrtf and / are correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is NOT:
    // it maps to the native 1-ulp v_sqrt_f32
    rms[r0 + j] = sqrtf(sum / (float)frame_length);
}

template <class T>
__global__ __launch_bounds__(256) void pcm_gather_kernel(const T* __restrict__ src, const int64_t* __restrict__ src_offsets,
                                                          const int64_t* __restrict__ dst_offsets, float* __restrict__ dst) {
    const int b = blockIdx.y;
    const int64_t d0 = dst_offsets[b];
    const int64_t n = dst_offsets[b + 1] - d0;
    const T* __restrict__ s = src + src_offsets[b];
    float* __restrict__ d = dst + d0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < n) d[i + k] = Sample<T>::get(s, i + k);
    }
}

}  // namespace

hipError_t launch_slicer_rms(const void* audio, int is_pcm16, const int64_t* sample_offsets, const int64_t* rms_offsets,
                             int B, int64_t max_rms_frames, int frame_length, int hop, float* rms, hipStream_t s) {
    if (B <= 0 || max_rms_frames <= 0) return hipSuccess;
    dim3 grid((unsigned)((max_rms_frames + 127) / 128), (unsigned)B);
    if (is_pcm16)
        hipLaunchKernelGGL(slicer_rms_kernel<int16_t>, grid, dim3(128), 0, s, static_cast<const int16_t*>(audio), sample_offsets,
                           rms_offsets, frame_length, hop, rms);
    else
        hipLaunchKernelGGL(slicer_rms_kernel<float>, grid, dim3(128), 0, s, static_cast<const float*>(audio), sample_offsets,
                           rms_offsets, frame_length, hop, rms);
    return hipGetLastError();
}

hipError_t launch_pcm_gather(const void* src, int is_pcm16, const int64_t* src_offsets, const int64_t* dst_offsets, int B,
                             int64_t max_len, float* dst, hipStream_t s) {
    if (B <= 0 || max_len <= 0) return hipSuccess;
    const int64_t blocks = (max_len + 1023) / 1024;
    dim3 grid((unsigned)(blocks < 4096 ? blocks : 4096), (unsigned)B);
    if (is_pcm16)
        hipLaunchKernelGGL(pcm_gather_kernel<int16_t>, grid, dim3(256), 0, s, static_cast<const int16_t*>(src), src_offsets,
                           dst_offsets, dst);
    else
        hipLaunchKernelGGL(pcm_gather_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(src), src_offsets,
                           dst_offsets, dst);
    return hipGetLastError();
}
