// Log-mel front end for the key-shift / speed augmentation of the training data:
// MelSpectrogram.forward(audio, keyshift != 0 | speed != 1 | center=False) (modules/rmvpe/spec.py:38-72, called by
// preprocessing/me_binarizer.py:235-246 and me_quant_binarizer.py:39-48 with key shifts in [-12, 12] semitones).
// A key shift of s semitones re-frames the audio with n_fft' = round(2048 * 2^(s/12)) - anything from 1024 to 4096 and
// almost never a power of two - keeps the first 1025 bins (zero-filled when n_fft' / 2 + 1 < 1025), rescales them by
// win / win' and applies the unchanged 80-band filterbank.
//
// The transform length is arbitrary, so this is a direct DFT, restricted to the bins the filterbank touches
// (k <= kmax, 371 for fmax = 8 kHz): one workgroup per frame, the windowed frame (fp32) and the n_fft' twiddles (fp64)
// in LDS, every thread accumulating two bins in fp64 (CDNA4 issues fp64 FMA at the fp32 rate) with the twiddle index
// (k n) mod n_fft' carried as an exact integer - no trigonometric recurrence, no fp32 accumulation error, so the
// result is closer to the exact transform than the reference's fp32 FFT is.  It is an offline-preprocessing path
// (8 shifted copies of every training clip): measured 0.45 ms (n_fft' = 1024) to 2.3 ms (4096) per 30 s clip,
// LDS-read bound; the inference front end is logmel.hip.
#include "internal.h"

namespace {

struct dbl2 { double x, y; };

__global__ __launch_bounds__(256) void logmel_shift_kernel(LogmelTables t, const float* __restrict__ window,
                                                            const dbl2* __restrict__ twiddle,
                                                            const float* __restrict__ audio,
                                                            const int64_t* __restrict__ sample_offsets,
                                                            const int32_t* __restrict__ frame_offsets,
                                                            float* __restrict__ units, int N, int hop, int pad_left,
                                                            int nbins, float scale_num, float scale_den, int rescale) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    dbl2* tw = reinterpret_cast<dbl2*>(lds_raw);                       // [N]
    float* xs = reinterpret_cast<float*>(lds_raw + (size_t)N * 16);    // [N]
    float* mag = xs + N;                                               // [kmax + 1]
    const int b = blockIdx.y, frame = blockIdx.x, tid = threadIdx.x;
    const int f0 = frame_offsets[b];
    if (frame >= frame_offsets[b + 1] - f0) return;
    const int64_t s0 = sample_offsets[b];
    const int64_t n = sample_offsets[b + 1] - s0;
    const float* __restrict__ x = audio + s0;
    const int64_t p0 = (int64_t)frame * hop - pad_left;               // spec.py:47-50: zeros either side
    for (int i = tid; i < N; i += 256) {
        const int64_t p = p0 + i;
        xs[i] = (p >= 0 && p < n) ? x[p] * window[i] : 0.f;
        tw[i] = twiddle[i];
    }
    for (int i = tid; i <= t.kmax; i += 256) mag[i] = 0.f;             // bins past n_fft' / 2: F.pad zeros (spec.py:66-67)
    __syncthreads();

    for (int k0 = tid; k0 < nbins; k0 += 512) {
        const int k1 = k0 + 256;                                       // may be >= nbins: computed, not stored
        const int st1 = k1 % N;
        double re0 = 0., im0 = 0., re1 = 0., im1 = 0.;
        int i0 = 0, i1 = 0;
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
            const double v = (double)xs[j];
            const dbl2 w0 = tw[i0], w1 = tw[i1];
            re0 = fma(v, w0.x, re0); im0 = fma(v, w0.y, im0);
            re1 = fma(v, w1.x, re1); im1 = fma(v, w1.y, im1);
            i0 += k0; i0 -= i0 >= N ? N : 0;
            i1 += st1; i1 -= i1 >= N ? N : 0;
        }
        float m0 = (float)sqrt(re0 * re0 + im0 * im0), m1 = (float)sqrt(re1 * re1 + im1 * im1);
        if (rescale) { m0 = m0 * scale_num / scale_den; m1 = m1 * scale_num / scale_den; }   // spec.py:68, fp32 like torch
        if (k0 <= t.kmax) mag[k0] = m0;
        if (k1 < nbins && k1 <= t.kmax) mag[k1] = m1;
    }
    __syncthreads();
    if (tid < kMels) {                                                 // spec.py:70-71 over the band's non-zeros
        const int st = t.mel_start[tid], len = t.mel_len[tid], off = t.mel_off[tid];
        float a = 0.f;
        for (int i = 0; i < len; ++i) a = fmaf(t.mel_w[off + i], mag[st + i], a);
        units[(size_t)(f0 + frame) * kMels + tid] = logf(fmaxf(a, 1e-5f));
    }
}

}  // namespace

size_t logmel_shift_lds_bytes(int N, int kmax) { return (size_t)N * 16 + (size_t)N * 4 + (size_t)(kmax + 1) * 4; }

hipError_t launch_logmel_shift(const LogmelTables& t, const float* window, const double* twiddle, const float* audio,
                               const int64_t* sample_offsets, const int32_t* frame_offsets, int B, int max_frames, int N,
                               int hop, int pad_left, int rescale, float scale_num, float scale_den, float* units,
                               hipStream_t s) {
    if (B <= 0 || max_frames <= 0) return hipSuccess;
    if (N < 2 || N > kMaxShiftFft) return hipErrorInvalidValue;
    const size_t lds = logmel_shift_lds_bytes(N, t.kmax);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&logmel_shift_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)logmel_shift_lds_bytes(kMaxShiftFft, 1024));
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    const int nbins = (t.kmax < N / 2 ? t.kmax : N / 2) + 1;
    hipLaunchKernelGGL(logmel_shift_kernel, dim3((unsigned)max_frames, (unsigned)B), dim3(256), lds, s, t, window,
                       reinterpret_cast<const dbl2*>(twiddle), audio, sample_offsets, frame_offsets, units, N, hop, pad_left,
                       nbins, scale_num, scale_den, rescale);
    return hipGetLastError();
}
