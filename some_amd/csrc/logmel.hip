// Fused log-mel front end: zero-padded framing -> Hann window -> 2048-point real FFT -> magnitude ->
// 80-band mel filterbank -> clamp(1e-5) -> log, written directly in the [frames, 80] layout the network
// reads.  Replaces MelSpectrogram.forward (modules/rmvpe/spec.py:38-72: F.pad, torch.stft, abs, matmul, clamp,
// log) and the transpose at inference/me_infer.py:31.  The 1025 x T complex spectrum and the magnitude
// never leave LDS.
//
// One workgroup (256 threads) per frame.  The real transform is one 1024-point complex Stockham FFT (radix 4,
// 5 passes, one butterfly per thread per pass, ping-pong LDS buffers) + the split step (fft_core.h).  The
// mel basis (librosa.filters.mel, htk=True, Slaney norm; built on the host in some_create) has only 727
// non-zeros in bins 2..371, so it is applied as 80 contiguous triangles and only the needed bins get a
// magnitude.  HBM-bound by design: 4 L bytes of audio in (each sample is touched by 4 frames; the re-reads
// hit L2) and 320 bytes per frame out.
#include "fft_core.h"
#include "internal.h"

namespace {

__global__ __launch_bounds__(256) void logmel_kernel(LogmelTables t, const float* __restrict__ audio,
                                                      const int64_t* __restrict__ sample_offsets,
                                                      const int32_t* __restrict__ frame_offsets,
                                                      float* __restrict__ units, int kmax, int pad_reflect) {
    __shared__ cpx bufA[FFT_N];
    __shared__ cpx bufB[FFT_N];
    __shared__ float mag[FFT_N + 8];
    const int b = blockIdx.y, frame = blockIdx.x, tid = threadIdx.x;
    const int f0 = frame_offsets[b];
    const int T = frame_offsets[b + 1] - f0;
    if (frame >= T) return;
    const int64_t s0 = sample_offsets[b];
    const int64_t n = sample_offsets[b + 1] - s0;
    const float* __restrict__ x = audio + s0;
    const cpx* __restrict__ tw = reinterpret_cast<const cpx*>(t.twiddle);

    // spec.py:47-50: pad win/2 zeros each side; frame f covers padded samples [512 f, 512 f + 2048)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int64_t p = (int64_t)frame * kHop + 2 * idx - kWin / 2;
        float x0, x1;
        if (pad_reflect) {       // torch.stft(center=True): reflect without repeating the edge sample (n > win/2)
            const int64_t q0 = p < 0 ? -p : (p >= n ? 2 * (n - 1) - p : p);
            const int64_t q1 = p + 1 < 0 ? -(p + 1) : (p + 1 >= n ? 2 * (n - 1) - (p + 1) : p + 1);
            x0 = x[q0];
            x1 = x[q1];
        } else {
            x0 = (p >= 0 && p < n) ? x[p] : 0.f;
            x1 = (p + 1 >= 0 && p + 1 < n) ? x[p + 1] : 0.f;
        }
        bufA[idx] = {x0 * t.window[2 * idx], x1 * t.window[2 * idx + 1]};
    }
    __syncthreads();
    fft_pass(tid, 1, bufA, bufB, tw);
    __syncthreads();
    fft_pass(tid, 4, bufB, bufA, tw);
    __syncthreads();
    fft_pass(tid, 16, bufA, bufB, tw);
    __syncthreads();
    fft_pass(tid, 64, bufB, bufA, tw);
    __syncthreads();
    fft_pass(tid, 256, bufA, bufB, tw);
    __syncthreads();
    for (int k = tid; k <= kmax; k += 256) mag[k] = rfft_mag(k, bufB, tw);
    __syncthreads();
    if (tid < kMels) {
        const int st = t.mel_start[tid], len = t.mel_len[tid];
        const float* __restrict__ w = t.mel_w + t.mel_off[tid];
        float acc = 0.f;
        for (int i = 0; i < len; ++i) acc = fmaf(w[i], mag[st + i], acc);
        units[(size_t)(f0 + frame) * kMels + tid] = logf(fmaxf(acc, 1e-5f));
    }
}

}  // namespace

hipError_t launch_logmel(const LogmelTables& t, const float* audio, const int64_t* sample_offsets,
                         const int32_t* frame_offsets, int B, int max_frames, int pad_reflect, float* units, hipStream_t s) {
    if (B <= 0 || max_frames <= 0) return hipSuccess;
    dim3 grid((unsigned)max_frames, (unsigned)B);
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), 0, s, t, audio, sample_offsets, frame_offsets, units, t.kmax, pad_reflect);
    return hipGetLastError();
}
