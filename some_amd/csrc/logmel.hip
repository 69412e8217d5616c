// Fused log-mel front end: zero-padded framing -> Hann window -> 2048-point real FFT -> magnitude ->
// 80-band mel filterbank -> clamp(1e-5) -> log, written directly in the [frames, 80] layout the network
// reads.  Replaces MelSpectrogram.forward (modules/rmvpe/spec.py:38-72: F.pad, torch.stft, abs, matmul, clamp,
// log) and the transpose at inference/me_infer.py:31.  The 1025 x T complex spectrum and the magnitude never leave
// the CU.
//
// HBM-bound by design (4 L bytes of audio in, 320 bytes per frame out).  One workgroup
// (4 wavefronts) owns kFramesPerWg = 12 CONSECUTIVE frames of one clip: the 2048 + 11 * 512 samples they cover are
// loaded into LDS ONCE with coalesced loads (each sample is used by 4 frames; HBM sees 1.25 x the algorithmic read,
// the overlap between neighbouring workgroups), then every wavefront transforms its 3 frames on its own - the
// 1024-point complex FFT keeps its data in registers, 16 x 16 x 4 with two LDS transposes and a DPP quad butterfly
// (fft_core.h), so there is no workgroup barrier after the load and two workgroups per CU overlap one's load with
// the other's arithmetic.  The mel basis (librosa.filters.mel, htk=True, Slaney norm; built on the host) has only
// 727 non-zeros in bins 2..371: it is applied as 80 contiguous triangles out of a zero-padded LDS copy (no
// predication in the inner loop) and only the needed bins get a magnitude.
#include "fft_core.h"
#include "internal.h"

namespace {

constexpr int kFramesPerWg = 12;
constexpr int kFramesPerWave = kFramesPerWg / 4;
constexpr int kTile = kWin + (kFramesPerWg - 1) * kHop;          // 7680 samples
constexpr int kMelPad = 32;                                      // padded band length of the LDS copy of the filterbank
constexpr int kMelRow = kMelPad + 1;                             // LDS row stride: lane l reads row l - a stride of 32 floats would put all 64 lanes on one bank
constexpr int kMelRows = kMels + 1;                              // + one all-zero row for lanes without a second band
constexpr size_t kLdsBytes = (size_t)kTile * 4 + 4 * (size_t)FFT_TBUF * 8 + (size_t)kMelRows * kMelRow * 4;   // 76 228 B: two per CU

__device__ __forceinline__ float quad_xor2(float v) {     // value of lane ^ 2: quad_perm [2, 3, 0, 1]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor1(float v) {     // value of lane ^ 1: quad_perm [1, 0, 3, 2]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// Two triangular bands per lane (bands l and l + 64), each a sequential fmaf chain in ascending bin order.  The loop
// runs to the wave-uniform longest band in steps of 8 with predicated loads (weight 0 past a band's end: fmaf(0, m, acc)
// = acc), so the 16 LDS loads of a step are in flight together instead of one dependent round trip per bin.
template <typename WP>
__device__ __forceinline__ void band_sums(WP w, const float* mag, int st0, int len0, int off0, int st1, int len1, int off1,
                                          int max_len, float& a0, float& a1) {
    a0 = 0.f;
    a1 = 0.f;
    for (int i0 = 0; i0 < max_len; i0 += 8) {
        float w0[8], m0[8], w1[8], m1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u;
            const bool ok0 = i < len0, ok1 = i < len1;
            const float x0 = w[off0 + (ok0 ? i : 0)], x1 = w[off1 + (ok1 ? i : 0)];
            w0[u] = ok0 ? x0 : 0.f;
            w1[u] = ok1 ? x1 : 0.f;
            m0[u] = mag[st0 + (ok0 ? i : 0)];
            m1[u] = mag[st1 + (ok1 ? i : 0)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = fmaf(w0[u], m0[u], a0);
            a1 = fmaf(w1[u], m1[u], a1);
        }
    }
}

// The usual case (every band <= kMelPad bins): weights come from a zero-padded [81][32] LDS table, so the chains need no
// predication - a term past a band's end is 0 * (some finite magnitude or stale spectrum word) = 0.
__device__ __forceinline__ void band_sums_padded(const float* w0, const float* w1, const float* m0, const float* m1, int steps,
                                                 float& a0, float& a1) {
    a0 = 0.f;
    a1 = 0.f;
    for (int s8 = 0; s8 < steps; ++s8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = fmaf(w0[8 * s8 + u], m0[8 * s8 + u], a0);
            a1 = fmaf(w1[8 * s8 + u], m1[8 * s8 + u], a1);
        }
    }
}

__global__ __launch_bounds__(256) void logmel_kernel(LogmelTables t, const float* __restrict__ audio,
                                                      const int64_t* __restrict__ sample_offsets,
                                                      const int32_t* __restrict__ frame_offsets,
                                                      float* __restrict__ units, int kmax, int pad_reflect) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* samp = lds;
    const int b = blockIdx.y, fbase = blockIdx.x * kFramesPerWg, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    cpx* tb = reinterpret_cast<cpx*>(lds + kTile) + wave * FFT_TBUF;
    float* melpad = lds + kTile + 4 * FFT_TBUF * 2;
    const int f0 = frame_offsets[b];
    const int T = frame_offsets[b + 1] - f0;
    if (fbase >= T) return;
    const int64_t s0 = sample_offsets[b];
    const int64_t n = sample_offsets[b + 1] - s0;
    const float* __restrict__ x = audio + s0;
    const cpx* __restrict__ tw = reinterpret_cast<const cpx*>(t.twiddle);

    // spec.py:47-50: pad win/2 zeros each side; frame f covers padded samples [512 f, 512 f + 2048)
    const int64_t p0 = (int64_t)fbase * kHop - kWin / 2;
    const int need = kWin + (min(kFramesPerWg, T - fbase) - 1) * kHop;
    for (int i = tid; i < need; i += 256) {
        const int64_t p = p0 + i;
        float v;
        if (pad_reflect) {       // torch.stft(center=True): reflect without repeating the edge sample (n > win/2)
            const int64_t q = p < 0 ? -p : (p >= n ? 2 * (n - 1) - p : p);
            v = x[q];
        } else {
            v = (p >= 0 && p < n) ? x[p] : 0.f;
        }
        samp[i] = v;
    }
    const bool padded = t.max_len <= kMelPad;
    if (padded)
        for (int i = tid; i < kMelRows * kMelPad; i += 256) melpad[(i / kMelPad) * kMelRow + i % kMelPad] = t.mel_wpad[i];

    // per-lane constants, kept in registers for all frames of this wavefront
    float w0[16], w1[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const float2 w = *reinterpret_cast<const float2*>(t.window + 128 * n1 + 2 * lane);
        w0[n1] = w.x;
        w1[n1] = w.y;
    }
    cpx twa[16], twb[16];
    fft_lane_twiddles(lane, tw, twa, twb);
    const int band0 = lane, band1 = lane + 64;
    const int st0 = t.mel_start[band0], len0 = t.mel_len[band0], off0 = t.mel_off[band0];
    int st1 = 0, len1 = 0, off1 = 0;
    if (band1 < kMels) { st1 = t.mel_start[band1]; len1 = t.mel_len[band1]; off1 = t.mel_off[band1]; }
    __syncthreads();

    const int k1 = lane >> 2, qj = lane & 3, k3 = fft_quad_k3(qj);
    const float* w0p = melpad + band0 * kMelRow;
    const float* w1p = melpad + (band1 < kMels ? band1 : kMels) * kMelRow;
    const int mel_steps = (t.max_len + 7) >> 3;
    for (int fi = 0; fi < kFramesPerWave; ++fi) {
        const int frame = fbase + wave * kFramesPerWave + fi;
        if (frame >= T) break;                                   // wave-uniform
        const float* xs = samp + (wave * kFramesPerWave + fi) * kHop;
        cpx v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const float2 s = *reinterpret_cast<const float2*>(xs + 128 * n1 + 2 * lane);
            v[n1] = {s.x * w0[n1], s.y * w1[n1]};
        }
        // LDS operations of one wavefront complete in issue order: a wave barrier (scheduling fence) is all the
        // transposes need
        fft_stage_a(lane, v, twa, tb);
        __builtin_amdgcn_wave_barrier();
        fft_stage_b(lane, tb, twb, v);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const cpx c1 = fft_stage_c1(qj, v[k2], {quad_xor2(v[k2].re), quad_xor2(v[k2].im)});
            tb[fft_zaddr(k1 + 16 * k2 + 256 * k3)] = fft_stage_c2(qj, c1, {quad_xor1(c1.re), quad_xor1(c1.im)});
        }
        __builtin_amdgcn_wave_barrier();
        // magnitudes of the bins the filterbank touches (k <= kmax), first into registers, then over the spectrum
        float mg[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = lane + 64 * j;
            mg[j] = (64 * j <= kmax && k <= kmax) ? rfft_mag(k, tb, tw) : 0.f;
        }
        float mg16 = 0.f;
        if (kmax >= 1024 && lane == 0) mg16 = rfft_mag(1024, tb, tw);
        __builtin_amdgcn_wave_barrier();
        float* mag = reinterpret_cast<float*>(tb);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (64 * j <= kmax) mag[lane + 64 * j] = mg[j];
        if (kmax >= 1024 && lane == 0) mag[1024] = mg16;
        __builtin_amdgcn_wave_barrier();
        float* out = units + (size_t)(f0 + frame) * kMels;
        float a0, a1;
        if (padded) band_sums_padded(w0p, w1p, mag + st0, mag + st1, mel_steps, a0, a1);
        else band_sums(t.mel_w, mag, st0, len0, off0, st1, len1, off1, t.max_len, a0, a1);
        out[band0] = logf(fmaxf(a0, 1e-5f));
        if (band1 < kMels) out[band1] = logf(fmaxf(a1, 1e-5f));
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

hipError_t launch_logmel(const LogmelTables& t, const float* audio, const int64_t* sample_offsets,
                         const int32_t* frame_offsets, int B, int max_frames, int pad_reflect, float* units, hipStream_t s) {
    if (B <= 0 || max_frames <= 0) return hipSuccess;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&logmel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        if (e != hipSuccess) return e;
        attr_once.mark();
    }
    dim3 grid((unsigned)((max_frames + kFramesPerWg - 1) / kFramesPerWg), (unsigned)B);
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), kLdsBytes, s, t, audio, sample_offsets, frame_offsets, units, t.kmax, pad_reflect);
    return hipGetLastError();
}
