// Note / boundary decoder: probs [M, N] + bounds [M]  ->  per-clip note sequence.
//
// Replaces utils/infer_utils.py:9-76 (decode_gaussian_blurred_probs, decode_bounds_to_alignment,
// decode_note_sequence) as driven by inference/me_infer.py:78-97 and inference/me_quant_infer.py:22-38.
// Integer results must be bit-exact with the reference's CPU path, so every floating-point reduction whose
// result feeds an integer decision is done in the reference's order and precision:
//   * bounds.cumsum(): torch-CPU accumulates fp32 cumsum in fp64 and rounds each prefix to fp32 -> a
//     sequential fp64 scan by one lane (T = 2584 -> ~10 us, one workgroup per clip, clips in parallel);
//   * .round(): round-half-even (rintf);
//   * the weighted mean over the <= 7-bin window and the per-note value sum: sequential fp32 in ascending
//     index / frame order, separate multiply and add (__fmul_rn / __fadd_rn: no FMA contraction);
//   * per-note counts and the 128-bin histogram are integers (LDS atomics, order-free), argmax = first max.
// HBM-bound: reads 4 (N + 1) bytes per frame, writes <= 13 bytes per frame.
#include "internal.h"

namespace {

struct FrameArgs {
    const float* probs; const uint8_t* mask; int64_t M; int nbins; int quantized;
    int width; float interval, vmin, threshold;
    float* values; uint8_t* rest;
};

// ---- stage A: one wave per frame -------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_frames_kernel(FrameArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_in_grid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t m = wave_in_grid; m < a.M; m += n_waves) {
        const float* __restrict__ p = a.probs + m * a.nbins;
        const bool on = a.mask == nullptr || a.mask[m] != 0;          // probs *= masks[..., None]
        // first-max argmax over nbins (<= 192): lane holds bins lane, lane + 64, lane + 128
        float best = -INFINITY;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = lane + 64 * i;
            if (c < a.nbins) {
                const float v = on ? p[c] : 0.f;
                if (v > best) { best = v; bidx = c; }                  // ascending c: keeps the first max
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bidx, o, 64);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (lane == 0) {
            float value;
            bool rest;
            if (a.quantized) {                                         // me_quant_infer.py:31-35
                rest = bidx == 128;
                value = (float)min(max(bidx, 0), 127);
            } else {                                                   // infer_utils.py:9-24
                const int s = max(bidx - a.width, 0), e = min(bidx + a.width + 1, a.nbins);
                float ps = 0.f, ws = 0.f;
                for (int j = s; j < e; ++j) {
                    const float pj = on ? p[j] : 0.f;
                    const float vj = __fadd_rn(__fmul_rn((float)j, a.interval), a.vmin);
                    ps = __fadd_rn(ps, __fmul_rn(pj, vj));
                    ws = __fadd_rn(ws, pj);
                }
                value = __fdiv_rn(ps, __fadd_rn(ws, ws == 0.f ? 1.f : 0.f));
                rest = best < a.threshold;
            }
            a.values[m] = value;
            a.rest[m] = rest ? 1 : 0;
        }
    }
}

// ---- stage B + C: one workgroup per clip ------------------------------------------------------------
constexpr int SCAN_TILE = 4096;

__global__ __launch_bounds__(256) void decode_notes_kernel(DecodeArgs a, const float* __restrict__ values,
                                                            const uint8_t* __restrict__ rest,
                                                            int32_t* __restrict__ f2i_s, int32_t* __restrict__ start_s) {
    __shared__ float sb[SCAN_TILE];
    __shared__ uint8_t sm[SCAN_TILE];
    __shared__ int hist[4][128];
    __shared__ int s_nnotes, s_nmax;
    __shared__ double s_acc;
    __shared__ long long s_prev_step;
    __shared__ int s_cur;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = a.frame_offsets[b];
    const int T = a.frame_offsets[b + 1] - f0;
    const float* __restrict__ bounds = a.bounds + f0;
    const uint8_t* __restrict__ mask = a.mask ? a.mask + f0 : nullptr;
    int32_t* __restrict__ f2i = f2i_s + f0;          // masked frame2item (0 for masked frames)
    int32_t* __restrict__ nstart = start_s + f0;      // nstart[n-1] = first frame of note n (unmasked scan)

    if (tid == 0) { s_acc = 0.0; s_prev_step = -1; s_cur = 0; s_nmax = 0; }
    __syncthreads();
    // infer_utils.py:27-39 on bounds * masks
    for (int t0 = 0; t0 < T; t0 += SCAN_TILE) {
        const int nt = min(SCAN_TILE, T - t0);
        for (int i = tid; i < nt; i += 256) {
            const bool on = mask == nullptr || mask[t0 + i] != 0;
            sb[i] = on ? bounds[t0 + i] : 0.f;
            sm[i] = on ? 1 : 0;
        }
        __syncthreads();
        if (tid == 0) {
            double acc = s_acc;
            long long prev = s_prev_step;
            int cur = s_cur, nmax = s_nmax;
            for (int i = 0; i < nt; ++i) {
                acc += (double)sb[i];
                const long long step = (long long)rintf((float)acc);
                if (step - prev > 0) { nstart[cur] = t0 + i; ++cur; }
                prev = step;
                const int item = sm[i] ? cur : 0;                          // ... * masks (me_infer.py:84)
                f2i[t0 + i] = item;
                nmax = max(nmax, item);
            }
            s_acc = acc; s_prev_step = prev; s_cur = cur; s_nmax = nmax;
        }
        __syncthreads();
    }
    if (tid == 0) { s_nnotes = s_cur; a.n_notes[b] = s_nmax; }
    __syncthreads();
    const int n_scan = s_nnotes;     // notes found by the scan
    const int n_out = s_nmax;        // space - 1 = frame2item.max() (infer_utils.py:52)

    if (a.frame2item != nullptr)
        for (int t = tid; t < T; t += 256) a.frame2item[f0 + t] = f2i[t];

    // infer_utils.py:42-76: one wave per note
    for (int n = 1 + wave; n <= n_out; n += 4) {
        const int ts = nstart[n - 1];
        const int te = (n < n_scan) ? nstart[n] : T;
        hist[wave][lane] = 0;
        hist[wave][lane + 64] = 0;
        int dur = 0, unm = 0;
        for (int t = ts + lane; t < te; t += 64) {
            if (f2i[t] == n) {
                ++dur;
                const bool on = rest[f0 + t] == 0;                      // masks_eff = ~rest & masks (frame is unmasked here)
                if (on) {
                    ++unm;
                    const int vq = (int)rintf(values[f0 + t]);
                    atomicAdd(&hist[wave][vq & 127], 1);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { dur += __shfl_xor(dur, o, 64); unm += __shfl_xor(unm, o, 64); }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // first-max argmax of the 128-bin histogram
        int hv = hist[wave][lane], hi_ = lane;
        { const int h2 = hist[wave][lane + 64]; if (h2 > hv) { hv = h2; hi_ = lane + 64; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(hv, o, 64), oi = __shfl_xor(hi_, o, 64);
            if (ov > hv || (ov == hv && oi < hi_)) { hv = ov; hi_ = oi; }
        }
        if (lane == 0) {
            const float center = (float)hi_;
            const float lo = __fsub_rn(center, 0.5f), hi2 = __fadd_rn(center, 0.5f);
            int valid = 0;
            float item_value;
            if (a.quantized) {
                long long acc = 0;
                for (int t = ts; t < te; ++t)
                    if (f2i[t] == n && rest[f0 + t] == 0) {
                        const float v = values[f0 + t];
                        if (v >= lo && v <= hi2) { ++valid; acc += (long long)v; }
                    }
                item_value = __fdiv_rn((float)acc, (float)(valid + (valid == 0 ? 1 : 0)));
            } else {
                float acc = 0.f;
                for (int t = ts; t < te; ++t)
                    if (f2i[t] == n && rest[f0 + t] == 0) {
                        const float v = values[f0 + t];
                        if (v >= lo && v <= hi2) { ++valid; acc = __fadd_rn(acc, v); }
                    }
                item_value = __fdiv_rn(acc, (float)(valid + (valid == 0 ? 1 : 0)));
            }
            a.note_midi[f0 + n - 1] = item_value;
            a.note_dur[f0 + n - 1] = dur;
            // item_masks = unmasked / dur >= 0.5 in fp32 (int64 / int64 true-divide -> fp32); 0/0 = nan -> False
            const bool keep = dur > 0 && __fdiv_rn((float)unm, (float)dur) >= 0.5f;
            a.note_rest[f0 + n - 1] = keep ? 0 : 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

size_t decode_scratch_bytes(int64_t total_frames) {
    const size_t m = (size_t)((total_frames + 63) / 64 * 64);
    return m * (4 + 4 + 4 + 1) + 1024;
}

hipError_t launch_decode(const DecodeArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.total_frames <= 0) return hipSuccess;
    if (a.nbins > 192) return hipErrorInvalidValue;
    const size_t m = (size_t)((a.total_frames + 63) / 64 * 64);
    char* sc = static_cast<char*>(a.scratch);
    int32_t* f2i_s = reinterpret_cast<int32_t*>(sc);
    int32_t* start_s = reinterpret_cast<int32_t*>(sc + 4 * m);
    float* values = a.values ? a.values : reinterpret_cast<float*>(sc + 8 * m);
    uint8_t* rest = a.rest ? a.rest : reinterpret_cast<uint8_t*>(sc + 12 * m);

    FrameArgs f;
    f.probs = a.probs; f.mask = a.mask; f.M = a.total_frames; f.nbins = a.nbins; f.quantized = a.quantized;
    const double interval = (a.vmax - a.vmin) / (double)(a.nbins - 1);                   // infer_utils.py:11-12
    f.width = a.quantized ? 0 : (int)(3.0 * a.deviation / interval);
    f.interval = (float)interval; f.vmin = (float)a.vmin; f.threshold = (float)a.threshold;
    f.values = values; f.rest = rest;
    int64_t blocks = (a.total_frames + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(decode_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, s, f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(decode_notes_kernel, dim3((unsigned)a.B), dim3(256), 0, s, a, values, rest, f2i_s, start_s);
    return hipGetLastError();
}
