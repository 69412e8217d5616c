// Note / boundary decoder: probs [M, N] + bounds [M]  ->  per-clip note sequence.
//
// Replaces utils/infer_utils.py:9-76 (decode_gaussian_blurred_probs, decode_bounds_to_alignment,
// decode_note_sequence) as driven by inference/me_infer.py:78-97 and inference/me_quant_infer.py:22-38.
// Integer results must be bit-exact with the reference's CPU path, so every floating-point reduction whose
// result feeds an integer decision is done in the reference's order and precision:
//   * bounds.cumsum(): torch-CPU accumulates fp32 cumsum in fp64 and rounds each prefix to fp32 -> a
//     sequential fp64 scan by one lane out of LDS (one workgroup per clip, clips in parallel);
//   * .round(): round-half-even (rintf);
//   * the weighted mean over the <= 7-bin window and the per-note value sum: sequential fp32 in ascending
//     index / frame order, separate multiply and add (plain operators under `fp contract(off)`: the
//     __fmul_rn / __fadd_rn wrappers are ordinary * and + compiled with the contract flag in this toolchain's
//     headers and DO fuse into FMAs after inlining);
//   * per-note counts and the 128-bin histogram are integers (LDS atomics, order-free), argmax = first max.
// HBM-bound: reads 4 (N + 1) bytes per frame, writes <= 13 bytes per frame.
#include "internal.h"

#pragma clang fp contract(off)

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
                    const float vj = (float)j * a.interval + a.vmin;      // two roundings (contract off)
                    ps = ps + pj * vj;
                    ws = ws + pj;
                }
                value = ps / (ws + (ws == 0.f ? 1.f : 0.f));
                rest = best < a.threshold;
            }
            a.values[m] = value;
            a.rest[m] = rest ? 1 : 0;
        }
    }
}

// ---- stage B + C: one workgroup per clip ------------------------------------------------------------
// decode_bounds_to_alignment (infer_utils.py:27-39).  Only the fp64 prefix sum has to be sequential to be
// bit-identical with torch-CPU's cumsum; it runs as ONE dependent v_add_f64 per frame on lane 0 over LDS-resident
// doubles.  Everything downstream - fp32 cast, round-half-even, diff > 0, the integer cumsum of the increments
// (an exact, order-free workgroup scan) and the note-start compaction - is done by all 256 threads.
// Clips of up to CAP frames (47 s at hop 512 / 44.1 kHz) keep every per-frame array in LDS for the note phase;
// longer clips stream through the same tiles and spill frame2item / note starts to the global scratch.
constexpr int CAP = 4096;
constexpr int PER = CAP / 256;     // frames per thread in the parallel phases

struct ClipLds {
    double acc[CAP];       // masked bounds as fp64, overwritten in place by their running sum
    float val[CAP];
    int f2i[CAP];
    int start[CAP];
    uint8_t flg[CAP];      // bit 0: frame unmasked, bit 1: rest
    int hist[4][128];
    int wsum[4];
    double carry_acc;
    float carry_step;
    int carry_cnt, nmax, n_long;
};

__global__ __launch_bounds__(256) void decode_notes_kernel(DecodeArgs a, const float* __restrict__ values,
                                                            const uint8_t* __restrict__ rest,
                                                            int32_t* __restrict__ f2i_s, int32_t* __restrict__ start_s,
                                                            const int64_t* __restrict__ f2i_in) {
    __shared__ ClipLds L;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = a.frame_offsets[b];
    const int T = a.frame_offsets[b + 1] - f0;
    const bool in_lds = T <= CAP;
    int* f2i_g = f2i_s + f0;          // used when the clip does not fit in LDS
    int* start_g = start_s + f0;
    if (tid == 0) { L.carry_acc = 0.0; L.carry_step = -1.0f; L.carry_cnt = 0; L.nmax = 0; }
    __syncthreads();
    int* nend = reinterpret_cast<int*>(L.acc);        // only used with an explicit frame2item (L.acc is free then)

    if (f2i_in != nullptr) {
        // decode_note_sequence on a caller-supplied frame2item (any order; T <= CAP checked by the launcher):
        // note n spans [first frame with f2i == n, last such frame]; `rest` carries ~masks here
        for (int i = tid; i < T; i += 256) {
            L.f2i[i] = (int)f2i_in[f0 + i];
            L.flg[i] = 1 | (rest[f0 + i] ? 2 : 0);
            L.val[i] = values[f0 + i];
            L.start[i] = 0x7fffffff;
            nend[i] = 0;
        }
        __syncthreads();
        int nmax = 0;
        for (int i = tid; i < T; i += 256) {
            const int n = L.f2i[i];
            nmax = max(nmax, n);
            if (n >= 1 && n <= CAP) { atomicMin(&L.start[n - 1], i); atomicMax(&nend[n - 1], i + 1); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
        if (lane == 0) atomicMax(&L.nmax, nmax);
        __syncthreads();
        if (tid == 0) L.carry_cnt = min(L.nmax, CAP);
        __syncthreads();
    }

    for (int t0 = 0; f2i_in == nullptr && t0 < T; t0 += CAP) {
        const int nt = min(CAP, T - t0);
        // (1) parallel: bounds * masks -> fp64; flags; values
        for (int i = tid; i < nt; i += 256) {
            const int t = f0 + t0 + i;
            const bool on = a.mask == nullptr || a.mask[t] != 0;
            L.acc[i] = on ? (double)a.bounds[t] : 0.0;                  // bounds *= masks (me_infer.py:83)
            L.flg[i] = (on ? 1 : 0) | (rest[t] ? 2 : 0);
            L.val[i] = values[t];
        }
        __syncthreads();
        // (2) sequential fp64 prefix sum (the only order-dependent step)
        if (tid == 0) {
            double acc = L.carry_acc;
            for (int i = 0; i < nt; i += 8) {
                double x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = (i + j < nt) ? L.acc[i + j] : 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc += x[j]; x[j] = acc; }
#pragma unroll
                for (int j = 0; j < 8; ++j) if (i + j < nt) L.acc[i + j] = x[j];
            }
            L.carry_acc = acc;
        }
        __syncthreads();
        // (3) parallel: step = round_half_even((float)acc); inc = step - step_prev > 0; frame2item = cumsum(inc)
        const int base = tid * PER;
        int inc_bits = 0, local = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = base + j;
            if (i < nt) {
                const float step = rintf((float)L.acc[i]);
                const float prev = i > 0 ? rintf((float)L.acc[i - 1]) : L.carry_step;
                if (step - prev > 0.f) { inc_bits |= 1 << j; ++local; }
            }
        }
        int incl = local;                                               // inclusive scan across the workgroup
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) L.wsum[wave] = incl;
        __syncthreads();
        int offset = L.carry_cnt + incl - local;
        for (int w = 0; w < wave; ++w) offset += L.wsum[w];
        int nmax = 0;
        int cur = offset;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = base + j;
            if (i < nt) {
                if (inc_bits & (1 << j)) {
                    if (in_lds) L.start[cur] = i; else start_g[cur] = t0 + i;
                    ++cur;
                }
                const int item = (L.flg[i] & 1) ? cur : 0;              // ... * masks (me_infer.py:84)
                if (in_lds) L.f2i[i] = item; else f2i_g[t0 + i] = item;
                if (a.frame2item != nullptr) a.frame2item[f0 + t0 + i] = item;
                nmax = max(nmax, item);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
        if (lane == 0) atomicMax(&L.nmax, nmax);
        __syncthreads();
        if (tid == 0) {
            L.carry_cnt += L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
            L.carry_step = rintf((float)L.acc[nt - 1]);
        }
        __syncthreads();
    }
    const int n_scan = L.carry_cnt;  // notes found by the scan
    const int n_out = f2i_in != nullptr ? min(L.nmax, CAP) : L.nmax;   // space - 1 = frame2item.max() (infer_utils.py:52)
    const bool explicit_f2i = f2i_in != nullptr;
    if (tid == 0) a.n_notes[b] = n_out;

    // infer_utils.py:42-76.  Notes are mostly a handful of frames: one THREAD per short note (mode of the rounded
    // values by pairwise counting, no histogram), one WAVE per long note (LDS histogram + wavefront argmax).
    const int* f2i = in_lds ? L.f2i : f2i_g;
    const int* nstart = in_lds ? L.start : start_g;
    auto frame_ok = [&](int t, int n) {      // frame belongs to note n and counts (masks_eff = ~rest & masks)
        if (f2i[t] != n) return false;
        return in_lds ? (L.flg[t] & 2) == 0 : rest[f0 + t] == 0;
    };
    auto frame_val = [&](int t) { return in_lds ? L.val[t] : values[f0 + t]; };
    auto finish = [&](int n, int ts, int te, int center_bin, int dur, int unm) {
        // mean of the values within +-0.5 of the mode, summed in ascending frame order like CPU scatter_add
        const float center = (float)center_bin;
        const float lo = center - 0.5f, hi2 = center + 0.5f;
        int valid = 0;
        long long iacc = 0;
        float facc = 0.f;
        for (int t = ts; t < te; ++t) {
            if (!frame_ok(t, n)) continue;
            const float v = frame_val(t);
            if (v >= lo && v <= hi2) {
                ++valid;
                if (a.quantized) iacc += (long long)v; else facc = facc + v;
            }
        }
        const float denom = (float)(valid + (valid == 0 ? 1 : 0));
        a.note_midi[f0 + n - 1] = (a.quantized ? (float)iacc : facc) / denom;
        a.note_dur[f0 + n - 1] = dur;
        // item_masks = unmasked / dur >= 0.5 in fp32 (int64 / int64 true-divide -> fp32); 0/0 = nan -> False
        const bool keep = dur > 0 && (float)unm / (float)dur >= 0.5f;
        a.note_rest[f0 + n - 1] = keep ? 0 : 1;
    };
    constexpr int SHORT = 24;
    if (tid == 0) L.n_long = 0;
    __syncthreads();
    for (int n = 1 + tid; n <= n_out; n += 256) {
        const int ts = explicit_f2i ? min(nstart[n - 1], T) : nstart[n - 1];
        const int te = explicit_f2i ? nend[n - 1] : ((n < n_scan) ? nstart[n] : T);
        if (!explicit_f2i && te - ts > SHORT) {                       // defer to the wave-cooperative path
            const int slot = atomicAdd(&L.n_long, 1);
            if (slot < CAP) {
                L.start[CAP - 1 - slot] = n;                          // long-note list grows down from the end of L.start
                continue;                                             // (cannot collide: a long note spans > SHORT frames)
            }                                                         // > CAP long notes in one clip: stay on this thread
        }
        int dur = 0, unm = 0, best_cnt = 0, best_bin = 0;
        for (int t = ts; t < te; ++t) {
            if (f2i[t] != n) continue;
            ++dur;
            if (!frame_ok(t, n)) continue;
            ++unm;
            const int vq = (int)rintf(frame_val(t)) & 127;
            int cnt = 0;
            for (int u = ts; u < te; ++u)
                if (frame_ok(u, n) && (((int)rintf(frame_val(u))) & 127) == vq) ++cnt;
            if (cnt > best_cnt || (cnt == best_cnt && vq < best_bin)) { best_cnt = cnt; best_bin = vq; }   // first max
        }
        finish(n, ts, te, best_cnt > 0 ? best_bin : 0, dur, unm);
    }
    __syncthreads();
    const int n_long = min(L.n_long, CAP);
    for (int k = wave; k < n_long; k += 4) {
        const int n = L.start[CAP - 1 - k];
        const int ts = explicit_f2i ? min(nstart[n - 1], T) : nstart[n - 1];
        const int te = explicit_f2i ? nend[n - 1] : ((n < n_scan) ? nstart[n] : T);
        L.hist[wave][lane] = 0;
        L.hist[wave][lane + 64] = 0;
        int dur = 0, unm = 0;
        for (int t = ts + lane; t < te; t += 64) {
            if (f2i[t] == n) {
                ++dur;
                if (frame_ok(t, n)) {
                    ++unm;
                    atomicAdd(&L.hist[wave][((int)rintf(frame_val(t))) & 127], 1);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { dur += __shfl_xor(dur, o, 64); unm += __shfl_xor(unm, o, 64); }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        int hv = L.hist[wave][lane], hi_ = lane;                      // first-max argmax of the 128-bin histogram
        { const int h2 = L.hist[wave][lane + 64]; if (h2 > hv) { hv = h2; hi_ = lane + 64; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(hv, o, 64), oi = __shfl_xor(hi_, o, 64);
            if (ov > hv || (ov == hv && oi < hi_)) { hv = ov; hi_ = oi; }
        }
        if (lane == 0) finish(n, ts, te, hi_, dur, unm);
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// decode_note_sequence on caller-supplied per-frame arrays (utils/infer_utils.py:42-76); clips of <= CAP frames
hipError_t launch_decode_notes(const DecodeArgs& a, const int64_t* frame2item, const float* values,
                               const uint8_t* not_masks, int max_frames, hipStream_t s) {
    if (a.B <= 0 || a.total_frames <= 0) return hipSuccess;
    if (max_frames > CAP) return hipErrorInvalidValue;
    const size_t m = (size_t)((a.total_frames + 63) / 64 * 64);
    char* sc = static_cast<char*>(a.scratch);
    hipLaunchKernelGGL(decode_notes_kernel, dim3((unsigned)a.B), dim3(256), 0, s, a, values, not_masks,
                       reinterpret_cast<int32_t*>(sc), reinterpret_cast<int32_t*>(sc + 4 * m), frame2item);
    return hipGetLastError();
}

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
    hipLaunchKernelGGL(decode_notes_kernel, dim3((unsigned)a.B), dim3(256), 0, s, a, values, rest, f2i_s, start_s,
                       static_cast<const int64_t*>(nullptr));
    return hipGetLastError();
}
