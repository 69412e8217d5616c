// Host-side emulation of the device index logic (no GPU needed): compiled by tests/test_host_emulation.py
// with g++ and driven through ctypes.  It executes the SAME inline functions the HIP kernels use
// (some_amd/csrc/fft_core.h), one "thread" at a time, so permutation / twiddle mistakes are caught on CPU.
#include <cmath>
#include <vector>
#include "../../some_amd/csrc/fft_core.h"

extern "C" void emu_rfft_mag(const float* x2048, const float* window2048, float* mag1025) {
    // one wavefront, lane by lane, stage by stage - exactly the order logmel.hip runs the stages in
    std::vector<cpx> tw(2048), tbuf(FFT_TBUF);
    for (int k = 0; k < 2048; ++k) {
        const double ang = -2.0 * M_PI * k / 2048.0;
        tw[k] = {(float)std::cos(ang), (float)std::sin(ang)};
    }
    static cpx v[64][16], twa[64][16], twb[64][16];
    for (int lane = 0; lane < 64; ++lane) fft_lane_twiddles(lane, tw.data(), twa[lane], twb[lane]);
    for (int lane = 0; lane < 64; ++lane) {
        for (int n1 = 0; n1 < 16; ++n1) {
            const int n = 64 * n1 + lane;
            v[lane][n1] = {x2048[2 * n] * window2048[2 * n], x2048[2 * n + 1] * window2048[2 * n + 1]};
        }
        fft_stage_a(lane, v[lane], twa[lane], tbuf.data());
    }
    for (int lane = 0; lane < 64; ++lane) fft_stage_b(lane, tbuf.data(), twb[lane], v[lane]);
    static cpx c1[64][16];
    for (int lane = 0; lane < 64; ++lane)
        for (int k2 = 0; k2 < 16; ++k2) c1[lane][k2] = fft_stage_c1(lane & 3, v[lane][k2], v[lane ^ 2][k2]);
    for (int lane = 0; lane < 64; ++lane) {
        const int k1 = lane >> 2, k3 = fft_quad_k3(lane & 3);
        for (int k2 = 0; k2 < 16; ++k2)
            tbuf[fft_zaddr(k1 + 16 * k2 + 256 * k3)] = fft_stage_c2(lane & 3, c1[lane][k2], c1[lane ^ 1][k2]);
    }
    for (int k = 0; k <= 1024; ++k) mag1025[k] = rfft_mag(k, tbuf.data(), tw.data());
}

// ---- slicer RMS: the device summation order (some_amd/csrc/rms_core.h), one frame at a time -----------------------
#include <cstdint>
#include "../../some_amd/csrc/rms_core.h"

extern "C" void emu_slicer_rms(const float* y, int64_t n, int frame_length, int hop, float* rms) {
    const int64_t frames = 1 + n / hop;
    for (int64_t j = 0; j < frames; ++j) {
        const int64_t first = j * hop - frame_length / 2;
        const float sum = rms_pairwise_sumsq([&](int i) {
            const int64_t q = first + i;
            return (q >= 0 && q < n) ? y[q] : 0.f;
        }, frame_length);
        rms[j] = std::sqrt(sum / (float)frame_length);
    }
}
