// Host-side emulation of the device index logic (no GPU needed): compiled by tests/test_host_emulation.py
// with g++ and driven through ctypes.  It executes the SAME inline functions the HIP kernels use
// (some_amd/csrc/fft_core.h), one "thread" at a time, so permutation / twiddle mistakes are caught on CPU.
#include <cmath>
#include <vector>
#include "../../some_amd/csrc/fft_core.h"

extern "C" void emu_rfft_mag(const float* x2048, const float* window2048, float* mag1025) {
    std::vector<cpx> tw(2048), a(1024), b(1024);
    for (int k = 0; k < 2048; ++k) {
        const double ang = -2.0 * M_PI * k / 2048.0;
        tw[k] = {(float)std::cos(ang), (float)std::sin(ang)};
    }
    for (int n = 0; n < 1024; ++n) a[n] = {x2048[2 * n] * window2048[2 * n], x2048[2 * n + 1] * window2048[2 * n + 1]};
    cpx* in = a.data();
    cpx* out = b.data();
    for (int Ns = 1; Ns < 1024; Ns *= 4) {
        for (int j = 0; j < 256; ++j) fft_pass(j, Ns, in, out, tw.data());
        cpx* t = in; in = out; out = t;
    }
    for (int k = 0; k <= 1024; ++k) mag1025[k] = rfft_mag(k, in, tw.data());
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
