// Index math and butterflies of the 2048-point real FFT used by the log-mel front end.
// Shared between the HIP kernel (logmel.hip) and a host-side emulation (tests/host/fft_emu.cpp) so the
// permutation logic can be verified on a machine without a GPU.
//
// Real FFT of x[0..2047] via one 1024-point complex FFT of z[n] = x[2n] + i x[2n+1]
// (Stockham autosort, radix 4, 5 passes, 256 butterflies per pass) followed by the split
//   X[k] = E[k] + e^{-2 pi i k / 2048} O[k],  E[k] = (Z[k] + conj Z[1024-k]) / 2,
//                                             O[k] = (Z[k] - conj Z[1024-k]) / (2i),   k = 0..1024.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FFT_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define FFT_HD inline
#endif

struct cpx { float re, im; };

FFT_HD cpx cmul(cpx a, cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
FFT_HD cpx cadd(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
FFT_HD cpx csub(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }

constexpr int FFT_N = 1024;       // complex points
constexpr int FFT_R = 4;
constexpr int FFT_THREADS = FFT_N / FFT_R;

// One radix-4 Stockham pass for butterfly j (0..255).  tw2048[k] = exp(-2 pi i k / 2048), k = 0..2047.
// in/out: 1024 complex values (distinct buffers).  Ns = 1, 4, 16, 64, 256.
FFT_HD void fft_pass(int j, int Ns, const cpx* in, cpx* out, const cpx* tw2048) {
    const int jm = j % Ns;
    cpx v0 = in[j], v1 = in[j + 256], v2 = in[j + 512], v3 = in[j + 768];
    if (Ns > 1) {
        // angle = -2 pi jm / (4 Ns); as an index into the 2048-entry table: jm * (2048 / (4 Ns)) * r
        const int step = jm * (512 / Ns);
        v1 = cmul(v1, tw2048[step]);
        v2 = cmul(v2, tw2048[2 * step]);
        v3 = cmul(v3, tw2048[3 * step]);
    }
    const cpx t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3);
    const cpx d = csub(v1, v3);
    const cpx t3 = {d.im, -d.re};                       // -i (v1 - v3)
    const int base = (j / Ns) * Ns * 4 + jm;
    out[base] = cadd(t0, t2);
    out[base + Ns] = cadd(t1, t3);
    out[base + 2 * Ns] = csub(t0, t2);
    out[base + 3 * Ns] = csub(t1, t3);
}

// |X[k]| of the 2048-point real transform from the 1024-point complex spectrum Z (natural order), k = 0..1024.
FFT_HD float rfft_mag(int k, const cpx* Z, const cpx* tw2048) {
    const cpx a = Z[k & 1023];
    const cpx bq = Z[(1024 - k) & 1023];
    const cpx b = {bq.re, -bq.im};                      // conj Z[1024-k]
    const cpx e = {0.5f * (a.re + b.re), 0.5f * (a.im + b.im)};
    const cpx dd = {0.5f * (a.re - b.re), 0.5f * (a.im - b.im)};
    const cpx o = {dd.im, -dd.re};                      // (Z - conj Z') / (2i)
    cpx w;
    if (k < 1024) w = tw2048[k]; else w = {-1.f, 0.f};  // e^{-i pi}
    const cpx x = cadd(e, cmul(w, o));
    return sqrtf(x.re * x.re + x.im * x.im);
}
