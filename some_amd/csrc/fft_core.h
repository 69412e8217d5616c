// Index math and butterflies of the 2048-point real FFT used by the log-mel front end.
// Shared between the HIP kernel (logmel.hip) and a host-side emulation (tests/host/host_emu.cpp) that runs the same
// inline functions one "lane" at a time, so permutation / twiddle mistakes are caught on a machine without a GPU.
//
// Real FFT of x[0..2047] via one 1024-point complex FFT of z[n] = x[2n] + i x[2n+1] followed by the split
//   X[k] = E[k] + e^{-2 pi i k / 2048} O[k],  E[k] = (Z[k] + conj Z[1024-k]) / 2,
//                                             O[k] = (Z[k] - conj Z[1024-k]) / (2i),   k = 0..1024.
//
// The 1024-point transform is done by ONE 64-lane wavefront with the data in registers (16 complex values per lane)
// as 16 x 16 x 4 (Cooley-Tukey, decimation in time):
//     n = 64 n1 + 4 n2 + n3,   k = k1 + 16 k2 + 256 k3      (n1, n2, k1, k2 < 16;  n3, k3 < 4)
//     Z[k] = sum_n3 W4^(n3 k3) W1024^(n3 (16 k2 + k1)) [ sum_n2 W16^(n2 k2) W256^(n2 k1) ( sum_n1 W16^(n1 k1) z[n] ) ]
//   stage A: lane l = 4 n2 + n3 (= n mod 64) holds z[64 n1 + l]: 16-point DFT over n1 in registers, twiddle W256^(n2 k1),
//            written to LDS as T[k1][l] (rows padded to 68 complex: both sides of the transpose are conflict-free);
//   stage B: lane l' = 4 k1 + n3 reads T[k1][4 n2 + n3] over n2: 16-point DFT over n2, twiddle W1024^(n3 (16 k2 + k1));
//   stage C: the radix-4 over n3 runs across the 4 lanes of a quad as two DPP exchange steps (no LDS): lane 4 k1 + j
//            ends with Z[k1 + 16 k2 + 256 k3], k3 = fft_quad_k3(j), in register k2.
// Two LDS round trips per frame instead of five, and no workgroup barrier.
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
FFT_HD cpx cmuli_neg(cpx a) { return {a.im, -a.re}; }          // -i a

constexpr int FFT_N = 1024;       // complex points
constexpr int FFT_TROW = 68;      // padded row of the stage A -> B transpose buffer (complex elements)
constexpr int FFT_TBUF = 16 * FFT_TROW;                         // 1088 complex = 8704 bytes per wave
FFT_HD int fft_zaddr(int k) { return k + 4 * (k >> 8); }        // padded natural-order index of Z[k] (fits in FFT_TBUF)

// radix-4 DFT of (a0, a1, a2, a3): out[k] = sum_n a_n (-i)^(n k)
FFT_HD void fft_radix4(cpx a0, cpx a1, cpx a2, cpx a3, cpx& o0, cpx& o1, cpx& o2, cpx& o3) {
    const cpx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmuli_neg(csub(a1, a3));
    o0 = cadd(t0, t2);
    o1 = cadd(t1, t3);
    o2 = csub(t0, t2);
    o3 = csub(t1, t3);
}

// in-register 16-point DFT: v[k] <- sum_n v[n] W16^(n k)   (n = 4 na + nb, k = ka + 4 kb)
FFT_HD void fft_dft16(cpx (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    cpx y[4][4];                                                // y[nb][ka]
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) fft_radix4(v[nb], v[4 + nb], v[8 + nb], v[12 + nb], y[nb][0], y[nb][1], y[nb][2], y[nb][3]);
    // twiddles W16^(nb ka)
    y[1][1] = cmul(y[1][1], {C1, -S1});
    y[1][2] = cmul(y[1][2], {R2, -R2});
    y[1][3] = cmul(y[1][3], {S1, -C1});
    y[2][1] = cmul(y[2][1], {R2, -R2});
    y[2][2] = cmuli_neg(y[2][2]);
    y[2][3] = cmul(y[2][3], {-R2, -R2});
    y[3][1] = cmul(y[3][1], {S1, -C1});
    y[3][2] = cmul(y[3][2], {-R2, -R2});
    y[3][3] = cmul(y[3][3], {-C1, S1});
#pragma unroll
    for (int ka = 0; ka < 4; ++ka) fft_radix4(y[0][ka], y[1][ka], y[2][ka], y[3][ka], v[ka], v[ka + 4], v[ka + 8], v[ka + 12]);
}

// twiddles a lane keeps in registers for all its frames; tw2048[k] = exp(-2 pi i k / 2048), k = 0..2047
FFT_HD void fft_lane_twiddles(int lane, const cpx* tw2048, cpx (&twa)[16], cpx (&twb)[16]) {
    const int n2 = lane >> 2;                                   // stage A: W256^(n2 k1)
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) twa[k1] = tw2048[(8 * n2 * k1) & 2047];
    const int k1 = lane >> 2, n3 = lane & 3;                    // stage B: W1024^(n3 (16 k2 + k1))
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) twb[k2] = tw2048[(2 * n3 * (16 * k2 + k1)) & 2047];
}

// stage A for one lane: v[n1] = z[64 n1 + lane] in, T[k1][lane] out
FFT_HD void fft_stage_a(int lane, cpx (&v)[16], const cpx (&twa)[16], cpx* tbuf) {
    fft_dft16(v);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) tbuf[k1 * FFT_TROW + lane] = k1 == 0 ? v[0] : cmul(v[k1], twa[k1]);
}

// stage B for one lane (l' = 4 k1 + n3): reads T[k1][4 n2 + n3], leaves the twiddled 16-point DFT over n2 in v[k2]
FFT_HD void fft_stage_b(int lane, const cpx* tbuf, const cpx (&twb)[16], cpx (&v)[16]) {
    const int k1 = lane >> 2, n3 = lane & 3;
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = tbuf[k1 * FFT_TROW + 4 * n2 + n3];
    fft_dft16(v);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) v[k2] = cmul(v[k2], twb[k2]);
}

// stage C: radix-4 over the four lanes of a quad as two exchange steps (each lane computes ONE butterfly output per
// step from its own value and its partner's):
//   step 1, partner = lane ^ 2:  lanes 0, 1: own + partner;  lanes 2, 3: partner - own;  lane 3 then multiplies by -i
//            -> the quad holds (t0, t2, t1, t3) = (q0 + q2, q1 + q3, q0 - q2, -i (q1 - q3))
//   step 2, partner = lane ^ 1:  even lanes: own + partner;  odd lanes: partner - own
//            -> lane j holds output k3 = fft_quad_k3(j) = (0, 2, 1, 3)[j]
FFT_HD int fft_quad_k3(int j) { return ((j & 1) << 1) | (j >> 1); }
FFT_HD cpx fft_stage_c1(int j, cpx own, cpx partner) {
    const float s = (j & 2) ? -1.f : 1.f;
    const cpx a = {fmaf(s, own.re, partner.re), fmaf(s, own.im, partner.im)};
    return j == 3 ? cmuli_neg(a) : a;
}
FFT_HD cpx fft_stage_c2(int j, cpx own, cpx partner) {
    const float s = (j & 1) ? -1.f : 1.f;
    return {fmaf(s, own.re, partner.re), fmaf(s, own.im, partner.im)};
}

// |X[k]| of the 2048-point real transform from the 1024-point complex spectrum, k = 0..1024; Z is addressed through
// fft_zaddr (padded natural order)
FFT_HD float rfft_mag(int k, const cpx* Z, const cpx* tw2048) {
    const cpx a = Z[fft_zaddr(k & 1023)];
    const cpx bq = Z[fft_zaddr((1024 - k) & 1023)];
    const cpx b = {bq.re, -bq.im};                      // conj Z[1024-k]
    const cpx e = {0.5f * (a.re + b.re), 0.5f * (a.im + b.im)};
    const cpx dd = {0.5f * (a.re - b.re), 0.5f * (a.im - b.im)};
    const cpx o = {dd.im, -dd.re};                      // (Z - conj Z') / (2i)
    cpx w;
    if (k < 1024) w = tw2048[k]; else w = {-1.f, 0.f};  // e^{-i pi}
    const cpx x = cadd(e, cmul(w, o));
    return sqrtf(x.re * x.re + x.im * x.im);
}
