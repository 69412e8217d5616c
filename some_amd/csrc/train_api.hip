// C ABI of the training operators (include/some_amd.h, "training operators"): argument checks + launches.
#include <string>

#include "internal.h"

namespace {

int tfail(SomeHandle* h, int code, const char* msg) {
    if (h) h->err = msg;
    return code;
}
#define T_TRY(h, expr)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            if (h) (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                         \
            return SOME_EHIP;                                                                            \
        }                                                                                                \
    } while (0)
#define T_CHECK(h, cond, msg) \
    do { if (!(cond)) return tfail((h), SOME_EINVAL, msg); } while (0)

inline hipStream_t st(void* s) { return static_cast<hipStream_t>(s); }

// Weight-gradient lanes (some_train_set_wgrad_stream).  wgrad_begin: the stream a weight-gradient GEMM issued on `s` runs on - `s` itself,
// or the side stream paired with it, made to wait for everything enqueued on `s` so far (the operands dY / X were produced there; one
// event per pair is enough: hipStreamWaitEvent captures the record it follows).  With deferred reductions the GEMM about to be launched
// must not write planes a waiting reduction still has to read, and no two waiting reductions may share an output: such a clash (a
// caller that reuses one partial buffer, a weight used twice) flushes what waits first - on the same side stream, so in order.
SomeHandle::WgradLane* wgrad_lane_of(SomeHandle* h, hipStream_t s) {
    for (auto& l : h->wgrad_lanes) if (l.lane == s) return &l;
    return nullptr;
}

hipError_t wgrad_flush(SomeHandle::WgradLane* l) {
    if (!l || l->pending.empty()) return hipSuccess;
    const hipError_t e = launch_reduce_wgrad_table(l->pending.data(), (int)l->pending.size(), l->side);
    l->pending.clear();
    return e;
}

inline bool overlap(const void* a, size_t na, const void* b, size_t nb) {
    const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
    return x < y + nb && y < x + na;
}

hipError_t wgrad_begin(SomeHandle* h, hipStream_t s, const WgradReduce& next, hipStream_t* out, SomeHandle::WgradLane** lane) {
    *out = s;
    SomeHandle::WgradLane* l = *lane = wgrad_lane_of(h, s);
    if (!l) return hipSuccess;
    hipError_t e = hipEventRecord(l->ev, s);
    if (e != hipSuccess) return e;
    e = hipStreamWaitEvent(l->side, l->ev, 0);
    if (e != hipSuccess) return e;
    *out = l->side;
    const size_t plane_bytes = (size_t)next.slices * next.stride * sizeof(float), dw_bytes = (size_t)next.M * next.N * sizeof(float);
    bool clash = (int)l->pending.size() >= kWgradTable;
    for (const WgradReduce& r : l->pending) {
        if (clash) break;
        clash = overlap(r.partial, (size_t)r.slices * r.stride * sizeof(float), next.partial, plane_bytes) ||
                overlap(r.dw, (size_t)r.M * r.N * sizeof(float), next.dw, dw_bytes) ||
                (r.db && next.db && overlap(r.db, (size_t)r.M * sizeof(float), next.db, (size_t)next.M * sizeof(float)));
    }
    return clash ? wgrad_flush(l) : hipSuccess;
}

// behind the GEMM: the reduction of its planes - now, or with the lane's other waiting reductions in one launch later
hipError_t wgrad_reduce(SomeHandle::WgradLane* l, const WgradReduce& r, hipStream_t ws) {
    if (l && l->defer) {
        l->pending.push_back(r);
        return hipSuccess;
    }
    return launch_reduce_wgrad(r.partial, r.slices, r.stride, r.M, r.N, r.ldc, r.dw, r.db, r.accumulate, ws);
}

}  // namespace

extern "C" {

size_t some_train_scratch_bytes(const SomeHandle* h, int64_t M, int32_t N) {
    (void)h;
    if (M <= 0 || N <= 0) return 256;
    const size_t col = (size_t)((M + 63) / 64) * (size_t)kConvK * (size_t)N * sizeof(float);       // tap partials (64-row chunks) >= column partials
    const size_t loss = 1024 * sizeof(double) + (size_t)M * sizeof(double) / 64 + 8192;             // loss partials (<= 1024 blocks or B rows)
    return (col > loss ? col : loss) + 256;
}

static int splitk_slices(int M, int N, int K) {
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256), nk = K >= 32 ? K / 32 : 1;      // (K < 32 is refused by the GEMM entry point; the sizing call must not divide by zero)
    int want = (512 + tiles - 1) / tiles;                       // ~2 workgroups per CU in flight
    if (want > nk) want = nk;
    if (want < 1) want = 1;
    const int per = (nk + want - 1) / want;
    return (nk + per - 1) / per;                                // no empty slice
}

size_t some_train_gemm_splitk_bytes(const SomeHandle* h, int32_t M, int32_t N, int32_t K) {
    (void)h;
    if (M <= 0 || N <= 0 || K <= 0) return 256;
    return (size_t)splitk_slices(M, N, K) * (size_t)M * N * sizeof(float) + 256;
}

int some_train_gemm_splitk(SomeHandle* h, const float* A_split_dev, int32_t lda, const float* W_split_dev, float* C_dev,
                           int32_t M, int32_t N, int32_t K, int32_t hi_only, void* partial_dev, size_t partial_bytes,
                           void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && N > 0 && K > 0 && (K % 32) == 0 && (lda % 32) == 0 && lda >= K && (N % 4) == 0, "some_train_gemm_splitk: bad shape (K, lda % 32, N % 4)");
    T_CHECK(h, A_split_dev && W_split_dev && C_dev && partial_dev, "some_train_gemm_splitk: null pointer");
    T_CHECK(h, partial_bytes >= some_train_gemm_splitk_bytes(h, M, N, K), "some_train_gemm_splitk: partial buffer too small");
    const int slices = splitk_slices(M, N, K);
    GemmArgs a{};
    a.g[0] = GemmGroup{A_split_dev, W_split_dev, nullptr, nullptr, slices > 1 ? static_cast<float*>(partial_dev) : C_dev, nullptr, N, 0};
    a.groups = 1; a.M = M; a.K = K; a.lda = lda; a.ldc = N; a.ldr = N; a.alpha = 1.f;
    a.k_slices = slices; a.slice_stride = (size_t)M * N;
    T_CHECK(h, hi_only >= 0 && hi_only <= 2, "some_train_gemm_splitk: hi_only must be 0, 1 (f16) or 2 (bf16)");
    if (hi_only) T_TRY(h, launch_gemm_f16x1(EPI_NONE, a, 2, st(stream), hi_only == 2));
    else T_TRY(h, launch_gemm_f16x3(EPI_NONE, a, false, 2, st(stream)));
    if (slices > 1) T_TRY(h, launch_reduce_slices(static_cast<const float*>(partial_dev), slices, (size_t)M * N, C_dev, st(stream)));
    return SOME_OK;
}

static int gemm16_slices(int M, int N, int K, int mode) {
    const int bn = gemm16_tile_n(mode);
    const int tiles = ((M + 127) / 128) * ((N + bn - 1) / bn), nk = (K + 31) / 32;
    int want = (512 + tiles - 1) / tiles;                       // two 128 x 256 workgroups per CU in flight
    // ... but at least 16 k-blocks per slice: every slice writes a whole [M, ldc] fp32 plane that the reduction reads back, whatever the
    // number of frames - at the reference's batch shape (~4 100 frames, 130 k-blocks) 16 slices of 8 k-blocks made the planes the cost
    // (measured: 13.7 -> 12.8 ms per step at 8 x 520 frames; unchanged from 8 x 2584 frames up, where slices are >= 40 k-blocks deep)
    if (want > nk / 16) want = nk / 16;
    if (want > nk) want = nk;
    if (want < 1) want = 1;
    const int per = (nk + want - 1) / want;
    return (nk + per - 1) / per;                                // no empty slice
}

size_t some_train_gemm16_bytes(const SomeHandle* h, int32_t M, int32_t N, int32_t K, int32_t ldc) {
    (void)h;
    if (M <= 0 || N <= 0 || K <= 0 || ldc <= 0) return 256;
    const int s1 = gemm16_slices(M, N, K, 1), s3 = gemm16_slices(M, N, K, 3);     // the slice count depends on the mode's tile width
    return (size_t)(s1 > s3 ? s1 : s3) * (size_t)M * ldc * sizeof(float) + 256;
}

int some_train_gemm16(SomeHandle* h, const float* A_dev, int32_t lda, int32_t ta, const float* B_dev, int32_t ldb, int32_t tb,
                      const float* bias_dev, float* C_dev, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t operand,
                      int32_t sum_col, void* partial_dev, size_t partial_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && N > 0 && K > 0 && A_dev && B_dev && C_dev, "some_train_gemm16: bad argument");
    T_CHECK(h, operand >= 1 && operand <= 3, "some_train_gemm16: operand must be 1 (f16), 2 (bf16) or 3 (split f16, fp32-equivalent)");
    T_CHECK(h, (ta == 0 || ta == 1) && (tb == 0 || tb == 1) && !(ta && !tb), "some_train_gemm16: layouts (0,0), (0,1), (1,1) only");
    T_CHECK(h, (lda % 4) == 0 && (ldb % 4) == 0 && lda >= (ta ? M : K) && ldb >= (tb ? N : K), "some_train_gemm16: leading dimensions (% 4, >= row length)");
    T_CHECK(h, (ta && tb) || (K % 32) == 0, "some_train_gemm16: a contraction-contiguous operand needs K % 32 == 0");
    T_CHECK(h, (!ta || (M % 2) == 0) && (!tb || (N % 4) == 0), "some_train_gemm16: M even (ta) / N % 4 == 0 (tb)");
    T_CHECK(h, ldc >= N && (sum_col < 0 || (ta && tb && sum_col >= N && sum_col < ldc)), "some_train_gemm16: ldc / sum_col");
    T_CHECK(h, !bias_dev || !(ta && tb), "some_train_gemm16: no bias epilogue on the weight-gradient layout");
    // 32-bit byte offsets behind buffer descriptors (2 GiB each): refuse what they cannot address instead of reading zeros
    const size_t lim = 0x7fffffffull;
    T_CHECK(h, (size_t)(ta ? K : M) * lda * 4 <= lim && (size_t)(tb ? K : N) * ldb * 4 <= lim && (size_t)M * ldc * 4 <= lim,
            "some_train_gemm16: an operand exceeds 2 GiB (split the batch)");
    const int slices = (ta && tb) ? gemm16_slices(M, N, K, operand) : 1;
    if (slices > 1) T_CHECK(h, partial_dev && partial_bytes >= (size_t)slices * (size_t)M * ldc * sizeof(float), "some_train_gemm16: partial buffer too small");
    float* out = slices > 1 ? static_cast<float*>(partial_dev) : C_dev;
    T_TRY(h, launch_gemm16(A_dev, lda, ta, B_dev, ldb, tb, bias_dev, out, ldc, M, N, K, operand, slices, (size_t)M * ldc, sum_col, st(stream)));
    if (slices > 1) T_TRY(h, launch_reduce_slices(static_cast<const float*>(partial_dev), slices, (size_t)M * ldc, C_dev, st(stream)));
    return SOME_OK;
}

int some_train_gemm16_wgrad(SomeHandle* h, const float* dY_dev, int32_t ldy, const float* X_dev, int32_t ldx, float* dW_dev, float* db_dev,
                            int32_t N, int32_t K, int32_t frames, int32_t operand, int32_t accumulate, void* partial_dev, size_t partial_bytes,
                            void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, N > 0 && K > 0 && frames > 0 && dY_dev && X_dev && dW_dev && partial_dev, "some_train_gemm16_wgrad: bad argument");
    T_CHECK(h, operand >= 1 && operand <= 3, "some_train_gemm16_wgrad: operand must be 1 (f16), 2 (bf16) or 3 (split f16, fp32-equivalent)");
    T_CHECK(h, (ldy % 4) == 0 && (ldx % 4) == 0 && ldy >= N && ldx >= K && (N % 2) == 0 && (K % 4) == 0, "some_train_gemm16_wgrad: shapes (ld % 4, N even, K % 4)");
    T_CHECK(h, (reinterpret_cast<uintptr_t>(dW_dev) & 15) == 0, "some_train_gemm16_wgrad: dW must be 16-byte aligned");
    const int ldc = K + 4;
    const size_t lim = 0x7fffffffull;
    T_CHECK(h, (size_t)frames * ldy * 4 <= lim && (size_t)frames * ldx * 4 <= lim && (size_t)N * ldc * 4 <= lim, "some_train_gemm16_wgrad: an operand exceeds 2 GiB (split the batch)");
    const int slices = gemm16_slices(N, K, frames, operand);
    T_CHECK(h, partial_bytes >= (size_t)slices * (size_t)N * ldc * sizeof(float), "some_train_gemm16_wgrad: partial buffer too small (some_train_gemm16_bytes(N, K, frames, K + 4))");
    float* planes = static_cast<float*>(partial_dev);
    const WgradReduce red{planes, dW_dev, db_dev, (size_t)N * ldc, slices, N, K, ldc, accumulate};
    hipStream_t ws;
    SomeHandle::WgradLane* wl;
    T_TRY(h, wgrad_begin(h, st(stream), red, &ws, &wl));
    T_TRY(h, launch_gemm16(dY_dev, ldy, 1, X_dev, ldx, 1, nullptr, planes, ldc, N, K, frames, operand, slices, (size_t)N * ldc, db_dev ? K : -1, ws));
    T_TRY(h, wgrad_reduce(wl, red, ws));
    return SOME_OK;
}

int some_train_set_wgrad_stream(SomeHandle* h, void* stream, void* wgrad_stream, int32_t defer_reductions) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, stream != wgrad_stream || !wgrad_stream, "some_train_set_wgrad_stream: the weight-gradient stream must differ from the stream it serves");
    for (size_t i = 0; i < h->wgrad_lanes.size(); ++i) {
        SomeHandle::WgradLane& l = h->wgrad_lanes[i];
        if (l.lane != st(stream)) continue;
        T_TRY(h, wgrad_flush(&l));                    // what waits goes out on the stream it was computed on
        if (wgrad_stream) { l.side = st(wgrad_stream); l.defer = defer_reductions != 0; return SOME_OK; }
        (void)hipEventDestroy(l.ev);
        h->wgrad_lanes.erase(h->wgrad_lanes.begin() + (long)i);
        return SOME_OK;
    }
    if (!wgrad_stream) return SOME_OK;
    SomeHandle::WgradLane l{st(stream), st(wgrad_stream), nullptr, defer_reductions != 0, {}};
    T_TRY(h, hipEventCreateWithFlags(&l.ev, hipEventDisableTiming));
    h->wgrad_lanes.push_back(l);
    return SOME_OK;
}

int some_train_wgrad_flush(SomeHandle* h, void* stream) {
    if (!h) return SOME_EINVAL;
    T_TRY(h, wgrad_flush(wgrad_lane_of(h, st(stream))));
    return SOME_OK;
}

int some_train_cast16(SomeHandle* h, const float* x_dev, void* y16_dev, int64_t n, int32_t operand, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, n >= 0 && (n % 8) == 0 && (n == 0 || (x_dev && y16_dev)), "some_train_cast16: n % 8 == 0, non-null arrays");
    T_CHECK(h, operand == 1 || operand == 2, "some_train_cast16: operand must be 1 (f16) or 2 (bf16)");
    T_CHECK(h, ((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(y16_dev)) & 15) == 0, "some_train_cast16: 16-byte aligned arrays");
    T_TRY(h, launch_cast16(x_dev, y16_dev, n, operand == 2, st(stream)));
    return SOME_OK;
}

int some_train_silu16(SomeHandle* h, const float* x_dev, void* y16_dev, int64_t n, int32_t operand, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, n >= 0 && (n % 8) == 0 && (n == 0 || (x_dev && y16_dev)), "some_train_silu16: n % 8 == 0, non-null arrays");
    T_CHECK(h, operand == 1 || operand == 2, "some_train_silu16: operand must be 1 (f16) or 2 (bf16)");
    T_CHECK(h, ((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(y16_dev)) & 15) == 0, "some_train_silu16: 16-byte aligned arrays");
    T_TRY(h, launch_silu16(x_dev, y16_dev, n, operand == 2, st(stream)));
    return SOME_OK;
}

int some_train_transpose16(SomeHandle* h, const float* w_dev, void* w16_dev, void* w16t_dev, int32_t N, int32_t K, int32_t operand, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, N > 0 && K > 0 && w_dev && (w16_dev || w16t_dev), "some_train_transpose16: bad argument");
    T_CHECK(h, operand == 1 || operand == 2, "some_train_transpose16: operand must be 1 (f16) or 2 (bf16)");
    T_TRY(h, launch_transpose16(w_dev, w16_dev, w16t_dev, N, K, operand == 2, st(stream)));
    return SOME_OK;
}

int some_train_transpose16_table(SomeHandle* h, const int64_t* table_dev, int32_t n, int32_t max_n, int32_t max_k, int32_t operand, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, n >= 0 && (n == 0 || (table_dev && max_n > 0 && max_k > 0)), "some_train_transpose16_table: bad argument");
    T_CHECK(h, n <= 65535, "some_train_transpose16_table: at most 65535 weights per call");
    T_CHECK(h, operand == 1 || operand == 2, "some_train_transpose16_table: operand must be 1 (f16) or 2 (bf16)");
    T_TRY(h, launch_transpose16_table(table_dev, n, max_n, max_k, operand == 2, st(stream)));
    return SOME_OK;
}

int some_train_gemm16s(SomeHandle* h, int32_t epilogue, const void* A16_dev, int32_t lda, const void* B16_dev, int32_t ldb, const float* bias_dev,
                       void* C_dev, int32_t ldc, const void* H16_dev, int32_t ldh, int64_t plane_elems, int32_t M, int32_t N, int32_t K,
                       int32_t operand, float p, uint64_t seed, float alpha, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && N > 0 && K > 0 && A16_dev && B16_dev && C_dev, "some_train_gemm16s: bad argument");
    T_CHECK(h, operand == 1 || operand == 2, "some_train_gemm16s: operand must be 1 (f16) or 2 (bf16)");
    T_CHECK(h, epilogue >= 0 && epilogue <= 3, "some_train_gemm16s: epilogue must be 0 (fp32), 1 (FFN first linear), 2 (SiLU / dropout gradient) or 3 (residual + dropout)");
    T_CHECK(h, (K % 32) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && lda >= K && ldb >= K, "some_train_gemm16s: K % 32, ld % 8 (16-byte rows), ld >= K");
    T_CHECK(h, (N % 2) == 0 && ldc >= N && (epilogue == 0 || epilogue == 3 || (ldc % 2) == 0), "some_train_gemm16s: N even, ldc >= N (even for 16-bit outputs)");
    T_CHECK(h, ((reinterpret_cast<uintptr_t>(A16_dev) | reinterpret_cast<uintptr_t>(B16_dev)) & 15) == 0 && (reinterpret_cast<uintptr_t>(C_dev) & 3) == 0,
            "some_train_gemm16s: operands must be 16-byte aligned");
    T_CHECK(h, p >= 0.f && p < 1.f, "some_train_gemm16s: dropout rate in [0, 1)");
    const size_t lim = 0x7fffffffull;
    size_t plane_bytes = 0;
    if (epilogue == 0) {
        T_CHECK(h, (size_t)M * ldc * 4 <= lim, "some_train_gemm16s: the output exceeds 2 GiB (split the batch)");
    } else if (epilogue == 1) {
        T_CHECK(h, bias_dev != nullptr, "some_train_gemm16s: the FFN epilogue needs the bias");
        T_CHECK(h, plane_elems >= (int64_t)M * ldc && (plane_elems % 2) == 0, "some_train_gemm16s: plane_elems (h16 -> a16 distance) must cover M * ldc, even");
        plane_bytes = (size_t)plane_elems * 2;
        T_CHECK(h, plane_bytes + (size_t)M * ldc * 2 <= lim, "some_train_gemm16s: the two output planes exceed 2 GiB (split the batch)");
    } else if (epilogue == 3) {
        T_CHECK(h, H16_dev != nullptr && ldh >= N && (reinterpret_cast<uintptr_t>(H16_dev) & 3) == 0, "some_train_gemm16s: the residual (fp32 [M, ldh], ldh >= N)");
        T_CHECK(h, (size_t)M * ldc * 4 <= lim && (size_t)M * ldh * 4 <= lim, "some_train_gemm16s: an array exceeds 2 GiB (split the batch)");
    } else {
        T_CHECK(h, H16_dev != nullptr && ldh >= N && (ldh % 2) == 0 && (reinterpret_cast<uintptr_t>(H16_dev) & 3) == 0, "some_train_gemm16s: h16 (ldh >= N, even)");
        T_CHECK(h, (size_t)M * ldc * 2 <= lim && (size_t)M * ldh * 2 <= lim, "some_train_gemm16s: an array exceeds 2 GiB (split the batch)");
    }
    T_CHECK(h, (uint64_t)((M + 1) / 2) * (uint64_t)N <= 0xffffffffull, "some_train_gemm16s: more than 2^32 dropout cells");
    T_TRY(h, launch_gemm16s(epilogue, A16_dev, lda, B16_dev, ldb, bias_dev, C_dev, ldc, H16_dev, ldh, plane_bytes, M, N, K, operand == 2, p, seed,
                            alpha, st(stream)));
    return SOME_OK;
}

int some_train_dropcast16(SomeHandle* h, const float* d_dev, void* y16_dev, int32_t M, int32_t N, float alpha, float p, uint64_t seed,
                          int32_t operand, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && N > 0 && (N % 4) == 0 && d_dev && y16_dev, "some_train_dropcast16: bad argument (N % 4 == 0)");
    T_CHECK(h, operand == 1 || operand == 2, "some_train_dropcast16: operand must be 1 (f16) or 2 (bf16)");
    T_CHECK(h, p >= 0.f && p < 1.f, "some_train_dropcast16: dropout rate in [0, 1)");
    T_CHECK(h, (reinterpret_cast<uintptr_t>(d_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(y16_dev) & 7) == 0, "some_train_dropcast16: alignment (16 / 8 bytes)");
    T_CHECK(h, (uint64_t)((M + 1) / 2) * (uint64_t)N <= 0xffffffffull, "some_train_dropcast16: more than 2^32 dropout cells");
    T_TRY(h, launch_dropcast16(d_dev, y16_dev, M, N, alpha, p, seed, operand == 2, st(stream)));
    return SOME_OK;
}

int some_train_gemm16_wgrad16(SomeHandle* h, const void* dY16_dev, int32_t ldy, const void* X16_dev, int32_t ldx, float* dW_dev, float* db_dev,
                              int32_t N, int32_t K, int32_t frames, int32_t operand, int32_t accumulate, void* partial_dev, size_t partial_bytes,
                              void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, N > 0 && K > 0 && frames > 0 && dY16_dev && X16_dev && dW_dev && partial_dev, "some_train_gemm16_wgrad16: bad argument");
    T_CHECK(h, operand == 1 || operand == 2, "some_train_gemm16_wgrad16: operand must be 1 (f16) or 2 (bf16)");
    T_CHECK(h, (ldy % 4) == 0 && (ldx % 4) == 0 && ldy >= N && ldx >= K && (N % 2) == 0 && (K % 4) == 0, "some_train_gemm16_wgrad16: shapes (ld % 4, N even, K % 4)");
    T_CHECK(h, (reinterpret_cast<uintptr_t>(dW_dev) & 15) == 0 && ((reinterpret_cast<uintptr_t>(dY16_dev) | reinterpret_cast<uintptr_t>(X16_dev)) & 7) == 0,
            "some_train_gemm16_wgrad16: dW must be 16-byte, the operands 8-byte aligned");
    const int ldc = K + 4;
    const size_t lim = 0x7fffffffull;
    T_CHECK(h, (size_t)frames * ldy * 2 <= lim && (size_t)frames * ldx * 2 <= lim && (size_t)N * ldc * 4 <= lim, "some_train_gemm16_wgrad16: an operand exceeds 2 GiB (split the batch)");
    const int slices = gemm16_slices(N, K, frames, operand);
    T_CHECK(h, partial_bytes >= (size_t)slices * (size_t)N * ldc * sizeof(float), "some_train_gemm16_wgrad16: partial buffer too small (some_train_gemm16_bytes(N, K, frames, K + 4))");
    float* planes = static_cast<float*>(partial_dev);
    const WgradReduce red{planes, dW_dev, db_dev, (size_t)N * ldc, slices, N, K, ldc, accumulate};
    hipStream_t ws;
    SomeHandle::WgradLane* wl;
    T_TRY(h, wgrad_begin(h, st(stream), red, &ws, &wl));
    T_TRY(h, launch_gemm16(static_cast<const float*>(dY16_dev), ldy, 1, static_cast<const float*>(X16_dev), ldx, 1, nullptr, planes, ldc, N, K, frames,
                           operand, slices, (size_t)N * ldc, db_dev ? K : -1, ws, 1));
    T_TRY(h, wgrad_reduce(wl, red, ws));
    return SOME_OK;
}

// ---- block-level operators (round 5): one C call per conformer sub-block and direction ---------------------------------------------
// The trainer's Python layer drove 3 (forward) + 7 (backward) library calls and 9 tensor allocations per FFN sub-block; at the reference's
// batch shape (8 x ~520 frames, configs/base.yaml:55-56) the step is bound by that host path (enqueue 7.5 - 8.4 ms of 8.9).  These entry
// points run the SAME launches in the same order (bit-identical results) behind one call; the caller owns one `save` block (activations kept
// for the backward) and one scratch block.  Layout of both: consecutive 256-byte aligned segments, sizes from the *_bytes functions.
static size_t up256(size_t n) { return (n + 255) / 256 * 256; }

size_t some_train_ffn_block_save_bytes(const SomeHandle* h, int32_t M, int32_t K, int32_t H) {
    (void)h;
    if (M <= 0 || K <= 0 || H <= 0) return 256;
    return up256((size_t)M * K * 2) + 2 * up256((size_t)M * 4) + up256(2 * (size_t)M * H * 2);       // n16 | mean | rstd | h16, a16 planes
}
size_t some_train_ffn_block_scratch_bytes(const SomeHandle* h, int32_t M, int32_t K, int32_t H, int32_t N) {
    (void)h;
    if (M <= 0 || K <= 0 || H <= 0 || N <= 0) return 256;
    return up256((size_t)M * N * 2) + up256((size_t)M * H * 2) + up256((size_t)M * K * 4);             // dy16 | dh16 | dn
}

int some_train_ffn_block_fwd(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev, const void* w1_16_dev,
                             const float* b1_dev, const void* w2_16_dev, const float* b2_dev, int32_t M, int32_t K, int32_t H, int32_t N,
                             int32_t operand, float alpha, float p_latent, uint64_t seed_latent, float p_out, uint64_t seed_out,
                             void* save_dev, size_t save_bytes, float* out_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && x_dev && save_dev && out_dev, "some_train_ffn_block_fwd: bad argument");
    T_CHECK(h, N == K, "some_train_ffn_block_fwd: the residual needs N == K");
    T_CHECK(h, save_bytes >= some_train_ffn_block_save_bytes(h, M, K, H) && (reinterpret_cast<uintptr_t>(save_dev) & 255) == 0,
            "some_train_ffn_block_fwd: save block too small / not 256-byte aligned (some_train_ffn_block_save_bytes)");
    char* p = static_cast<char*>(save_dev);
    void* n16 = p; p += up256((size_t)M * K * 2);
    float* mean = reinterpret_cast<float*>(p); p += up256((size_t)M * 4);
    float* rstd = reinterpret_cast<float*>(p); p += up256((size_t)M * 4);
    char* ha = p;                                                             // h16 plane, then the a16 plane
    int rc;
    if ((rc = some_train_layernorm_fwd16(h, x_dev, gamma_dev, beta_dev, n16, mean, rstd, M, operand, stream))) return rc;
    if ((rc = some_train_gemm16s(h, 1, n16, K, w1_16_dev, K, b1_dev, ha, H, nullptr, 0, (int64_t)M * H, M, H, K, operand, p_latent, seed_latent, 1.0f, stream)))
        return rc;
    return some_train_gemm16s(h, 3, ha + (size_t)M * H * 2, H, w2_16_dev, H, b2_dev, out_dev, N, x_dev, K, 0, M, N, H, operand, p_out, seed_out, alpha, stream);
}

int some_train_ffn_block_bwd(SomeHandle* h, const float* d_dev, const float* x_dev, const float* gamma_dev, const void* save_dev,
                             const void* w1t_16_dev, const void* w2t_16_dev, int32_t M, int32_t K, int32_t H, int32_t N, int32_t operand,
                             float alpha, float p_latent, uint64_t seed_latent, float p_out, uint64_t seed_out,
                             float* dw1_dev, float* db1_dev, float* dw2_dev, float* db2_dev, float* dgamma_dev, float* dbeta_dev,
                             int32_t add_residual, float* dx_dev, void* scratch_dev, size_t scratch_bytes, void* ln_scratch_dev, size_t ln_scratch_bytes,
                             void* partial_dev, size_t partial_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && d_dev && x_dev && save_dev && dx_dev && dw1_dev && dw2_dev && dgamma_dev && dbeta_dev, "some_train_ffn_block_bwd: bad argument");
    T_CHECK(h, scratch_dev && scratch_bytes >= some_train_ffn_block_scratch_bytes(h, M, K, H, N) && (reinterpret_cast<uintptr_t>(scratch_dev) & 255) == 0,
            "some_train_ffn_block_bwd: scratch block too small / not 256-byte aligned (some_train_ffn_block_scratch_bytes)");
    const char* s = static_cast<const char*>(save_dev);
    const void* n16 = s; s += up256((size_t)M * K * 2);
    const float* mean = reinterpret_cast<const float*>(s); s += up256((size_t)M * 4);
    const float* rstd = reinterpret_cast<const float*>(s); s += up256((size_t)M * 4);
    const char* ha = s;
    char* q = static_cast<char*>(scratch_dev);
    void* dy16 = q; q += up256((size_t)M * N * 2);
    void* dh16 = q; q += up256((size_t)M * H * 2);
    float* dn = reinterpret_cast<float*>(q);
    int rc;
    // the order of _FfnBlock16.backward (some_amd/training/ops.py): dy16, dh16, dn, weight gradients (first linear, then second), LayerNorm
    if ((rc = some_train_dropcast16(h, d_dev, dy16, M, N, alpha, p_out, seed_out, operand, stream))) return rc;
    if ((rc = some_train_gemm16s(h, 2, dy16, N, w2t_16_dev, N, nullptr, dh16, H, ha, H, 0, M, H, N, operand, p_latent, seed_latent, 1.0f, stream))) return rc;
    if ((rc = some_train_gemm16s(h, 0, dh16, H, w1t_16_dev, H, nullptr, dn, K, nullptr, 0, 0, M, K, H, operand, 0.f, 0, 1.0f, stream))) return rc;
    if ((rc = some_train_gemm16_wgrad16(h, dh16, H, n16, K, dw1_dev, db1_dev, H, K, M, operand, 1, partial_dev, partial_bytes, stream))) return rc;
    // deferred reductions (weight-gradient lanes): the second product gets planes of its own behind the first's when the caller's buffer
    // holds both (some_train_gemm16_bytes of each, the first rounded up to 256 bytes) - sharing them would only force a flush in between
    char* part2 = static_cast<char*>(partial_dev);
    size_t bytes2 = partial_bytes;
    if (const SomeHandle::WgradLane* wl = wgrad_lane_of(h, st(stream)); wl && wl->defer) {
        const size_t first = up256((size_t)gemm16_slices(H, K, M, operand) * (size_t)H * (K + 4) * sizeof(float));
        if (partial_bytes >= first + (size_t)gemm16_slices(N, H, M, operand) * (size_t)N * (H + 4) * sizeof(float)) { part2 += first; bytes2 -= first; }
    }
    if ((rc = some_train_gemm16_wgrad16(h, dy16, N, ha + (size_t)M * H * 2, H, dw2_dev, db2_dev, N, H, M, operand, 1, part2, bytes2, stream))) return rc;
    return some_train_layernorm_bwd_add(h, dn, x_dev, gamma_dev, mean, rstd, add_residual ? d_dev : nullptr, dx_dev, dgamma_dev, dbeta_dev, 1, M,
                                        ln_scratch_dev, ln_scratch_bytes, stream);
}

int some_train_transpose(SomeHandle* h, const float* in_dev, int32_t M, int32_t N, int32_t ld_in, float* out_dev,
                         int32_t ld_out, int32_t split_out, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && N >= 0 && ld_in >= N && ld_out >= M, "some_train_transpose: bad shape");
    T_CHECK(h, split_out >= 0 && split_out <= 2, "some_train_transpose: split_out must be 0 (fp32), 1 (SPLIT32) or 2 (bf16 hi)");
    T_CHECK(h, !split_out || (ld_out % 32) == 0, "some_train_transpose: SPLIT32 output needs ld_out % 32 == 0");
    if (M == 0 || N == 0) return SOME_OK;
    T_CHECK(h, in_dev && out_dev, "some_train_transpose: null pointer");
    T_TRY(h, launch_transpose(in_dev, M, N, ld_in, out_dev, ld_out, split_out, st(stream)));
    return SOME_OK;
}

int some_train_colsum(SomeHandle* h, const float* x_dev, int32_t M, int32_t N, int32_t ld, float* out_dev,
                      int32_t accumulate, void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && N > 0 && ld >= N && out_dev, "some_train_colsum: bad argument");
    if (M == 0) return SOME_OK;
    T_CHECK(h, x_dev && scratch_dev && scratch_bytes >= train_col_scratch_bytes(M, N), "some_train_colsum: scratch too small");
    T_TRY(h, launch_colsum(x_dev, M, N, ld, out_dev, accumulate, static_cast<float*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_weighted_colsum(SomeHandle* h, const float* w_dev, int32_t ldw, const float* x_dev, int32_t M, int32_t N, int32_t ld,
                               float* out_dev, float* wsum_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && N > 0 && ld >= N && ldw >= 1 && out_dev, "some_train_weighted_colsum: bad argument");
    if (M == 0) return SOME_OK;
    T_CHECK(h, w_dev && x_dev && scratch_dev && scratch_bytes >= train_col_scratch_bytes(M, N), "some_train_weighted_colsum: scratch too small");
    T_TRY(h, launch_weighted_colsum(w_dev, ldw, x_dev, M, N, ld, out_dev, wsum_dev, static_cast<float*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_layernorm_fwd(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                             float* y_dev, float* mean_dev, float* rstd_dev, int32_t M, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && x_dev && gamma_dev && beta_dev && y_dev && mean_dev && rstd_dev, "some_train_layernorm_fwd: bad argument");
    T_TRY(h, launch_ln_fwd(x_dev, gamma_dev, beta_dev, y_dev, mean_dev, rstd_dev, M, 0, st(stream)));
    return SOME_OK;
}

int some_train_layernorm_fwd16(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                               void* y16_dev, float* mean_dev, float* rstd_dev, int32_t M, int32_t operand, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && x_dev && gamma_dev && beta_dev && y16_dev && mean_dev && rstd_dev, "some_train_layernorm_fwd16: bad argument");
    T_CHECK(h, operand == 1 || operand == 2, "some_train_layernorm_fwd16: operand must be 1 (f16) or 2 (bf16)");
    T_TRY(h, launch_ln_fwd(x_dev, gamma_dev, beta_dev, y16_dev, mean_dev, rstd_dev, M, operand, st(stream)));
    return SOME_OK;
}

int some_train_layernorm_bwd(SomeHandle* h, const float* dy_dev, const float* x_dev, const float* gamma_dev,
                             const float* mean_dev, const float* rstd_dev, float* dx_dev, float* dgamma_dev,
                             float* dbeta_dev, int32_t accumulate, int32_t M, void* scratch_dev, size_t scratch_bytes,
                             void* stream) {
    return some_train_layernorm_bwd_add(h, dy_dev, x_dev, gamma_dev, mean_dev, rstd_dev, nullptr, dx_dev, dgamma_dev, dbeta_dev, accumulate, M,
                                        scratch_dev, scratch_bytes, stream);
}

int some_train_layernorm_bwd_add(SomeHandle* h, const float* dy_dev, const float* x_dev, const float* gamma_dev,
                                 const float* mean_dev, const float* rstd_dev, const float* add_dev, float* dx_dev, float* dgamma_dev,
                                 float* dbeta_dev, int32_t accumulate, int32_t M, void* scratch_dev, size_t scratch_bytes,
                                 void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && dy_dev && x_dev && gamma_dev && mean_dev && rstd_dev && dx_dev && dgamma_dev && dbeta_dev,
            "some_train_layernorm_bwd: bad argument");
    if (M == 0) return SOME_OK;
    T_CHECK(h, scratch_dev && scratch_bytes >= train_ln_scratch_bytes(M), "some_train_layernorm_bwd: scratch too small (some_train_scratch_bytes(M, 512))");
    T_TRY(h, launch_ln_bwd(dy_dev, x_dev, gamma_dev, mean_dev, rstd_dev, add_dev, dx_dev, dgamma_dev, dbeta_dev, accumulate, M,
                           static_cast<float*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_batchnorm_fwd(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                             int32_t M, int32_t C, float eps, float momentum, float* running_mean_dev,
                             float* running_var_dev, float* y_dev, float* save_mean_dev, float* save_rstd_dev,
                             void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && C > 0 && (C % 4) == 0 && x_dev && gamma_dev && beta_dev && y_dev && save_mean_dev && save_rstd_dev,
            "some_train_batchnorm_fwd: bad argument");
    T_CHECK(h, (running_mean_dev == nullptr) == (running_var_dev == nullptr), "some_train_batchnorm_fwd: running stats come in pairs");
    T_CHECK(h, scratch_dev && scratch_bytes >= train_col_scratch_bytes(M, C), "some_train_batchnorm_fwd: scratch too small");
    T_TRY(h, launch_bn_fwd(x_dev, gamma_dev, beta_dev, M, C, eps, momentum, running_mean_dev, running_var_dev, y_dev, save_mean_dev,
                           save_rstd_dev, static_cast<float*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_batchnorm_bwd(SomeHandle* h, const float* dy_dev, const float* x_dev, const float* gamma_dev,
                             const float* save_mean_dev, const float* save_rstd_dev, int32_t M, int32_t C,
                             float* dx_dev, float* dgamma_dev, float* dbeta_dev, void* scratch_dev,
                             size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && C > 0 && (C % 4) == 0 && dy_dev && x_dev && gamma_dev && save_mean_dev && save_rstd_dev && dx_dev && dgamma_dev && dbeta_dev,
            "some_train_batchnorm_bwd: bad argument");
    T_CHECK(h, scratch_dev && scratch_bytes >= train_col_scratch_bytes(M, C), "some_train_batchnorm_bwd: scratch too small");
    T_TRY(h, launch_bn_bwd(dy_dev, x_dev, gamma_dev, save_mean_dev, save_rstd_dev, M, C, dx_dev, dgamma_dev, dbeta_dev,
                           static_cast<float*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_eltwise(SomeHandle* h, int32_t op, const float* a_dev, const float* b_dev, float* out_dev, int64_t n,
                       float alpha, float p, uint64_t seed, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, op >= 0 && op <= SOME_ELT_AXPY_DROP && n >= 0, "some_train_eltwise: bad op or size");
    if (n == 0) return SOME_OK;
    const bool needs_b = op == SOME_ELT_SILU_BWD || op == SOME_ELT_SIGMOID_BWD || op == SOME_ELT_SILU_DROP_BWD;
    T_CHECK(h, a_dev && out_dev && (!needs_b || b_dev), "some_train_eltwise: null pointer");
    T_CHECK(h, p >= 0.f && p < 1.f, "some_train_eltwise: dropout probability must be in [0, 1)");
    T_TRY(h, launch_eltwise(op, a_dev, b_dev, out_dev, n, alpha, p, seed, st(stream)));
    return SOME_OK;
}

int some_train_glu(SomeHandle* h, const float* dy_dev, const float* x_dev, float* out_dev, int64_t M, int32_t C,
                   int32_t backward, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && C > 0 && (C % 4) == 0 && x_dev && out_dev && (!backward || dy_dev), "some_train_glu: bad argument (C % 4 == 0, 16-byte aligned arrays)");
    T_TRY(h, launch_glu(dy_dev, x_dev, out_dev, M, C, backward, st(stream)));
    return SOME_OK;
}

int some_train_mask_rows(SomeHandle* h, const float* x_dev, const uint8_t* mask_dev, float* y_dev, int64_t M,
                         int32_t C, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && C > 0 && x_dev && mask_dev && y_dev, "some_train_mask_rows: bad argument");
    T_TRY(h, launch_mask_rows(x_dev, mask_dev, y_dev, M, C, st(stream)));
    return SOME_OK;
}

int some_train_dwconv(SomeHandle* h, const float* x_dev, const float* taps_dev, const float* bias_dev,
                      const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, float* y_dev, int32_t C,
                      int32_t flip, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, B >= 0 && max_frames >= 0 && C > 0 && (C % 4) == 0 && B <= 65535, "some_train_dwconv: bad shape");
    if (B == 0 || max_frames == 0) return SOME_OK;
    T_CHECK(h, x_dev && taps_dev && frame_offsets_dev && y_dev, "some_train_dwconv: null pointer");
    T_TRY(h, launch_dwconv_train(x_dev, taps_dev, bias_dev, frame_offsets_dev, B, max_frames, y_dev, C, flip, st(stream)));
    return SOME_OK;
}

int some_train_dwconv_bwd_taps(SomeHandle* h, const float* dy_dev, const float* x_dev, const int32_t* clip_of_row_dev,
                               const int32_t* frame_offsets_dev, int32_t M, int32_t C, float* dtaps_dev,
                               int32_t accumulate, void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && C > 0 && dtaps_dev, "some_train_dwconv_bwd_taps: bad argument");
    if (M == 0) return SOME_OK;
    T_CHECK(h, dy_dev && x_dev && clip_of_row_dev && frame_offsets_dev, "some_train_dwconv_bwd_taps: null pointer");
    T_CHECK(h, scratch_dev && scratch_bytes >= train_dwconv_w_scratch_bytes(M, C), "some_train_dwconv_bwd_taps: scratch too small");
    T_TRY(h, launch_dwconv_bwd_w(dy_dev, x_dev, clip_of_row_dev, frame_offsets_dev, M, C, dtaps_dev, accumulate,
                                 static_cast<float*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_dwconv_bwd_params(SomeHandle* h, const float* dy_dev, const float* x_dev, const int32_t* clip_of_row_dev,
                                 const int32_t* frame_offsets_dev, int32_t M, int32_t C, float* dweight_dev, float* dbias_dev,
                                 void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && C > 0 && dweight_dev, "some_train_dwconv_bwd_params: bad argument");
    if (M == 0) return SOME_OK;
    T_CHECK(h, dy_dev && x_dev && clip_of_row_dev && frame_offsets_dev, "some_train_dwconv_bwd_params: null pointer");
    T_CHECK(h, scratch_dev && scratch_bytes >= train_dwconv_w_scratch_bytes(M, C) && scratch_bytes >= train_col_scratch_bytes(M, C),
            "some_train_dwconv_bwd_params: scratch too small (some_train_scratch_bytes(M, C))");
    // nothing downstream reads these gradients: they run on the stream's weight-gradient side stream when it has one (the tap sums' two
    // launches and the bias column sum's two share the scratch block in stream order)
    hipStream_t ws = st(stream);
    if (SomeHandle::WgradLane* l = wgrad_lane_of(h, ws)) {
        T_TRY(h, hipEventRecord(l->ev, ws));
        T_TRY(h, hipStreamWaitEvent(l->side, l->ev, 0));
        ws = l->side;
    }
    T_TRY(h, launch_dwconv_bwd_w(dy_dev, x_dev, clip_of_row_dev, frame_offsets_dev, M, C, dweight_dev, 1, static_cast<float*>(scratch_dev), ws, 1));
    if (dbias_dev) T_TRY(h, launch_colsum(dy_dev, M, C, C, dbias_dev, 1, static_cast<float*>(scratch_dev), ws));
    return SOME_OK;
}

int some_train_bce_with_logits(SomeHandle* h, const float* logits_dev, const float* target_dev, int64_t n,
                               float* dlogits_dev, float* loss_dev, void* scratch_dev, size_t scratch_bytes,
                               void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, n > 0 && logits_dev && target_dev && loss_dev, "some_train_bce_with_logits: bad argument");
    T_CHECK(h, scratch_dev && scratch_bytes >= 1024 * sizeof(double), "some_train_bce_with_logits: scratch too small");
    T_TRY(h, launch_bce(logits_dev, target_dev, n, dlogits_dev, loss_dev, static_cast<double*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_cross_entropy(SomeHandle* h, const float* logits_dev, const int64_t* target_dev, int32_t M, int32_t N, int64_t ignore_index,
                             float* dlogits_dev, float* loss_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M > 0 && N > 0 && logits_dev && target_dev && loss_dev, "some_train_cross_entropy: bad argument");
    T_CHECK(h, scratch_dev && scratch_bytes >= 1025 * sizeof(double), "some_train_cross_entropy: scratch too small");
    T_TRY(h, launch_cross_entropy(logits_dev, target_dev, M, N, ignore_index, dlogits_dev, loss_dev, static_cast<double*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_binary_emd(SomeHandle* h, const float* pred_dev, const float* gt_dev, int32_t B, int32_t T,
                          float* dpred_dev, float* loss_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, B > 0 && T > 0 && pred_dev && gt_dev && loss_dev, "some_train_binary_emd: bad argument");
    T_CHECK(h, scratch_dev && scratch_bytes >= (size_t)B * sizeof(double), "some_train_binary_emd: scratch too small");
    T_TRY(h, launch_emd(pred_dev, gt_dev, B, T, dpred_dev, loss_dev, static_cast<double*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_sumsq(SomeHandle* h, const float* x_dev, int64_t n, double* out_dev, void* scratch_dev, size_t scratch_bytes,
                     void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, n >= 0 && out_dev && (n == 0 || x_dev), "some_train_sumsq: bad argument");
    T_CHECK(h, scratch_dev && scratch_bytes >= 1024 * sizeof(double), "some_train_sumsq: scratch too small");
    T_TRY(h, launch_sumsq(x_dev, n, out_dev, static_cast<double*>(scratch_dev), st(stream)));
    return SOME_OK;
}

int some_train_adamw(SomeHandle* h, float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                     int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                     float grad_scale, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, n >= 0 && step >= 1, "some_train_adamw: bad argument (step counts from 1)");
    if (n == 0) return SOME_OK;
    T_CHECK(h, param_dev && grad_dev && exp_avg_dev && exp_avg_sq_dev, "some_train_adamw: null pointer");
    T_TRY(h, launch_adamw(param_dev, grad_dev, exp_avg_dev, exp_avg_sq_dev, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, st(stream)));
    return SOME_OK;
}

int some_train_adamw_clip(SomeHandle* h, float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                          int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                          const double* sumsq_dev, double clip_norm, double grad_denominator, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, n >= 0 && step >= 1 && grad_denominator > 0.0 && clip_norm >= 0.0, "some_train_adamw_clip: bad argument (step counts from 1)");
    if (n == 0) return SOME_OK;
    T_CHECK(h, param_dev && grad_dev && exp_avg_dev && exp_avg_sq_dev && sumsq_dev, "some_train_adamw_clip: null pointer");
    T_TRY(h, launch_adamw_clip(param_dev, grad_dev, exp_avg_dev, exp_avg_sq_dev, n, lr, beta1, beta2, eps, weight_decay, step, sumsq_dev, clip_norm,
                               grad_denominator, st(stream)));
    return SOME_OK;
}

int some_train_attention_fwd(SomeHandle* h, const float* qkv_dev, const int32_t* frame_offsets_dev, int32_t B,
                             int32_t max_frames, int32_t M, float* out_dev, float* lse_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, B >= 0 && max_frames >= 0 && M >= 0, "some_train_attention_fwd: negative size");
    if (B == 0 || M == 0) return SOME_OK;
    T_CHECK(h, qkv_dev && frame_offsets_dev && out_dev && lse_dev, "some_train_attention_fwd: null pointer");
    AttnArgs a{};
    a.qkv[0] = qkv_dev; a.out[0] = out_dev; a.frame_offsets = frame_offsets_dev; a.groups = 1; a.B = B; a.max_frames = max_frames;
    a.lse[0] = lse_dev; a.M = M;
    T_TRY(h, launch_attention(a, st(stream)));
    return SOME_OK;
}

int some_train_attention_bwd(SomeHandle* h, const float* qkv_dev, const float* out_dev, const float* dout_dev,
                             const float* lse_dev, const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames,
                             int32_t M, float* dqkv_dev, float* dsum_scratch_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, B >= 0 && max_frames >= 0 && M >= 0, "some_train_attention_bwd: negative size");
    if (B == 0 || M == 0) return SOME_OK;
    T_CHECK(h, qkv_dev && out_dev && dout_dev && lse_dev && frame_offsets_dev && dqkv_dev && dsum_scratch_dev, "some_train_attention_bwd: null pointer");
    AttnBwdArgs a{qkv_dev, out_dev, dout_dev, lse_dev, dsum_scratch_dev, dqkv_dev, frame_offsets_dev, B, max_frames, M};
    T_TRY(h, launch_attention_bwd(a, st(stream)));
    return SOME_OK;
}

int some_train_attention_fwd_f16x3(SomeHandle* h, const float* qkv_split_dev, const float* qkv_t_split_dev,
                                   const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, int32_t M,
                                   int32_t Mp, int32_t hi_only, float* out_dev, float* lse_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, B >= 0 && max_frames >= 0 && M >= 0, "some_train_attention_fwd_f16x3: negative size");
    if (B == 0 || M == 0) return SOME_OK;
    T_CHECK(h, Mp >= M && (Mp % 64) == 0, "some_train_attention_fwd_f16x3: Mp must be M rounded up to a multiple of 64");
    T_CHECK(h, qkv_split_dev && qkv_t_split_dev && frame_offsets_dev && out_dev && lse_dev, "some_train_attention_fwd_f16x3: null pointer");
    Attn3Args a{};
    a.q[0] = qkv_split_dev; a.k[0] = qkv_split_dev + kDim;
    a.vt[0] = qkv_t_split_dev + (size_t)2 * kDim * Mp;            // V rows of the frame-major split tensor
    a.out32[0] = out_dev; a.lse[0] = lse_dev; a.hi_only = hi_only;
    a.frame_offsets = frame_offsets_dev; a.groups = 1; a.B = B; a.max_frames = max_frames; a.M = M; a.ldv = Mp;
    T_TRY(h, launch_attention_f16x3(a, st(stream)));
    return SOME_OK;
}

int some_train_attention_bwd_f16x3(SomeHandle* h, const float* qkv_split_dev, const float* qkv_t_split_dev,
                                   const float* dout_split_dev, const float* dout_t_split_dev, const float* out_dev,
                                   const float* dout_dev, const float* lse_dev, const int32_t* frame_offsets_dev,
                                   int32_t B, int32_t max_frames, int32_t M, int32_t Mp, int32_t hi_only,
                                   float* dqkv_dev, float* dsum_scratch_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, B >= 0 && max_frames >= 0 && M >= 0, "some_train_attention_bwd_f16x3: negative size");
    if (B == 0 || M == 0) return SOME_OK;
    T_CHECK(h, Mp >= M && (Mp % 32) == 0, "some_train_attention_bwd_f16x3: Mp must be M rounded up to a multiple of 32");
    T_CHECK(h, qkv_split_dev && qkv_t_split_dev && dout_split_dev && dout_t_split_dev && out_dev && dout_dev && lse_dev && frame_offsets_dev &&
                   dqkv_dev && dsum_scratch_dev, "some_train_attention_bwd_f16x3: null pointer");
    T_TRY(h, launch_attention_dsum(out_dev, dout_dev, dsum_scratch_dev, M, st(stream)));
    T_TRY(h, launch_attention_bwd_f16x3(qkv_split_dev, qkv_t_split_dev, dout_split_dev, dout_t_split_dev, lse_dev, dsum_scratch_dev,
                                        frame_offsets_dev, B, max_frames, M, Mp, dqkv_dev, hi_only, st(stream)));
    return SOME_OK;
}

int some_train_attention_bwd_f16x3_out16(SomeHandle* h, const float* qkv_split_dev, const float* qkv_t_split_dev,
                                         const float* dout_split_dev, const float* dout_t_split_dev, const float* out_dev,
                                         const float* dout_dev, const float* lse_dev, const int32_t* frame_offsets_dev,
                                         int32_t B, int32_t max_frames, int32_t M, int32_t Mp, int32_t hi_only,
                                         void* dqkv16_dev, const float* out_scale_dev, float* dsum_scratch_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, B >= 0 && max_frames >= 0 && M >= 0, "some_train_attention_bwd_f16x3_out16: negative size");
    if (B == 0 || M == 0) return SOME_OK;
    T_CHECK(h, hi_only == 1 || hi_only == 2, "some_train_attention_bwd_f16x3_out16: mixed precision only (hi_only 1 = f16, 2 = bf16: also the output format)");
    T_CHECK(h, Mp >= M && (Mp % 32) == 0, "some_train_attention_bwd_f16x3_out16: Mp must be M rounded up to a multiple of 32");
    T_CHECK(h, qkv_split_dev && qkv_t_split_dev && dout_split_dev && dout_t_split_dev && out_dev && dout_dev && lse_dev && frame_offsets_dev &&
                   dqkv16_dev && out_scale_dev && dsum_scratch_dev, "some_train_attention_bwd_f16x3_out16: null pointer");
    T_CHECK(h, (reinterpret_cast<uintptr_t>(dqkv16_dev) & 7) == 0, "some_train_attention_bwd_f16x3_out16: dqkv16 must be 8-byte aligned");
    T_TRY(h, launch_attention_dsum(out_dev, dout_dev, dsum_scratch_dev, M, st(stream)));
    T_TRY(h, launch_attention_bwd_f16x3(qkv_split_dev, qkv_t_split_dev, dout_split_dev, dout_t_split_dev, lse_dev, dsum_scratch_dev,
                                        frame_offsets_dev, B, max_frames, M, Mp, nullptr, hi_only, st(stream), dqkv16_dev, out_scale_dev, hi_only));
    return SOME_OK;
}

int some_train_split_transpose(SomeHandle* h, const float* x_dev, int32_t M, int32_t N, float* rows_split_dev, float* t_split_dev, int32_t Mp,
                               int32_t format, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, M >= 0 && N >= 0 && (N % 32) == 0 && Mp >= M && (Mp % 32) == 0, "some_train_split_transpose: bad shape (N % 32, Mp % 32, Mp >= M)");
    T_CHECK(h, format == SOME_OPERAND_F16X2 || format == SOME_OPERAND_BF16, "some_train_split_transpose: unknown operand format");
    if (M == 0 || N == 0) return SOME_OK;
    T_CHECK(h, x_dev && rows_split_dev && t_split_dev, "some_train_split_transpose: null pointer");
    T_TRY(h, launch_split_transpose(x_dev, M, N, rows_split_dev, t_split_dev, Mp, format == SOME_OPERAND_BF16, 0, nullptr, nullptr, st(stream)));
    return SOME_OK;
}

namespace {
struct Bwd16Work { size_t d, dt, dsum, absmax, factor, total; };
Bwd16Work bwd16_work(int M, int Mp) {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    Bwd16Work w{};
    w.d = 0;
    w.dt = w.d + up((size_t)M * kDim * sizeof(float));
    w.dsum = w.dt + up((size_t)kDim * Mp * sizeof(float));
    w.absmax = w.dsum + up((size_t)8 * M * sizeof(float));
    w.factor = w.absmax + up(256 * sizeof(uint32_t));
    w.total = w.factor + 256;
    return w;
}
}  // namespace

size_t some_train_attention_bwd16_work_bytes(const SomeHandle* h, int32_t M, int32_t Mp) {
    (void)h;
    if (M <= 0 || Mp < M) return 0;
    return bwd16_work(M, Mp).total;
}

int some_train_attention_bwd_f16x3_auto16(SomeHandle* h, const float* qkv_split_dev, const float* qkv_t_split_dev, const float* out_dev,
                                          const float* dout_dev, const float* lse_dev, const int32_t* frame_offsets_dev, int32_t B,
                                          int32_t max_frames, int32_t M, int32_t Mp, int32_t hi_only, void* dqkv16_dev, void* work_dev,
                                          size_t work_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    T_CHECK(h, B >= 0 && max_frames >= 0 && M >= 0, "some_train_attention_bwd_f16x3_auto16: negative size");
    if (B == 0 || M == 0) return SOME_OK;
    T_CHECK(h, hi_only == 1 || hi_only == 2, "some_train_attention_bwd_f16x3_auto16: mixed precision only (hi_only 1 = f16, 2 = bf16: also the output format)");
    T_CHECK(h, Mp >= M && (Mp % 32) == 0, "some_train_attention_bwd_f16x3_auto16: Mp must be M rounded up to a multiple of 32");
    T_CHECK(h, qkv_split_dev && qkv_t_split_dev && out_dev && dout_dev && lse_dev && frame_offsets_dev && dqkv16_dev && work_dev,
            "some_train_attention_bwd_f16x3_auto16: null pointer");
    T_CHECK(h, (reinterpret_cast<uintptr_t>(dqkv16_dev) & 7) == 0 && (reinterpret_cast<uintptr_t>(work_dev) & 255) == 0,
            "some_train_attention_bwd_f16x3_auto16: dqkv16 must be 8-byte, the work area 256-byte aligned");
    const Bwd16Work w = bwd16_work(M, Mp);
    T_CHECK(h, work_bytes >= w.total, "some_train_attention_bwd_f16x3_auto16: work area smaller than some_train_attention_bwd16_work_bytes");
    char* base = static_cast<char*>(work_dev);
    float* D = reinterpret_cast<float*>(base + w.d);
    float* Dt = reinterpret_cast<float*>(base + w.dt);
    float* dsum = reinterpret_cast<float*>(base + w.dsum);
    uint32_t* absmax = reinterpret_cast<uint32_t*>(base + w.absmax);
    float* factor = reinterpret_cast<float*>(base + w.factor);
    T_TRY(h, launch_split_transpose(dout_dev, M, kDim, D, Dt, Mp, hi_only == 2, 1, absmax, factor, st(stream)));
    T_TRY(h, launch_attention_dsum(out_dev, dout_dev, dsum, M, st(stream), factor));
    T_TRY(h, launch_attention_bwd_f16x3(qkv_split_dev, qkv_t_split_dev, D, Dt, lse_dev, dsum, frame_offsets_dev, B, max_frames, M, Mp, nullptr, hi_only,
                                        st(stream), dqkv16_dev, factor + 1, hi_only));
    return SOME_OK;
}

}  // extern "C"
