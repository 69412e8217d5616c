// C ABI of libsome_amd.so (include/some_amd.h): handle lifecycle, weight packing, and the launch sequences
// that replace the reference's Python-level op chains.  Host code only; kernels live in the sibling files.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <thread>

#include "internal.h"
#include "split.h"

namespace {

thread_local std::string g_create_error;

int fail(SomeHandle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}
int fail_hip(SomeHandle* h, hipError_t e, const char* what) {
    return fail(h, SOME_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(h, expr)                                             \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return fail_hip((h), _e, #expr);       \
    } while (0)

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// fp32 k-blocks of 32 -> SPLIT32 in place with the x86 F16C conversions (round to nearest even, subnormals kept: the same
// values as split.h's scalar split_f16, which falls back to a software conversion routine per element on the host).
// Returns false when the CPU lacks F16C / AVX: the caller uses the scalar path.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx,f16c"))) void split_blocks_f16c_impl(float* blk, size_t n_blocks) {
    for (size_t i = 0; i < n_blocks; ++i, blk += 32) {
        __m128i hi[4], lo[4];
        for (int q = 0; q < 4; ++q) {
            const __m256 x = _mm256_loadu_ps(blk + 8 * q);
            hi[q] = _mm256_cvtps_ph(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
            lo[q] = _mm256_cvtps_ph(_mm256_sub_ps(x, _mm256_cvtph_ps(hi[q])), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        }
        for (int q = 0; q < 4; ++q) {
            _mm_storeu_si128(reinterpret_cast<__m128i*>(blk) + q, hi[q]);
            _mm_storeu_si128(reinterpret_cast<__m128i*>(blk) + 4 + q, lo[q]);
        }
    }
}
bool split_blocks_f16c(float* blk, size_t n_blocks) {
    static const bool ok = __builtin_cpu_supports("f16c") && __builtin_cpu_supports("avx");
    if (ok) split_blocks_f16c_impl(blk, n_blocks);
    return ok;
}
#else
bool split_blocks_f16c(float*, size_t) { return false; }
#endif
constexpr size_t kWsSlack = 1 << 20;   // per stream: slack behind the last array

// Workspace of some_forward: per model stream X [M,512] | H [M,512] | G [M,512] | U, then the attention plan (f16x3 mode).
// U holds the FFN hidden rows [M,2048] and, in f16x3 mode, the attention operands in clip-aligned rows (Mc = attn_rows_cover):
// Q | K SPLIT32 planes [Mc,512] and the V^T f16 planes 2 x [512, vt_ld(Mc)] - larger than the FFN rows only for batches of
// very short clips.
struct WsLayout {
    size_t per_stream;     // floats
    size_t plan_off;       // bytes: pad_offsets int32 [B + 1], then row_map int32 [Mc]
    size_t map_off;        // bytes
    size_t total;          // bytes
    int64_t Mc;
};
WsLayout ws_layout(int precision, int64_t total_frames, int32_t B) {
    WsLayout w{};
    const size_t m = (size_t)total_frames;
    w.Mc = attn_rows_cover(total_frames, B);
    size_t u_bytes = m * (size_t)kFfn * sizeof(float);
    if (precision == SOME_PRECISION_F16X3)
        u_bytes = std::max(u_bytes, (size_t)w.Mc * 2 * kDim * sizeof(float) + (size_t)2 * kDim * vt_ld(w.Mc) * 2);
    w.per_stream = (align_up(m * (size_t)(kDim * 3) * sizeof(float) + u_bytes, 256) + kWsSlack) / sizeof(float);
    w.plan_off = kStreams * w.per_stream * sizeof(float);
    w.map_off = w.plan_off + align_up(((size_t)B + 1) * 4, 256);
    w.total = w.map_off + align_up((size_t)w.Mc * 4, 256) + 1024;
    return w;
}

// ---- arena layout ------------------------------------------------------------------------------------
struct Cursor {
    size_t pos = 0;
    size_t take(size_t n) { size_t p = pos; pos = align_up(pos + n, 64); return p; }   // 256-byte aligned
};

void build_layout(const SomeConfig& c, ArenaLayout& L) {
    Cursor cur;
    for (int g = 0; g < kStreams; ++g) { L.in_w[g] = cur.take((size_t)kDim * c.indim); L.in_b[g] = cur.take(kDim); }
    L.out_w = cur.take((size_t)c.outdim * kDim);
    L.out_b = cur.take(c.outdim);
    L.cut_w = cur.take(kDim);
    L.cut_b = cur.take(1);
    L.blocks.resize((size_t)(c.lay + 1) * 2);
    for (auto& b : L.blocks) {
        for (int i = 0; i < 5; ++i) { b.ln_g[i] = cur.take(kDim); b.ln_b[i] = cur.take(kDim); }
        for (int f = 0; f < 2; ++f) {
            b.ffn_w1[f] = cur.take((size_t)kFfn * kDim); b.ffn_b1[f] = cur.take(kFfn);
            b.ffn_w2[f] = cur.take((size_t)kDim * kFfn); b.ffn_b2[f] = cur.take(kDim);
        }
        b.wqkv = cur.take((size_t)3 * kDim * kDim);
        b.wo = cur.take((size_t)kDim * kDim); b.bo = cur.take(kDim);
        b.pw1_w = cur.take((size_t)2 * kDim * kDim); b.pw1_b = cur.take(2 * kDim);
        b.dw_w = cur.take((size_t)kConvK * kDim); b.dw_b = cur.take(kDim);
        b.pw2_w = cur.take((size_t)kDim * kDim); b.pw2_b = cur.take(kDim);
    }
    L.glu_w.resize((size_t)c.lay * 2);
    L.glu_b.resize((size_t)c.lay * 2);
    for (size_t i = 0; i < L.glu_w.size(); ++i) { L.glu_w[i] = cur.take((size_t)2 * kDim * kDim); L.glu_b[i] = cur.take(2 * kDim); }
    L.total_floats = cur.pos;
}

// ---- mel filterbank (librosa.filters.mel, htk=True, norm='slaney'; call site spec.py:22-28) ----------
double hz_to_mel(double f) { return 2595.0 * std::log10(1.0 + f / 700.0); }
double mel_to_hz(double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); }

void mel_basis(const SomeConfig& c, std::vector<float>& out) {
    const int n_bins = 1 + c.win_size / 2, n_mels = kMels;
    const double fmax = c.fmax > 0 ? (double)c.fmax : c.sample_rate / 2.0;
    std::vector<double> fftf(n_bins), melf(n_mels + 2);
    const double fstep = (c.sample_rate / 2.0) / (n_bins - 1);
    for (int i = 0; i < n_bins; ++i) fftf[i] = i * fstep;
    fftf[n_bins - 1] = c.sample_rate / 2.0;
    const double m0 = hz_to_mel(c.fmin), m1 = hz_to_mel(fmax);
    const double mstep = (m1 - m0) / (n_mels + 1);
    for (int i = 0; i < n_mels + 2; ++i) melf[i] = mel_to_hz(i == n_mels + 1 ? m1 : m0 + i * mstep);
    out.assign((size_t)n_mels * n_bins, 0.f);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = melf[i + 1] - melf[i], fd1 = melf[i + 2] - melf[i + 1];
        const double enorm = 2.0 / (melf[i + 2] - melf[i]);
        for (int j = 0; j < n_bins; ++j) {
            const double lower = -(melf[i] - fftf[j]) / fd0;
            const double upper = (melf[i + 2] - fftf[j]) / fd1;
            // librosa builds the triangles in a float32 array and scales them in place: two fp32 roundings
            const float w = (float)std::max(0.0, std::min(lower, upper));
            out[(size_t)i * n_bins + j] = (float)((double)w * enorm);
        }
    }
}

int ensure_mel_tables(SomeHandle* h) {
    if (h->mel_blob) return SOME_OK;
    const int n_bins = 1 + kWin / 2;
    std::vector<float> basis;
    mel_basis(h->cfg, basis);
    std::vector<int32_t> start(kMels), len(kMels), off(kMels);
    std::vector<float> packed;
    int kmax = 0;
    for (int m = 0; m < kMels; ++m) {
        int first = -1, last = -1;
        for (int j = 0; j < n_bins; ++j)
            if (basis[(size_t)m * n_bins + j] != 0.f) { if (first < 0) first = j; last = j; }
        if (first < 0) { first = 0; last = -1; }
        start[m] = first; len[m] = last - first + 1; off[m] = (int32_t)packed.size();
        for (int j = first; j <= last; ++j) packed.push_back(basis[(size_t)m * n_bins + j]);
        kmax = std::max(kmax, last);
    }
    std::vector<float> window(kWin), tw(2 * 2048);
    for (int n = 0; n < kWin; ++n) window[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / kWin));   // periodic Hann
    for (int k = 0; k < 2048; ++k) {
        const double ang = -2.0 * M_PI * k / 2048.0;
        tw[2 * k] = (float)std::cos(ang);
        tw[2 * k + 1] = (float)std::sin(ang);
    }
    // one blob: window | twiddle | mel_w | start | len | off
    const size_t o_win = 0, o_tw = o_win + kWin * 4, o_w = o_tw + tw.size() * 4;
    const size_t o_st = align_up(o_w + packed.size() * 4, 16), o_len = o_st + kMels * 4, o_off = o_len + kMels * 4;
    constexpr int kPad = 32;
    const size_t o_pad = align_up(o_off + kMels * 4, 16);
    const size_t total = o_pad + (size_t)(kMels + 1) * kPad * 4;
    std::vector<char> host(total, 0);
    for (int m = 0; m < kMels; ++m)
        for (int i = 0; i < len[m] && i < kPad; ++i)
            memcpy(host.data() + o_pad + ((size_t)m * kPad + i) * 4, &packed[(size_t)off[m] + i], 4);
    memcpy(host.data() + o_win, window.data(), kWin * 4);
    memcpy(host.data() + o_tw, tw.data(), tw.size() * 4);
    memcpy(host.data() + o_w, packed.data(), packed.size() * 4);
    memcpy(host.data() + o_st, start.data(), kMels * 4);
    memcpy(host.data() + o_len, len.data(), kMels * 4);
    memcpy(host.data() + o_off, off.data(), kMels * 4);
    void* dev = nullptr;
    HIP_TRY(h, hipMalloc(&dev, total));
    hipError_t e = hipMemcpy(dev, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(dev); return fail_hip(h, e, "hipMemcpy(mel tables)"); }
    char* d = static_cast<char*>(dev);
    h->mel_blob = dev;
    h->mel.window = reinterpret_cast<float*>(d + o_win);
    h->mel.twiddle = reinterpret_cast<float*>(d + o_tw);
    h->mel.mel_w = reinterpret_cast<float*>(d + o_w);
    h->mel.mel_start = reinterpret_cast<int32_t*>(d + o_st);
    h->mel.mel_len = reinterpret_cast<int32_t*>(d + o_len);
    h->mel.mel_off = reinterpret_cast<int32_t*>(d + o_off);
    h->mel.mel_wpad = reinterpret_cast<float*>(d + o_pad);
    h->mel.kmax = std::min(kmax, 1024);
    h->mel.nnz = (int)packed.size();
    h->mel.max_len = *std::max_element(len.begin(), len.end());
    return SOME_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------
struct Scope {
    SomeHandle* h; hipStream_t s; ProfRecord rec; bool on;
    Scope(SomeHandle* h_, hipStream_t s_, const char* name, double flops, double bytes) : h(h_), s(s_), on(h_->profiling) {
        if (!on) return;
        rec.name = name; rec.flops = flops; rec.bytes = bytes;
        auto get = [&]() {
            hipEvent_t e;
            if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        rec.e0 = get(); rec.e1 = get();
        (void)hipEventRecord(rec.e0, s);
    }
    ~Scope() {
        if (!on) return;
        (void)hipEventRecord(rec.e1, s);
        h->prof.push_back(rec);
    }
};

}  // namespace

extern "C" {

const char* some_version(void) { return "some_amd 0.6 gfx950 (f32 / split-f16 MFMA conformer, HIP)"; }

const char* some_last_error(const SomeHandle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int some_create(const SomeConfig* cfg, SomeHandle** out) {
    if (!cfg || !out) return fail(nullptr, SOME_EINVAL, "some_create: null argument");
    char msg[256];
#define REQUIRE(cond, ...)                                                                   \
    if (!(cond)) { snprintf(msg, sizeof msg, __VA_ARGS__); return fail(nullptr, SOME_EINVAL, msg); }
    REQUIRE(cfg->lay >= 0 && cfg->lay <= 64, "unsupported lay=%d", cfg->lay);
    REQUIRE(cfg->dim == kDim, "unsupported dim=%d (compiled for %d)", cfg->dim, kDim);
    REQUIRE(cfg->heads == kHeads && cfg->head_dim == kHeadDim, "unsupported attention %dx%d (compiled for %dx%d)",
            cfg->heads, cfg->head_dim, kHeads, kHeadDim);
    REQUIRE(cfg->kernel_size == kConvK, "unsupported kernel_size=%d (compiled for %d)", cfg->kernel_size, kConvK);
    REQUIRE(cfg->indim > 0 && cfg->indim % 4 == 0 && cfg->indim <= 1024, "unsupported units_dim=%d (need a multiple of 4)", cfg->indim);
    REQUIRE(cfg->outdim >= 2 && cfg->outdim <= 192, "unsupported midi_num_bins=%d", cfg->outdim);
    REQUIRE(cfg->win_size == kWin && cfg->hop_size == kHop, "unsupported win/hop %d/%d (compiled for %d/%d)",
            cfg->win_size, cfg->hop_size, kWin, kHop);
    REQUIRE(cfg->indim == kMels, "front end is compiled for %d mel bands, got units_dim=%d", kMels, cfg->indim);
    REQUIRE(cfg->sample_rate > 0 && cfg->fmin >= 0, "bad sample_rate/fmin");
#undef REQUIRE
    if (cfg->precision != SOME_PRECISION_F32 && cfg->precision != SOME_PRECISION_F16X3 && cfg->precision != SOME_PRECISION_F16X3_FAST)
        return fail(nullptr, SOME_EINVAL, "unsupported precision (0 = f32, 1 = f16x3, 2 = f16x3_fast)");
    SomeHandle* h = new SomeHandle();
    h->cfg = *cfg;
    h->precision = cfg->precision;
    if (cfg->precision == SOME_PRECISION_F16X3_FAST) {     // everything is the f16x3 path; only the attention kernel differs
        h->precision = SOME_PRECISION_F16X3;
        h->attn_fast = 1;
        if (const char* t = getenv("SOME_AMD_ATTN_FAST")) h->attn_fast = atoi(t) == 2 ? 2 : 1;      // (2: the measured-only variant without kh * ql)
    }
    h->tile = -1;                                    // -1: pick per launch from the grid size
    if (const char* t = getenv("SOME_AMD_TILE")) h->tile = atoi(t);
    if (const char* t = getenv("SOME_AMD_GEMM_FLAGS")) h->gemm_flags = atoi(t);
    if (const char* t = getenv("SOME_AMD_DUAL_STREAM")) h->dual_stream = atoi(t) != 0;
    build_layout(h->cfg, h->lay);
    *out = h;
    return SOME_OK;
}

void some_destroy(SomeHandle* h) {
    if (!h) return;
    if (h->mel_blob) (void)hipFree(h->mel_blob);
    for (auto& t : h->shift_tables) (void)hipFree(t.blob);
    for (auto& a : h->aux_sets) { (void)hipStreamDestroy(a.aux); (void)hipEventDestroy(a.fork); (void)hipEventDestroy(a.join); }
    for (auto& r : h->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    for (auto& l : h->wgrad_lanes) (void)hipEventDestroy(l.ev);
    delete h;
}

size_t some_arena_bytes(const SomeHandle* h) { return h ? h->lay.total_floats * sizeof(float) : 0; }

int some_mel_filterbank(const SomeHandle* h, float* basis_host) {
    if (!h || !basis_host) return SOME_EINVAL;
    std::vector<float> b;
    mel_basis(h->cfg, b);
    memcpy(basis_host, b.data(), b.size() * sizeof(float));
    return SOME_OK;
}

int some_pack_weights(SomeHandle* h, const SomeTensorDesc* tensors, int32_t n, float* arena) {
    if (!h) return SOME_EINVAL;
    if (!tensors || !arena || n < 0) return fail(h, SOME_EINVAL, "some_pack_weights: null argument");
    const SomeConfig& c = h->cfg;
    std::map<std::string, const SomeTensorDesc*> have;
    for (int i = 0; i < n; ++i) have[tensors[i].name] = &tensors[i];
    std::set<std::string> used;
    std::string missing, bad_shape;
    auto get = [&](const std::string& name, std::initializer_list<int64_t> shape) -> const float* {
        auto it = have.find(name);
        if (it == have.end()) { if (missing.size() < 400) missing += " " + name; return nullptr; }
        used.insert(name);
        const SomeTensorDesc* t = it->second;
        bool ok = t->dtype == 0 && t->ndim == (int)shape.size();
        int d = 0;
        for (int64_t s : shape) { if (ok && t->shape[d] != s) ok = false; ++d; }
        if (!ok) { if (bad_shape.size() < 400) bad_shape += " " + name; return nullptr; }
        return static_cast<const float*>(t->data);
    };
    std::fill(arena, arena + h->lay.total_floats, 0.f);
    auto copy = [&](size_t off, const float* src, size_t cnt) { if (src) memcpy(arena + off, src, cnt * sizeof(float)); };
    // GLU-producing GEMMs: packed row p <- source row (p%64 < 32 ? a-row : gate-row) so that one wave's two
    // 32-column MFMA tiles hold an output column and its gate (gemm.hip, EPI_GLU)
    auto copy_glu = [&](size_t w_off, size_t b_off, const float* w, const float* b) {
        if (!w || !b) return;
        for (int p = 0; p < 2 * kDim; ++p) {
            const int c64 = p / 64, wi = p % 64;
            const int src = wi < 32 ? c64 * 32 + wi : kDim + c64 * 32 + (wi - 32);
            memcpy(arena + w_off + (size_t)p * kDim, w + (size_t)src * kDim, kDim * sizeof(float));
            arena[b_off + p] = b[src];
        }
    };
    const int64_t D = kDim, F = kFfn, K = kConvK;
    const ArenaLayout& L = h->lay;
    copy(L.in_w[0], get("model.inln.weight", {D, c.indim}), (size_t)D * c.indim);
    copy(L.in_b[0], get("model.inln.bias", {D}), D);
    copy(L.in_w[1], get("model.inln1.weight", {D, c.indim}), (size_t)D * c.indim);
    copy(L.in_b[1], get("model.inln1.bias", {D}), D);
    copy(L.out_w, get("model.outln.weight", {c.outdim, D}), (size_t)c.outdim * D);
    copy(L.out_b, get("model.outln.bias", {c.outdim}), c.outdim);
    copy(L.cut_w, get("model.cutheard.weight", {1, D}), D);
    copy(L.cut_b, get("model.cutheard.bias", {1}), 1);
    for (int layer = 0; layer <= c.lay; ++layer) {
        for (int g = 0; g < kStreams; ++g) {
            const std::string p = (layer < c.lay ? "model.cf_lay." + std::to_string(layer) : std::string("model")) +
                                  (g == 0 ? ".att1." : ".att2.");
            const BlockOff& b = L.blocks[(size_t)layer * 2 + g];
            for (int i = 0; i < 5; ++i) {
                copy(b.ln_g[i], get(p + "norm" + std::to_string(i + 1) + ".weight", {D}), D);
                copy(b.ln_b[i], get(p + "norm" + std::to_string(i + 1) + ".bias", {D}), D);
            }
            for (int f = 0; f < 2; ++f) {
                const std::string q = p + (f == 0 ? "ffn1." : "ffn2.");
                copy(b.ffn_w1[f], get(q + "ln1.weight", {F, D}), (size_t)F * D);
                copy(b.ffn_b1[f], get(q + "ln1.bias", {F}), F);
                copy(b.ffn_w2[f], get(q + "ln2.weight", {D, F}), (size_t)D * F);
                copy(b.ffn_b2[f], get(q + "ln2.bias", {D}), D);
            }
            copy(b.wqkv, get(p + "att.to_q.weight", {D, D}), (size_t)D * D);                 // rows 0..511   = q
            copy(b.wqkv + (size_t)D * D, get(p + "att.to_kv.weight", {2 * D, D}), (size_t)2 * D * D);   // k | v
            copy(b.wo, get(p + "att.to_out.0.weight", {D, D}), (size_t)D * D);
            copy(b.bo, get(p + "att.to_out.0.bias", {D}), D);
            copy_glu(b.pw1_w, b.pw1_b, get(p + "conv.pointwise_conv1.weight", {2 * D, D, 1}),
                     get(p + "conv.pointwise_conv1.bias", {2 * D}));
            const float* dw = get(p + "conv.depthwise_conv.weight", {D, 1, K});
            const float* db = get(p + "conv.depthwise_conv.bias", {D});
            const float* bn_w = get(p + "conv.norm.weight", {D});
            const float* bn_b = get(p + "conv.norm.bias", {D});
            const float* bn_m = get(p + "conv.norm.running_mean", {D});
            const float* bn_v = get(p + "conv.norm.running_var", {D});
            if (have.count(p + "conv.norm.num_batches_tracked")) used.insert(p + "conv.norm.num_batches_tracked");
            if (dw && db && bn_w && bn_b && bn_m && bn_v) {
                for (int ch = 0; ch < kDim; ++ch) {          // BatchNorm1d eval fold, eps 1e-5 (base_conv.py:52,66)
                    const double scale = (double)bn_w[ch] / std::sqrt((double)bn_v[ch] + 1e-5);
                    for (int j = 0; j < kConvK; ++j)
                        arena[b.dw_w + (size_t)j * kDim + ch] = (float)((double)dw[(size_t)ch * kConvK + j] * scale);
                    arena[b.dw_b + ch] = (float)(((double)db[ch] - (double)bn_m[ch]) * scale + (double)bn_b[ch]);
                }
            }
            copy(b.pw2_w, get(p + "conv.pointwise_conv2.weight", {D, D, 1}), (size_t)D * D);
            copy(b.pw2_b, get(p + "conv.pointwise_conv2.bias", {D}), D);
        }
        if (layer < c.lay) {
            for (int k = 0; k < 2; ++k) {
                const std::string p = "model.cf_lay." + std::to_string(layer) + (k == 0 ? ".glu1.0." : ".glu2.0.");
                copy_glu(L.glu_w[(size_t)layer * 2 + k], L.glu_b[(size_t)layer * 2 + k],
                         get(p + "weight", {2 * D, D}), get(p + "bias", {2 * D}));
            }
        }
    }
    if (h->precision == SOME_PRECISION_F16X3) {
        // every GEMM weight except the K = units_dim input projections -> SPLIT32 in place (same byte size)
        // 117 M weights (lay 8): the in-place conversion is spread over a few host threads (a cold start spends ~1 s here)
        auto to_split = [&](size_t off, size_t rows, size_t K) {
            const size_t n_blk = rows * (K / 32);
            const unsigned hw = std::thread::hardware_concurrency();
            const size_t n_thr = n_blk < 4096 ? 1 : std::min<size_t>(8, hw ? hw : 1);
            auto work = [&](size_t b0, size_t b1) {
                if (split_blocks_f16c(arena + off + b0 * 32, b1 - b0)) return;      // hardware conversions where the host has them
                half_t tmp[64];
                for (size_t i = b0; i < b1; ++i) {
                    float* blk = arena + off + i * 32;
                    for (int j = 0; j < 32; ++j) split_f16(blk[j], tmp[j], tmp[32 + j]);
                    memcpy(blk, tmp, 128);
                }
            };
            if (n_thr <= 1) { work(0, n_blk); return; }
            std::vector<std::thread> pool;
            const size_t per = (n_blk + n_thr - 1) / n_thr;
            for (size_t t = 0; t < n_thr; ++t) pool.emplace_back(work, std::min(n_blk, t * per), std::min(n_blk, (t + 1) * per));
            for (auto& th : pool) th.join();
        };
        to_split(L.out_w, (size_t)c.outdim, kDim);
        // (L.cut_w stays fp32: the bound head runs on the exact-f32 kernel in both modes - see some_forward, "heads")
        for (const BlockOff& b : L.blocks) {
            for (int f = 0; f < 2; ++f) { to_split(b.ffn_w1[f], kFfn, kDim); to_split(b.ffn_w2[f], kDim, kFfn); }
            to_split(b.wqkv, 3 * kDim, kDim);
            to_split(b.wo, kDim, kDim);
            to_split(b.pw1_w, 2 * kDim, kDim);
            to_split(b.pw2_w, kDim, kDim);
        }
        for (size_t off : L.glu_w) to_split(off, 2 * kDim, kDim);
    }
    if (!missing.empty()) return fail(h, SOME_EKEY, "Missing key(s) in state_dict:" + missing);
    std::string unexpected;
    for (auto& kv : have)
        if (!used.count(kv.first) && unexpected.size() < 400) unexpected += " " + kv.first;
    if (!unexpected.empty()) return fail(h, SOME_EKEY, "Unexpected key(s) in state_dict:" + unexpected);
    if (!bad_shape.empty()) return fail(h, SOME_ESHAPE, "size mismatch for:" + bad_shape);
    return SOME_OK;
}

int some_attach_arena(SomeHandle* h, const float* arena_dev, size_t bytes) {
    if (!h) return SOME_EINVAL;
    if (!arena_dev || bytes < some_arena_bytes(h)) return fail(h, SOME_EINVAL, "some_attach_arena: buffer too small");
    if (reinterpret_cast<uintptr_t>(arena_dev) & 255) return fail(h, SOME_EINVAL, "some_attach_arena: arena must be 256-byte aligned");
    h->arena = arena_dev;
    return SOME_OK;
}

int some_logmel(SomeHandle* h, const float* audio_dev, const int64_t* sample_offsets_dev,
                const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, int32_t pad_mode, float* units_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || max_frames < 0) return fail(h, SOME_EINVAL, "some_logmel: negative size");
    if (B == 0 || max_frames == 0) return SOME_OK;
    if (!audio_dev || !sample_offsets_dev || !frame_offsets_dev || !units_dev) return fail(h, SOME_EINVAL, "some_logmel: null pointer");
    if (B > 65535) return fail(h, SOME_EINVAL, "some_logmel: B > 65535");
    if (pad_mode != SOME_PAD_ZERO && pad_mode != SOME_PAD_REFLECT) return fail(h, SOME_EINVAL, "some_logmel: bad pad_mode");
    int rc = ensure_mel_tables(h);
    if (rc != SOME_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "logmel", 0.0, 0.0);
    HIP_TRY(h, launch_logmel(h->mel, audio_dev, sample_offsets_dev, frame_offsets_dev, B, max_frames, pad_mode, units_dev, s));
    return SOME_OK;
}

int some_logmel_shifted(SomeHandle* h, const float* audio_dev, const int64_t* sample_offsets_dev,
                        const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, int32_t n_fft_new,
                        int32_t win_length_new, int32_t hop_length_new, int32_t center, int32_t rescale, float* units_dev,
                        void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || max_frames < 0) return fail(h, SOME_EINVAL, "some_logmel_shifted: negative size");
    if (n_fft_new < 2 || n_fft_new > kMaxShiftFft) return fail(h, SOME_EINVAL, "some_logmel_shifted: n_fft_new must be in [2, 4096] (key shifts up to +12 semitones)");
    if (win_length_new < 1 || win_length_new > n_fft_new) return fail(h, SOME_EINVAL, "some_logmel_shifted: win_length_new must be in [1, n_fft_new]");
    if (hop_length_new < 1) return fail(h, SOME_EINVAL, "some_logmel_shifted: hop_length_new must be positive");
    if (B == 0 || max_frames == 0) return SOME_OK;
    if (!audio_dev || !sample_offsets_dev || !frame_offsets_dev || !units_dev) return fail(h, SOME_EINVAL, "some_logmel_shifted: null pointer");
    if (B > 65535) return fail(h, SOME_EINVAL, "some_logmel_shifted: B > 65535");
    int rc = ensure_mel_tables(h);
    if (rc != SOME_OK) return rc;
    SomeHandle::ShiftTables tab{};
    {
        std::lock_guard<std::mutex> lock(h->shift_mu);
        bool found = false;
        for (auto& t : h->shift_tables)
            if (t.n_fft == n_fft_new && t.win == win_length_new) { tab = t; found = true; break; }
        if (!found) {
            // torch.hann_window(win') (periodic), centred in the n_fft' frame as torch.stft does (spec.py:44-46, 52-60)
            const size_t o_tw = 0, o_win = (size_t)n_fft_new * 16, total = o_win + (size_t)n_fft_new * 4;
            std::vector<char> host(total, 0);
            double* tw = reinterpret_cast<double*>(host.data() + o_tw);
            float* win = reinterpret_cast<float*>(host.data() + o_win);
            for (int k = 0; k < n_fft_new; ++k) {
                const double ang = 2.0 * M_PI * k / n_fft_new;
                tw[2 * k] = std::cos(ang);
                tw[2 * k + 1] = -std::sin(ang);
            }
            const int left = (n_fft_new - win_length_new) / 2;
            for (int n = 0; n < win_length_new; ++n) win[left + n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / win_length_new));
            void* dev = nullptr;
            HIP_TRY(h, hipMalloc(&dev, total));
            hipError_t e = hipMemcpy(dev, host.data(), total, hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(dev); return fail_hip(h, e, "hipMemcpy(shifted mel tables)"); }
            tab = {n_fft_new, win_length_new, dev, reinterpret_cast<float*>(static_cast<char*>(dev) + o_win),
                   reinterpret_cast<double*>(static_cast<char*>(dev) + o_tw)};
            h->shift_tables.push_back(tab);
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "logmel_shifted", 0.0, 0.0);
    // spec.py:47-50: pad_left = win' // 2 zeros when center; spec.py:68: magnitude * win / win'
    HIP_TRY(h, launch_logmel_shift(h->mel, tab.window, tab.twiddle, audio_dev, sample_offsets_dev, frame_offsets_dev, B, max_frames,
                                   n_fft_new, hop_length_new, center ? win_length_new / 2 : 0, rescale, (float)kWin,
                                   (float)win_length_new, units_dev, s));
    return SOME_OK;
}

int some_slicer_rms(SomeHandle* h, const void* audio_dev, int32_t sample_format, const int64_t* sample_offsets_dev,
                    const int64_t* rms_offsets_dev, int32_t B, int64_t max_rms_frames, int32_t frame_length,
                    int32_t hop_length, float* rms_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || max_rms_frames < 0) return fail(h, SOME_EINVAL, "some_slicer_rms: negative size");
    if (frame_length <= 0 || hop_length <= 0) return fail(h, SOME_EINVAL, "some_slicer_rms: frame_length and hop_length must be positive");
    if (frame_length > (1 << 19)) return fail(h, SOME_EINVAL, "some_slicer_rms: frame_length > 524288");
    if (sample_format != SOME_SAMPLE_F32 && sample_format != SOME_SAMPLE_PCM16) return fail(h, SOME_EINVAL, "some_slicer_rms: bad sample_format");
    if (B == 0 || max_rms_frames == 0) return SOME_OK;
    if (B > 65535) return fail(h, SOME_EINVAL, "some_slicer_rms: B > 65535");
    if (!audio_dev || !sample_offsets_dev || !rms_offsets_dev || !rms_dev) return fail(h, SOME_EINVAL, "some_slicer_rms: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "slicer_rms", 0.0, 0.0);
    HIP_TRY(h, launch_slicer_rms(audio_dev, sample_format == SOME_SAMPLE_PCM16, sample_offsets_dev, rms_offsets_dev, B,
                                 max_rms_frames, frame_length, hop_length, rms_dev, s));
    return SOME_OK;
}

int some_pcm_gather(SomeHandle* h, const void* src_dev, int32_t sample_format, const int64_t* src_offsets_dev,
                    const int64_t* dst_offsets_dev, int32_t B, int64_t max_len, float* audio_out_dev, void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || max_len < 0) return fail(h, SOME_EINVAL, "some_pcm_gather: negative size");
    if (sample_format != SOME_SAMPLE_F32 && sample_format != SOME_SAMPLE_PCM16) return fail(h, SOME_EINVAL, "some_pcm_gather: bad sample_format");
    if (B == 0 || max_len == 0) return SOME_OK;
    if (B > 65535) return fail(h, SOME_EINVAL, "some_pcm_gather: B > 65535");
    if (!src_dev || !src_offsets_dev || !dst_offsets_dev || !audio_out_dev) return fail(h, SOME_EINVAL, "some_pcm_gather: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "pcm_gather", 0.0, 0.0);
    HIP_TRY(h, launch_pcm_gather(src_dev, sample_format == SOME_SAMPLE_PCM16, src_offsets_dev, dst_offsets_dev, B, max_len,
                                 audio_out_dev, s));
    return SOME_OK;
}

size_t some_workspace_bytes(const SomeHandle* h, int64_t total_frames, int32_t B) {
    if (!h || total_frames <= 0) return 0;
    return ws_layout(h->precision, total_frames, B < 0 ? 0 : B).total;
}

int some_forward(SomeHandle* h, const float* units_dev, const int32_t* frame_offsets_dev, int32_t B,
                 int64_t total_frames, int32_t max_frames, const uint8_t* row_mask_dev, int32_t head_mode,
                 float* midi_dev, float* bound_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    if (!h->arena) return fail(h, SOME_ESTATE, "some_forward: no weights attached (call some_attach_arena first)");
    if (B < 0 || total_frames < 0) return fail(h, SOME_EINVAL, "some_forward: negative size");
    if (B == 0 || total_frames == 0) return SOME_OK;
    // the widest activation row is 8 KiB (FFN hidden) and operands are addressed through 2 GiB buffer descriptors
    if (total_frames > 262143) return fail(h, SOME_EINVAL, "some_forward: total_frames too large (at most 262143 frames = 50 minutes of audio per call; split the batch)");
    if (!units_dev || !frame_offsets_dev || !midi_dev || !bound_dev || !workspace_dev) return fail(h, SOME_EINVAL, "some_forward: null pointer");
    if (head_mode < 0 || head_mode > 2) return fail(h, SOME_EINVAL, "some_forward: bad head_mode");
    if (workspace_bytes < some_workspace_bytes(h, total_frames, B)) return fail(h, SOME_ENOMEM, "some_forward: workspace too small");
    if (reinterpret_cast<uintptr_t>(workspace_dev) & 255) return fail(h, SOME_EINVAL, "some_forward: workspace must be 256-byte aligned");
    if ((size_t)B * kHeads * kStreams > 0x7fffffffu / 64) return fail(h, SOME_EINVAL, "some_forward: B too large");
    // the attention operands live in clip-aligned rows (every clip padded to a multiple of 16 rows) behind 2 GiB buffer descriptors of 2 KiB rows
    if (h->precision == SOME_PRECISION_F16X3 && attn_rows_cover(total_frames, B) > 1048575)
        return fail(h, SOME_EINVAL, "some_forward: too many short clips in one call (total_frames + 15 * B must stay below 1048450; split the batch)");

    const SomeConfig& c = h->cfg;
    const ArenaLayout& L = h->lay;
    const float* W = h->arena;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = (int)total_frames;
    const size_t m = (size_t)M;
    const WsLayout wl = ws_layout(h->precision, total_frames, B);
    const size_t per_stream = wl.per_stream;
    float* ws = static_cast<float*>(workspace_dev);
    int32_t* pad_off = reinterpret_cast<int32_t*>(static_cast<char*>(workspace_dev) + wl.plan_off);
    int32_t* row_map = reinterpret_cast<int32_t*>(static_cast<char*>(workspace_dev) + wl.map_off);
    const int Mc = (int)wl.Mc;
    float *X[2], *H[2], *U[2], *G[2];
    for (int g = 0; g < kStreams; ++g) {
        float* base = ws + g * per_stream;
        X[g] = base; H[g] = X[g] + m * kDim; G[g] = H[g] + m * kDim; U[g] = G[g] + m * kDim;
    }
    const double Md = (double)M;
    double sumT2 = 0.0;
    if (h->profiling) {   // measurement only: exact attention FLOPs need the per-clip lengths
        std::vector<int32_t> off((size_t)B + 1);
        HIP_TRY(h, hipMemcpy(off.data(), frame_offsets_dev, off.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) { const double t = off[b + 1] - off[b]; sumT2 += t * t; }
    }

    const bool f16x3 = h->precision == SOME_PRECISION_F16X3;
    auto pick_tile = [&](int n_max, int ng) -> int {
        if (h->tile >= 0) return h->tile;
        auto blocks = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((n_max + bn - 1) / bn) * ng; };
        if (blocks(256, 256) >= 512) return 2;       // >= 2 waves of workgroups over 256 CUs
        if (blocks(256, 128) >= 512) return 1;
        if (blocks(128, 128) >= 256) return 0;       // 128 x 128 tiles run two per CU (measured: 2 x 30 s clips 7.7 vs 8.5 ms)
        return 4;                                    // 64 x 128: single-clip latency regime (1 x 30 s: 5.07 vs 5.13 ms)
    };
    // Every launch covers the model streams [g0, g0 + ng) as blockIdx.y groups: both (grouped, one HIP stream), or one
    // model stream per HIP stream (dual-stream mode below).
    auto gemm = [&](const char* name, GemmEpi epi, GemmArgs& a, int ng, int n_out, hipStream_t st, bool out_split = false, int rows = -1) -> int {
        a.groups = ng; a.M = rows < 0 ? M : rows; a.flags = h->gemm_flags;
        const double flops = 2.0 * Md * a.K * n_out * ng;
        Scope sc(h, st, name, flops, 0.0);
        hipError_t e = f16x3 ? launch_gemm_f16x3(epi, a, out_split, pick_tile(n_out, ng), st) : launch_gemm(epi, a, st);
        if (e != hipSuccess) return fail_hip(h, e, name);
        return SOME_OK;
    };
    // f32 mode: dst gets fp32.  f16x3 mode: dst gets SPLIT32 (GEMM operand) and, if dst32 != null, an fp32 copy too.
    auto ln = [&](int layer, int idx, int g0, int ng, hipStream_t st, float* const* src, float* const* dst, float* const* dst32 = nullptr) -> int {
        LnArgs a{};
        for (int gi = 0; gi < ng; ++gi) {
            const int g = g0 + gi;
            const BlockOff& b = L.blocks[(size_t)layer * 2 + g];
            a.x[gi] = src[g]; a.gamma[gi] = W + b.ln_g[idx]; a.beta[gi] = W + b.ln_b[idx];
            if (f16x3) { a.ys[gi] = dst[g]; a.y[gi] = dst32 ? dst32[g] : nullptr; }
            else { a.y[gi] = dst[g]; a.ys[gi] = nullptr; }
        }
        a.groups = ng; a.M = M;
        Scope sc(h, st, "layernorm", 0.0, 2.0 * ng * Md * kDim * 4);
        hipError_t e = launch_layernorm(a, st);
        if (e != hipSuccess) return fail_hip(h, e, "layernorm");
        return SOME_OK;
    };
    auto ffn = [&](int layer, int f, int g0, int ng, hipStream_t st) -> int {
        GemmArgs a{};
        for (int gi = 0; gi < ng; ++gi) {
            const int g = g0 + gi;
            const BlockOff& b = L.blocks[(size_t)layer * 2 + g];
            a.g[gi] = GemmGroup{H[g], W + b.ffn_w1[f], W + b.ffn_b1[f], nullptr, U[g], nullptr, kFfn, 0};
        }
        a.K = kDim; a.lda = kDim; a.ldc = kFfn;
        int rc = gemm("gemm_bias_silu[512->2048]", EPI_BIAS_SILU, a, ng, kFfn, st, /*out_split=*/f16x3);
        if (rc) return rc;
        GemmArgs d{};
        for (int gi = 0; gi < ng; ++gi) {
            const int g = g0 + gi;
            const BlockOff& b = L.blocks[(size_t)layer * 2 + g];
            d.g[gi] = GemmGroup{U[g], W + b.ffn_w2[f], W + b.ffn_b2[f], X[g], X[g], nullptr, kDim, 0};
        }
        d.K = kFfn; d.lda = kFfn; d.ldc = kDim; d.ldr = kDim; d.alpha = 0.5f;
        return gemm("gemm_bias_res[2048->512]", EPI_BIAS_RES, d, ng, kDim, st);
    };
    // conform_blocke.forward (Gconform.py:56-63) for the model streams [g0, g0 + ng) of `layer`, enqueued on `st`
    auto run_block = [&](int layer, int g0, int ng, hipStream_t st) -> int {
        int rc;
        if ((rc = ln(layer, 0, g0, ng, st, X, H))) return rc;
        if ((rc = ffn(layer, 0, g0, ng, st))) return rc;
        if ((rc = ln(layer, 1, g0, ng, st, X, H))) return rc;
        if (f16x3) {
            // QKV projection writes Q | K as SPLIT32 planes and V transposed, all in clip-aligned rows (row gather through the
            // attention plan: a clip's key tiles start at its own first frame); split-f16 attention consumes them
            const int ldv = vt_ld(Mc);
            GemmArgs a{};
            Attn3Args t{};
            for (int gi = 0; gi < ng; ++gi) {
                const int g = g0 + gi;
                float* qp = U[g];
                float* kp = U[g] + (size_t)Mc * kDim;
                void* vt = U[g] + 2 * (size_t)Mc * kDim;
                a.g[gi] = GemmGroup{H[g], W + L.blocks[(size_t)layer * 2 + g].wqkv, nullptr, nullptr, qp, nullptr, 3 * kDim, 0, kp, vt, ldv};
                t.q[gi] = qp; t.k[gi] = kp; t.vt[gi] = vt; t.out[gi] = H[g];
            }
            a.K = kDim; a.lda = kDim; a.ldc = kDim; a.row_map = row_map; a.a_rows = M;
            if ((rc = gemm("gemm[512->1536 qkv]", EPI_QKV, a, ng, 3 * kDim, st, false, Mc))) return rc;
            t.frame_offsets = frame_offsets_dev; t.pad_offsets = pad_off; t.groups = ng; t.B = B; t.max_frames = max_frames; t.M = Mc; t.ldv = ldv;
            t.fast = h->attn_fast;
            Scope sc(h, st, "attention", 4.0 * kHeadDim * kHeads * ng * sumT2, 0.0);
            HIP_TRY(h, launch_attention_f16x3(t, st));
        } else {
            GemmArgs a{};
            AttnArgs t{};
            for (int gi = 0; gi < ng; ++gi) {
                const int g = g0 + gi;
                a.g[gi] = GemmGroup{H[g], W + L.blocks[(size_t)layer * 2 + g].wqkv, nullptr, nullptr, U[g], nullptr, 3 * kDim, 0};
                t.qkv[gi] = U[g]; t.out[gi] = H[g];
            }
            a.K = kDim; a.lda = kDim; a.ldc = 3 * kDim;
            if ((rc = gemm("gemm[512->1536 qkv]", EPI_NONE, a, ng, 3 * kDim, st))) return rc;
            t.frame_offsets = frame_offsets_dev; t.groups = ng; t.B = B; t.max_frames = max_frames;
            t.out_split = 0;
            Scope sc(h, st, "attention", 4.0 * kHeadDim * kHeads * ng * sumT2, 0.0);
            HIP_TRY(h, launch_attention(t, st));
        }
        {
            GemmArgs a{};
            for (int gi = 0; gi < ng; ++gi) {
                const int g = g0 + gi;
                const BlockOff& b = L.blocks[(size_t)layer * 2 + g];
                a.g[gi] = GemmGroup{H[g], W + b.wo, W + b.bo, X[g], X[g], nullptr, kDim, 0};
            }
            a.K = kDim; a.lda = kDim; a.ldc = kDim; a.ldr = kDim; a.alpha = 1.0f;
            if ((rc = gemm("gemm_bias_res[512->512]", EPI_BIAS_RES, a, ng, kDim, st))) return rc;
        }
        if ((rc = ln(layer, 2, g0, ng, st, X, H))) return rc;
        {
            GemmArgs a{};
            for (int gi = 0; gi < ng; ++gi) {
                const int g = g0 + gi;
                const BlockOff& b = L.blocks[(size_t)layer * 2 + g];
                a.g[gi] = GemmGroup{H[g], W + b.pw1_w, W + b.pw1_b, nullptr, G[g], nullptr, 2 * kDim, 0};
            }
            a.K = kDim; a.lda = kDim; a.ldc = kDim;
            if ((rc = gemm("gemm_glu[512->2x512]", EPI_GLU, a, ng, 2 * kDim, st))) return rc;
        }
        {
            DwArgs a{};
            for (int gi = 0; gi < ng; ++gi) {
                const int g = g0 + gi;
                const BlockOff& b = L.blocks[(size_t)layer * 2 + g];
                a.x[gi] = G[g]; a.y[gi] = H[g]; a.w[gi] = W + b.dw_w; a.b[gi] = W + b.dw_b;
            }
            a.frame_offsets = frame_offsets_dev; a.groups = ng; a.B = B; a.max_frames = max_frames;
            a.out_split = f16x3 ? 1 : 0;
            Scope sc(h, st, "dwconv_bn_silu", 0.0, 2.0 * ng * Md * kDim * 4);
            HIP_TRY(h, launch_dwconv(a, st));
        }
        {
            GemmArgs a{};
            for (int gi = 0; gi < ng; ++gi) {
                const int g = g0 + gi;
                const BlockOff& b = L.blocks[(size_t)layer * 2 + g];
                a.g[gi] = GemmGroup{H[g], W + b.pw2_w, W + b.pw2_b, X[g], X[g], nullptr, kDim, 0};
            }
            a.K = kDim; a.lda = kDim; a.ldc = kDim; a.ldr = kDim; a.alpha = 1.0f;
            if ((rc = gemm("gemm_bias_res[512->512]", EPI_BIAS_RES, a, ng, kDim, st))) return rc;
        }
        if ((rc = ln(layer, 3, g0, ng, st, X, H))) return rc;
        if ((rc = ffn(layer, 1, g0, ng, st))) return rc;
        // block output y: f32 mode keeps it in H; f16x3 mode needs it twice - SPLIT32 in H as the next GEMM's
        // operand and fp32 in X (in place) as the exact residual of the cross gate
        return ln(layer, 4, g0, ng, st, X, H, X);
    };

    // Dual-stream mode: the midi and the bound stream of a layer are independent until the cross gate
    // (Gconform.py:82-87), so each runs on its own HIP stream (fork / join with events around every layer).  While one
    // chain sits in an HBM-bound kernel (LayerNorm, depthwise conv, a GEMM's epilogue tail) the other's MFMA-bound GEMM
    // or attention shares the CUs - LayerNorm waves fit beside a GEMM workgroup (no LDS, ~30 VGPRs).  Kernel-level
    // HIP-event profiling needs serial execution: it uses the grouped single-stream path.
    bool dual = h->dual_stream && !h->profiling;
    std::unique_lock<std::mutex> fwd_lock(h->fwd_mu, std::defer_lock);      // the helper stream and its events are per handle
    if (dual) fwd_lock.lock();
    SomeHandle::AuxSet* aux = nullptr;
    if (dual) {
        for (auto& a : h->aux_sets) if (a.caller == s) aux = &a;
        if (!aux) {     // first use on this stream: the helper stream cannot be created while the caller's stream is being captured
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
                dual = false;
            } else {
                SomeHandle::AuxSet a{s, nullptr, nullptr, nullptr};
                HIP_TRY(h, hipStreamCreateWithFlags(&a.aux, hipStreamNonBlocking));
                HIP_TRY(h, hipEventCreateWithFlags(&a.fork, hipEventDisableTiming));
                HIP_TRY(h, hipEventCreateWithFlags(&a.join, hipEventDisableTiming));
                h->aux_sets.push_back(a);
                aux = &h->aux_sets.back();
            }
        }
    }
    hipStream_t s2 = dual ? aux->aux : s;

    int rc;
    if (f16x3) {    // clip-aligned attention coordinates, once per call (18 attention launches read them)
        Scope sc(h, s, "attn_plan", 0.0, 0.0);
        HIP_TRY(h, launch_attn_plan(frame_offsets_dev, B, Mc, pad_off, row_map, s));
    }
    {   // Gconform.py:124-127: the two input projections (+ masked_fill on the midi stream)
        GemmArgs a{};
        for (int g = 0; g < kStreams; ++g)
            a.g[g] = GemmGroup{units_dev, W + L.in_w[g], W + L.in_b[g], nullptr, X[g], g == 0 ? row_mask_dev : nullptr, kDim, 0};
        a.K = c.indim; a.lda = c.indim; a.ldc = kDim; a.groups = kStreams; a.M = M;
        {   // K = units_dim (80) is not a multiple of 32 and the operand is raw fp32: always the exact-f32 kernel
            Scope sc(h, s, "gemm_bias[in->512]", 2.0 * Md * c.indim * kStreams * kDim, 0.0);
            HIP_TRY(h, launch_gemm(EPI_BIAS, a, s));
        }
    }
    for (int layer = 0; layer <= c.lay; ++layer) {
        if (dual) {
            HIP_TRY(h, hipEventRecord(aux->fork, s));
            HIP_TRY(h, hipStreamWaitEvent(s2, aux->fork, 0));
            rc = run_block(layer, 0, 1, s);
            const int rc2 = rc ? 0 : run_block(layer, 1, 1, s2);
            // join ALWAYS, also on an error: a fork left open would stay in a stream capture of the caller's (and the helper
            // stream would run on behind the failed call)
            const hipError_t e1 = hipEventRecord(aux->join, s2);
            const hipError_t e2 = hipStreamWaitEvent(s, aux->join, 0);
            if (rc) return rc;
            if (rc2) return rc2;
            HIP_TRY(h, e1);
            HIP_TRY(h, e2);
        } else if ((rc = run_block(layer, 0, kStreams, s))) {
            return rc;
        }
        if (layer < c.lay) {
            // Gcf.forward (Gconform.py:82-87): midi' = y0 + GLU(glu2(y1)), bound' = y1 + GLU(glu1(y0));
            // then masked_fill on the midi stream (Gconform.py:131-132)
            GemmArgs a{};
            const float* res0 = f16x3 ? X[0] : H[0];
            const float* res1 = f16x3 ? X[1] : H[1];
            a.g[0] = GemmGroup{H[1], W + L.glu_w[(size_t)layer * 2 + 1], W + L.glu_b[(size_t)layer * 2 + 1], res0, X[0], row_mask_dev, 2 * kDim, 0};
            a.g[1] = GemmGroup{H[0], W + L.glu_w[(size_t)layer * 2 + 0], W + L.glu_b[(size_t)layer * 2 + 0], res1, X[1], nullptr, 2 * kDim, 0};
            a.K = kDim; a.lda = kDim; a.ldc = kDim; a.ldr = kDim;
            if ((rc = gemm("gemm_glu_res[512->2x512 gate]", EPI_GLU_RES, a, kStreams, 2 * kDim, s))) return rc;
        }
    }
    {   // heads (Gconform.py:135-138, Gmidi_conform.py:33-37)
        GemmArgs a{};
        a.g[0] = GemmGroup{H[0], W + L.out_w, W + L.out_b, nullptr, midi_dev, nullptr, c.outdim, head_mode == SOME_HEAD_SIGMOID ? 1 : 0};
        a.g[1] = GemmGroup{H[1], W + L.cut_w, W + L.cut_b, nullptr, bound_dev, nullptr, 1, 1};
        a.K = kDim; a.lda = kDim; a.ldc = 0;   // per-group ldc below
        // the two heads have different widths: run them as two launches of one group each
        GemmArgs a0 = a; a0.g[0] = a.g[0]; a0.groups = 1; a0.M = M; a0.ldc = c.outdim;
        {
            Scope sc(h, s, "gemm_bias[512->outdim]", 2.0 * Md * kDim * c.outdim, 0.0);
            HIP_TRY(h, f16x3 ? launch_gemm_f16x3(EPI_BIAS, a0, false, 0, s) : launch_gemm(EPI_BIAS, a0, s));
        }
        // The bound head is ONE output column whose per-frame values are summed over the whole clip by the decoder
        // (decode_bounds_to_alignment, utils/infer_utils.py:27-39).  The split-f16 product drops a_lo * w_lo; for a single fixed weight
        // column and LayerNorm outputs whose channels keep their rough value from frame to frame that term is not noise but a CONSTANT of
        // ~2e-7 on the logit - +5e-8 on every bound probability, 1.4e-4 .. 4.9e-4 on the 2584-frame cumsum, and 16 of 50 724 note
        // boundaries moved at 32 x 30 s where the exact-f32 mode moves none (round 5: profiles/r05_experiments.md, knock-out table).
        // So this head always takes the exact-f32 kernel on LayerNorm 5's fp32 copy (X) - 0.17 GFLOP of 27 300.
        GemmArgs a1 = a; a1.g[0] = a.g[1]; a1.groups = 1; a1.M = M; a1.ldc = 1;
        if (f16x3) a1.g[0].A = X[1];
        {
            Scope sc(h, s, "gemm_bias[512->1]", 2.0 * Md * kDim, 0.0);
            HIP_TRY(h, launch_gemm(EPI_BIAS, a1, s));
        }
        if (head_mode == SOME_HEAD_SOFTMAX) {
            Scope sc(h, s, "row_softmax", 0.0, 2.0 * Md * c.outdim * 4);
            HIP_TRY(h, launch_row_softmax(midi_dev, M, c.outdim, s));
        }
    }
    return SOME_OK;
}

size_t some_decode_scratch_bytes(const SomeHandle* h, int64_t total_frames) {
    if (!h || total_frames <= 0) return 0;
    return decode_scratch_bytes(total_frames);
}

int some_decode(SomeHandle* h, const float* probs_dev, const float* bounds_dev, const uint8_t* row_mask_dev,
                const int32_t* frame_offsets_dev, int32_t B, int64_t total_frames, int32_t quantized,
                float* note_midi_dev, int64_t* note_dur_dev, uint8_t* note_rest_dev, int32_t* n_notes_dev,
                int64_t* frame2item_dev, float* values_dev, uint8_t* rest_dev,
                void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || total_frames < 0) return fail(h, SOME_EINVAL, "some_decode: negative size");
    if (B == 0 || total_frames == 0) return SOME_OK;
    if (!probs_dev || !bounds_dev || !frame_offsets_dev || !note_midi_dev || !note_dur_dev || !note_rest_dev ||
        !n_notes_dev || !scratch_dev)
        return fail(h, SOME_EINVAL, "some_decode: null pointer");
    if (scratch_bytes < decode_scratch_bytes(total_frames)) return fail(h, SOME_ENOMEM, "some_decode: scratch too small");
    if (quantized && h->cfg.outdim != 129) return fail(h, SOME_EINVAL, "some_decode: quantized decode expects 129 bins (rest = 128)");
    DecodeArgs a{};
    a.probs = probs_dev; a.bounds = bounds_dev; a.mask = row_mask_dev; a.frame_offsets = frame_offsets_dev;
    a.B = B; a.total_frames = total_frames; a.nbins = h->cfg.outdim; a.quantized = quantized ? 1 : 0;
    a.vmin = h->cfg.midi_min; a.vmax = h->cfg.midi_max; a.deviation = h->cfg.midi_deviation; a.threshold = h->cfg.rest_threshold;
    a.note_midi = note_midi_dev; a.note_dur = note_dur_dev; a.note_rest = note_rest_dev; a.n_notes = n_notes_dev;
    a.frame2item = frame2item_dev; a.values = values_dev; a.rest = rest_dev; a.scratch = scratch_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "decode", 0.0, (double)total_frames * (4.0 * (h->cfg.outdim + 1) + 13.0));
    HIP_TRY(h, launch_decode(a, s));
    return SOME_OK;
}

int some_op_gemm(SomeHandle* h, int32_t epilogue, const float* A_dev, int32_t lda, const float* W_dev,
                 const float* bias_dev, const float* res_dev, int32_t ldr, float* C_dev, int32_t ldc,
                 int32_t M, int32_t N, int32_t K, float alpha, int32_t act, const uint8_t* row_mask_dev, int32_t flags,
                 void* stream) {
    if (!h) return SOME_EINVAL;
    if (epilogue < 0 || epilogue > 5 || M < 0 || N <= 0 || K <= 0 || !A_dev || !W_dev || !C_dev)
        return fail(h, SOME_EINVAL, "some_op_gemm: bad argument");
    if (epilogue != EPI_NONE && !bias_dev) return fail(h, SOME_EINVAL, "some_op_gemm: epilogue needs a bias");
    if ((epilogue == EPI_BIAS_RES || epilogue == EPI_GLU_RES) && !res_dev) return fail(h, SOME_EINVAL, "some_op_gemm: epilogue needs a residual");
    if ((epilogue == EPI_GLU || epilogue == EPI_GLU_RES) && (N % 64)) return fail(h, SOME_EINVAL, "some_op_gemm: GLU epilogue needs N % 64 == 0");
    GemmArgs a{};
    a.g[0] = GemmGroup{A_dev, W_dev, bias_dev, res_dev, C_dev, row_mask_dev, N, act};
    a.groups = 1; a.M = M; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr; a.alpha = alpha; a.flags = h->gemm_flags;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "op_gemm", 2.0 * M * (double)N * K, 0.0);
    if (flags & SOME_GEMM_SPLIT_IN) {
        if ((K & 31) || (lda & 31)) return fail(h, SOME_EINVAL, "some_op_gemm: SPLIT32 operands need K % 32 == 0 and lda % 32 == 0");
        if (flags & SOME_GEMM_HI_ONLY) {
            if (epilogue != EPI_NONE && epilogue != EPI_BIAS) return fail(h, SOME_EINVAL, "some_op_gemm: HI_ONLY supports EPI_NONE / EPI_BIAS");
            HIP_TRY(h, launch_gemm_f16x1(static_cast<GemmEpi>(epilogue), a, (flags >> 8) & 7, s, (flags & SOME_GEMM_HI_BF16) != 0));
        } else {
            HIP_TRY(h, launch_gemm_f16x3(static_cast<GemmEpi>(epilogue), a, (flags & SOME_GEMM_SPLIT_OUT) != 0, (flags >> 8) & 7, s));
        }
    } else {
        HIP_TRY(h, launch_gemm(static_cast<GemmEpi>(epilogue), a, s));
    }
    return SOME_OK;
}

int some_op_split_rows(SomeHandle* h, const float* x_dev, float* out_dev, int64_t rows, int32_t K, void* stream) {
    return some_op_split_rows_fmt(h, x_dev, out_dev, rows, K, SOME_OPERAND_F16X2, stream);
}

int some_op_split_rows_fmt(SomeHandle* h, const float* x_dev, float* out_dev, int64_t rows, int32_t K, int32_t format, void* stream) {
    if (!h) return SOME_EINVAL;
    if (rows < 0 || K <= 0 || (K & 31) || !x_dev || !out_dev) return fail(h, SOME_EINVAL, "some_op_split_rows: bad argument (K % 32 == 0)");
    if (format != SOME_OPERAND_F16X2 && format != SOME_OPERAND_BF16) return fail(h, SOME_EINVAL, "some_op_split_rows_fmt: bad format");
    HIP_TRY(h, launch_split_rows(x_dev, out_dev, rows, K, static_cast<hipStream_t>(stream), format == SOME_OPERAND_BF16));
    return SOME_OK;
}

int some_op_layernorm(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                      float* y_dev, float* y_split_dev, int32_t M, void* stream) {
    if (!h) return SOME_EINVAL;
    if (M < 0 || !x_dev || !gamma_dev || !beta_dev || (!y_dev && !y_split_dev)) return fail(h, SOME_EINVAL, "some_op_layernorm: bad argument");
    LnArgs a{};
    a.x[0] = x_dev; a.y[0] = y_dev; a.ys[0] = y_split_dev; a.gamma[0] = gamma_dev; a.beta[0] = beta_dev; a.groups = 1; a.M = M;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "op_layernorm", 0.0, 2.0 * M * kDim * 4.0);
    HIP_TRY(h, launch_layernorm(a, s));
    return SOME_OK;
}

int some_op_attention(SomeHandle* h, const float* qkv_dev, const int32_t* frame_offsets_dev, int32_t B,
                      int32_t max_frames, float* out_dev, int32_t out_split, void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || max_frames < 0 || !qkv_dev || !frame_offsets_dev || !out_dev) return fail(h, SOME_EINVAL, "some_op_attention: bad argument");
    AttnArgs a{};
    a.qkv[0] = qkv_dev; a.out[0] = out_dev; a.frame_offsets = frame_offsets_dev; a.groups = 1; a.B = B; a.max_frames = max_frames;
    a.out_split = out_split ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "op_attention", 0.0, 0.0);
    HIP_TRY(h, launch_attention(a, s));
    return SOME_OK;
}

size_t some_op_qkv_attention_f16x3_bytes(int32_t M, int32_t B) {
    if (M <= 0 || B < 0) return 0;
    const int64_t Mc = attn_rows_cover(M, B);
    return align_up((size_t)Mc * kDim * 4 * 2 + (size_t)2 * kDim * vt_ld(Mc) * 2, 256) + align_up(((size_t)B + 1) * 4, 256) +
           align_up((size_t)Mc * 4, 256);
}

int some_op_qkv_attention_f16x3(SomeHandle* h, const float* h_split_dev, const float* wqkv_split_dev,
                                const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, int32_t M,
                                float* out_split_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || M < 0 || !h_split_dev || !wqkv_split_dev || !frame_offsets_dev || !out_split_dev || !workspace_dev)
        return fail(h, SOME_EINVAL, "some_op_qkv_attention_f16x3: bad argument");
    if (B == 0 || M == 0) return SOME_OK;
    const int Mc = (int)attn_rows_cover(M, B);
    const int ldv = vt_ld(Mc);
    const size_t plan_off = align_up((size_t)Mc * kDim * 4 * 2 + (size_t)2 * kDim * ldv * 2, 256);
    const size_t map_off = plan_off + align_up(((size_t)B + 1) * 4, 256);
    if (workspace_bytes < some_op_qkv_attention_f16x3_bytes(M, B)) return fail(h, SOME_ENOMEM, "some_op_qkv_attention_f16x3: workspace too small");
    if (reinterpret_cast<uintptr_t>(workspace_dev) & 255) return fail(h, SOME_EINVAL, "some_op_qkv_attention_f16x3: workspace must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* qp = static_cast<float*>(workspace_dev);
    float* kp = qp + (size_t)Mc * kDim;
    void* vt = kp + (size_t)Mc * kDim;
    int32_t* pad_off = reinterpret_cast<int32_t*>(static_cast<char*>(workspace_dev) + plan_off);
    int32_t* row_map = reinterpret_cast<int32_t*>(static_cast<char*>(workspace_dev) + map_off);
    HIP_TRY(h, launch_attn_plan(frame_offsets_dev, B, Mc, pad_off, row_map, s));
    GemmArgs a{};
    a.g[0] = GemmGroup{h_split_dev, wqkv_split_dev, nullptr, nullptr, qp, nullptr, 3 * kDim, 0, kp, vt, ldv};
    a.groups = 1; a.M = Mc; a.K = kDim; a.lda = kDim; a.ldc = kDim; a.row_map = row_map; a.a_rows = M;
    HIP_TRY(h, launch_gemm_f16x3(EPI_QKV, a, false, h->tile >= 0 ? h->tile : 0, s));
    Attn3Args t{};
    t.q[0] = qp; t.k[0] = kp; t.vt[0] = vt; t.out[0] = out_split_dev;
    t.frame_offsets = frame_offsets_dev; t.pad_offsets = pad_off; t.groups = 1; t.B = B; t.max_frames = max_frames; t.M = Mc; t.ldv = ldv;
    t.fast = h->attn_fast;
    HIP_TRY(h, launch_attention_f16x3(t, s));
    return SOME_OK;
}

int some_op_dwconv_silu(SomeHandle* h, const float* x_dev, const float* taps_dev, const float* bias_dev,
                        const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, float* y_dev, int32_t out_split,
                        void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || max_frames < 0 || !x_dev || !taps_dev || !bias_dev || !frame_offsets_dev || !y_dev)
        return fail(h, SOME_EINVAL, "some_op_dwconv_silu: bad argument");
    DwArgs a{};
    a.x[0] = x_dev; a.y[0] = y_dev; a.w[0] = taps_dev; a.b[0] = bias_dev;
    a.frame_offsets = frame_offsets_dev; a.groups = 1; a.B = B; a.max_frames = max_frames;
    a.out_split = out_split ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(h, s, "op_dwconv_silu", 0.0, 0.0);
    HIP_TRY(h, launch_dwconv(a, s));
    return SOME_OK;
}

int some_decode_notes(SomeHandle* h, const int64_t* frame2item_dev, const float* values_dev, const uint8_t* masks_dev,
                      const int32_t* frame_offsets_dev, int32_t B, int64_t total_frames, int32_t max_frames,
                      int32_t values_are_integers, float* note_midi_dev, int64_t* note_dur_dev, uint8_t* note_rest_dev,
                      int32_t* n_notes_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!h) return SOME_EINVAL;
    if (B < 0 || total_frames < 0) return fail(h, SOME_EINVAL, "some_decode_notes: negative size");
    if (B == 0 || total_frames == 0) return SOME_OK;
    if (!frame2item_dev || !values_dev || !masks_dev || !frame_offsets_dev || !note_midi_dev || !note_dur_dev ||
        !note_rest_dev || !n_notes_dev || !scratch_dev)
        return fail(h, SOME_EINVAL, "some_decode_notes: null pointer");
    if (max_frames > 4096) return fail(h, SOME_EINVAL, "some_decode_notes: clips longer than 4096 frames are not supported by this entry point");
    if (scratch_bytes < decode_scratch_bytes(total_frames)) return fail(h, SOME_ENOMEM, "some_decode_notes: scratch too small");
    DecodeArgs a{};
    a.frame_offsets = frame_offsets_dev; a.B = B; a.total_frames = total_frames; a.nbins = h->cfg.outdim;
    a.quantized = values_are_integers ? 1 : 0;
    a.note_midi = note_midi_dev; a.note_dur = note_dur_dev; a.note_rest = note_rest_dev; a.n_notes = n_notes_dev;
    a.scratch = scratch_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the kernel's flag byte wants "rest" semantics (1 = frame does not count): pass ~masks through the scratch tail
    HIP_TRY(h, launch_decode_notes(a, frame2item_dev, values_dev, masks_dev, max_frames, s));
    return SOME_OK;
}

int some_profile_enable(SomeHandle* h, int32_t on) {
    if (!h) return SOME_EINVAL;
    h->profiling = on != 0;
    return SOME_OK;
}

int some_profile_collect(SomeHandle* h, SomeKernelStat* stats, int32_t max_stats, int32_t* n_stats) {
    if (!h || !stats || !n_stats || max_stats < 0) return SOME_EINVAL;
    std::vector<SomeKernelStat> acc;
    std::map<std::string, size_t> index;
    for (auto& r : h->prof) {
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) return fail_hip(h, e, "hipEventSynchronize");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.e0, r.e1);
        if (e != hipSuccess) return fail_hip(h, e, "hipEventElapsedTime");
        auto it = index.find(r.name);
        if (it == index.end()) {
            SomeKernelStat st{};
            snprintf(st.name, sizeof st.name, "%s", r.name.c_str());
            index[r.name] = acc.size();
            acc.push_back(st);
            it = index.find(r.name);
        }
        SomeKernelStat& st = acc[it->second];
        st.launches += 1; st.total_ms += ms; st.flops += r.flops; st.bytes += r.bytes;
        h->event_pool.push_back(r.e0);
        h->event_pool.push_back(r.e1);
    }
    h->prof.clear();
    const int32_t n = std::min<int32_t>((int32_t)acc.size(), max_stats);
    for (int32_t i = 0; i < n; ++i) stats[i] = acc[i];
    *n_stats = n;
    return SOME_OK;
}

}  // extern "C"
