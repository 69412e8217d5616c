// Internal declarations shared by the HIP translation units of libsome_amd.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/some_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipFuncSetAttribute is PER DEVICE: a process-wide `static bool` latch is wrong the day one process drives two GPUs (a server with two
// handles) and racy between threads.  One bit per device, set after the attributes are in place; two threads that race both set the
// (idempotent) attributes.
struct DeviceOnce {
    std::atomic<uint64_t> done{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool need() const { return !((done.load(std::memory_order_acquire) >> dev()) & 1ull); }
    void mark() { done.fetch_or(1ull << dev(), std::memory_order_release); }
};

// Compiled shapes (validated in some_create): the reference's configs all use them
// (configs/midi_conformer.yaml:22-33, configs/base.yaml:11-15).
constexpr int kDim = 512;
constexpr int kHeads = 8;
constexpr int kHeadDim = 64;
constexpr int kFfn = 2048;
constexpr int kConvK = 31;
constexpr int kWin = 2048;
constexpr int kHop = 512;
constexpr int kMels = 80;
constexpr int kStreams = 2;   // 0 = midi stream (att1 / x), 1 = bound stream (att2 / x1)

// ---- GEMM -----------------------------------------------------------------------------------------
enum GemmEpi { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_SILU = 2, EPI_BIAS_RES = 3, EPI_GLU = 4, EPI_GLU_RES = 5,
               EPI_QKV = 6 /* f16x3 only: Q | K planes in SPLIT32, V transposed (attention_f16x3.hip) */ };

struct GemmGroup {
    const float* A;        // [M, lda]
    const float* W;        // [N, K] row-major (K contiguous); GLU epilogues: rows interleaved 32 a / 32 gate
    const float* bias;     // [N] (same interleave) or nullptr
    const float* res;      // residual [M, ldr] or nullptr
    float* C;              // [M, ldc]
    const uint8_t* mask;   // optional row mask (0 -> output row := 0)
    int N;                 // number of W rows for this group
    int act;               // EPI_BIAS only: 0 none, 1 sigmoid
    float* C2;             // EPI_QKV: K plane [M, 512] SPLIT32 (C is the Q plane)
    void* C3;              // EPI_QKV: V^T f16 planes: hi [512, ldv] then lo [512, ldv]
    int ldv;               // EPI_QKV: frames per V^T row (multiple of 256, >= M rounded up to the M tile)
};

struct GemmArgs {
    GemmGroup g[kStreams];
    int groups;
    int M, K;
    int lda, ldc, ldr;
    float alpha;           // EPI_BIAS_RES: C = res + alpha * (acc + bias)
    int n_tiles;           // filled by launch_gemm
    int m_begin;           // first row of this launch (rows [m_begin, M) are covered); filled by the launcher
    int k_slices;          // split-K (f16x3, EPI_NONE): blockIdx.z = slice of the contraction; 0 / 1 = off
    size_t slice_stride;   // floats between the partial C planes of consecutive slices
    int flags;             // GEMM_FLAG_* (f16x3 path)
    // f16x3 path, optional row gather: output row p reads A row row_map[p] (< 0: a row of zeros); M then counts OUTPUT rows and
    // a_rows the rows of A.  The QKV projection uses it to write Q | K | V^T in clip-aligned coordinates (launch_attn_plan).
    const int32_t* row_map;
    int a_rows;            // rows of A (0: M)
    int vb_count;          // persistent kernel (hgemm3p_kernel): virtual blocks per group, filled by its launcher (last: the other kernels' argument offsets stay put)
};
constexpr int GEMM_FLAG_TR = 1;   // row-per-lane (transposed accumulator) epilogues where the epilogue has one (gemm_f16x3.hip)
constexpr int GEMM_FLAG_LINES = 4;     // persistent FFN1: the SPLIT32 epilogue through a wave-private LDS patch, whole 128-byte lines per store (round 6)
constexpr int GEMM_FLAG_PERSIST = 2;   // 256 x 256 launches through the persistent stream kernel (hgemm3p_kernel, round 6)

hipError_t launch_gemm(GemmEpi epi, const GemmArgs& a, hipStream_t s);
// 3-term split-f16 path (gemm_f16x3.hip): A and W in SPLIT32 format (split.h); out_split: C written in SPLIT32
// (EPI_BIAS_SILU only); tile: 0 = 128x128, 1 = 256x128, 2 = 256x256.
hipError_t launch_gemm_f16x3(GemmEpi epi, const GemmArgs& a, bool out_split, int tile, hipStream_t s);
// hi halves only (plain f16 x f16 -> fp32): mixed-precision training; EPI_NONE (optional split-K) / EPI_BIAS, 256 x 256 tile
hipError_t launch_gemm_f16x1(GemmEpi epi, const GemmArgs& a, int tile, hipStream_t s, int bf16 = 0);    // bf16: hi slots hold bf16 (split.h)
// GEMM on fp32 operands rounded / split in the staging path, either operand stored contraction-major (gemm16_kernel);
// mode 1 f16, 2 bf16 (one product), 3 split-f16 (three products, fp32-equivalent)
hipError_t launch_gemm16(const float* A, int lda, int ta, const float* B, int ldb, int tb, const float* bias, float* C, int ldc,
                         int M, int N, int K, int mode, int slices, size_t slice_stride, int sum_col, hipStream_t s, int stored16 = 0);
int gemm16_tile_n(int mode);
// GEMMs on operands stored as 16-bit values (train_gemm16s.hip): C = A16 [M, K] B16 [N, K]^T through a DMA ring; epi 0 fp32 (+ bias),
// 1 the FFN's first linear (bias, h16 | a16 = dropout(silu(h16)) planes), 2 the data gradient through dropout / SiLU (reads h16)
hipError_t launch_gemm16s(int epi, const void* A16, int lda, const void* B16, int ldb, const float* bias, void* C, int ldc, const void* H16,
                          int ldh, size_t plane_bytes, int M, int N, int K, int bf16, float p, uint64_t seed, float alpha, hipStream_t s);
hipError_t launch_dropcast16(const float* d, void* y16, int M, int N, float alpha, float p, uint64_t seed, int bf16, hipStream_t s);
hipError_t launch_cast16(const float* x, void* y16, int64_t n, int bf16, hipStream_t s);
hipError_t launch_silu16(const float* x, void* y16, int64_t n, int bf16, hipStream_t s);
hipError_t launch_transpose16(const float* w, void* w16, void* w16t, int N, int K, int bf16, hipStream_t s);
hipError_t launch_transpose16_table(const int64_t* table_dev, int n, int max_n, int max_k, int bf16, hipStream_t s);

// ---- row ops --------------------------------------------------------------------------------------
// y[g][m][:] = LayerNorm(x[g][m][:]) * gamma[g] + beta[g], eps 1e-5, rows of 512.
struct LnArgs {
    const float* x[kStreams];
    float* y[kStreams];            // fp32 output (may be nullptr)
    float* ys[kStreams];           // SPLIT32 output (may be nullptr)
    const float* gamma[kStreams];
    const float* beta[kStreams];
    int groups;
    int M;
};
hipError_t launch_layernorm(const LnArgs& a, hipStream_t s);
hipError_t launch_split_rows(const float* x, float* out, int64_t rows, int K, hipStream_t s, int bf16 = 0);
// in-place softmax over rows of width n (head_mode SOFTMAX)
hipError_t launch_row_softmax(float* x, int64_t rows, int n, hipStream_t s);

// ---- attention ------------------------------------------------------------------------------------
// qkv[g]: [M, 1536] (q | k | v, each 8 heads x 64); out[g]: [M, 512]; per clip, unmasked, scale 1/8.
struct AttnArgs {
    const float* qkv[kStreams];
    float* out[kStreams];
    const int32_t* frame_offsets;  // device [B+1]
    int groups, B, max_frames;
    int out_split;                 // 1: write `out` in SPLIT32 format
    float* lse[kStreams];          // optional (training): [8 heads][M] base-2 log-sum-exp of the scaled scores
    int M;                         // total frames (row count of lse planes); only read when lse is set
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);

// attention backward (train_attention.hip): dqkv [M, 1536] from dout [M, 512]; dsum [8][M] is scratch
struct AttnBwdArgs {
    const float* qkv; const float* out; const float* dout; const float* lse;
    float* dsum; float* dqkv;
    const int32_t* frame_offsets;
    int B, max_frames, M;
};
hipError_t launch_attention_bwd(const AttnBwdArgs& a, hipStream_t s);
hipError_t launch_attention_dsum(const float* out, const float* dout, float* dsum, int M, hipStream_t s, const float* factor = nullptr);
// split-f16 backward (train_attention_f16x3.hip): R / D row-major SPLIT32, Rt / Dt SPLIT32 over frames ([C, Mp])
hipError_t launch_attention_bwd_f16x3(const float* R, const float* Rt, const float* D, const float* Dt, const float* lse, const float* dsum,
                                      const int32_t* frame_offsets, int B, int max_frames, int M, int Mp, float* dqkv, int hi_only, hipStream_t s,
                                      void* dqkv16 = nullptr, const float* out_scale = nullptr, int out16 = 0);

// split-f16 attention (attention_f16x3.hip): operands as written by the EPI_QKV GEMM epilogue; out is SPLIT32.
struct Attn3Args {
    const float* q[kStreams];      // [M, 512] SPLIT32
    const float* k[kStreams];      // [M, 512] SPLIT32
    const void* vt[kStreams];      // f16 hi plane [512, ldv] followed by lo plane [512, ldv]
    float* out[kStreams];          // [M, 512] SPLIT32
    const int32_t* frame_offsets;
    int groups, B, max_frames, M, ldv;
    // inference: clip b's Q / K rows and V^T columns start at pad_offsets[b] (multiples of 16, launch_attn_plan) and its key tiles
    // are counted from there - a clip's result does not depend on where it lies in the batch; M = rows of the Q / K planes.
    // nullptr (training forward): operands in frame_offsets coordinates, key tiles aligned in global 64-frame blocks.
    const int32_t* pad_offsets;
    // training forward (out32 != nullptr): q / k point into split_rows(qkv) (row stride 1536 floats), vt into the V rows of
    // transpose(qkv, split) ([.., ldv] SPLIT32 over frames, ldv % 64 == 0); fp32 output + base-2 log-sum-exp [8, M]
    float* out32[kStreams];
    float* lse[kStreams];
    int hi_only;                   // training forward only: 1 = plain f16 operands (one product instead of three), 2 = bf16 operands
    int fast;                      // inference only (SOME_PRECISION_F16X3_FAST): 1 = P V without vh * pl, 2 = also Q K^T without kh * ql (attention3i_kernel<DMA, FAST>)
};
hipError_t launch_attention_f16x3(const Attn3Args& a, hipStream_t s);
inline int vt_ld(int64_t M) { return (int)((M + 255) / 256 * 256); }
// Clip-aligned attention coordinates (inference): clip b owns rows [pad_offsets[b], pad_offsets[b] + T_b) with
// pad_offsets[b] = sum over earlier clips of roundup(T, 16).  attn_rows_cover bounds the row count from the host's side (the
// offsets live on the device): M + 15 B rows of clips, + 63 so that the last key tile of the last clip stays inside, as a
// multiple of 64.  launch_attn_plan fills pad_offsets [B + 1] and row_map [rows_cover] (source row or -1).
#ifndef SOME_CLIP_ALIGN
#define SOME_CLIP_ALIGN 16      // 16 = the permuted V^T group (gemm_f16x3.hip EPI_QKV); build variants: 32 / 64
#endif
constexpr int kClipAlign = SOME_CLIP_ALIGN;
inline int64_t attn_rows_cover(int64_t M, int64_t B) { return (M + (kClipAlign - 1) * B + 63 + 63) / 64 * 64; }
hipError_t launch_attn_plan(const int32_t* frame_offsets, int B, int rows_cover, int32_t* pad_offsets, int32_t* row_map, hipStream_t s);

// ---- depthwise conv + folded BN + SiLU ------------------------------------------------------------
struct DwArgs {
    const float* x[kStreams];      // [M, 512] (GLU output)
    float* y[kStreams];            // [M, 512]
    const float* w[kStreams];      // [31, 512] folded
    const float* b[kStreams];      // [512] folded
    const int32_t* frame_offsets;
    int groups, B, max_frames;
    int out_split;                 // 1: write `y` in SPLIT32 format
};
hipError_t launch_dwconv(const DwArgs& a, hipStream_t s);

// ---- front end ------------------------------------------------------------------------------------
struct LogmelTables {
    float* window;       // [2048] periodic Hann
    float* twiddle;      // [2048] complex exp(-2 pi i k / 2048), interleaved (re, im)
    float* mel_w;        // packed non-zero filter weights
    float* mel_wpad;     // [81][32]: band m's weights zero-padded to 32 (row 80 = zeros); valid when max_len <= 32
    int32_t* mel_start;  // [80] first bin of each band
    int32_t* mel_len;    // [80] number of bins
    int32_t* mel_off;    // [80] offset into mel_w
    int kmax;            // highest FFT bin any band touches
    int nnz;             // number of packed filter weights
    int max_len;         // longest band (bins)
};
hipError_t launch_logmel(const LogmelTables& t, const float* audio, const int64_t* sample_offsets,
                         const int32_t* frame_offsets, int B, int max_frames, int pad_reflect, float* units, hipStream_t s);

// general transform length (key-shift / speed augmentation, logmel_shift.hip); window: [N] fp32 (already centred in the
// frame), twiddle: [N] (cos, -sin) fp64
constexpr int kMaxShiftFft = 4096;
size_t logmel_shift_lds_bytes(int N, int kmax);
hipError_t launch_logmel_shift(const LogmelTables& t, const float* window, const double* twiddle, const float* audio,
                               const int64_t* sample_offsets, const int32_t* frame_offsets, int B, int max_frames, int N,
                               int hop, int pad_left, int rescale, float scale_num, float scale_den, float* units,
                               hipStream_t s);

// ---- decode ---------------------------------------------------------------------------------------
struct DecodeArgs {
    const float* probs; const float* bounds; const uint8_t* mask;
    const int32_t* frame_offsets; int B; int64_t total_frames; int nbins; int quantized;
    double vmin, vmax, deviation, threshold;
    float* note_midi; int64_t* note_dur; uint8_t* note_rest; int32_t* n_notes;
    int64_t* frame2item; float* values; uint8_t* rest;
    void* scratch;
};
size_t decode_scratch_bytes(int64_t total_frames);
hipError_t launch_decode(const DecodeArgs& a, hipStream_t s);
hipError_t launch_decode_notes(const DecodeArgs& a, const int64_t* frame2item, const float* values,
                               const uint8_t* not_masks, int max_frames, hipStream_t s);

// ---- host ingest (ingest.hip) ------------------------------------------------------------------------
hipError_t launch_slicer_rms(const void* audio, int is_pcm16, const int64_t* sample_offsets, const int64_t* rms_offsets,
                             int B, int64_t max_rms_frames, int frame_length, int hop, float* rms, hipStream_t s);
hipError_t launch_pcm_gather(const void* src, int is_pcm16, const int64_t* src_offsets, const int64_t* dst_offsets, int B,
                             int64_t max_len, float* dst, hipStream_t s);

// ---- training operators (train_ops.hip) -------------------------------------------------------------------
size_t train_col_scratch_bytes(int M, int N);
size_t train_ln_scratch_bytes(int M);
size_t train_dwconv_w_scratch_bytes(int M, int C);
hipError_t launch_reduce_slices(const float* partial, int slices, size_t n, float* out, hipStream_t s);
// One reduction of weight-gradient planes, as launch_reduce_wgrad's arguments; launch_reduce_wgrad_table runs up to kWgradTable of them
// in ONE launch (same per-element summation order: bit-identical to one launch_reduce_wgrad each).  The entries must not share outputs.
struct WgradReduce { const float* partial; float* dw; float* db; size_t stride; int slices, M, N, ldc, accumulate; };
constexpr int kWgradTable = 32;
hipError_t launch_reduce_wgrad_table(const WgradReduce* entries, int n, hipStream_t s);
hipError_t launch_reduce_wgrad(const float* partial, int slices, size_t stride, int M, int N, int ldc, float* dw, float* db, int accumulate,
                               hipStream_t s);
hipError_t launch_transpose(const float* in, int M, int N, int ld_in, float* out, int ld_out, int split_out, hipStream_t s);
// row-major SPLIT32 and transposed SPLIT32 copies of in [M, N] in one pass; prescale: both multiplied by 2^floor(10 - log2 max|in|), computed on
// the device through absmax[256] and left as factor_out = {factor, 1 / factor}
hipError_t launch_split_transpose(const float* in, int M, int N, float* out_rows, float* out_t, int ld_t, int bf16, int prescale,
                                  uint32_t* absmax, float* factor_out, hipStream_t s);
hipError_t launch_colsum(const float* x, int M, int N, int ld, float* out, int accumulate, float* scratch, hipStream_t s);
hipError_t launch_weighted_colsum(const float* w, int ldw, const float* x, int M, int N, int ld, float* out, float* wsum, float* scratch, hipStream_t s);
hipError_t launch_ln_fwd(const float* x, const float* g, const float* b, void* y, float* mean, float* rstd, int M, int out16, hipStream_t s);
hipError_t launch_ln_bwd(const float* dy, const float* x, const float* g, const float* mean, const float* rstd, const float* add, float* dx,
                         float* dgamma, float* dbeta, int accumulate, int M, float* scratch, hipStream_t s);
hipError_t launch_bn_fwd(const float* x, const float* g, const float* b, int M, int C, float eps, float momentum, float* running_mean,
                         float* running_var, float* y, float* save_mean, float* save_rstd, float* scratch, hipStream_t s);
hipError_t launch_bn_bwd(const float* dy, const float* x, const float* g, const float* save_mean, const float* save_rstd, int M, int C,
                         float* dx, float* dgamma, float* dbeta, float* scratch, hipStream_t s);
hipError_t launch_eltwise(int op, const float* a, const float* b, float* out, int64_t n, float alpha, float p, uint64_t seed, hipStream_t s);
hipError_t launch_glu(const float* dy, const float* x, float* out, int64_t M, int C, int backward, hipStream_t s);
hipError_t launch_mask_rows(const float* x, const uint8_t* mask, float* y, int64_t M, int C, hipStream_t s);
hipError_t launch_dwconv_train(const float* x, const float* w, const float* bias, const int32_t* frame_offsets, int B, int max_frames, float* y,
                               int C, int flip, hipStream_t s);
hipError_t launch_dwconv_bwd_w(const float* dy, const float* x, const int32_t* clip_of_row, const int32_t* frame_offsets, int M, int C, float* dw,
                               int accumulate, float* scratch, hipStream_t s, int weight_layout = 0);
hipError_t launch_bce(const float* x, const float* t, int64_t n, float* dx, float* loss, double* scratch, hipStream_t s);
hipError_t launch_cross_entropy(const float* x, const int64_t* target, int M, int N, int64_t ignore, float* dx, float* loss, double* scratch, hipStream_t s);
hipError_t launch_emd(const float* pred, const float* gt, int B, int T, float* dpred, float* loss, double* scratch, hipStream_t s);
hipError_t launch_sumsq(const float* x, int64_t n, double* out, double* scratch, hipStream_t s);
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, float grad_scale, hipStream_t s);
hipError_t launch_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, const double* sumsq, double clip, double denom, hipStream_t s);

// ---- profiling ------------------------------------------------------------------------------------
struct ProfRecord { std::string name; hipEvent_t e0, e1; double flops, bytes; };

// ---- arena layout ---------------------------------------------------------------------------------
struct BlockOff {           // offsets in floats into the arena
    size_t ln_g[5], ln_b[5];
    size_t ffn_w1[2], ffn_b1[2], ffn_w2[2], ffn_b2[2];
    size_t wqkv, wo, bo;
    size_t pw1_w, pw1_b, dw_w, dw_b, pw2_w, pw2_b;
};

struct ArenaLayout {
    size_t in_w[kStreams], in_b[kStreams];
    size_t out_w, out_b, cut_w, cut_b;
    std::vector<BlockOff> blocks;       // index: (layer * 2 + stream), layer == lay -> final att1/att2
    std::vector<size_t> glu_w, glu_b;   // index: layer * 2 + (0: glu1, 1: glu2)
    size_t total_floats;
};

struct SomeHandle {
    SomeConfig cfg;
    ArenaLayout lay;
    const float* arena = nullptr;
    LogmelTables mel{};
    void* mel_blob = nullptr;
    std::string err;
    bool profiling = false;
    int precision = 0;          // SOME_PRECISION_*
    int attn_fast = 0;          // SOME_PRECISION_F16X3_FAST: attention3i_kernel<DMA, FAST> variant (precision itself then reads F16X3)
    int tile = 2;               // f16x3 GEMM tile selector (tuning knob)
    int gemm_flags = GEMM_FLAG_TR | GEMM_FLAG_PERSIST;   // SOME_AMD_GEMM_FLAGS overrides (A/B runs: 1 = one tile per workgroup)
    bool dual_stream = true;         // midi / bound model streams on two HIP streams (SOME_AMD_DUAL_STREAM=0: grouped launches)
    // helper stream + fork / join events of the dual-stream forward, one set per caller stream (two forwards enqueued on
    // different streams must not share a helper stream); enqueues are serialised by fwd_mu
    struct AuxSet { hipStream_t caller; hipStream_t aux; hipEvent_t fork, join; };
    std::vector<AuxSet> aux_sets;
    std::mutex fwd_mu;
    std::vector<ProfRecord> prof;
    std::vector<hipEvent_t> event_pool;
    // training: weight-gradient lanes (some_train_set_wgrad_stream) - the split-K weight-gradient GEMM + its reduction issued on `lane`
    // run on `side` behind an event of `lane`: they leave the backward pass's dependent chain (nothing downstream reads a weight gradient)
    // defer: the reductions of the planes wait in `pending` until some_train_wgrad_flush (or a full table / a clash of buffers) and
    // go out in one launch
    struct WgradLane { hipStream_t lane; hipStream_t side; hipEvent_t ev; bool defer; std::vector<WgradReduce> pending; };
    std::vector<WgradLane> wgrad_lanes;
    // window + twiddles of the key-shifted front end, one entry per (n_fft', win') seen
    struct ShiftTables { int n_fft, win; void* blob; float* window; double* twiddle; };
    std::vector<ShiftTables> shift_tables;
    std::mutex shift_mu;
};
