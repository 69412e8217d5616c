/*
 * some_amd.h - C ABI of libsome_amd.so: the MI355X (gfx950) implementation of the openvpi/SOME inference
 * hot path.  Plain C, raw pointers and sizes only; no torch / HIP types in any signature.
 *
 * The reference has no native layer: every entry point below replaces a sequence of
 * PyTorch ATen calls issued by the reference's Python code.  The Python classes in some_amd/ keep the
 * reference's operator API (names, argument meaning, error behaviour) and call this library via ctypes;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *  - All *_dev pointers are device (HBM) pointers owned by the CALLER (e.g. the PyTorch caching allocator).
 *    The library allocates device memory only in some_create (constant tables, a few hundred KB).
 *  - Clips are packed back to back ("var-len"): clip b owns frames [frame_offsets[b], frame_offsets[b+1]) of
 *    every [total_frames, C] array and samples [sample_offsets[b], sample_offsets[b+1]) of the audio array.
 *    Every operator that looks across time (STFT padding, attention, depthwise conv, decode) treats each
 *    clip independently, exactly as the reference's per-chunk B=1 loop does (inference/base_infer.py:46-53).
 *  - Every call only ENQUEUES work on `stream` (a hipStream_t passed as void*, NULL = default stream) and
 *    returns; nothing synchronises except where stated.
 *  - Return value: 0 on success, negative SOME_E* on failure; some_last_error() gives the message.  The
 *    library never aborts the process.
 */
#ifndef SOME_AMD_H
#define SOME_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOME_OK 0
#define SOME_EINVAL (-1)   /* bad argument / unsupported configuration            */
#define SOME_EKEY (-2)     /* state-dict key set mismatch (strict load)            */
#define SOME_ESHAPE (-3)   /* state-dict tensor shape mismatch                     */
#define SOME_EHIP (-4)     /* HIP runtime error (message carries hipGetErrorString) */
#define SOME_ESTATE (-5)   /* call order error (e.g. forward before attach_arena)  */
#define SOME_ENOMEM (-6)   /* workspace too small                                  */

typedef struct SomeHandle SomeHandle;

/* Hot-path keys of the reference's config.yaml (configs/base.yaml:11-28, configs/midi_conformer.yaml:16-33). */
typedef struct SomeConfig {
    int32_t lay;            /* midi_extractor_args.lay                  */
    int32_t dim;            /* midi_extractor_args.dim        (512)     */
    int32_t heads;          /* attention_heads                (8)       */
    int32_t head_dim;       /* attention_heads_dim            (64)      */
    int32_t kernel_size;    /* depthwise kernel               (31)      */
    int32_t indim;          /* units_dim                      (80)      */
    int32_t outdim;         /* midi_num_bins                  (128|129) */
    int32_t sample_rate;    /* audio_sample_rate              (44100)   */
    int32_t hop_size;       /* hop_size                       (512)     */
    int32_t win_size;       /* win_size == n_fft              (2048)    */
    float fmin;             /* fmin                           (40)      */
    float fmax;             /* fmax                           (8000)    */
    double midi_min;        /* midi_min                       (0)       */
    double midi_max;        /* midi_max                       (127)     */
    double midi_deviation;  /* midi_prob_deviation            (1.0)     */
    double rest_threshold;  /* rest_threshold                 (0.1)     */
    int32_t precision;      /* SOME_PRECISION_* : arithmetic of the dense GEMMs (library knob, not a reference key) */
    int32_t reserved;
} SomeConfig;

#define SOME_PRECISION_F32 0    /* exact fp32 on the f32 matrix pipe (v_mfma_f32_32x32x2_f32)                      */
#define SOME_PRECISION_F16X3 1  /* fp32-equivalent 3-term split on the f16 matrix pipe: x = hi + lo (two f16),      */
                                /* a*b = ah*bh + ah*bl + al*bh, fp32 accumulate; needs |GEMM inputs| < 65504        */
#define SOME_PRECISION_F16X3_FAST 2 /* OPT-IN: f16x3 everywhere except the attention product P V, which drops its vh*pl term  */
                                /* (P enters as rn_f16(2^11 p), normalised by the sum of the rounded values): attention  */
                                /* output within 2^-12 relative instead of 2^-21; GEMMs unchanged.  Never the default.    */

/* One entry of a PyTorch state_dict, host memory, contiguous, as produced by
 * torch.load(ckpt)['state_dict'] after the 'model.' prefix strip (inference/base_infer.py:27-32). */
typedef struct SomeTensorDesc {
    const char* name;       /* e.g. "model.cf_lay.0.att1.ffn1.ln1.weight"  */
    const void* data;       /* host pointer                                */
    int32_t dtype;          /* 0 = float32, 1 = int64                      */
    int32_t ndim;
    int64_t shape[4];
} SomeTensorDesc;

/* ---- lifecycle ---------------------------------------------------------------------------------- */

/* Replaces: model construction in BaseInference.build_model (inference/base_infer.py:24-26) and the
 * MelSpectrogram constructor (modules/rmvpe/spec.py:8-36; mel filterbank, Hann window, FFT twiddles).
 * Validates the configuration (dim 512, 8x64 heads, k=31, win 2048 / hop 512 are the compiled shapes). */
int some_create(const SomeConfig* cfg, SomeHandle** out);
void some_destroy(SomeHandle* h);
/* Message for the last failing call on this handle (h may be NULL: error of the last failed some_create). */
const char* some_last_error(const SomeHandle* h);
/* Library / build identification string ("some_amd <ver> gfx950 ..."). */
const char* some_version(void);

/* ---- weights ------------------------------------------------------------------------------------ */

/* Replaces: nn.Module.load_state_dict(strict=True) (inference/base_infer.py:33).
 * Packs the n named host tensors into the library's flat fp32 arena layout (host memory, some_arena_bytes()
 * long): [N,K] GEMM weights, fused QKV, GLU halves interleaved per 32 columns, BatchNorm folded into the
 * depthwise weights/bias (eval mode, eps 1e-5; modules/conv/base_conv.py:52,66).  Missing / unexpected
 * keys -> SOME_EKEY, wrong shapes -> SOME_ESHAPE, like strict=True.  `num_batches_tracked` is accepted
 * and ignored.  The caller uploads the arena (and, multi-GPU, broadcasts it with RCCL) and attaches it. */
size_t some_arena_bytes(const SomeHandle* h);
int some_pack_weights(SomeHandle* h, const SomeTensorDesc* tensors, int32_t n, float* host_arena);
/* Borrow a device copy of the packed arena; it must stay alive and unchanged while the handle is used. */
int some_attach_arena(SomeHandle* h, const float* arena_dev, size_t bytes);

/* ---- front end ---------------------------------------------------------------------------------- */

/* Replaces: librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=True) at modules/rmvpe/spec.py:22-28.
 * Host-only: writes the float32 [n_mels(80), 1 + win_size/2] basis the device kernel applies. */
int some_mel_filterbank(const SomeHandle* h, float* basis_host);

/* Replaces: MelSpectrogram.forward(keyshift=0, speed=1, center=True) + transpose(1,2)
 * (modules/rmvpe/spec.py:38-72, inference/me_infer.py:31).
 * audio_dev: fp32 packed clips; clip b has n_b = sample_offsets[b+1]-sample_offsets[b] samples and
 * produces T_b = 1 + n_b / hop frames at rows frame_offsets[b]... of units_dev [total_frames, n_mels].
 * sample_offsets_dev / frame_offsets_dev: device int64 / int32 arrays of B+1 entries.
 * max_frames: max_b T_b (host value, sizes the launch). */
#define SOME_PAD_ZERO 0      /* inference path: F.pad(audio, (win/2, win/2)) zeros        (modules/rmvpe/spec.py:47-50)   */
#define SOME_PAD_REFLECT 1   /* deployment path: torch.stft(center=True) reflect padding (deployment/base_onnx_module.py:68-76); needs n_b > win/2 */
int some_logmel(SomeHandle* h, const float* audio_dev, const int64_t* sample_offsets_dev,
                const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, int32_t pad_mode,
                float* units_dev, void* stream);

/* Replaces: MelSpectrogram.forward(audio, keyshift, speed, center) for keyshift != 0, speed != 1 or center=False
 * (modules/rmvpe/spec.py:38-72) - the key-shift augmentation of the training data
 * (preprocessing/me_binarizer.py:235-246, me_quant_binarizer.py:39-48).  The caller evaluates spec.py:39-42 itself:
 *   n_fft_new = round(n_fft * 2^(keyshift/12)), win_length_new likewise, hop_length_new = round(hop * speed);
 * clip b yields T_b = 1 + (n_b + (center ? win_length_new : 0) - n_fft_new) / hop_length_new frames (the caller sizes
 * frame_offsets with that; n_b + pad < n_fft_new is the caller's error, as it is torch.stft's).  rescale != 0 applies
 * spec.py:63-68 (keep 1 + n_fft/2 bins, zero-fill, scale by win_length / win_length_new).  n_fft_new <= 4096 (key shifts
 * up to +12 semitones, the range the reference's configs use).  Same packed layout and outputs as some_logmel. */
int some_logmel_shifted(SomeHandle* h, const float* audio_dev, const int64_t* sample_offsets_dev,
                        const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, int32_t n_fft_new,
                        int32_t win_length_new, int32_t hop_length_new, int32_t center, int32_t rescale, float* units_dev,
                        void* stream);

/* ---- network ------------------------------------------------------------------------------------ */

#define SOME_HEAD_LOGITS 0   /* midi = outln(x)                  (Gmidi_conform.py:30-32)            */
#define SOME_HEAD_SIGMOID 1  /* sig=True:  sigmoid(midi)         (Gmidi_conform.py:33-34)            */
#define SOME_HEAD_SOFTMAX 2  /* softmax=True: softmax over bins  (Gmidi_conform.py:36-37)            */

size_t some_workspace_bytes(const SomeHandle* h, int64_t total_frames, int32_t B);

/* Replaces: midi_conforms.forward(x, f0, mask, softmax, sig) (modules/model/Gmidi_conform.py:30-40 over
 * modules/conform/Gconform.py:119-140).  f0 is ignored by the reference model and has no parameter here.
 * units_dev [total_frames, indim]; row_mask_dev: optional uint8 [total_frames] (0 = masked frame ->
 * masked_fill on the midi stream, Gconform.py:126-132), NULL = all ones (what inference passes).
 * Outputs: midi_dev [total_frames, outdim], bound_dev [total_frames] (already sigmoid).
 * workspace_dev: at least some_workspace_bytes(total_frames, B) bytes, 256-byte aligned.  total_frames <= 262143 per call
 * (50 minutes of audio; SOME_EINVAL beyond - split the batch); in f16x3 mode additionally total_frames + 15 * B < 1 048 450.
 * Batch invariance: clip b's output rows depend on clip b alone - the same bits whether it is passed alone, at any position of a packed
 * batch or beside any neighbours (the reference runs every chunk by itself, inference/base_infer.py:46-53): attention key tiles are
 * counted from the clip's own first frame, every other kernel is row- or clip-local.
 * Streams: everything is ordered after the work already on `stream` and complete before anything enqueued on it
 * afterwards.  Inside, the two model streams of a layer (midi / bound, Gconform.py:82-87) run on `stream` and on a helper
 * stream the handle owns, forked and joined with events around every layer (SOME_AMD_DUAL_STREAM=0, kernel profiling, or a
 * first call made while `stream` is being captured: everything on `stream`).  Concurrent calls on one handle are serialised while they enqueue. */
int some_forward(SomeHandle* h, const float* units_dev, const int32_t* frame_offsets_dev, int32_t B,
                 int64_t total_frames, int32_t max_frames, const uint8_t* row_mask_dev, int32_t head_mode,
                 float* midi_dev, float* bound_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- decode ------------------------------------------------------------------------------------- */

/* Replaces: MIDIExtractionInference.postprocess (inference/me_infer.py:78-97) or, with quantized != 0,
 * QuantizedMIDIExtractionInference.postprocess (inference/me_quant_infer.py:22-38), i.e.
 * decode_bounds_to_alignment + decode_gaussian_blurred_probs|argmax + decode_note_sequence
 * (utils/infer_utils.py:9-76), per clip.
 * probs_dev [total_frames, outdim], bounds_dev [total_frames], row_mask_dev optional (NULL = all ones).
 * Outputs (device): clip b's notes are written at rows frame_offsets[b] ... frame_offsets[b]+n_notes[b]-1
 * of note_midi_dev (fp32), note_dur_dev (int64, in FRAMES; the caller multiplies by hop/sr in float64 as
 * me_infer.py:95 does) and note_rest_dev (uint8); n_notes_dev int32 [B].
 * Optional per-frame intermediates (may be NULL): frame2item_dev int64 [total_frames],
 * values_dev fp32 [total_frames] (quantized: the clipped argmax as fp32), rest_dev uint8 [total_frames].
 * scratch_dev: some_decode_scratch_bytes(total_frames) bytes. */
size_t some_decode_scratch_bytes(const SomeHandle* h, int64_t total_frames);
int some_decode(SomeHandle* h, const float* probs_dev, const float* bounds_dev, const uint8_t* row_mask_dev,
                const int32_t* frame_offsets_dev, int32_t B, int64_t total_frames, int32_t quantized,
                float* note_midi_dev, int64_t* note_dur_dev, uint8_t* note_rest_dev, int32_t* n_notes_dev,
                int64_t* frame2item_dev, float* values_dev, uint8_t* rest_dev,
                void* scratch_dev, size_t scratch_bytes, void* stream);

/* Replaces: decode_note_sequence(frame2item, values, masks) (utils/infer_utils.py:42-76) on caller-supplied
 * per-frame arrays (any frame2item, not necessarily produced from bounds).  not_masks_dev: uint8, 1 where the
 * frame must NOT count (= ~masks).  values_are_integers: the quantised head's int64 semantics (exact integer
 * sums).  Clips of at most 4096 frames.  Outputs as some_decode. */
int some_decode_notes(SomeHandle* h, const int64_t* frame2item_dev, const float* values_dev, const uint8_t* not_masks_dev,
                      const int32_t* frame_offsets_dev, int32_t B, int64_t total_frames, int32_t max_frames,
                      int32_t values_are_integers, float* note_midi_dev, int64_t* note_dur_dev, uint8_t* note_rest_dev,
                      int32_t* n_notes_dev, void* scratch_dev, size_t scratch_bytes, void* stream);

/* ---- host ingest either side of the silence slicer ---------------------- */

#define SOME_SAMPLE_F32 0     /* float32 samples in [-1, 1]                                            */
#define SOME_SAMPLE_PCM16 1   /* int16 PCM as stored in the WAV file; value = x / 32768 (exact in fp32) */

/* Replaces: get_rms(y, frame_length, hop_length) (utils/slicer2.py:5-38), the numpy reduction Slicer.slice runs
 * over every file (slicer2.py:87): zero centre padding, mean of squares over hop-strided frames, sqrt - with numpy's
 * float32 pairwise summation order, so rms < threshold and argmin decide exactly as on the host.
 * audio_dev: packed clips of `sample_format`; clip b = [sample_offsets[b], sample_offsets[b+1]) (device int64 [B+1]).
 * Clip b yields 1 + n_b / hop_length values at rms_dev[rms_offsets[b] ...] (device int64 [B+1]);
 * max_rms_frames = max_b of that count (host value, sizes the launch). */
int some_slicer_rms(SomeHandle* h, const void* audio_dev, int32_t sample_format, const int64_t* sample_offsets_dev,
                    const int64_t* rms_offsets_dev, int32_t B, int64_t max_rms_frames, int32_t frame_length,
                    int32_t hop_length, float* rms_dev, void* stream);

/* Replaces: the chunk cut of Slicer.slice (utils/slicer2.py:73-82) fused with librosa.load's sample conversion
 * (infer.py:34, batch_infer.py:51): span b = src_dev[src_offsets[b] ... + n_b) is written as fp32 to
 * audio_out_dev[dst_offsets[b] ...), n_b = dst_offsets[b+1] - dst_offsets[b]; the result is the packed layout
 * some_logmel reads with sample_offsets = dst_offsets.  src_offsets_dev: device int64 [B]; dst_offsets_dev: device
 * int64 [B+1]; max_len = max_b n_b (host value). */
int some_pcm_gather(SomeHandle* h, const void* src_dev, int32_t sample_format, const int64_t* src_offsets_dev,
                    const int64_t* dst_offsets_dev, int32_t B, int64_t max_len, float* audio_out_dev, void* stream);

/* ---- training operators (train.py, training/me_task.py:79-111) ------------------
 * Forward + backward building blocks of midi_conforms in train mode on packed [M, C] fp32 activations (M = B * T_max
 * frames of the padded training batch).  The dense contractions reuse some_op_gemm (weight gradients contract over M:
 * dW = dY^T X is some_op_gemm on the two transposes).  `scratch_dev` arguments need some_train_scratch_bytes(M, N)
 * bytes; column reductions are two-pass and deterministic. */
size_t some_train_scratch_bytes(const SomeHandle* h, int64_t M, int32_t N);

/* Weight-gradient GEMM  C[M, N] = A[M, K] W[N, K]^T  with a LONG contraction (K = all frames of the batch) and a small
 * output: split-f16 kernel with the contraction cut into slices across workgroups (blockIdx.z), partial planes in
 * partial_dev (some_train_gemm_splitk_bytes), summed in slice order - deterministic.  A_split / W_split: SPLIT32 rows
 * (some_op_split_rows), K % 32 == 0, lda % 32 == 0.  hi_only = 1: plain f16 operands (one product instead of three);
 * hi_only = 2: bf16 operands (made with SOME_OPERAND_BF16 / split_out = 2), the reference's pl_trainer_precision 'bf16'
 * (configs/midi_conformer.yaml:35).  The same 0 / 1 / 2 applies to the attention entry points below. */
size_t some_train_gemm_splitk_bytes(const SomeHandle* h, int32_t M, int32_t N, int32_t K);
int some_train_gemm_splitk(SomeHandle* h, const float* A_split_dev, int32_t lda, const float* W_split_dev, float* C_dev,
                           int32_t M, int32_t N, int32_t K, int32_t hi_only, void* partial_dev, size_t partial_bytes,
                           void* stream);
/* Mixed-precision GEMM on fp32 operands (the reference's autocast matmuls under pl_trainer_precision 'bf16' / '16-mixed',
 * configs/midi_conformer.yaml:35, training/me_task.py:79-111 through Lightning):
 *     C[M, N] = Aop[M, K] Bop[N, K]^T (+ bias[N]),   Aop / Bop = the operand rounded to bf16 (operand = 2) or f16 (1),
 * fp32 accumulation.  The rounding - and, where needed, the transposition - happens in the kernel's staging path, so no
 * split / transposed copy of an activation is ever written:
 *     ta = 0: A_dev is [M, lda] (contraction index contiguous);   ta = 1: A_dev is [K, lda] (A stored transposed)
 *     tb = 0: B_dev is [N, ldb];                                   tb = 1: B_dev is [K, ldb]
 * (ta, tb) = (0, 0) nn.Linear forward, (0, 1) its data gradient dX = dY W, (1, 1) its weight gradient dW = dY^T X - then the
 * contraction runs over all frames and is cut into slices across workgroups, summed in slice order (deterministic) through
 * partial_dev (some_train_gemm16_bytes), and sum_col >= 0 writes the fp32 column sums of A_dev (the bias gradient dY^T 1)
 * into column sum_col of C (ldc > sum_col >= N).  Needs lda % 4 == ldb % 4 == 0; K % 32 == 0 unless ta = tb = 1; M even
 * when ta; N % 4 == 0 when tb.  (1, 0) is not provided. */
size_t some_train_gemm16_bytes(const SomeHandle* h, int32_t M, int32_t N, int32_t K, int32_t ldc);
int some_train_gemm16(SomeHandle* h, const float* A_dev, int32_t lda, int32_t ta, const float* B_dev, int32_t ldb, int32_t tb,
                      const float* bias_dev, float* C_dev, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t operand,
                      int32_t sum_col, void* partial_dev, size_t partial_bytes, void* stream);
/* The weight gradient of nn.Linear (what autograd computes for every Linear / k = 1 Conv1d of modules/conform/Gconform.py in
 * training/me_task.py:79-111's backward pass) written where it belongs: dW[N, K] (+)= dY[frames, N]^T X[frames, K] and, if db_dev, db[N] (+)= the
 * fp32 column sums of dY - the (1, 1) layout of some_train_gemm16 with the slice reduction storing straight into the parameter-
 * gradient arrays (accumulate = 1: added to their contents, i.e. to the flat gradient buffer across micro-batches), so that no
 * intermediate tensor, slice copy or autograd accumulation launch remains.  partial_dev: some_train_gemm16_bytes(N, K, frames, K + 4). */
int some_train_gemm16_wgrad(SomeHandle* h, const float* dY_dev, int32_t ldy, const float* X_dev, int32_t ldx, float* dW_dev, float* db_dev,
                            int32_t N, int32_t K, int32_t frames, int32_t operand, int32_t accumulate, void* partial_dev, size_t partial_bytes,
                            void* stream);
/* ---- 16-bit STORED operands (mixed-precision training: under the reference's autocast, training/base_task.py:260-283 with
 * pl_trainer_precision 'bf16' / '16-mixed', nn.Linear reads and writes 16-bit tensors) ----------------------------------------
 * operand: 1 = f16, 2 = bf16 - the storage format of every `16` array of a call.
 * some_train_cast16:      y16[i] = rn16(x[i]), n % 8 == 0, 16-byte aligned arrays.
 * some_train_transpose16: W[N, K] fp32 -> W16[N, K] (forward operand) and / or W16T[K, N] (data-gradient operand); either may be NULL. */
int some_train_cast16(SomeHandle* h, const float* x_dev, void* y16_dev, int64_t n, int32_t operand, void* stream);
/* y16[i] = rn16(silu(x[i])): conform_conv's activation (modules/conv/base_conv.py:68) as the 16-bit operand of pointwise_conv2; n % 8 == 0. */
int some_train_silu16(SomeHandle* h, const float* x_dev, void* y16_dev, int64_t n, int32_t operand, void* stream);
int some_train_transpose16(SomeHandle* h, const float* w_dev, void* w16_dev, void* w16t_dev, int32_t N, int32_t K, int32_t operand, void* stream);
/* The same for n weights in ONE launch: table_dev holds five int64 per weight - the device addresses of W, W16, W16T (either image may be
 * 0) and N, K; max_n / max_k bound the table's N / K.  What the trainer runs once per optimiser step for all 16-bit weight images. */
int some_train_transpose16_table(SomeHandle* h, const int64_t* table_dev, int32_t n, int32_t max_n, int32_t max_k, int32_t operand, void* stream);
/* C[M, N] = A16[M, K] B16[N, K]^T, fp32 accumulation, both operands contraction-contiguous 16-bit arrays (K % 32 == 0, lda % 8 ==
 * ldb % 8 == 0, 16-byte aligned) moved global -> LDS by DMA.  epilogue:
 *   0  C fp32 [M, ldc] = acc (+ bias_dev[N] if not NULL)                                 nn.Linear forward / data gradient (B16 = W16T)
 *   1  conform_ffn.forward's ln1 + act + drop1 (modules/conform/Gconform.py:29-32): C is TWO 16-bit planes, h16 = rn16(acc + bias) at C
 *      and a16 = rn16(dropout_p(silu(h16))) at C + plane_elems elements (plane_elems >= M * ldc, even)
 *   2  the gradient back through drop1 / act: C 16-bit [M, ldc] = rn16(acc * mask / (1 - p') * silu'(h16)), h16 = H16_dev [M, ldh] of the
 *      forward call with the SAME (p, seed)
 * Dropout: element (m, n) is zeroed when its 16 pseudo-random bits - a pure function of (seed, m, n) - are below round(65536 p); kept
 * values are scaled by 1 / (1 - p'), p' = round(65536 p) / 65536, the rate actually applied.  p = 0: no dropout. */
int some_train_gemm16s(SomeHandle* h, int32_t epilogue, const void* A16_dev, int32_t lda, const void* B16_dev, int32_t ldb, const float* bias_dev,
                       void* C_dev, int32_t ldc, const void* H16_dev, int32_t ldh, int64_t plane_elems, int32_t M, int32_t N, int32_t K,
                       int32_t operand, float p, uint64_t seed, float alpha, void* stream);
/* epilogue 3 of some_train_gemm16s: C fp32 [M, ldc] = R + alpha * dropout_p(acc + bias), R = H16_dev read as fp32 [M, ldh] - the conformer
 * block's `x = ffn(x) * 0.5 + x` with conform_ffn's output dropout (modules/conform/Gconform.py:33,57,60) folded into the second linear.
 * some_train_dropcast16 is its gradient w.r.t. the linear's output, written as the 16-bit operand of the data-gradient GEMM:
 * y16[m, n] = rn16(alpha * mask(m, n) / (1 - p') * d[m, n]) with the SAME (p, seed); N % 4 == 0. */
int some_train_dropcast16(SomeHandle* h, const float* d_dev, void* y16_dev, int32_t M, int32_t N, float alpha, float p, uint64_t seed,
                          int32_t operand, void* stream);
/* some_train_gemm16_wgrad on 16-bit stored operands: dW[N, K] (+)= dY16[frames, N]^T X16[frames, K], db[N] (+)= column sums of dY16
 * (fp32 sums of the stored values).  ld % 4 == 0, 8-byte aligned operands; partial_dev as for some_train_gemm16_wgrad. */
int some_train_gemm16_wgrad16(SomeHandle* h, const void* dY16_dev, int32_t ldy, const void* X16_dev, int32_t ldx, float* dW_dev, float* db_dev,
                              int32_t N, int32_t K, int32_t frames, int32_t operand, int32_t accumulate, void* partial_dev, size_t partial_bytes,
                              void* stream);
/* Weight-gradient lanes (round 5).  Nothing downstream in a backward pass reads a weight gradient, yet on one stream every split-K
 * weight-gradient GEMM + reduction (64 + 73 launches per step at the reference's batch shape, configs/base.yaml:55-56) sits in the
 * dependent chain of the data gradients.  After some_train_set_wgrad_stream(h, stream, wgrad_stream) the launches of
 * some_train_gemm16_wgrad / some_train_gemm16_wgrad16 (also those inside some_train_ffn_block_bwd) that are issued on `stream` run on
 * `wgrad_stream` instead, behind an event recorded on `stream` at the call (their operands were produced there).  Same kernels, same
 * summation order: the gradients are bit-identical.  The CALLER then owns three orderings: (1) whatever reads the gradient arrays
 * (gradient norm, optimiser, an all-reduce) must wait for `wgrad_stream`; (2) dY / X, the save / scratch blocks of the block call and
 * partial_dev must stay untouched until `wgrad_stream` has passed the call - partial_dev may be shared by the calls of ONE wgrad_stream
 * (they serialise there) but not with calls that stay on `stream`; (3) writers of the same gradient array must use the same pair.
 * defer_reductions != 0: the reduction behind each weight-gradient GEMM (planes -> dW / db) is not launched with it but waits until
 * some_train_wgrad_flush(h, stream) - which the caller issues BEFORE it orders a reader behind `wgrad_stream` - and then goes out with
 * the pair's other waiting reductions in ONE launch (up to 32 per launch; 73 launches per step become 4); every call then needs planes of
 * its own until the flush (a partial_dev that overlaps waiting planes, or an output that overlaps a waiting output, flushes first: correct,
 * but nothing is saved).  wgrad_stream = NULL flushes and removes the pairing of `stream`.  Not thread-safe against concurrent training
 * calls on the same handle. */
int some_train_set_wgrad_stream(SomeHandle* h, void* stream, void* wgrad_stream, int32_t defer_reductions);
int some_train_wgrad_flush(SomeHandle* h, void* stream);
/* out[n, m] = in[m, n] for m < M, 0 for M <= m < ld_out (the zero padding makes ld_out a valid contraction length).
 * split_out = 1: rows are written in SPLIT32 format (ready as a split-f16 GEMM operand; ld_out % 32 == 0);
 * split_out = 2: the same slots with bf16 hi halves (SOME_OPERAND_BF16). */
int some_train_transpose(SomeHandle* h, const float* in_dev, int32_t M, int32_t N, int32_t ld_in, float* out_dev,
                         int32_t ld_out, int32_t split_out, void* stream);
/* out[n] (+)= sum_m x[m, n]: bias gradient of nn.Linear / Conv1d. */
int some_train_colsum(SomeHandle* h, const float* x_dev, int32_t M, int32_t N, int32_t ld, float* out_dev,
                      int32_t accumulate, void* scratch_dev, size_t scratch_bytes, void* stream);
/* out[n] = sum_m w[m * ldw] x[m, n], wsum[n] = sum_m w[m * ldw] (same value for every n; may be NULL): weight and bias gradient
 * of a ONE-output nn.Linear (the bound head `cutheard`, modules/conform/Gconform.py:116) - a column reduction, not a GEMM. */
int some_train_weighted_colsum(SomeHandle* h, const float* w_dev, int32_t ldw, const float* x_dev, int32_t M, int32_t N, int32_t ld,
                               float* out_dev, float* wsum_dev, void* scratch_dev, size_t scratch_bytes, void* stream);
/* nn.LayerNorm(512, eps 1e-5) forward with saved statistics, and its backward (Gconform.py:48-52). */
int some_train_layernorm_fwd(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                             float* y_dev, float* mean_dev, float* rstd_dev, int32_t M, void* stream);
int some_train_layernorm_bwd(SomeHandle* h, const float* dy_dev, const float* x_dev, const float* gamma_dev,
                             const float* mean_dev, const float* rstd_dev, float* dx_dev, float* dgamma_dev,
                             float* dbeta_dev, int32_t accumulate, int32_t M, void* scratch_dev, size_t scratch_bytes,
                             void* stream);
/* Block-level operators (round 5): the trainer's FFN sub-block `x + alpha * dropout(ffn(LayerNorm(x)))` (modules/conform/Gconform.py:29-34,
 * 57, 60) forward and backward as ONE call each - the same launches, in the same order, as some_train_layernorm_fwd16 + 2 x some_train_gemm16s
 * (forward) and some_train_dropcast16 + 2 x some_train_gemm16s + 2 x some_train_gemm16_wgrad16 + some_train_layernorm_bwd_add (backward):
 * bit-identical results, a third of the host time.  save_dev: caller-owned block the forward fills for the backward (n16 | mean | rstd | h16 |
 * a16); scratch_dev: temporaries of the backward (dy16 | dh16 | dn); both 256-byte aligned, sized by the *_bytes functions.  w*_16: the
 * weights' 16-bit images [N, K] (forward) / their transposes [K, N] (backward) from some_train_transpose16*.  The weight / bias / LayerNorm
 * gradients are ACCUMULATED into d*_dev (the flat gradient buffer); dx_dev = LayerNorm'(dn) (+ d when add_residual). */
size_t some_train_ffn_block_save_bytes(const SomeHandle* h, int32_t M, int32_t K, int32_t H);
size_t some_train_ffn_block_scratch_bytes(const SomeHandle* h, int32_t M, int32_t K, int32_t H, int32_t N);
int some_train_ffn_block_fwd(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev, const void* w1_16_dev,
                             const float* b1_dev, const void* w2_16_dev, const float* b2_dev, int32_t M, int32_t K, int32_t H, int32_t N,
                             int32_t operand, float alpha, float p_latent, uint64_t seed_latent, float p_out, uint64_t seed_out,
                             void* save_dev, size_t save_bytes, float* out_dev, void* stream);
int some_train_ffn_block_bwd(SomeHandle* h, const float* d_dev, const float* x_dev, const float* gamma_dev, const void* save_dev,
                             const void* w1t_16_dev, const void* w2t_16_dev, int32_t M, int32_t K, int32_t H, int32_t N, int32_t operand,
                             float alpha, float p_latent, uint64_t seed_latent, float p_out, uint64_t seed_out,
                             float* dw1_dev, float* db1_dev, float* dw2_dev, float* db2_dev, float* dgamma_dev, float* dbeta_dev,
                             int32_t add_residual, float* dx_dev, void* scratch_dev, size_t scratch_bytes, void* ln_scratch_dev, size_t ln_scratch_bytes,
                             void* partial_dev, size_t partial_bytes, void* stream);
/* LayerNorm forward with the output written in 16 bits (operand 1 = f16, 2 = bf16): the FFN's GEMM operand without a cast pass. */
int some_train_layernorm_fwd16(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                               void* y16_dev, float* mean_dev, float* rstd_dev, int32_t M, int32_t operand, void* stream);
/* some_train_layernorm_bwd with a second gradient of x summed in: dx = add + (LayerNorm gradient); add_dev may be NULL.  The residual
 * connection around the LayerNorm of a conformer sub-block (Gconform.py:57-61) reaches x twice; this removes the separate addition. */
int some_train_layernorm_bwd_add(SomeHandle* h, const float* dy_dev, const float* x_dev, const float* gamma_dev,
                                 const float* mean_dev, const float* rstd_dev, const float* add_dev, float* dx_dev, float* dgamma_dev,
                                 float* dbeta_dev, int32_t accumulate, int32_t M, void* scratch_dev, size_t scratch_bytes,
                                 void* stream);
/* nn.BatchNorm1d(C) in train mode over the M rows (base_conv.py:56): batch statistics, running-stat update
 * (momentum, unbiased variance), saved mean / rstd for the backward. */
int some_train_batchnorm_fwd(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                             int32_t M, int32_t C, float eps, float momentum, float* running_mean_dev,
                             float* running_var_dev, float* y_dev, float* save_mean_dev, float* save_rstd_dev,
                             void* scratch_dev, size_t scratch_bytes, void* stream);
int some_train_batchnorm_bwd(SomeHandle* h, const float* dy_dev, const float* x_dev, const float* gamma_dev,
                             const float* save_mean_dev, const float* save_rstd_dev, int32_t M, int32_t C,
                             float* dx_dev, float* dgamma_dev, float* dbeta_dev, void* scratch_dev,
                             size_t scratch_bytes, void* stream);
#define SOME_ELT_SILU_FWD 0     /* out = a silu                                       */
#define SOME_ELT_SILU_BWD 1     /* out = a * silu'(b)        a = dy, b = pre-activation */
#define SOME_ELT_SIGMOID_FWD 2  /* out = sigmoid(a)                                   */
#define SOME_ELT_SIGMOID_BWD 3  /* out = a * b * (1 - b)     a = dy, b = sigmoid output */
#define SOME_ELT_AXPY 4         /* out = alpha * a + b       (x = f(x) * 0.5 + x, Gconform.py:57,60); b NULL: alpha * a */
#define SOME_ELT_DROPOUT 5        /* out = drop(a): a * keep / (1 - p), keep ~ Bernoulli(1 - p) from (seed, index)       */
#define SOME_ELT_SILU_DROP_FWD 6  /* out = drop(silu(a))                       (conform_ffn: act + drop1, Gconform.py:31-32) */
#define SOME_ELT_SILU_DROP_BWD 7  /* out = drop(a) * silu'(b)                  a = dy, b = pre-activation                    */
#define SOME_ELT_AXPY_DROP 8      /* out = alpha * drop(a) + b (b NULL: none)  (x = drop(f(x)) * 0.5 + x, Gconform.py:57-61)  */
int some_train_eltwise(SomeHandle* h, int32_t op, const float* a_dev, const float* b_dev, float* out_dev, int64_t n,
                       float alpha, float p, uint64_t seed, void* stream);
/* GLU over channel halves of [M, 2C] (Gconform.py:11-17, base_conv.py:7-15); backward = 1: dy [M, C], x -> dx [M, 2C]. */
int some_train_glu(SomeHandle* h, const float* dy_dev, const float* x_dev, float* out_dev, int64_t M, int32_t C,
                   int32_t backward, void* stream);
/* x.masked_fill(~mask[:, None], 0) (Gconform.py:128-133); the same call maps dy -> dx. */
int some_train_mask_rows(SomeHandle* h, const float* x_dev, const uint8_t* mask_dev, float* y_dev, int64_t M,
                         int32_t C, void* stream);
/* Depthwise Conv1d(C, C, 31, padding 15, groups C) over time inside each clip (base_conv.py:48-53); taps [31, C]
 * (tap-major), bias may be NULL.  flip = 1 applies the taps reversed: the data gradient dx = conv(dy, flip(w)). */
int some_train_dwconv(SomeHandle* h, const float* x_dev, const float* taps_dev, const float* bias_dev,
                      const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, float* y_dev, int32_t C,
                      int32_t flip, void* stream);
/* dtaps[k, c] (+)= sum_t dy[t, c] x[t + k - 15, c]; clip_of_row_dev: int32 [M], the clip index of every row. */
int some_train_dwconv_bwd_taps(SomeHandle* h, const float* dy_dev, const float* x_dev, const int32_t* clip_of_row_dev,
                               const int32_t* frame_offsets_dev, int32_t M, int32_t C, float* dtaps_dev,
                               int32_t accumulate, void* scratch_dev, size_t scratch_bytes, void* stream);
/* The depthwise convolution's PARAMETER gradients written where they belong: dweight_dev [C][31] - the Conv1d weight's own layout, i.e. the
 * parameter's gradient array - += the tap sums of some_train_dwconv_bwd_taps, dbias_dev [C] (may be NULL) += the column sums of dy.  The
 * same kernels and sums as some_train_dwconv_bwd_taps(accumulate 0) + a transposed add and some_train_colsum + an add: bit-identical,
 * four launches (two of them torch's) fewer.  When `stream` is paired with a weight-gradient stream (some_train_set_wgrad_stream) the
 * launches run there - the caller keeps dy / x and the scratch block (some_train_scratch_bytes(M, C), not shared with calls that stay on
 * `stream`) untouched until that stream has been joined. */
int some_train_dwconv_bwd_params(SomeHandle* h, const float* dy_dev, const float* x_dev, const int32_t* clip_of_row_dev,
                                 const int32_t* frame_offsets_dev, int32_t M, int32_t C, float* dweight_dev, float* dbias_dev,
                                 void* scratch_dev, size_t scratch_bytes, void* stream);
/* nn.BCEWithLogitsLoss() (mean over all n elements; training/me_task.py:74,105): loss_dev[0] and, when dlogits_dev is
 * not NULL, d loss / d logits. */
int some_train_bce_with_logits(SomeHandle* h, const float* logits_dev, const float* target_dev, int64_t n,
                               float* dlogits_dev, float* loss_dev, void* scratch_dev, size_t scratch_bytes,
                               void* stream);
/* nn.CrossEntropyLoss(ignore_index) over the rows of logits [M, N] with int64 class targets [M] (training/me_quant_task.py:42,77:
 * QuantizedMIDIExtractionTask's 129-way midi loss, padding frames carry -1): loss_dev[0] = mean over the rows whose target is not
 * ignore_index (NaN when there is none, as torch) and, when dlogits_dev is not NULL, d loss / d logits.  scratch_dev: 1025 doubles. */
int some_train_cross_entropy(SomeHandle* h, const float* logits_dev, const int64_t* target_dev, int32_t M, int32_t N, int64_t ignore_index,
                             float* dlogits_dev, float* loss_dev, void* scratch_dev, size_t scratch_bytes, void* stream);
/* modules.losses.BinaryEMDLoss() (modules/losses/bound_loss.py:6-19, bidirectional=False) on [B, T] rows. */
int some_train_binary_emd(SomeHandle* h, const float* pred_dev, const float* gt_dev, int32_t B, int32_t T,
                          float* dpred_dev, float* loss_dev, void* scratch_dev, size_t scratch_bytes, void* stream);
/* out_dev[0] (double) = sum_i x[i]^2: the squared global gradient norm for clip_grad_norm (configs/base.yaml:49,
 * train.py:88); non-finite inputs give a non-finite result, which doubles as the loss-scale overflow check.
 * scratch_dev: 1024 doubles. */
int some_train_sumsq(SomeHandle* h, const float* x_dev, int64_t n, double* out_dev, void* scratch_dev, size_t scratch_bytes,
                     void* stream);
/* torch.optim.AdamW step (configs/two_head_model.yaml:42-47) on flat arrays; step counts from 1; the gradient is
 * multiplied by grad_scale first (1 / world_size after a summing all-reduce). */
int some_train_adamw(SomeHandle* h, float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                     int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                     float grad_scale, void* stream);

/* The same AdamW step with Lightning's gradient_clip_val (torch.nn.utils.clip_grad_norm_; configs/base.yaml:49, train.py:88) applied on the
 * DEVICE: grad_scale = min(1, clip_norm / (sqrt(*sumsq_dev) / grad_denominator + 1e-6)) / grad_denominator in IEEE double arithmetic, where
 * *sumsq_dev is some_train_sumsq's result for the (summed, scaled) gradient and grad_denominator = world_size * loss_scale; clip_norm 0 = no
 * clipping.  A non-finite *sumsq_dev leaves parameters and moments untouched.  Lets a training step end without a host synchronisation. */
int some_train_adamw_clip(SomeHandle* h, float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                          int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                          const double* sumsq_dev, double clip_norm, double grad_denominator, void* stream);

/* F.scaled_dot_product_attention (8 heads x 64, scale 1/8, per clip, unmasked; base_attention.py:34-44) on the fused
 * projection output qkv [M, 1536] (q | k | v), exact fp32 MFMA.  Forward also writes lse_dev [8, M] (base-2
 * log-sum-exp per head and frame) for the backward; backward recomputes the probabilities flash-style and writes
 * dqkv_dev [M, 1536] completely.  dsum_scratch_dev: 8 * M floats. */
int some_train_attention_fwd(SomeHandle* h, const float* qkv_dev, const int32_t* frame_offsets_dev, int32_t B,
                             int32_t max_frames, int32_t M, float* out_dev, float* lse_dev, void* stream);
int some_train_attention_bwd(SomeHandle* h, const float* qkv_dev, const float* out_dev, const float* dout_dev,
                             const float* lse_dev, const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames,
                             int32_t M, float* dqkv_dev, float* dsum_scratch_dev, void* stream);

/* The forward on the f16 matrix pipe with 3-term split operands (fp32-equivalent): qkv_split = some_op_split_rows(qkv),
 * qkv_t_split = some_train_transpose(qkv, split_out = 1) with Mp = M rounded up to a multiple of 64.  Writes fp32 out
 * [M, 512] and lse [8, M]; the same two split tensors feed some_train_attention_bwd_f16x3. */
int some_train_attention_fwd_f16x3(SomeHandle* h, const float* qkv_split_dev, const float* qkv_t_split_dev,
                                   const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, int32_t M,
                                   int32_t Mp, int32_t hi_only, float* out_dev, float* lse_dev, void* stream);
/* The same backward on the f16 matrix pipe with 3-term split operands (fp32-equivalent).  qkv_split / dout_split:
 * some_op_split_rows of qkv [M, 1536] / dout [M, 512]; qkv_t_split / dout_t_split: some_train_transpose(split_out = 1)
 * of the same tensors ([1536, Mp] / [512, Mp], Mp = M rounded up to 32).  out_dev / dout_dev (fp32) are only used for
 * D = rowsum(dO O). */
int some_train_attention_bwd_f16x3(SomeHandle* h, const float* qkv_split_dev, const float* qkv_t_split_dev,
                                   const float* dout_split_dev, const float* dout_t_split_dev, const float* out_dev,
                                   const float* dout_dev, const float* lse_dev, const int32_t* frame_offsets_dev,
                                   int32_t B, int32_t max_frames, int32_t M, int32_t Mp, int32_t hi_only,
                                   float* dqkv_dev, float* dsum_scratch_dev, void* stream);
/* The same with dq | dk | dv written as 16-bit values (the format of hi_only: 1 = f16, 2 = bf16) and multiplied by *out_scale_dev on the
 * way out (a device scalar - the inverse of the power of two the caller multiplied dO with): the operand of the projection's data- and
 * weight-gradient GEMMs (some_train_gemm16s / some_train_gemm16_wgrad16) without an fp32 array, a rescaling pass and a cast pass. */
int some_train_attention_bwd_f16x3_out16(SomeHandle* h, const float* qkv_split_dev, const float* qkv_t_split_dev,
                                         const float* dout_split_dev, const float* dout_t_split_dev, const float* out_dev,
                                         const float* dout_dev, const float* lse_dev, const int32_t* frame_offsets_dev,
                                         int32_t B, int32_t max_frames, int32_t M, int32_t Mp, int32_t hi_only,
                                         void* dqkv16_dev, const float* out_scale_dev, float* dsum_scratch_dev, void* stream);

/* Both operand layouts of the split attention kernels in one pass over x [M, N] (N % 32 == 0): rows_split = some_op_split_rows_fmt(x),
 * t_split [N, Mp] = some_train_transpose(x, split_out) - bit for bit. */
int some_train_split_transpose(SomeHandle* h, const float* x_dev, int32_t M, int32_t N, float* rows_split_dev, float* t_split_dev, int32_t Mp,
                               int32_t format, void* stream);
/* some_train_attention_bwd_f16x3_out16 taking dO itself [M, 512] fp32: the power-of-two factor 2^floor(10 - log2 max|dO|) (f16 halves have an
 * absolute floor of 2^-25), both split layouts of factor * dO, D = rowsum(dO O) and the 1 / factor on the way out are made on the device inside
 * this call (5 launches, no host synchronisation).  work_dev: >= some_train_attention_bwd16_work_bytes(h, M, Mp) bytes, 256-byte aligned. */
size_t some_train_attention_bwd16_work_bytes(const SomeHandle* h, int32_t M, int32_t Mp);
int some_train_attention_bwd_f16x3_auto16(SomeHandle* h, const float* qkv_split_dev, const float* qkv_t_split_dev, const float* out_dev,
                                          const float* dout_dev, const float* lse_dev, const int32_t* frame_offsets_dev, int32_t B,
                                          int32_t max_frames, int32_t M, int32_t Mp, int32_t hi_only, void* dqkv16_dev, void* work_dev,
                                          size_t work_bytes, void* stream);

/* ---- single-operator entry points (kernel-level parity tests and micro-benchmarks) ------------------ */

#define SOME_EPI_NONE 0       /* C = A W^T                                                          */
#define SOME_EPI_BIAS 1       /* C = act(A W^T + b), act: 0 none / 1 sigmoid; optional row mask      */
#define SOME_EPI_BIAS_SILU 2  /* C = silu(A W^T + b)                       (Gconform.py:30-31)       */
#define SOME_EPI_BIAS_RES 3   /* C = res + alpha (A W^T + b)               (Gconform.py:57,60-62)    */
#define SOME_EPI_GLU 4        /* C = (a + ba) sigmoid(g + bg), W rows packed 32 a | 32 gate          */
#define SOME_EPI_GLU_RES 5    /* C = res + GLU(...), optional row mask     (Gconform.py:82-87)       */

/* One nn.Linear (+ fused epilogue) on the f32 matrix pipe.  A [M,lda], W [N,K] (K contiguous), C [M,ldc].
 * For the GLU epilogues N counts the packed rows (2 x output width) and C gets N/2 columns. */
#define SOME_GEMM_SPLIT_IN 1    /* A and W are in SPLIT32 format -> 3-term split-f16 kernel (K % 32 == 0)            */
#define SOME_GEMM_SPLIT_OUT 2   /* C written in SPLIT32 format (SOME_EPI_BIAS_SILU only)                            */
#define SOME_GEMM_HI_ONLY 4      /* with SPLIT_IN: use the f16 hi halves only - plain f16 x f16 -> fp32 (mixed-precision training); EPI_NONE / EPI_BIAS */
#define SOME_GEMM_HI_BF16 8      /* with HI_ONLY: the hi slots hold bf16 (SOME_OPERAND_BF16) - bf16 x bf16 -> fp32 */
#define SOME_GEMM_TILE(t) (((t) & 7) << 8)   /* split kernel tile: 0 = 128x128, 1 = 256x128, 2 = 256x256, 3 = DMA ring 128x256, 4 = 64x128 */
int some_op_gemm(SomeHandle* h, int32_t epilogue, const float* A_dev, int32_t lda, const float* W_dev,
                 const float* bias_dev, const float* res_dev, int32_t ldr, float* C_dev, int32_t ldc,
                 int32_t M, int32_t N, int32_t K, float alpha, int32_t act, const uint8_t* row_mask_dev,
                 int32_t flags, void* stream);
/* fp32 rows [rows, K] -> SPLIT32 format (per 32-element k-block: 32 f16 hi | 32 f16 lo; same byte size). */
int some_op_split_rows(SomeHandle* h, const float* x_dev, float* out_dev, int64_t rows, int32_t K, void* stream);
#define SOME_OPERAND_F16X2 0     /* hi = f16(x), lo = f16(x - hi): operands of the 3-product and the f16 one-product kernels */
#define SOME_OPERAND_BF16 1      /* hi slot = bf16(x), lo = 0: operands of the bf16 one-product kernels (hi_only = 2 / SOME_GEMM_HI_BF16) */
int some_op_split_rows_fmt(SomeHandle* h, const float* x_dev, float* out_dev, int64_t rows, int32_t K, int32_t format, void* stream);
/* nn.LayerNorm(512), eps 1e-5 (Gconform.py:49-53): y = LN(x) * gamma + beta, rows of 512. */
int some_op_layernorm(SomeHandle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                      float* y_dev, float* y_split_dev, int32_t M, void* stream);   /* either output may be NULL */
/* Unmasked per-clip attention, 8 heads x 64, scale 1/8 (base_attention.py:34-44): qkv [M,1536] -> out [M,512]. */
int some_op_attention(SomeHandle* h, const float* qkv_dev, const int32_t* frame_offsets_dev, int32_t B,
                      int32_t max_frames, float* out_dev, int32_t out_split, void* stream);
/* Split-f16 pair: QKV projection (h [M,512] SPLIT32 x Wqkv [1536,512] SPLIT32) writing Q | K SPLIT32 planes and V
 * transposed, followed by the split-f16 flash attention (attention_f16x3.hip); out [M,512] SPLIT32.
 * Operands are laid out in clip-aligned rows (every clip's keys start a 64-key tile: a clip's result is bit-identical
 * whatever else is in the batch - inference/base_infer.py:46-53 runs every chunk alone).
 * workspace: >= some_op_qkv_attention_f16x3_bytes(M, B) bytes, 256-byte aligned. */
size_t some_op_qkv_attention_f16x3_bytes(int32_t M, int32_t B);
int some_op_qkv_attention_f16x3(SomeHandle* h, const float* h_split_dev, const float* wqkv_split_dev,
                                const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, int32_t M,
                                float* out_split_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Depthwise k=31 conv (taps [31,512], BatchNorm already folded) + bias + SiLU, per-clip zero padding
 * (base_conv.py:66-68): x [M,512] -> y [M,512]. */
int some_op_dwconv_silu(SomeHandle* h, const float* x_dev, const float* taps_dev, const float* bias_dev,
                        const int32_t* frame_offsets_dev, int32_t B, int32_t max_frames, float* y_dev,
                        int32_t out_split, void* stream);

/* ---- per-kernel timing (measurement only; off by default) ----------------------------------------- */

typedef struct SomeKernelStat {
    char name[48];          /* kernel family, e.g. "gemm_bias_silu[512->2048]"                         */
    int64_t launches;
    double total_ms;        /* sum of hipEventElapsedTime over the launches since the last reset       */
    double flops;           /* algorithmic FLOPs summed over those launches (0 for byte-bound kernels) */
    double bytes;           /* algorithmic HBM bytes summed over those launches                        */
} SomeKernelStat;

/* on != 0: bracket every kernel launch of subsequent calls with hipEvents on the launch stream. */
int some_profile_enable(SomeHandle* h, int32_t on);
/* Synchronises the recorded events, accumulates and returns up to max_stats entries; resets the log. */
int some_profile_collect(SomeHandle* h, SomeKernelStat* stats, int32_t max_stats, int32_t* n_stats);

/* ---- box calibration (measurement only) -------------------------------------------------------------- */

/* What THIS GPU sustains on the two resources the hot path is bound by, for comparing bench lines across boxes (bench.py `box`):
 * a pure v_mfma_f32_32x32x16_f16 stream on random operands under the package power limit (issued TFLOP/s and the effective clock
 * it implies: TF / (1024 SIMDs x 1024 FLOP per SIMD-cycle)) and a float4 copy of 1 GiB (GB/s, read + written).  Runs for about
 * `seconds` (0 < seconds <= 30) on `stream`, allocates ~2 GiB of its own for the duration of the call, synchronises. */
int some_box_calibrate(double seconds, double* mfma_tflops, double* mfma_mhz, double* copy_gbs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SOME_AMD_H */
