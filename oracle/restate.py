"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's inference hot path.  It exists to
check the HIP kernels; it is never shipped, never imported by the ``some_amd`` package, and only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may call it.

Parity pin: every function below is checked by ``tests/test_oracle_golden.py`` against
``tests/golden/*.npz``, which were produced by running the REFERENCE'S OWN modules from
``/root/reference`` (``oracle/make_golden.py``, committed).  The reference has no tests or golden vectors
of its own; the only textual known-answer, the commented example at
``utils/infer_utils.py:103-113``, is included in the goldens.  Third-party arithmetic the reference imports
but does not vendor (``librosa.filters.mel``; pinned ``librosa<0.10.0`` in ``requirements.txt:10``) is
restated from its published algorithm; the reference itself has no test that asserts its values, so it is pinned
to an independent third-party implementation instead (``transformers.audio_utils.mel_filter_bank``, written to
reproduce librosa: same support, every weight within one fp32 ulp - ``tests/test_oracle_golden.py``); see
``mel_filterbank``.

Arithmetic: fp32 torch-CPU ops for the network (same ATen kernels the reference's CPU path runs),
explicit sequential numpy for the integer/decode logic so results are machine independent.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------
# front end  (reference modules/rmvpe/spec.py)
# --------------------------------------------------------------------------------------------------


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(sr=44100, n_fft=2048, n_mels=80, fmin=40.0, fmax=8000.0) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=True, norm='slaney') -> float32 [n_mels, 1+n_fft/2].

    Call site: reference modules/rmvpe/spec.py:22-28.  Published algorithm (librosa 0.9.x filters.py):
    n_mels+2 points equally spaced on the HTK mel scale, triangular ramps evaluated on
    ``fftfreqs = linspace(0, sr/2, 1+n_fft/2)``, each band scaled by ``2/(f[i+2]-f[i])`` (Slaney area
    normalisation).  The two fp32 roundings of librosa's own code (row assignment into a float32 array, then the
    in-place scale) are reproduced, so the basis is librosa's bit for bit as far as its published source goes;
    librosa itself is absent here; the independent pin is ``transformers.audio_utils.mel_filter_bank(norm='slaney',
    mel_scale='htk')`` (an fp64 re-implementation of the same published filters): identical support, 193 of 82 000 weights
    one fp32 ulp apart (``test_mel_filterbank_equals_an_independent_librosa_compatible_implementation``).
    """
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins)
    mel_f = mel_to_hz_htk(np.linspace(hz_to_mel_htk(fmin), hz_to_mel_htk(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    # librosa 0.9.x allocates the result in its `dtype` (float32) up front: every triangle row is rounded to fp32 when it
    # is assigned, and the Slaney scale is then applied IN PLACE (fp32 row x fp64 enorm, product rounded to fp32 again)
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def logmel(audio: np.ndarray, config: dict, dtype=torch.float32, keep_dtype=False, keyshift=0, speed=1,
           center=True) -> np.ndarray:
    """reference modules/rmvpe/spec.py:38-72, then the transpose of inference/me_infer.py:31.
    audio [L] -> units [T, n_mels]; T = 1 + L // hop for the inference configuration (keyshift=0, speed=1,
    center=True); keyshift / speed are the training-data augmentation (preprocessing/me_binarizer.py:235-246)."""
    win, hop = config['win_size'], config['hop_size']
    basis = torch.from_numpy(mel_filterbank(config['audio_sample_rate'], win, config['units_dim'],
                                            config['fmin'], config['fmax'])).to(dtype)
    factor = 2 ** (keyshift / 12)                                              # spec.py:39-42
    n_fft_new = int(np.round(win * factor))
    win_new = int(np.round(win * factor))
    hop_new = int(np.round(hop * speed))
    x = torch.from_numpy(np.ascontiguousarray(audio)).to(dtype)[None]
    if center:
        x = F.pad(x, (win_new // 2, (win_new + 1) // 2))                       # spec.py:47-50
    window = torch.hann_window(win_new, dtype=dtype)                           # spec.py:44-46 (periodic)
    spec = torch.stft(x, n_fft=n_fft_new, hop_length=hop_new, win_length=win_new, window=window,
                      center=False, return_complex=True)                       # spec.py:52-60
    mag = spec.abs()                                                           # spec.py:61
    if keyshift != 0:                                                          # spec.py:63-68
        size = win // 2 + 1
        if mag.size(1) < size:
            mag = F.pad(mag, (0, 0, 0, size - mag.size(1)))
        mag = mag[:, :size, :] * win / win_new
    mel = torch.matmul(basis, mag)                                             # spec.py:70
    out = torch.log(torch.clamp(mel, min=1e-5))                                # spec.py:71
    out = out[0].transpose(0, 1).contiguous()
    return (out if keep_dtype else out.to(torch.float32)).numpy()       # keep_dtype: the fp64 yardstick of the error-budget test


# --------------------------------------------------------------------------------------------------
# network  (reference modules/conform/Gconform.py, modules/attention/base_attention.py,
#           modules/conv/base_conv.py, modules/model/Gmidi_conform.py)
# --------------------------------------------------------------------------------------------------


def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _ffn(sd, p, x):
    """Gconform.py:29-34 (dropouts are identity in eval)."""
    x = F.linear(x, _t(sd, p + '.ln1.weight'), _t(sd, p + '.ln1.bias'))
    x = F.silu(x)
    return F.linear(x, _t(sd, p + '.ln2.weight'), _t(sd, p + '.ln2.bias'))


def _attention(sd, p, x, heads):
    """base_attention.py:23-46: bias-free q / kv projections, 8 heads, SDPA (scale = head_dim**-0.5,
    no mask - Gconform.py:56-60 never passes one), merged heads, to_out with bias."""
    b, t, _ = x.shape
    q = F.linear(x, _t(sd, p + '.to_q.weight'))
    k, v = F.linear(x, _t(sd, p + '.to_kv.weight')).chunk(2, dim=2)
    q, k, v = (z.reshape(b, t, heads, -1).permute(0, 2, 1, 3) for z in (q, k, v))
    # the reference calls F.scaled_dot_product_attention (base_attention.py:41-43; default scale = head_dim ** -0.5); on
    # CPU that is the fused flash kernel - the explicit softmax(q k^T) v form materialises [8, T, T] scores and made this
    # port 2.3 x slower than the reference it stands in for as `cpu_baseline` (round-2 cross-check)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.permute(0, 2, 1, 3).reshape(b, t, -1)
    return F.linear(o, _t(sd, p + '.to_out.0.weight'), _t(sd, p + '.to_out.0.bias'))


def _conv_module(sd, p, x):
    """base_conv.py:63-70: pw1 -> GLU(channels) -> depthwise k31 pad15 -> BatchNorm1d(eval) -> SiLU -> pw2."""
    x = x.transpose(1, 2)
    x = F.conv1d(x, _t(sd, p + '.pointwise_conv1.weight'), _t(sd, p + '.pointwise_conv1.bias'))
    a, g = x.chunk(2, dim=1)
    x = a * torch.sigmoid(g)
    w = _t(sd, p + '.depthwise_conv.weight')
    x = F.conv1d(x, w, _t(sd, p + '.depthwise_conv.bias'), padding=(w.shape[-1] - 1) // 2, groups=w.shape[0])
    x = F.batch_norm(x, _t(sd, p + '.norm.running_mean'), _t(sd, p + '.norm.running_var'),
                     _t(sd, p + '.norm.weight'), _t(sd, p + '.norm.bias'), training=False, eps=1e-5)
    x = F.silu(x)
    x = F.conv1d(x, _t(sd, p + '.pointwise_conv2.weight'), _t(sd, p + '.pointwise_conv2.bias'))
    return x.transpose(1, 2)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), _t(sd, p + '.weight'), _t(sd, p + '.bias'), eps=1e-5)


def conformer_block(sd, p, x, heads):
    """Gconform.py:56-63."""
    x = _ffn(sd, p + '.ffn1', _ln(sd, p + '.norm1', x)) * 0.5 + x
    x = _attention(sd, p + '.att', _ln(sd, p + '.norm2', x), heads) + x
    x = _conv_module(sd, p + '.conv', _ln(sd, p + '.norm3', x)) + x
    x = _ffn(sd, p + '.ffn2', _ln(sd, p + '.norm4', x)) * 0.5 + x
    return _ln(sd, p + '.norm5', x)


def _glu_gate(sd, p, x):
    """Gconform.py:79-80: Linear(dim, 2*dim) then GLU over the last dim (out * sigmoid(gate))."""
    y = F.linear(x, _t(sd, p + '.0.weight'), _t(sd, p + '.0.bias'))
    a, g = y.chunk(2, dim=-1)
    return a * torch.sigmoid(g)


@torch.no_grad()
def model_forward(sd, config, units, mask=None, softmax=False, sig=False):
    """midi_conforms.forward (Gmidi_conform.py:30-40) over Gmidi_conform.forward (Gconform.py:119-140).

    sd: state dict with the reference's keys (``model.inln.weight`` ...), numpy or torch values.
    units [B,T,indim] (or [T,indim]) fp32.  Returns (midi [B,T,outdim], bound [B,T]) as torch tensors.
    """
    a = config['midi_extractor_args']
    heads, lay = a['attention_heads'], a['lay']
    x_in = torch.as_tensor(units)
    squeeze = x_in.dim() == 2
    if squeeze:
        x_in = x_in[None]
    x = F.linear(x_in, _t(sd, 'model.inln.weight'), _t(sd, 'model.inln.bias'))
    x1 = F.linear(x_in, _t(sd, 'model.inln1.weight'), _t(sd, 'model.inln1.bias'))
    if mask is not None:
        mask = torch.as_tensor(mask).reshape(x.shape[0], x.shape[1]).bool()
        x = x.masked_fill(~mask.unsqueeze(-1), 0)
    for i in range(lay):
        p = f'model.cf_lay.{i}'
        m = conformer_block(sd, p + '.att1', x, heads)
        b = conformer_block(sd, p + '.att2', x1, heads)
        ms = _glu_gate(sd, p + '.glu1', m)
        bs = _glu_gate(sd, p + '.glu2', b)
        x, x1 = m + bs, b + ms                                                  # Gconform.py:82-87
        if mask is not None:
            x = x.masked_fill(~mask.unsqueeze(-1), 0)
    x = conformer_block(sd, 'model.att1', x, heads)
    x1 = conformer_block(sd, 'model.att2', x1, heads)
    bound = torch.sigmoid(F.linear(x1, _t(sd, 'model.cutheard.weight'), _t(sd, 'model.cutheard.bias'))).squeeze(-1)
    midi = F.linear(x, _t(sd, 'model.outln.weight'), _t(sd, 'model.outln.bias'))
    if sig:
        midi = torch.sigmoid(midi)
    if softmax:
        midi = F.softmax(midi, dim=2)
    if squeeze:
        midi, bound = midi[0], bound[0]
    return midi, bound


# --------------------------------------------------------------------------------------------------
# decode  (reference utils/infer_utils.py) - explicit sequential numpy, one clip at a time
# --------------------------------------------------------------------------------------------------


def _round_half_even_f32(x: np.ndarray) -> np.ndarray:
    return np.rint(x.astype(np.float32))  # numpy rint == IEEE roundTiesToEven == torch.round


def decode_bounds_to_alignment(bounds: np.ndarray) -> np.ndarray:
    """infer_utils.py:27-39 (use_diff=True).  torch-CPU ``cumsum`` of fp32 accumulates in fp64 and casts each
    prefix back to fp32 (ATen acc_type<float, /*cuda=*/false> = double), then round-half-even -> int64."""
    b = np.asarray(bounds, dtype=np.float32)
    step = _round_half_even_f32(np.cumsum(b.astype(np.float64)).astype(np.float32)).astype(np.int64)
    inc = np.diff(step, prepend=np.int64(-1)) > 0
    return np.cumsum(inc.astype(np.int64))


def decode_gaussian_blurred_probs(probs: np.ndarray, vmin, vmax, deviation, threshold):
    """infer_utils.py:9-24.  probs [T, N] fp32 -> (values [T] fp32, rest [T] bool).

    Summation order: ascending bin index inside the argmax-centred window, fp32 (the reference's
    ``torch.sum`` lane order is CPU-ISA dependent; the window holds <= 2*width+1 non-zero terms, so the
    two orders agree to 1 ulp of the sum - asserted against the goldens with that tolerance)."""
    p = np.asarray(probs, dtype=np.float32)
    t, n = p.shape
    interval = (vmax - vmin) / (n - 1)
    width = int(3 * deviation / interval)
    values = np.zeros(t, dtype=np.float32)
    rest = np.zeros(t, dtype=bool)
    for i in range(t):
        c = int(np.argmax(p[i]))  # first maximum, as torch.argmax
        s, e = max(c - width, 0), min(c + width + 1, n)
        ps = np.float32(0.0)
        ws = np.float32(0.0)
        for j in range(s, e):
            # idx_values = idx * interval + vmin: int64 * python float -> fp32 tensor (infer_utils.py:14)
            v = np.float32(np.float32(j) * np.float32(interval) + np.float32(vmin))
            ps = np.float32(ps + np.float32(p[i, j] * v))
            ws = np.float32(ws + p[i, j])
        values[i] = np.float32(ps / np.float32(ws + np.float32(ws == 0)))
        rest[i] = p[i, c] < np.float32(threshold)
    return values, rest


def decode_note_sequence(frame2item: np.ndarray, values: np.ndarray, masks: np.ndarray, threshold=0.5):
    """infer_utils.py:42-76 for one clip.  ``values`` is fp32 (continuous head) or int64 (quantised head,
    me_quant_infer.py:33-35).  Returns (item_values fp32 [N], item_dur int64 [N], item_masks bool [N]),
    N = frame2item.max().  scatter_add on CPU visits frames in ascending order -> sequential sums."""
    f2i = np.asarray(frame2item, dtype=np.int64)
    masks = np.asarray(masks, dtype=bool)
    n = int(f2i.max()) if f2i.size else 0
    space = n + 1
    is_int = np.issubdtype(np.asarray(values).dtype, np.integer)
    vals = np.asarray(values, dtype=np.int64 if is_int else np.float32)
    dur = np.zeros(space, dtype=np.int64)
    unmasked = np.zeros(space, dtype=np.int64)
    hist = np.zeros((space, 128), dtype=np.int64)
    vq = vals if is_int else _round_half_even_f32(vals).astype(np.int64)
    for t in range(f2i.shape[0]):
        dur[f2i[t]] += 1
        unmasked[f2i[t]] += int(masks[t])
        flat = f2i[t] * 128 + vq[t]               # the reference scatters into a flattened [space*128] row
        hist[flat // 128, flat % 128] += int(masks[t])
    with np.errstate(divide='ignore', invalid='ignore'):
        item_masks = (unmasked.astype(np.float32) / dur.astype(np.float32)) >= np.float32(threshold)
    center = np.argmax(hist, axis=1)              # first maximum (histogram.float().argmax)
    center[0] = 0                                  # F.pad(item_values_center, [1, 0]) puts 0 in slot 0
    center = center.astype(np.int64 if is_int else np.float32)
    valid = np.zeros(space, dtype=np.int64)
    acc = np.zeros(space, dtype=np.int64 if is_int else np.float32)
    for t in range(f2i.shape[0]):
        c = center[f2i[t]]
        if is_int:
            near = masks[t] and (np.float32(vals[t]) >= np.float32(c) - np.float32(0.5)) \
                and (np.float32(vals[t]) <= np.float32(c) + np.float32(0.5))
        else:
            near = masks[t] and (vals[t] >= np.float32(c - np.float32(0.5))) and (vals[t] <= np.float32(c + np.float32(0.5)))
        valid[f2i[t]] += int(near)
        if is_int:
            acc[f2i[t]] += vals[t] * int(near)
        else:
            acc[f2i[t]] = np.float32(acc[f2i[t]] + np.float32(vals[t] * np.float32(near)))
    denom = (valid + (valid == 0)).astype(np.float32)
    item_values = (acc.astype(np.float32) / denom).astype(np.float32)
    return item_values[1:], dur[1:], item_masks[1:]


def postprocess(probs, bounds, config, quantized=False, masks=None):
    """inference/me_infer.py:78-97 / inference/me_quant_infer.py:22-38 for one clip (probs [T,N], bounds [T])."""
    probs = np.asarray(probs, dtype=np.float32)
    bounds = np.asarray(bounds, dtype=np.float32)
    t = bounds.shape[0]
    masks = np.ones(t, dtype=bool) if masks is None else np.asarray(masks, dtype=bool)
    probs = probs * masks[:, None]
    bounds = bounds * masks
    f2i = decode_bounds_to_alignment(bounds) * masks
    if quantized:
        midi = np.argmax(probs, axis=-1).astype(np.int64)
        rest = midi == 128
        vals = np.clip(midi, 0, 127)
    else:
        vals, rest = decode_gaussian_blurred_probs(probs, config['midi_min'], config['midi_max'],
                                                   config['midi_prob_deviation'], config['rest_threshold'])
    item_values, item_dur, item_masks = decode_note_sequence(f2i, vals, ~rest & masks)
    timestep = config['hop_size'] / config['audio_sample_rate']
    return {
        'note_midi': item_values,
        'note_dur': item_dur * timestep,          # int64 * python float -> float64 (me_infer.py:95)
        'note_rest': ~item_masks,
        '_frame2item': f2i, '_values': vals, '_rest': rest,
    }


def infer_clip(sd, config, waveform: np.ndarray, quantized=None):
    """BaseInference.infer body for one waveform (base_infer.py:46-53): preprocess -> forward -> postprocess."""
    if quantized is None:
        quantized = config['task_cls'].endswith('QuantizedMIDIExtractionTask')
    import time
    t0 = time.perf_counter()
    units = logmel(waveform, config)
    t1 = time.perf_counter()
    probs, bounds = model_forward(sd, config, units, mask=np.ones(units.shape[0], dtype=bool),
                                  softmax=quantized, sig=not quantized)
    t2 = time.perf_counter()
    res = postprocess(probs.numpy(), bounds.numpy(), config, quantized=quantized)
    t3 = time.perf_counter()
    res['_units'], res['_probs'], res['_bounds'] = units, probs.numpy(), bounds.numpy()
    res['_stage_s'] = {'mel': t1 - t0, 'forward': t2 - t1, 'decode': t3 - t2}       # the CPU path's stages timed separately
    return res
