"""WAV decoding standing in for ``librosa.load(path, sr=config['audio_sample_rate'], mono=True)`` (call sites
infer.py:34, batch_infer.py:51).  librosa is absent; for PCM / float WAV files already at the target rate that
call is decode-to-float32 (+ channel mean).  Files at another rate are resampled the way the pinned
``librosa<0.10.0`` (requirements.txt:10) does by default - ``res_type='kaiser_best'`` = resampy's band-limited sinc
interpolation - restated below from the published algorithm ("parity unpinned": neither package is available here to
compare against; the test checks it against analytic sinusoids and scipy's polyphase resampler)."""
import functools

import numpy as np

# resampy 'kaiser_best': 64 zero crossings, 2**9 table entries per crossing, Kaiser(beta) taper, roll-off at 0.9476 Nyquist
_KB_ZEROS, _KB_PRECISION, _KB_ROLLOFF, _KB_BETA = 64, 9, 0.9475937167399596, 14.769656459379492


@functools.lru_cache(maxsize=1)
def _kaiser_best():
    """resampy.filters.sinc_window: right wing of the windowed sinc, tabulated at 2**precision points per zero crossing."""
    num_bits = 2 ** _KB_PRECISION
    n = num_bits * _KB_ZEROS
    sinc_win = _KB_ROLLOFF * np.sinc(_KB_ROLLOFF * np.linspace(0, _KB_ZEROS, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, _KB_BETA)[n:]
    return taper * sinc_win, num_bits


def resample(y: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    """librosa.resample(y, orig_sr, target_sr) with the 0.9.x defaults (res_type='kaiser_best', fix=True, scale=False):
    resampy.resample (interpn.resample_f: per output sample, left and right filter wings walked through the table with
    linear interpolation between entries) followed by fix_length to ceil(len * ratio).  float32 in, float32 out."""
    if orig_sr == target_sr:
        return np.ascontiguousarray(y, dtype=np.float32)
    x = np.asarray(y, dtype=np.float32)
    ratio = float(target_sr) / float(orig_sr)
    n_orig = x.shape[-1]
    n_out = int(n_orig * ratio)
    interp_win, num_table = _kaiser_best()
    interp_win = interp_win.copy()
    if ratio < 1:
        interp_win *= ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    out = np.zeros(n_out, dtype=np.float32)
    xd = x.astype(np.float64)
    chunk = 1 << 16
    for t0 in range(0, n_out, chunk):
        t = np.arange(t0, min(t0 + chunk, n_out), dtype=np.float64)
        time_register = t * time_increment
        n = time_register.astype(np.int64)
        acc = np.zeros(t.shape[0], dtype=np.float64)
        for wing in (0, 1):
            frac = scale * (time_register - n)
            if wing:
                frac = scale - frac
            index_frac = frac * num_table
            offset = index_frac.astype(np.int64)
            eta = index_frac - offset
            reach = (nwin - offset) // index_step                       # taps of this wing inside the table
            limit = np.minimum(n + 1, reach) if wing == 0 else np.minimum(n_orig - n - 1, reach)
            for i in range(int(limit.max()) if limit.size else 0):
                ok = i < limit
                idx = np.where(ok, offset + i * index_step, 0)
                w = interp_win[idx] + eta * interp_delta[idx]
                src = np.where(ok, n - i if wing == 0 else n + i + 1, 0)
                acc += np.where(ok, w * xd[src], 0.0)
        out[t0:t0 + t.shape[0]] = acc.astype(np.float32)
    n_fix = int(np.ceil(n_orig * ratio))                               # librosa.util.fix_length
    if n_fix > n_out:
        out = np.concatenate([out, np.zeros(n_fix - n_out, dtype=np.float32)])
    return out[:n_fix]


def load_wav(path, sr: int, mono: bool = True):
    from scipy.io import wavfile
    file_sr, data = wavfile.read(str(path))
    return _decode(path, file_sr, data, sr, mono)


def _decode(path, file_sr: int, data: np.ndarray, sr: int, mono: bool):
    if data.dtype == np.int16:
        y = data.astype(np.float32) / np.float32(32768.0)
    elif data.dtype == np.int32:
        y = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        y = (data.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif data.dtype in (np.float32, np.float64):
        y = data.astype(np.float32)
    else:
        raise ValueError(f'unsupported WAV sample type {data.dtype}')
    if y.ndim > 1:
        y = y.T                      # librosa layout [channels, samples]
        if mono:
            y = np.mean(y, axis=0)
    y = np.ascontiguousarray(y, dtype=np.float32)
    if file_sr != sr:
        if y.ndim > 1:
            y = np.stack([resample(ch, file_sr, sr) for ch in y])
        else:
            y = resample(y, file_sr, sr)
    return y, sr


def load_pcm(path, sr: int):
    """The same file as ``load_wav(path, sr, mono=True)`` but WITHOUT the host-side sample conversion when the payload
    is mono int16 PCM: returns the int16 samples as stored (value = x / 32768, converted on the device by
    some_pcm_gather / some_slicer_rms).  Any other layout goes through ``load_wav`` and comes back float32."""
    from scipy.io import wavfile
    file_sr, data = wavfile.read(str(path))
    if data.dtype == np.int16 and data.ndim == 1 and file_sr == sr:
        return np.ascontiguousarray(data), sr
    return _decode(path, file_sr, data, sr, mono=True)


def save_wav(path, y: np.ndarray, sr: int):
    from scipy.io import wavfile
    pcm = np.clip(np.round(np.asarray(y, dtype=np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    wavfile.write(str(path), sr, pcm)
