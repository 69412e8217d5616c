"""WAV decoding standing in for ``librosa.load(path, sr=config['audio_sample_rate'], mono=True)`` (call sites
infer.py:34, batch_infer.py:51).  librosa is absent; for PCM / float WAV files already at the target rate that
call is decode-to-float32 (+ channel mean).  Files at another rate are resampled the way the pinned
``librosa<0.10.0`` (requirements.txt:10) does by default - ``res_type='kaiser_best'`` = resampy's band-limited sinc
interpolation - restated below from the published algorithm ("parity unpinned": neither package is available here to
compare against; the test checks it against analytic sinusoids and scipy's polyphase resampler)."""
import functools
from struct import error as struct_error

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


class PcmPool:
    """Reusable int16 sample buffers for ``load_pcm``.  A fresh 2.6 MB array per 30 s file costs ~650 page faults when the kernel copies
    the file into it and an munmap (TLB shootdown to every thread of the process) when it dies; with 8 rank processes x 8 reader
    threads on one host that kernel time is what a file load consists of (tools/host_scaling_bench.py: 12 ms per file against 0.9 ms
    alone).  Buffers come in 1 Mi-sample steps, are handed out by ``take`` and come back through ``give`` once their samples have been
    staged for the device; ``max_bytes`` bounds what the pool keeps (beyond it ``take`` returns a plain array that ``give`` drops)."""
    STEP = 1 << 20

    def __init__(self, max_bytes: int = 6 << 30):
        import threading
        self.max_bytes = int(max_bytes)
        self.bytes = 0
        self._free = {}                  # capacity (samples) -> [arrays]
        self._owned = {}                 # id(base array) -> base array, for the views handed out
        self._lock = threading.Lock()

    def take(self, n: int) -> np.ndarray:
        cap = max(1, (n + self.STEP - 1) // self.STEP) * self.STEP
        with self._lock:
            lst = self._free.get(cap)
            if lst:
                base = lst.pop()
            elif self.bytes + 2 * cap <= self.max_bytes:
                base = np.empty(cap, dtype='<i2')
                self.bytes += 2 * cap
            else:
                return np.empty(n, dtype='<i2')
            self._owned[id(base)] = base
        return base[:n]

    def give(self, arr: np.ndarray):
        base = arr.base if arr.base is not None else arr
        with self._lock:
            if self._owned.pop(id(base), None) is not None:
                self._free.setdefault(base.shape[0], []).append(base)


def _read_pcm16_mono(path, sr: int, pool: 'PcmPool' = None):
    """Fast path of ``load_pcm``: a canonical RIFF / WAVE file whose ``fmt `` chunk says PCM (format tag 1), one channel, 16 bits,
    ``sr`` Hz is read with TWO system calls - the first 4 KiB for the chunk headers, then ``os.preadv`` of the ``data`` chunk straight
    into the result array.  Both release the GIL for the whole copy; scipy's reader walks the header through a dozen buffered
    ``fid.read`` calls and holds the GIL between them, which caps a rank's reader threads at one file at a time once its main thread is
    busy (tools/host_scaling_bench.py: 13 ms per file with 8 reader threads, 1.2 ms alone).  Returns None for anything else (other
    formats, extensible headers, a data chunk beyond the first 4 KiB or truncated): the caller falls back to scipy."""
    import os
    import struct
    fd = os.open(str(path), os.O_RDONLY)
    try:
        head = os.pread(fd, 4096, 0)
        if len(head) < 44 or head[:4] != b'RIFF' or head[8:12] != b'WAVE':
            return None
        pos, fmt_ok = 12, False
        while pos + 8 <= len(head):
            tag, size = head[pos:pos + 4], struct.unpack_from('<I', head, pos + 4)[0]
            if tag == b'fmt ':
                if size < 16 or pos + 24 > len(head):
                    return None
                fmt, ch, rate, _, _, bits = struct.unpack_from('<HHIIHH', head, pos + 8)
                if fmt != 1 or ch != 1 or bits != 16 or rate != sr:
                    return None
                fmt_ok = True
            elif tag == b'data':
                if not fmt_ok or size % 2 or size == 0 or pos + 8 + size > os.fstat(fd).st_size:
                    return None                    # (a streamed / truncated file: no buffer is allocated for a size the file does not have)
                data = pool.take(size // 2) if pool is not None else np.empty(size // 2, dtype='<i2')
                if os.preadv(fd, [memoryview(data).cast('B')], pos + 8) != size:
                    if pool is not None:
                        pool.give(data)
                    return None                    # truncated file: let scipy produce its own diagnosis
                return data
            pos += 8 + size + (size & 1)           # chunks are word aligned
        return None
    finally:
        os.close(fd)


def load_pcm(path, sr: int, pool: 'PcmPool' = None):
    """The same file as ``load_wav(path, sr, mono=True)`` but WITHOUT the host-side sample conversion when the payload
    is mono int16 PCM: returns the int16 samples as stored (value = x / 32768, converted on the device by
    some_pcm_gather / some_slicer_rms).  Any other layout goes through ``load_wav`` and comes back float32.
    ``pool``: take the sample buffer from a ``PcmPool`` (the caller gives it back with ``pool.give`` when done)."""
    try:
        data = _read_pcm16_mono(path, sr, pool)
    except (OSError, ValueError, struct_error):
        data = None
    if data is not None:
        return data, sr
    from scipy.io import wavfile
    file_sr, data = wavfile.read(str(path))
    if data.dtype == np.int16 and data.ndim == 1 and file_sr == sr:
        return np.ascontiguousarray(data), sr
    return _decode(path, file_sr, data, sr, mono=True)


def save_wav(path, y: np.ndarray, sr: int):
    from scipy.io import wavfile
    pcm = np.clip(np.round(np.asarray(y, dtype=np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    wavfile.write(str(path), sr, pcm)
