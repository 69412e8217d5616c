"""WAV decoding standing in for ``librosa.load(path, sr=config['audio_sample_rate'], mono=True)`` (call sites
infer.py:34, batch_infer.py:51).  librosa is absent; for PCM / float WAV files already at the target rate that
call is decode-to-float32 (+ channel mean), which is what is implemented.  Resampling is not (librosa would use
a polyphase / soxr resampler whose exact arithmetic is unpinned): a different source rate raises."""
import numpy as np


def load_wav(path, sr: int, mono: bool = True):
    from scipy.io import wavfile
    file_sr, data = wavfile.read(str(path))
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
    if file_sr != sr:
        raise NotImplementedError(
            f'{path}: sample rate {file_sr} != {sr}; resampling is not implemented (convert the file to {sr} Hz first)')
    return np.ascontiguousarray(y, dtype=np.float32), sr


def load_pcm(path, sr: int):
    """The same file as ``load_wav(path, sr, mono=True)`` but WITHOUT the host-side sample conversion when the payload
    is mono int16 PCM: returns the int16 samples as stored (value = x / 32768, converted on the device by
    some_pcm_gather / some_slicer_rms).  Any other layout goes through ``load_wav`` and comes back float32."""
    from scipy.io import wavfile
    file_sr, data = wavfile.read(str(path))
    if data.dtype == np.int16 and data.ndim == 1 and file_sr == sr:
        return np.ascontiguousarray(data), sr
    return load_wav(path, sr, mono=True)


def save_wav(path, y: np.ndarray, sr: int):
    from scipy.io import wavfile
    pcm = np.clip(np.round(np.asarray(y, dtype=np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    wavfile.write(str(path), sr, pcm)
