"""Cached flat weight file: the packed device arena of a checkpoint (BatchNorm folded, QKV
concatenated, GLU rows interleaved, SPLIT32 in f16x3 mode) is written next to the ``.ckpt`` after the first strict load
and reused while the checkpoint file is unchanged, so a process start costs one sequential read instead of
``torch.load`` + ``some_pack_weights``.

File: 64-byte header (magic, format version, arena element count, precision, checkpoint size and mtime in ns, a CRC of
the hot-path config keys, a CRC of the ``libsome_amd.so`` that packed it) followed by the raw little-endian fp32 arena.
The library CRC ties a cache file to the exact binary whose ``some_pack_weights`` laid it out: any rebuild that could
have changed the arena layout or the folding arithmetic invalidates every cache by itself, with no version to bump.  Any mismatch - or any I/O error - means "no
cache": the caller packs from the checkpoint as before."""
import os
import pathlib
import struct
import zlib
from typing import Optional

import numpy as np

MAGIC = b'SOMEAMD1'
_HEADER = struct.Struct('<8sIQIQQII16x')         # magic, version, numel, precision, ckpt size, ckpt mtime_ns, config crc, library crc
VERSION = 3
_lib_crc_cache = {}


def _library_crc() -> int:
    from . import _lib
    path = pathlib.Path(_lib.LIB_PATH)
    st = path.stat()
    key = (str(path), st.st_size, st.st_mtime_ns)
    if key not in _lib_crc_cache:
        _lib_crc_cache.clear()
        _lib_crc_cache[key] = zlib.crc32(path.read_bytes())
    return _lib_crc_cache[key]


def _config_crc(config: dict) -> int:
    a = config.get('midi_extractor_args', {})
    key = repr((a.get('lay'), a.get('dim'), a.get('attention_heads'), a.get('attention_heads_dim'), a.get('kernel_size'),
                config.get('units_dim'), config.get('midi_num_bins')))
    return zlib.crc32(key.encode())


def cache_path(ckpt: pathlib.Path, precision: int) -> pathlib.Path:
    return ckpt.with_name(ckpt.name + f'.some_amd-p{precision}.arena')


def load(ckpt: pathlib.Path, numel: int, precision: int, config: dict) -> Optional[np.ndarray]:
    path = cache_path(pathlib.Path(ckpt), precision)
    try:
        st = os.stat(ckpt)
        with open(path, 'rb') as f:
            head = f.read(_HEADER.size)
            if len(head) != _HEADER.size:
                return None
            if _HEADER.unpack(head) != (MAGIC, VERSION, numel, precision, st.st_size, st.st_mtime_ns, _config_crc(config), _library_crc()):
                return None
            arena = np.fromfile(f, dtype='<f4', count=numel)
        return arena if arena.shape[0] == numel else None
    except OSError:
        return None


def store(ckpt: pathlib.Path, arena: np.ndarray, precision: int, config: dict) -> bool:
    """Best effort (read-only checkpoint directories are fine): write to a temporary name, then rename."""
    path = cache_path(pathlib.Path(ckpt), precision)
    tmp = path.with_name(path.name + f'.tmp{os.getpid()}')
    try:
        st = os.stat(ckpt)
        with open(tmp, 'wb') as f:
            f.write(_HEADER.pack(MAGIC, VERSION, int(arena.shape[0]), precision, st.st_size, st.st_mtime_ns, _config_crc(config), _library_crc()))
            np.ascontiguousarray(arena, dtype='<f4').tofile(f)
        os.replace(tmp, path)
        return True
    except OSError:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        return False
