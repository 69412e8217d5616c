"""Host restatement (numpy) of the dropout mask the fused FFN epilogues draw on the GPU (csrc/train_gemm16s.hip: drop_bits, the key
derivation in launch_gemm16s).  Not used by the training path - it documents the function and lets tests check the kernels' masks cell
by cell and the generator's statistics without a GPU.

Element (m, n) of an [M, N] activation is KEPT when its 16 bits are >= round(65536 p); rows 2 q and 2 q + 1 of a column share one
32-bit word (low half: the even row)."""
import numpy as np

_M64 = (1 << 64) - 1


def key_words(seed: int):
    """splitmix64 finaliser of the call site's seed -> (k0, k1)."""
    z = (seed + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    z ^= z >> 31
    return z & 0xffffffff, z >> 32


def drop_words(seed: int, cells: np.ndarray) -> np.ndarray:
    """32 pseudo-random bits per cell index (uint32 array): a two-multiply integer finaliser, second key word injected between rounds."""
    k0, k1 = key_words(seed)
    x = (cells.astype(np.uint64) + k0) & 0xffffffff
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xffffffff
    x ^= k1
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x.astype(np.uint32)


def threshold(p: float) -> int:
    return min(int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)), 65535) if p > 0 else 0


def ffn_keep_mask(seed: int, M: int, N: int, p: float) -> np.ndarray:
    """bool [M, N]: True where the element survives dropout."""
    m = np.arange(M, dtype=np.uint64)[:, None]
    n = np.arange(N, dtype=np.uint64)[None, :]
    words = drop_words(seed, ((m >> 1) * N + n) & 0xffffffff)
    bits = np.where((m & 1) == 1, words >> 16, words & 0xffff)
    return bits >= threshold(p)
