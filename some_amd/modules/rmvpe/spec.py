"""``MelSpectrogram`` with the reference's constructor / forward signature (modules/rmvpe/spec.py:8-72), backed
by the fused HIP front end: csrc/logmel.hip for the inference configuration (``keyshift=0, speed=1, center=True``),
csrc/logmel_shift.hip (arbitrary transform length) for the key-shift / speed augmentation the reference's
binarizers request (preprocessing/me_binarizer.py:235-246)."""
import numpy as np
import torch
from torch import nn

from ...engine import ClipBatch, Engine


class MelSpectrogram(nn.Module):
    def __init__(self, n_mel_channels, sampling_rate, win_length, hop_length, n_fft=None, mel_fmin=0,
                 mel_fmax=None, clamp=1e-5):
        super().__init__()
        n_fft = win_length if n_fft is None else n_fft
        if n_fft != win_length:
            raise NotImplementedError('n_fft != win_length is not supported')
        if clamp != 1e-5:
            raise NotImplementedError('clamp is compiled as 1e-5 (the reference default)')
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.win_length = win_length
        self.sampling_rate = sampling_rate
        self.n_mel_channels = n_mel_channels
        self.clamp = clamp
        self._config = {
            'units_dim': n_mel_channels, 'audio_sample_rate': sampling_rate, 'win_size': win_length,
            'hop_size': hop_length, 'fmin': mel_fmin, 'fmax': mel_fmax,
        }
        self._engine = None

    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0))
        if probe.device.type == 'cuda':
            self._engine = Engine(self._config, device=probe.device)
        return super()._apply(fn, recurse)

    @property
    def mel_basis(self) -> torch.Tensor:
        eng = self._engine or Engine(self._config, host_only=True)
        return torch.from_numpy(eng.mel_filterbank())

    @torch.no_grad()
    def forward(self, audio, keyshift=0, speed=1, center=True):
        """audio [B, L] fp32 -> log-mel [B, n_mels, T], T = 1 + L // hop (inference configuration)."""
        if self._engine is None or (audio.is_cuda and self._engine.device != audio.device):
            self._engine = Engine(self._config, device=audio.device)    # raises for CPU tensors: no fallback
        eng = self._engine
        b, length = audio.shape
        if keyshift != 0 or speed != 1 or not center:
            factor = 2 ** (keyshift / 12)                                 # spec.py:39-42
            n_fft_new = int(np.round(self.n_fft * factor))
            win_length_new = int(np.round(self.win_length * factor))
            hop_length_new = int(np.round(self.hop_length * speed))
            flat = audio.to(device=eng.device, dtype=torch.float32).reshape(-1).contiguous()
            units, _ = eng.logmel_shifted(flat, [length] * b, n_fft_new, win_length_new, hop_length_new, center=center,
                                          rescale=keyshift != 0)
            return units.view(b, -1, self.n_mel_channels).transpose(1, 2)
        batch = ClipBatch.from_sample_counts([length] * b, self.hop_length, eng.device)
        flat = audio.to(device=eng.device, dtype=torch.float32).reshape(-1).contiguous()
        units = eng.logmel(flat, batch)                                   # [B*T, n_mels]
        return units.view(b, -1, self.n_mel_channels).transpose(1, 2)     # reference layout [B, n_mels, T]
