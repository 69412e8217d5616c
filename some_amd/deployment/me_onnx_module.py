"""Deployment twin: the waveform -> notes modules of the reference's ``deployment/``
package (``base_onnx_module.py:11-79``, ``me_onnx_module.py:8-39``, ``me_quant_onnx_module.py``) as callable
``nn.Module``s on the HIP kernels.  Same forward contract - ``forward(waveform [B, L]) -> (note_midi [B, N],
note_rest [B, N], note_dur [B, N] in seconds)`` - and the same numerics, which differ from the inference path in ONE
place: the STFT is ``torch.stft(center=True)``, i.e. REFLECT padding instead of zeros.  (``use_diff=False`` in the
alignment decode is equivalent to the default for non-negative bounds.)  ONNX export itself is out of scope."""
import pathlib
from collections import OrderedDict

import torch
from torch import nn

from .. import _lib
from ..engine import ClipBatch, Engine
from ..utils import build_object_from_class_name


class MelSpectrogram_ONNX(nn.Module):
    """deployment/base_onnx_module.py:38-79."""

    def __init__(self, n_mel_channels, sampling_rate, win_length, hop_length, n_fft=None, mel_fmin=0, mel_fmax=None, clamp=1e-5):
        super().__init__()
        if (n_fft or win_length) != win_length or clamp != 1e-5:
            raise NotImplementedError('n_fft == win_length and clamp == 1e-5 are the compiled configuration')
        self.hop_length, self.n_mel_channels = hop_length, n_mel_channels
        self._config = {'units_dim': n_mel_channels, 'audio_sample_rate': sampling_rate, 'win_size': win_length,
                        'hop_size': hop_length, 'fmin': mel_fmin, 'fmax': mel_fmax}
        self._engine = None

    @torch.no_grad()
    def forward(self, audio, center=True):
        if not center:
            raise NotImplementedError('center=False is not used by the deployment modules')
        if self._engine is None or self._engine.device != audio.device:
            self._engine = Engine(self._config, device=audio.device)
        b, length = audio.shape
        batch = ClipBatch.from_sample_counts([length] * b, self.hop_length, self._engine.device)
        units = self._engine.logmel(audio.to(torch.float32).reshape(-1).contiguous(), batch, reflect=True)
        return units.view(b, -1, self.n_mel_channels).transpose(1, 2)


class MIDIExtractionONNXModule(nn.Module):
    quantized = False

    def __init__(self, config: dict, model_path: pathlib.Path, device=None):
        super().__init__()
        if device is None:
            device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.config, self.model_path, self.device = config, model_path, device
        self.timestep = config['hop_size'] / config['audio_sample_rate']
        self.model = build_object_from_class_name(config['model_cls'], nn.Module, config=config).eval().to(device)
        sd = torch.load(model_path, map_location='cpu')['state_dict']
        self.model.load_state_dict(OrderedDict((k[6:], v) for k, v in sd.items() if k.startswith('model.')), strict=True)
        print(f"| load 'model' from '{model_path}'.")
        self.engine = self.model.engine

    @torch.no_grad()
    def forward(self, waveform: torch.Tensor):
        """me_onnx_module.py:23-39 / me_quant_onnx_module.py: padded [B, N] outputs, N = max notes in the batch."""
        eng = self.engine
        b, length = waveform.shape
        batch = ClipBatch.from_sample_counts([length] * b, eng.hop, eng.device)
        units = eng.logmel(waveform.to(device=eng.device, dtype=torch.float32).reshape(-1).contiguous(), batch, reflect=True)
        probs, bounds = eng.forward(units, batch, head_mode=_lib.HEAD_SOFTMAX if self.quantized else _lib.HEAD_SIGMOID)
        out = eng.decode(probs, bounds, batch, quantized=self.quantized)
        t = int(batch.frame_counts[0])
        n = int(out['n_notes'].max())
        idx = torch.arange(n, device=eng.device)[None, :]
        valid = idx < out['n_notes'][:, None]
        midi = torch.where(valid, out['note_midi'].view(b, t)[:, :n], torch.zeros((), device=eng.device))
        dur = torch.where(valid, out['note_dur'].view(b, t)[:, :n], torch.zeros((), dtype=torch.int64, device=eng.device))
        rest = torch.where(valid, out['note_rest'].view(b, t)[:, :n].bool(), torch.ones((), dtype=torch.bool, device=eng.device))
        return midi, rest, dur * self.timestep


class QuantizedMIDIExtractionONNXModule(MIDIExtractionONNXModule):
    quantized = True
