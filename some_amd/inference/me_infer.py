"""``MIDIExtractionInference`` (reference inference/me_infer.py:15-97) on the HIP kernels.

The three per-clip methods keep the reference's tensor contracts; ``infer`` (and ``infer_batch``) run whole
lists of clips as packed var-len batches: one log-mel launch, one forward, one decode, one D2H copy."""
import pathlib
from typing import Dict, List

import numpy as np
import torch

from .. import _lib
from ..engine import ClipBatch
from .base_infer import BaseInference


class MIDIExtractionInference(BaseInference):
    quantized = False
    head_mode = _lib.HEAD_SIGMOID          # forward_model passes sig=True (me_infer.py:70)
    max_batch_frames = 131072              # frames per packed device batch in infer()

    def __init__(self, config: dict, model_path: pathlib.Path, device=None):
        super().__init__(config, model_path, device=device)
        self.engine = self.model.engine    # shares the front-end tables and the decode kernels
        self.mel_spec = None               # the reference keeps a MelSpectrogram module here (me_infer.py:18-22)
        self.rmvpe = None
        self.midi_min = self.config['midi_min']
        self.midi_max = self.config['midi_max']
        self.midi_deviation = self.config.get('midi_prob_deviation', 1.0)
        self.rest_threshold = self.config.get('rest_threshold', 0.1)
        self._pinned = {}
        self._copy_stream = None

    # ---- reference-shaped per-clip API -------------------------------------------------------------
    def preprocess(self, waveform: np.ndarray) -> Dict[str, torch.Tensor]:
        """me_infer.py:29-63: waveform [L] -> units [1,T,80], pitch zeros [1,T], masks ones [1,T]."""
        wav = torch.from_numpy(np.ascontiguousarray(waveform, dtype=np.float32)).to(self.device)
        batch = ClipBatch.from_sample_counts([wav.numel()], self.engine.hop, self.engine.device)
        units = self.engine.logmel(wav, batch).unsqueeze(0)
        pitch = torch.zeros(units.shape[:2], dtype=torch.float32, device=self.device)
        return {'units': units, 'pitch': pitch, 'masks': torch.ones_like(pitch, dtype=torch.bool)}

    @torch.no_grad()
    def forward_model(self, sample: Dict[str, torch.Tensor]):
        """me_infer.py:65-76."""
        probs, bounds = self.model(x=sample['units'], f0=sample['pitch'], mask=sample['masks'],
                                   sig=not self.quantized, softmax=self.quantized)
        return {'probs': probs, 'bounds': bounds, 'masks': sample['masks']}

    def postprocess(self, results: Dict[str, torch.Tensor]) -> Dict[str, np.ndarray]:
        """me_infer.py:78-97 / me_quant_infer.py:22-38 for a [1,T,...] result."""
        probs, bounds, masks = results['probs'], results['bounds'], results['masks']
        b, t = bounds.shape
        batch = ClipBatch([t] * b, self.engine.device)
        out = self.engine.decode(probs.reshape(b * t, -1).contiguous(), bounds.reshape(-1).contiguous(), batch,
                                 quantized=self.quantized, mask=masks)
        return self._collect(out, batch)[0]

    # ---- batched path -----------------------------------------------------------------------------
    def _collect(self, out: Dict[str, torch.Tensor], batch: ClipBatch) -> List[Dict[str, np.ndarray]]:
        n_notes = out['n_notes'].cpu().numpy()
        midi = out['note_midi'].cpu().numpy()
        dur = out['note_dur'].cpu().numpy()
        rest = out['note_rest'].cpu().numpy()
        res = []
        for b in range(batch.B):
            s, n = int(batch.frame_offsets[b]), int(n_notes[b])
            res.append({
                'note_midi': midi[s:s + n].copy(),
                'note_dur': dur[s:s + n] * self.timestep,            # int64 * python float -> float64 (me_infer.py:95)
                'note_rest': rest[s:s + n].astype(bool),
            })
        return res

    # ---- host ingest pipeline: pinned double buffers, copy stream, deferred D2H ----
    def _stage(self, waveforms: List[np.ndarray], slot: int):
        """Pack the clips into a pinned host buffer (one memcpy per clip, no intermediate concatenate) and start the
        H2D copy on the copy stream.  Returns (audio_dev, batch, ready_event)."""
        lens = [int(w.shape[0]) for w in waveforms]
        total = int(sum(lens))
        batch = ClipBatch.from_sample_counts(lens, self.engine.hop, self.engine.device)
        pin = self._pinned.get(slot)
        if pin is None or pin.numel() < total:
            pin = self._pinned[slot] = torch.empty(max(total, 1), dtype=torch.float32).pin_memory()
        view = pin.numpy()
        pos = 0
        for w, n in zip(waveforms, lens):
            np.copyto(view[pos:pos + n], w, casting='same_kind')
            pos += n
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._copy_stream):
            audio = pin[:total].to(self.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        return audio, batch, ready

    def _launch(self, audio, batch, ready):
        """Enqueue front end + network + decode for one staged batch on the current stream (no host sync)."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        audio.record_stream(cur)
        units = self.engine.logmel(audio, batch)
        probs, bounds = self.engine.forward(units, batch, mask=None,
                                            head_mode=_lib.HEAD_SOFTMAX if self.quantized else _lib.HEAD_SIGMOID)
        out = self.engine.decode(probs, bounds, batch, quantized=self.quantized)
        finite = torch.isfinite(bounds).all()
        return out, finite, {'units': units, 'probs': probs, 'bounds': bounds, 'batch': batch}

    def _finish(self, out, finite, batch) -> List[Dict[str, np.ndarray]]:
        res = self._collect(out, batch)                 # the only host synchronisation of the batch
        if not bool(finite):
            raise FloatingPointError(
                'non-finite model outputs: with some_amd_precision=f16x3 every GEMM input must stay below 65504; '
                "set `some_amd_precision: f32` in config.yaml (or SOME_AMD_PRECISION=f32) for exact-fp32 GEMMs")
        return res

    @torch.no_grad()
    def infer_batch(self, waveforms: List[np.ndarray], return_device_outputs: bool = False):
        """All clips in ONE packed device batch.  Results equal running the clips one by one."""
        if not waveforms:
            return []
        audio, batch, ready = self._stage(waveforms, 0)
        out, finite, dev = self._launch(audio, batch, ready)
        res = self._finish(out, finite, batch)
        if return_device_outputs:
            return res, dev
        return res

    @torch.no_grad()
    def infer(self, waveforms: List[np.ndarray]) -> List[Dict[str, np.ndarray]]:
        """base_infer.py:46-53 semantics, executed as packed batches of at most ``max_batch_frames`` frames.  While
        the GPU works on batch i the host packs batch i + 1 into the other pinned buffer and its H2D copy runs on
        the copy stream; results of batch i are read back after batch i + 1 has been enqueued."""
        groups: List[List[np.ndarray]] = []
        group, frames = [], 0
        for w in waveforms:
            t = 1 + int(w.shape[0]) // self.engine.hop
            if group and frames + t > self.max_batch_frames:
                groups.append(group)
                group, frames = [], 0
            group.append(w)
            frames += t
        if group:
            groups.append(group)
        results: List[Dict[str, np.ndarray]] = []
        pending = None
        for i, grp in enumerate(groups):
            audio, batch, ready = self._stage(grp, i & 1)
            launched = self._launch(audio, batch, ready)
            if pending is not None:
                results.extend(self._finish(pending[0], pending[1], pending[2]))
            pending = (launched[0], launched[1], batch)
        if pending is not None:
            results.extend(self._finish(pending[0], pending[1], pending[2]))
        return results
