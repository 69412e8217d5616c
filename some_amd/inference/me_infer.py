"""``MIDIExtractionInference`` (reference inference/me_infer.py:15-97) on the HIP kernels.

The three per-clip methods keep the reference's tensor contracts; ``infer`` (and ``infer_batch``) run whole
lists of clips as packed var-len batches: one log-mel launch, one forward, one decode, one D2H copy."""
import pathlib
from typing import Dict, List, Tuple

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

    # ---- precision guard --------------------------------------------------------------------------------------
    def _guarded(self, fn, *args, **kw):
        """Run ``fn``; if the split-f16 path reports a range overflow and the precision was not pinned by the user
        (config key / environment), rebuild the model in the exact-f32 mode ONCE, say so, and run again."""
        import os
        try:
            return fn(*args, **kw)
        except FloatingPointError:
            pinned = self.config.get('some_amd_precision') or os.environ.get('SOME_AMD_PRECISION')
            if pinned:
                raise
            print('WARNING: activations left the f16 range of the split-f16 matrix path; switching this model to the exact-f32 '
                  'kernels (set `some_amd_precision: f32` in config.yaml to start there)')
            self.config = dict(self.config, some_amd_precision='f32')
            # a range overflow depends on the rows THIS rank was dealt: the rebuild must not enter a collective the
            # other ranks of a sharded job never join (they are on their way to gather_object / barrier)
            self.model = self.build_model(local=True)
            self.engine = self.model.engine
            return fn(*args, **kw)

    def infer_batch(self, waveforms: List[np.ndarray], return_device_outputs: bool = False):
        """All clips in ONE packed device batch.  Results equal running the clips one by one."""
        return self._guarded(self._infer_batch_impl, waveforms, return_device_outputs)

    def infer(self, waveforms: List[np.ndarray]) -> List[Dict[str, np.ndarray]]:
        """base_infer.py:46-53 semantics on packed device batches (see ``_infer_impl``)."""
        return self._guarded(self._infer_impl, waveforms)

    def infer_files(self, clips: List[np.ndarray], slicer) -> List[List[Tuple[float, Dict[str, np.ndarray]]]]:
        """Whole mono files -> per file [(chunk offset in seconds, notes)] (see ``_infer_files_impl``)."""
        return self._guarded(self._infer_files_impl, clips, slicer)

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
        finite = torch.isfinite(bounds).all() & torch.isfinite(probs).all()      # either stream can leave the f16 range
        return out, finite, {'units': units, 'probs': probs, 'bounds': bounds, 'batch': batch}

    def _finish(self, out, finite, batch) -> List[Dict[str, np.ndarray]]:
        res = self._collect(out, batch)                 # the only host synchronisation of the batch
        if not bool(finite):
            raise FloatingPointError(
                'non-finite model outputs: with some_amd_precision=f16x3 every GEMM input must stay below 65504; '
                "set `some_amd_precision: f32` in config.yaml (or SOME_AMD_PRECISION=f32) for exact-fp32 GEMMs")
        return res

    @torch.no_grad()
    def _infer_batch_impl(self, waveforms: List[np.ndarray], return_device_outputs: bool = False):
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
    def _infer_impl(self, waveforms: List[np.ndarray]) -> List[Dict[str, np.ndarray]]:
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

    # ---- device-side ingest of whole files -----------------------------------------
    def _stage_files(self, clips: List[np.ndarray], slicer, slot: int):
        """Upload whole files (int16 PCM as stored in the WAV, or fp32) through a pinned buffer and compute the slicer's
        RMS curve on the copy stream; the curve comes back through pinned memory.  No host-side float conversion."""
        dtype = np.int16 if all(c.dtype == np.int16 for c in clips) else np.float32
        tdtype = torch.int16 if dtype == np.int16 else torch.float32
        lens = np.asarray([int(c.shape[0]) for c in clips], dtype=np.int64)
        total = int(lens.sum())
        key = ('files', slot, tdtype)
        pin = self._pinned.get(key)
        if pin is None or pin.numel() < total:
            pin = self._pinned[key] = torch.empty(max(total, 1), dtype=tdtype).pin_memory()
        view = pin.numpy()
        pos = 0
        for c, n in zip(clips, lens):
            if c.dtype == np.int16:
                # int16 PCM as stored in the WAV: value = x / 32768 (librosa.load's scaling); stays int16 when every file is
                view[pos:pos + n] = c if dtype == np.int16 else c.astype(np.float32) / np.float32(32768.0)
            elif c.dtype.kind == 'f':
                view[pos:pos + n] = c                      # float waveforms are already in [-1, 1]: no scaling
            else:
                raise TypeError(f'infer_files takes int16 PCM or floating-point waveforms, got {c.dtype}')
            pos += int(n)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        need = np.asarray([slicer.needs_rms(int(n)) for n in lens])
        with torch.cuda.stream(self._copy_stream):
            audio = pin[:total].to(self.device, non_blocking=True)
            uploaded = torch.cuda.Event()
            uploaded.record(self._copy_stream)
            rms_host, ro = None, None
            if need.any():
                rms, ro = self.engine.slicer_rms(audio, lens, slicer.win_size, slicer.hop_size)
                rkey = ('rms', slot)
                rpin = self._pinned.get(rkey)
                if rpin is None or rpin.numel() < rms.numel():
                    rpin = self._pinned[rkey] = torch.empty(max(rms.numel(), 1), dtype=torch.float32).pin_memory()
                rpin[:rms.numel()].copy_(rms, non_blocking=True)
                rms_host = rpin.numpy()
            curve_ready = torch.cuda.Event()
            curve_ready.record(self._copy_stream)
        return {'audio': audio, 'lens': lens, 'need': need, 'rms': rms_host, 'ro': ro, 'uploaded': uploaded,
                'curve_ready': curve_ready}

    def _launch_files(self, st, slicer):
        """Silence decisions on the host from the device RMS curve, then cut + front end + network + decode."""
        st['curve_ready'].synchronize()                 # waits for the copy stream only, not for the compute stream
        so = np.zeros(st['lens'].shape[0] + 1, dtype=np.int64)
        np.cumsum(st['lens'], out=so[1:])
        spans_per_file, src, ln = [], [], []
        for f, n in enumerate(st['lens']):
            n = int(n)
            if st['need'][f]:
                spans = slicer.spans_from_rms(st['rms'][int(st['ro'][f]):int(st['ro'][f + 1])], n)
            else:
                spans = [(0, n)]
            spans_per_file.append(spans)
            for a, b in spans:
                src.append(int(so[f]) + a)
                ln.append(b - a)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(st['uploaded'])
        st['audio'].record_stream(cur)
        audio, batch = self.engine.pcm_gather(st['audio'], src, ln)
        units = self.engine.logmel(audio, batch)
        probs, bounds = self.engine.forward(units, batch, mask=None,
                                            head_mode=_lib.HEAD_SOFTMAX if self.quantized else _lib.HEAD_SIGMOID)
        out = self.engine.decode(probs, bounds, batch, quantized=self.quantized)
        finite = torch.isfinite(bounds).all() & torch.isfinite(probs).all()      # either stream can leave the f16 range
        return out, finite, batch, spans_per_file

    def _group_files(self, clips: List[np.ndarray]) -> List[List[int]]:
        """Consecutive files per device batch: at most ``max_batch_frames`` whole-file frames each."""
        groups: List[List[int]] = []
        group, frames = [], 0
        for i, c in enumerate(clips):
            if c.ndim != 1:
                raise ValueError('infer_files takes mono [L] arrays')
            t = 1 + int(c.shape[0]) // self.engine.hop
            if group and frames + t > self.max_batch_frames:
                groups.append(group)
                group, frames = [], 0
            group.append(i)
            frames += t
        if group:
            groups.append(group)
        return groups

    @torch.no_grad()
    def _infer_files_impl(self, clips: List[np.ndarray], slicer) -> List[List[Tuple[float, Dict[str, np.ndarray]]]]:
        """Whole mono files -> per file [(chunk offset in seconds, {'note_midi', 'note_dur', 'note_rest'})]: the
        ``Slicer(...).slice(waveform)`` + ``infer(chunks)`` pair of infer.py:35-38 / batch_infer.py:52-57 with the file
        uploaded once as it is stored (int16 PCM), the slicer's RMS curve and the chunk cut computed on the device and
        only the silence state machine left on the host.  Chunk boundaries and results equal the host path's."""
        groups = self._group_files(clips)
        sr = slicer.sr
        results: List[List[Tuple[float, Dict[str, np.ndarray]]]] = [None] * len(clips)

        def finish(p):
            out, finite, batch, spans_per_file, idx = p
            res = self._finish(out, finite, batch)
            pos = 0
            for i, spans in zip(idx, spans_per_file):
                results[i] = [(a / sr, res[pos + k]) for k, (a, _) in enumerate(spans)]
                pos += len(spans)

        pending = None
        staged = self._stage_files([clips[i] for i in groups[0]], slicer, 0) if groups else None
        for g, idx in enumerate(groups):
            launched = self._launch_files(staged, slicer)
            if g + 1 < len(groups):                    # upload + RMS of the next group run under this group's forward
                staged = self._stage_files([clips[i] for i in groups[g + 1]], slicer, (g + 1) & 1)
            if pending is not None:
                finish(pending)
            pending = launched + (idx,)
        if pending is not None:
            finish(pending)
        return results
