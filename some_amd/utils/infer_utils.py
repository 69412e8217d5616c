"""Decode / MIDI-file helpers with the reference's names and signatures (utils/infer_utils.py).

The three decode functions run on the GPU through ``some_decode`` (csrc/decode.hip); they accept and return
torch tensors shaped like the reference's ([B, T, ...]).  ``build_midi_file`` is host-side integer logic."""
from typing import Dict, List

import numpy as np
import torch

from . import smf


def _engine_for(device, nbins, vmin=0, vmax=127, deviation=1.0, threshold=0.1):
    from ..engine import Engine
    key = (str(device), int(nbins), float(vmin), float(vmax), float(deviation), float(threshold))
    eng = _engine_for.cache.get(key)
    if eng is None:
        cfg = {'midi_num_bins': int(nbins), 'midi_min': vmin, 'midi_max': vmax,
               'midi_prob_deviation': deviation, 'rest_threshold': threshold}
        eng = _engine_for.cache[key] = Engine(cfg, device=device)
    return eng


_engine_for.cache = {}


def decode_gaussian_blurred_probs(probs, vmin, vmax, deviation, threshold):
    """utils/infer_utils.py:9-24.  probs [B,T,N] -> (values [B,T] fp32, rest [B,T] bool)."""
    from ..engine import ClipBatch
    b, t, n = probs.shape
    eng = _engine_for(probs.device, n, vmin, vmax, deviation, threshold)
    batch = ClipBatch([t] * b, eng.device)
    p = probs.to(torch.float32).reshape(b * t, n).contiguous()
    zeros = torch.zeros(b * t, dtype=torch.float32, device=eng.device)
    out = eng.decode(p, zeros, batch, quantized=False, debug=True)
    return out['values'].view(b, t), out['rest'].view(b, t).bool()


def decode_bounds_to_alignment(bounds, use_diff=True):
    """utils/infer_utils.py:27-39.  bounds [B,T] -> frame2item int64 [B,T].  (``use_diff=False`` is the ONNX
    deployment variant, deployment/me_onnx_module.py:30 - out of scope.)"""
    from ..engine import ClipBatch
    if not use_diff:
        raise NotImplementedError('use_diff=False (ONNX export variant) is not implemented')
    b, t = bounds.shape
    eng = _engine_for(bounds.device, 128)
    batch = ClipBatch([t] * b, eng.device)
    probs = torch.zeros((b * t, 128), dtype=torch.float32, device=eng.device)
    out = eng.decode(probs, bounds.to(torch.float32).reshape(-1).contiguous(), batch, quantized=False, debug=True)
    return out['frame2item'].view(b, t)


def decode_note_sequence(frame2item, values, masks, threshold=0.5):
    """utils/infer_utils.py:42-76.  frame2item int64 [B,T], values fp32 | int64 [B,T], masks bool [B,T] ->
    (item_values fp32 [B,N], item_dur int64 [B,N], item_masks bool [B,N]) with N = frame2item.max(), padded with
    zeros / False for clips that have fewer notes (exactly what the reference's batched scatter yields)."""
    import ctypes as C
    from .. import _lib
    from ..engine import ClipBatch
    if threshold != 0.5:
        raise NotImplementedError('threshold is compiled as 0.5 (the reference default)')
    b, t = frame2item.shape
    eng = _engine_for(frame2item.device, 128)
    batch = ClipBatch([t] * b, eng.device)
    dev = eng.device
    is_int = not torch.is_floating_point(values)
    f2i = frame2item.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
    vals = values.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    not_masks = (~masks.to(device=dev, dtype=torch.bool)).to(torch.uint8).reshape(-1).contiguous()
    m = b * t
    note_midi = torch.zeros(m, dtype=torch.float32, device=dev)
    note_dur = torch.zeros(m, dtype=torch.int64, device=dev)
    note_rest = torch.ones(m, dtype=torch.uint8, device=dev)
    n_notes = torch.zeros(b, dtype=torch.int32, device=dev)
    sc = eng._decode_scratch(int(eng.lib.some_decode_scratch_bytes(eng.handle, m)))
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    _lib.check(eng.handle, eng.lib.some_decode_notes(
        eng.handle, p(f2i), p(vals), p(not_masks), p(batch.frame_offsets_dev), b, m, t, 1 if is_int else 0,
        p(note_midi), p(note_dur), p(note_rest), p(n_notes), p(sc), sc.numel(), eng._stream()))
    n = int(n_notes.max()) if b else 0
    iv = note_midi.view(b, t)[:, :n].clone()
    idur = note_dur.view(b, t)[:, :n].clone()
    imask = (note_rest.view(b, t)[:, :n] == 0)
    # rows beyond a clip's own note count keep their initial zeros / rest flags
    return iv, idur, imask


def build_midi_file(offsets: List[float], segments: List[Dict[str, np.ndarray]], tempo=120) -> smf.MidiFile:
    """utils/infer_utils.py:79-100: one track, 480 ticks per beat => ``tempo * 8`` ticks per second; each
    chunk's notes are laid end to end from the chunk offset, clipped at the next chunk's offset; rests and
    empty notes emit nothing."""
    midi_file = smf.MidiFile(charset='utf8')
    track = smf.MidiTrack()
    track.append(smf.MetaMessage('set_tempo', tempo=smf.bpm2tempo(tempo), time=0))
    tps = tempo * 8
    starts = [round(o * tempo * 8) for o in offsets]
    cursor = 0                                         # tick of the last emitted event
    for i, (t0, seg) in enumerate(zip(starts, segments)):
        pitches = np.round(seg['note_midi']).astype(np.int64).tolist()
        ends_rel = np.round(np.cumsum(seg['note_dur']) * tps).astype(np.int64)
        ticks = np.diff(ends_rel, prepend=0).tolist()
        rests = seg['note_rest'].tolist()
        limit = starts[i + 1] if i + 1 < len(starts) else None
        begin = t0
        for pitch, tick, is_rest in zip(pitches, ticks, rests):
            end = begin + tick
            if limit is not None and end > limit:
                end = limit
            if begin < end and not is_rest:
                track.append(smf.Message('note_on', note=pitch, time=begin - cursor))
                track.append(smf.Message('note_off', note=pitch, time=end - begin))
                cursor = end
            begin = end
    midi_file.tracks.append(track)
    return midi_file
