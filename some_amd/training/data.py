"""Training data of the HIP training path: ``MIDIExtractionDataset`` (training/me_task.py:13-52 over
training/base_task.py:31-76) reading the reference's binarised datasets - ``<prefix>.lengths`` + the HDF5 container
``<prefix>.data`` (utils/indexed_datasets.py), through some_amd/utils/indexed_datasets.py - its collater, and a
synthetic singing-note source with known notes (``SyntheticNoteDataset``) so that ``train.py`` also runs without a
dataset on disk.  Per-sample fields follow preprocessing/me_binarizer.py:202-223 (note_midi / note_rest / note_dur /
unit2note); synthetic units are computed by the HIP log-mel front end (some_logmel)."""
import pathlib
from typing import Dict, List

import numpy as np
import torch

from ..utils.indexed_datasets import IndexedDataset


def note_alignment(note_dur_sec: np.ndarray, length: int, timestep: float):
    """me_binarizer.py:215-222 + utils/binarizer_utils.py:75-84: frames per note (round(cumsum / timestep + 0.5)
    differences) and the 1-based frame -> note map, padded with the last note / clipped to ``length`` frames."""
    acc = torch.round(torch.cumsum(torch.as_tensor(note_dur_sec, dtype=torch.float32), dim=0) / timestep + 0.5).long().numpy()
    dur = np.diff(acc, prepend=0)
    unit2note = np.repeat(np.arange(1, len(dur) + 1), dur)
    if unit2note.shape[0] < length:
        unit2note = np.concatenate([unit2note, np.full(length - unit2note.shape[0], unit2note[-1])])
    return dur.astype(np.int64), unit2note[:length].astype(np.int64)


def synth_note_clip(index: int, seconds: float, sr: int = 44100):
    """A sung-like clip with KNOWN notes: piecewise-constant pitch with vibrato, 6 harmonics, occasional rests.
    Returns (waveform float32 [L], note_midi float32 [n], note_dur_sec float64 [n], note_rest bool [n])."""
    rng = np.random.default_rng(5000 + index)
    n = int(round(seconds * sr))
    midis, durs, rests = [], [], []
    t, cur = 0.0, rng.uniform(50, 70)
    while t < seconds:
        d = min(float(rng.uniform(0.15, 0.8)), seconds - t)
        rest = bool(rng.uniform() < 0.12)
        midis.append(cur)
        durs.append(d)
        rests.append(rest)
        t += d
        cur = float(np.clip(cur + rng.integers(-5, 6), 48, 72))
    tt = np.arange(n, dtype=np.float64) / sr
    edges = np.cumsum(durs)
    idx = np.minimum(np.searchsorted(edges, tt, side='right'), len(durs) - 1)
    midi_t = np.asarray(midis)[idx] + 0.3 * np.sin(2 * np.pi * 5.5 * tt)
    f0 = 440.0 * 2.0 ** ((midi_t - 69.0) / 12.0)
    phase = 2 * np.pi * np.cumsum(f0) / sr
    y = sum(np.sin(k * phase) / k for k in range(1, 7))
    y *= 0.3 / np.max(np.abs(y))
    y[np.asarray(rests)[idx]] = 0.0
    y += rng.standard_normal(n) * 0.003
    return y.astype(np.float32), np.asarray(midis, np.float32), np.asarray(durs, np.float64), np.asarray(rests, bool)


def make_sample(engine, clip, timestep: float) -> Dict[str, torch.Tensor]:
    """One dataset item (me_binarizer.py:144-223 with units_encoder: mel) on the device."""
    from ..engine import ClipBatch
    wave, note_midi, note_dur_sec, note_rest = clip
    wav = torch.from_numpy(wave).to(engine.device)
    batch = ClipBatch.from_sample_counts([wav.numel()], engine.hop, engine.device)
    units = engine.logmel(wav, batch)                                   # [T, 80]
    length = units.shape[0]
    note_dur, unit2note = note_alignment(note_dur_sec, length, timestep)
    return {'units': units, 'pitch': torch.zeros(length, device=engine.device), 'note_midi': torch.from_numpy(note_midi).to(engine.device),
            'note_rest': torch.from_numpy(note_rest).to(engine.device), 'note_dur': torch.from_numpy(note_dur).to(engine.device),
            'unit2note': torch.from_numpy(unit2note).to(engine.device)}


def _collate_nd(values: List[torch.Tensor], pad_value=0):
    """utils.collate_nd for 1-D / 2-D items: pad along the first axis to the longest."""
    size = max(v.shape[0] for v in values)
    out = values[0].new_full((len(values), size) + tuple(values[0].shape[1:]), pad_value)
    for i, v in enumerate(values):
        out[i, :v.shape[0]] = v
    return out


def collater(samples: List[Dict[str, torch.Tensor]], config: dict) -> Dict[str, torch.Tensor]:
    """MIDIExtractionDataset.collater (training/me_task.py:26-52): gaussian-blurred per-frame targets and boundaries."""
    num_bins = config['midi_num_bins']
    interval = (config['midi_max'] - config['midi_min']) / (num_bins - 1)
    sigma = config['midi_prob_deviation'] / interval
    batch = {'size': len(samples)}
    batch['units'] = _collate_nd([s['units'] for s in samples])
    batch['pitch'] = _collate_nd([s['pitch'] for s in samples])
    batch['note_midi'] = _collate_nd([s['note_midi'] for s in samples])
    batch['note_rest'] = _collate_nd([s['note_rest'] for s in samples])
    batch['note_dur'] = _collate_nd([s['note_dur'] for s in samples])
    miu = ((batch['note_midi'] - config['midi_min']) / interval)[:, :, None]
    x = torch.arange(num_bins, device=miu.device).float().reshape(1, 1, -1)
    probs = ((x - miu) / sigma).pow(2).div(-2).exp()
    note_mask = _collate_nd([torch.ones_like(s['note_rest']) for s in samples], pad_value=False)
    probs = probs * (note_mask[..., None] & ~batch['note_rest'][..., None])
    probs = torch.nn.functional.pad(probs, [0, 0, 1, 0])
    unit2note = _collate_nd([s['unit2note'] for s in samples])
    batch['probs'] = torch.gather(probs, 1, unit2note[..., None].repeat([1, 1, num_bins]))
    batch['unit2note'] = unit2note
    batch['bounds'] = (torch.diff(unit2note, dim=1, prepend=unit2note.new_zeros((len(samples), 1))) > 0).float()
    return batch


def quant_collater(samples: List[Dict[str, torch.Tensor]], config: dict) -> Dict[str, torch.Tensor]:
    """QuantizedMIDIExtractionDataset.collater (training/me_quant_task.py:14-27): ``note_midi`` holds integer classes (0 - 127, 128 = rest,
    preprocessing/me_quant_binarizer.py:16,32), padded with -1; the per-frame class ``midi_idx`` is gathered through unit2note with -1
    in front (frames of the padding, unit2note 0) - the ignore_index of the cross-entropy loss."""
    batch = {'size': len(samples)}
    batch['units'] = _collate_nd([s['units'] for s in samples])
    batch['pitch'] = _collate_nd([s['pitch'] for s in samples])
    batch['note_midi'] = _collate_nd([s['note_midi'] for s in samples], pad_value=-1)
    batch['note_dur'] = _collate_nd([s['note_dur'] for s in samples])
    unit2note = _collate_nd([s['unit2note'] for s in samples])
    batch['unit2note'] = unit2note
    batch['midi_idx'] = torch.gather(torch.nn.functional.pad(batch['note_midi'], [1, 0], value=-1), 1, unit2note)
    batch['bounds'] = (torch.diff(unit2note, dim=1, prepend=unit2note.new_zeros((len(samples), 1))) > 0).float()
    return batch


def quantize_item(item: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """A continuous item -> the quantised binarizer's fields (preprocessing/me_quant_binarizer.py:25-32): round_midi, rests -> class 128."""
    midi = torch.round(item['note_midi'].float()).long()
    midi[item['note_rest'].bool()] = 128
    return {k: v for k, v in {**item, 'note_midi': midi}.items() if k != 'note_rest'}


class MIDIExtractionDataset:
    """training/base_task.py:31-76 + training/me_task.py:13-52.  ``sizes`` are the frame counts the binarizer saved
    (preprocessing/base_binarizer.py:196-199); the samplers read ``_sizes`` / ``num_frames``.  Items are moved to
    ``device`` by the collater."""

    def __init__(self, config: dict, data_dir, prefix: str, allow_aug: bool = False, device=None):
        self.config, self.prefix, self.allow_aug, self.device = config, prefix, allow_aug, device
        self.data_dir = pathlib.Path(data_dir)
        self.sizes = np.load(self.data_dir / f'{prefix}.lengths')
        self.indexed_ds = IndexedDataset(self.data_dir, prefix)
        if len(self.indexed_ds) != len(self.sizes):
            raise ValueError(f'{self.data_dir}/{prefix}: {len(self.indexed_ds)} items in .data but {len(self.sizes)} lengths')

    @property
    def _sizes(self):
        return self.sizes

    def __getitem__(self, index):
        return self.indexed_ds[index]

    def __len__(self):
        return len(self.sizes)

    def num_frames(self, index):
        return self.sizes[index]

    size = num_frames

    def collater(self, samples: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        if self.device is not None:
            samples = [{k: v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v for k, v in s.items()} for s in samples]
        return collater(samples, self.config)


class QuantizedMIDIExtractionDataset(MIDIExtractionDataset):
    """training/me_quant_task.py:13-27 over the quantised binarizer's items."""
    quantized = True

    def collater(self, samples: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        if self.device is not None:
            samples = [{k: v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v for k, v in s.items()} for s in samples]
        return quant_collater(samples, self.config)


class SyntheticNoteDataset(MIDIExtractionDataset):
    """The same interface over synthetic sung clips (``synth_note_clip``) held in device memory.  ``quantized``: items and batches of
    QuantizedMIDIExtractionDataset (integer note classes, rest = 128)."""

    def __init__(self, config: dict, engine, indices, seconds, allow_aug: bool = False, quantized: bool = False):
        self.config, self.prefix, self.allow_aug, self.device = config, 'synthetic', allow_aug, None
        self.quantized = quantized
        timestep = config['hop_size'] / config['audio_sample_rate']
        self.items = [make_sample(engine, synth_note_clip(i, sec), timestep) for i, sec in zip(indices, seconds)]
        if quantized:
            self.items = [quantize_item(it) for it in self.items]
        self.sizes = np.asarray([int(s['units'].shape[0]) for s in self.items], dtype=np.int64)

    def __getitem__(self, index):
        return self.items[index]

    def collater(self, samples):
        return quant_collater(samples, self.config) if self.quantized else collater(samples, self.config)
