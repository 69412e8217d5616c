#!/usr/bin/env python
"""A synthetic binarised training set in the reference's on-disk format, at BASELINE configs[4]'s size: "synthetic 3 h dataset".

    python tools/make_train_dataset.py --dir /tmp/some_ds --hours 3 [--valid 8]

Writes ``<dir>/train.data`` + ``train.lengths`` (and ``valid.*``): one HDF5 group per item with the binarizer's attributes
(preprocessing/me_binarizer.py:22-29: units float32 [T, 80], pitch, note_midi, note_rest, note_dur, unit2note), through libhdf5
itself (tools/h5_write.py).  Items are synthetic sung phrases with known notes (some_amd/training/data.synth_note_clip); phrase
lengths are log-normal around 6 s clipped to 1 - 20 s - the shape of a sliced singing corpus (the distribution the sampler
fixtures use); their mel units come from the HIP front end (some_logmel), as a binarizer run on this GPU would produce them."""
import argparse
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import ClipBatch, Engine  # noqa: E402
from some_amd.training import data  # noqa: E402
from tools.h5_write import write_items  # noqa: E402


def phrase_seconds(hours: float, seed: int):
    rng = np.random.default_rng(seed)
    out, total = [], 0.0
    while total < hours * 3600.0:
        s = float(np.clip(rng.lognormal(mean=np.log(6.0), sigma=0.55), 1.0, 20.0))
        out.append(s)
        total += s
    return out


def make_items(engine, cfg, first_index, seconds, group=64):
    timestep = cfg['hop_size'] / cfg['audio_sample_rate']
    items = []
    for g in range(0, len(seconds), group):
        clips = [data.synth_note_clip(first_index + g + i, s) for i, s in enumerate(seconds[g:g + group])]
        waves = [c[0] for c in clips]
        batch = ClipBatch.from_sample_counts([len(w) for w in waves], engine.hop, engine.device)
        units = engine.logmel(torch.from_numpy(np.concatenate(waves)).to(engine.device), batch).cpu().numpy()
        for b, (wave, note_midi, note_dur_sec, note_rest) in enumerate(clips):
            s, e = int(batch.frame_offsets[b]), int(batch.frame_offsets[b + 1])
            length = e - s
            note_dur, unit2note = data.note_alignment(note_dur_sec, length, timestep)
            items.append({'units': units[s:e].astype(np.float32), 'pitch': np.zeros(length, np.float32), 'note_midi': note_midi.astype(np.float32),
                          'note_rest': note_rest.astype(bool), 'note_dur': note_dur.astype(np.int64), 'unit2note': unit2note.astype(np.int64)})
    return items


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', required=True)
    ap.add_argument('--hours', type=float, default=3.0)
    ap.add_argument('--valid', type=int, default=8)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    out = pathlib.Path(a.dir)
    out.mkdir(parents=True, exist_ok=True)
    cfg = get_config('two_head_model')
    engine = Engine(get_config('two_head_model', lay=0), device='cuda')
    t0 = time.perf_counter()
    for prefix, first, seconds in (('train', 0, phrase_seconds(a.hours, a.seed)), ('valid', 10 ** 6, [6.0] * a.valid)):
        items = make_items(engine, cfg, first, seconds)
        write_items(out / f'{prefix}.data', items)
        with open(out / f'{prefix}.lengths', 'wb') as fh:                  # preprocessing/base_binarizer.py:197-199
            np.save(fh, [it['units'].shape[0] for it in items])
        frames = sum(it['units'].shape[0] for it in items)
        print(f'{prefix}: {len(items)} items, {frames} frames = {frames * cfg["hop_size"] / cfg["audio_sample_rate"] / 3600:.3f} h, '
              f'{(out / (prefix + ".data")).stat().st_size / 2 ** 20:.1f} MiB', flush=True)
    print(f'written in {time.perf_counter() - t0:.1f} s')


if __name__ == '__main__':
    main()
