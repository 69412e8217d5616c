"""round 5, VERDICT r04 item 7: the sign structure of the f16x3 path's bound error at full size (tests/golden/fullsize.npz, clip by clip):
a systematic bias moves the fp64 cumsum of decode_bounds_to_alignment (utils/infer_utils.py:27-39) far more than zero-mean rounding noise."""
import json
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from some_amd import _lib, synth  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import ClipBatch, Engine  # noqa: E402

g = np.load(ROOT / 'tests' / 'golden' / 'fullsize.npz')
meta = json.loads((ROOT / 'tests' / 'golden' / 'fullsize.json').read_text())
for name in ('full_conf',):
    m = meta[name]
    for precision in ('f16x3', 'f32'):
        cfg = get_config(m['config'], some_amd_precision=precision)
        eng = Engine(cfg, device='cuda')
        eng.load_state_dict(synth.synth_state_dict(cfg, m['seed']))
        clips = [synth.synth_clip(m['clip0'] + i, m['seconds']) for i in range(m['clips'])]
        batch = ClipBatch.from_sample_counts([len(w) for w in clips], cfg['hop_size'], 'cuda')
        units = eng.logmel(torch.from_numpy(np.concatenate(clips)).cuda(), batch)
        ref_units = None
        probs, bounds = eng.forward(units, batch, head_mode=_lib.HEAD_SIGMOID)
        bounds = bounds.cpu().numpy().astype(np.float64)
        for c in range(m['clips']):
            s = int(batch.frame_offsets[c])
            ref = g[f'{name}.clip{c}.bounds'].astype(np.float64)
            d = bounds[s:s + 2584] - ref
            cs = np.cumsum(d)
            print(f'{name} {precision} clip {c}: max|d| {np.abs(d).max():.2e} mean d {d.mean():+.2e} rms {np.sqrt((d * d).mean()):.2e} '
                  f'sum d {d.sum():+.2e} max|cumsum d| {np.abs(cs).max():.2e} (random walk of this rms: {np.sqrt((d * d).sum()):.2e}); '
                  f'sum(bounds) {ref.sum():.1f}')
