"""Synthetic inputs for benchmarking and testing (no network, no datasets, no released checkpoints).

Two generators, both pure numpy and fully determined by their seed so that the benchmark, the tests, the
golden-vector generator (``oracle/make_golden.py``) and the CPU baseline all see identical data:

* :func:`synth_clip` - 44.1 kHz mono singing-like audio (recipe: SURVEY.md section 8(d), "Synthetic audio").
* :func:`synth_state_dict` - random weights with the exact key set / shapes of the reference model's
  ``state_dict()`` (reference ``modules/conform/Gconform.py:93-116``; key list in SURVEY.md section 2b),
  wrapped by :func:`save_checkpoint` into the Lightning-style ``{'state_dict': {'model.'+k: v}}`` file
  plus sibling ``config.yaml`` that ``BaseInference.build_model`` consumes (reference
  ``inference/base_infer.py:23-35``, ``infer.py:20-22``).
"""
import pathlib
from collections import OrderedDict

import numpy as np


def synth_clip(index: int, seconds: float, sr: int = 44100, silence_every: float = 0.0) -> np.ndarray:
    """Piecewise-constant f0 random walk over MIDI 48-72 with vibrato, 6 harmonics, -50 dB noise floor."""
    rng = np.random.default_rng(1000 + index)
    n = int(round(seconds * sr))
    t = np.arange(n, dtype=np.float64) / sr
    midi = np.empty(n, dtype=np.float64)
    pos = 0
    cur = rng.uniform(48, 72)
    while pos < n:
        length = int(rng.uniform(0.15, 0.8) * sr)
        midi[pos:pos + length] = cur
        pos += length
        cur = float(np.clip(cur + rng.integers(-5, 6), 48, 72))
    midi += 0.3 * np.sin(2 * np.pi * 5.5 * t)
    f0 = 440.0 * 2.0 ** ((midi - 69.0) / 12.0)
    phase = 2 * np.pi * np.cumsum(f0) / sr
    y = np.zeros(n, dtype=np.float64)
    for k in range(1, 7):
        y += np.sin(k * phase) / k
    y *= 0.3 / np.max(np.abs(y))
    if silence_every > 0:
        # 0.6 s gaps: exercise the Slicer (reference utils/slicer2.py)
        gap = int(0.6 * sr)
        step = int(silence_every * sr)
        for s in range(step, n - gap, step):
            y[s:s + gap] = 0.0
        y += rng.standard_normal(n) * 1e-4
    else:
        y += rng.standard_normal(n) * 0.003
    return y.astype(np.float32)


def _block_shapes(dim, heads, head_dim, ksize):
    hid = heads * head_dim
    shapes = OrderedDict()
    for f in ('ffn1', 'ffn2'):
        shapes[f'{f}.ln1.weight'] = (4 * dim, dim)
        shapes[f'{f}.ln1.bias'] = (4 * dim,)
        shapes[f'{f}.ln2.weight'] = (dim, 4 * dim)
        shapes[f'{f}.ln2.bias'] = (dim,)
    shapes['att.to_q.weight'] = (hid, dim)
    shapes['att.to_kv.weight'] = (2 * hid, dim)
    shapes['att.to_out.0.weight'] = (dim, hid)
    shapes['att.to_out.0.bias'] = (dim,)
    shapes['conv.pointwise_conv1.weight'] = (2 * dim, dim, 1)
    shapes['conv.pointwise_conv1.bias'] = (2 * dim,)
    shapes['conv.depthwise_conv.weight'] = (dim, 1, ksize)
    shapes['conv.depthwise_conv.bias'] = (dim,)
    shapes['conv.norm.weight'] = (dim,)
    shapes['conv.norm.bias'] = (dim,)
    shapes['conv.norm.running_mean'] = (dim,)
    shapes['conv.norm.running_var'] = (dim,)
    shapes['conv.norm.num_batches_tracked'] = ()
    shapes['conv.pointwise_conv2.weight'] = (dim, dim, 1)
    shapes['conv.pointwise_conv2.bias'] = (dim,)
    for i in range(1, 6):
        shapes[f'norm{i}.weight'] = (dim,)
        shapes[f'norm{i}.bias'] = (dim,)
    return shapes


def state_dict_shapes(config: dict) -> 'OrderedDict[str, tuple]':
    """Key -> shape of ``midi_conforms(config).state_dict()`` (after the ckpt's ``model.`` prefix strip)."""
    a = config['midi_extractor_args']
    dim, lay = a['dim'], a['lay']
    blk = _block_shapes(dim, a['attention_heads'], a['attention_heads_dim'], a['kernel_size'])
    indim, outdim = config['units_dim'], config['midi_num_bins']
    shapes = OrderedDict()
    shapes['model.inln.weight'] = (dim, indim)
    shapes['model.inln.bias'] = (dim,)
    shapes['model.inln1.weight'] = (dim, indim)
    shapes['model.inln1.bias'] = (dim,)
    shapes['model.outln.weight'] = (outdim, dim)
    shapes['model.outln.bias'] = (outdim,)
    shapes['model.cutheard.weight'] = (1, dim)
    shapes['model.cutheard.bias'] = (1,)
    for i in range(lay):
        for s in ('att1', 'att2'):
            for k, v in blk.items():
                shapes[f'model.cf_lay.{i}.{s}.{k}'] = v
        for g in ('glu1', 'glu2'):
            shapes[f'model.cf_lay.{i}.{g}.0.weight'] = (2 * dim, dim)
            shapes[f'model.cf_lay.{i}.{g}.0.bias'] = (2 * dim,)
    for s in ('att1', 'att2'):
        for k, v in blk.items():
            shapes[f'model.{s}.{k}'] = v
    return shapes


def synth_state_dict(config: dict, seed: int = 114514) -> 'OrderedDict[str, np.ndarray]':
    """Random fp32 weights (numpy).  Linear/conv weights ~ U(-1,1)/sqrt(fan_in), biases ~ U(-.1,.1);
    LayerNorm/BatchNorm affine ~ N(1,.1)/N(0,.1); BN running_mean ~ N(0,.1), running_var ~ U(.5,1.5)
    so that no parameter is left at a value that would hide a wiring mistake."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for key, shape in state_dict_shapes(config).items():
        leaf = key.rsplit('.', 1)[-1]
        is_norm = '.norm' in key  # norm1..5 (LayerNorm) and conv.norm (BatchNorm)
        if leaf == 'num_batches_tracked':
            v = np.array(1000, dtype=np.int64)
        elif leaf == 'running_mean':
            v = rng.standard_normal(shape) * 0.1
        elif leaf == 'running_var':
            v = rng.uniform(0.5, 1.5, shape)
        elif is_norm and leaf == 'weight':
            v = 1.0 + rng.standard_normal(shape) * 0.1
        elif is_norm and leaf == 'bias':
            v = rng.standard_normal(shape) * 0.1
        elif leaf == 'weight':
            fan_in = int(np.prod(shape[1:]))
            v = rng.uniform(-1.0, 1.0, shape) / np.sqrt(fan_in)
        else:
            v = rng.uniform(-0.1, 0.1, shape)
        out[key] = v.astype(np.float32) if v.dtype != np.int64 else v
    return out


def save_checkpoint(config: dict, path, seed: int = 114514):
    """Write ``<path>`` (torch ckpt with ``model.``-prefixed keys) and ``config.yaml`` beside it."""
    import torch
    import yaml
    path = pathlib.Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    sd = synth_state_dict(config, seed)
    ckpt = {'state_dict': OrderedDict(('model.' + k, torch.from_numpy(np.asarray(v))) for k, v in sd.items())}
    torch.save(ckpt, path)
    with open(path.with_name('config.yaml'), 'w', encoding='utf8') as f:
        yaml.safe_dump(config, f)
    return path


def synth_train_batch(B=2, T=96, nb=128, seed=21):
    """A padded training batch shaped like MIDIExtractionDataset.collater's (training/me_task.py:26-52)."""
    rng = np.random.default_rng(seed)
    units = (rng.standard_normal((B, T, 80)) * 1.5 - 4.0).astype(np.float32)
    unit2note = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        valid = T if b == 0 else T - 26
        edges = np.sort(rng.choice(np.arange(1, valid), size=6, replace=False))
        unit2note[b, :valid] = 1 + np.searchsorted(edges, np.arange(valid), side='right')
    units[unit2note == 0] = 0.0                                   # collate_nd pads with zeros
    centers = rng.uniform(40, 80, (B, 8))
    idx = np.arange(nb, dtype=np.float32)[None, None, :]
    probs = np.exp(-0.5 * (idx - centers[np.arange(B)[:, None], np.clip(unit2note, 0, 7)][..., None]) ** 2).astype(np.float32)
    probs *= (unit2note > 0)[..., None]
    bounds = (np.diff(unit2note, axis=1, prepend=0) > 0).astype(np.float32)
    return {'units': units, 'unit2note': unit2note, 'probs': probs, 'bounds': bounds}
