"""round 5: which sub-block of a conformer block carries the split-f16 path's bound bias?  lay 0 (input projection -> one block -> head);
sub-blocks are knocked out IN THE WEIGHTS (their last linear map zeroed: both the GPU run and the fp64 run then skip them)."""
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import restate  # noqa: E402  (measurement tool, not the product path)
from some_amd import _lib, synth  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import ClipBatch, Engine  # noqa: E402

torch.set_num_threads(64)
clip = synth.synth_clip(1000, 30.0)
cfg0 = get_config('midi_conformer', lay=0)
base = synth.synth_state_dict(cfg0, 11)
LAST = {'ffn1': 'ffn1.ln2', 'att': 'att.to_out.0', 'conv': 'conv.pointwise_conv2', 'ffn2': 'ffn2.ln2'}
units = None
for keep in (('ffn1', 'att', 'conv', 'ffn2'), (), ('ffn1',), ('att',), ('conv',), ('ffn2',)):
    sd = {k: np.array(v) for k, v in base.items()}
    for name, last in LAST.items():
        if name not in keep:
            for blk in ('att1', 'att2'):
                for leaf in ('weight', 'bias'):
                    sd[f'model.{blk}.{last}.{leaf}'][...] = 0
    res = {}
    for precision in ('f16x3', 'f32'):
        eng = Engine(dict(cfg0, some_amd_precision=precision), device='cuda')
        eng.load_state_dict(sd)
        batch = ClipBatch.from_sample_counts([len(clip)], cfg0['hop_size'], 'cuda')
        if units is None:
            units = eng.logmel(torch.from_numpy(clip).cuda(), batch)
        res[precision] = eng.forward(units, batch, head_mode=_lib.HEAD_SIGMOID)[1].cpu().numpy().astype(np.float64)
    sd64 = {k: torch.from_numpy(v).double() if v.dtype.kind == 'f' else torch.from_numpy(v) for k, v in sd.items()}
    ref = restate.model_forward(sd64, cfg0, units.cpu().double().numpy(), sig=True)[1].numpy()
    line = f'active sub-blocks {"+".join(keep) or "none (LayerNorm 5 + head only)":34s}'
    for k, b in res.items():
        d = b - ref
        line += f' | {k}: mean d {d.mean():+.2e} rms {np.sqrt((d * d).mean()):.2e}'
    print(line)
