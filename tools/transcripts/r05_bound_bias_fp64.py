"""round 5: bound-probability bias of the GPU paths and of the fp32 CPU oracle against an fp64 run of the same model on the same
units (one 30 s clip), by depth.  Which arithmetic is biased - the split-f16 path, or everything that computes in fp32?"""
import pathlib
import sys
import time

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
for lay in [int(a) for a in sys.argv[1:]] or (0, 2, 8):
    cfg0 = get_config('midi_conformer', lay=lay)
    sd = synth.synth_state_dict(cfg0, 11)
    units = None
    res = {}
    for precision in ('f16x3', 'f32'):
        cfg = dict(cfg0, some_amd_precision=precision)
        eng = Engine(cfg, device='cuda')
        eng.load_state_dict(sd)
        batch = ClipBatch.from_sample_counts([len(clip)], cfg['hop_size'], 'cuda')
        if units is None:
            units = eng.logmel(torch.from_numpy(clip).cuda(), batch)
        probs, bounds = eng.forward(units, batch, head_mode=_lib.HEAD_SIGMOID)
        res[precision] = bounds.cpu().numpy().astype(np.float64)
    u = units.cpu()
    t0 = time.perf_counter()
    res['cpu fp32 oracle'] = restate.model_forward(sd, cfg0, u.numpy(), sig=True)[1].numpy().astype(np.float64)
    sd64 = {k: torch.from_numpy(np.asarray(v)).double() if np.asarray(v).dtype.kind == 'f' else torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    ref = restate.model_forward(sd64, cfg0, u.double().numpy(), sig=True)[1].numpy()
    print(f'lay {lay}: T = {len(ref)}, sum(bounds) = {ref.sum():.1f}  (CPU runs {time.perf_counter() - t0:.1f} s)')
    for k, b in res.items():
        d = b - ref
        print(f'    {k:16s} vs fp64: mean d {d.mean():+.2e}  rms {np.sqrt((d * d).mean()):.2e}  sum d {d.sum():+.2e}  max|cumsum d| {np.abs(np.cumsum(d)).max():.2e}')
    d = res['f16x3'] - res['cpu fp32 oracle']
    print(f'    f16x3 vs cpu fp32 oracle: mean d {d.mean():+.2e}  sum d {d.sum():+.2e};   f32 vs cpu fp32 oracle: sum d {(res["f32"] - res["cpu fp32 oracle"]).sum():+.2e}')
