#!/usr/bin/env python
"""Front-end kernel alone: log-mel of B x S-second clips resident in HBM, HIP-event timed; prints GB/s of algorithmic
traffic (4 L bytes in + 320 bytes per frame out) against the 8 TB/s HBM peak."""
import argparse
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))

import numpy as np
import torch

from some_amd import synth
from some_amd.configs import get_config
from some_amd.engine import ClipBatch, Engine

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--seconds', type=float, default=30.0)
ap.add_argument('--iters', type=int, default=50)
args = ap.parse_args()
eng = Engine(get_config('midi_conformer', lay=0), device='cuda')
clips = [synth.synth_clip(i, args.seconds) for i in range(min(8, args.batch))]
waves = [clips[i % len(clips)] for i in range(args.batch)]
batch = ClipBatch.from_sample_counts([len(w) for w in waves], eng.hop, 'cuda')
audio = torch.from_numpy(np.concatenate(waves)).cuda()
for _ in range(5):
    eng.logmel(audio, batch)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    eng.logmel(audio, batch)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
nbytes = 4.0 * audio.numel() + 320.0 * batch.total_frames
print(f'logmel {args.batch} x {args.seconds:g} s: {ms:.4f} ms/launch, {nbytes / ms / 1e6:.1f} GB/s algorithmic ({nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s)')
