"""Host-side logic of the training path (no GPU): the WarmupLR schedule against the reference's own class (golden), the
sample / collater restatement (training/me_task.py:26-52, me_binarizer.py:215-222), the batch planner, and the
data-parallel gradient synchronisation on gloo with world size 2."""
import json
import subprocess
import sys

import numpy as np
import pytest
import torch

from some_amd.configs import get_config
from some_amd.training import data
from some_amd.training.task import warmup_lr


def test_warmup_lr_matches_reference_scheduler(golden_dir):
    g = json.loads((golden_dir / 'lr_schedule.json').read_text())
    for key, lrs in g.items():
        warmup, min_lr = key.split(',')
        for step, lr in lrs.items():
            assert warmup_lr(int(step), 1e-4, int(warmup), float(min_lr)) == pytest.approx(lr, rel=1e-12), (key, step)


def test_note_alignment_and_collater():
    cfg = get_config('two_head_model')
    ts = cfg['hop_size'] / cfg['audio_sample_rate']
    samples = []
    for i, secs in enumerate((3.0, 5.5)):
        wave, midi, dur_sec, rest = data.synth_note_clip(i, secs)
        length = 1 + wave.shape[0] // 512
        dur, u2n = data.note_alignment(dur_sec, length, ts)
        assert u2n.shape == (length,) and u2n[0] == 1 and (np.diff(u2n) >= 0).all() and u2n.max() <= len(dur)
        assert abs(int(dur.sum()) - length) <= 1
        samples.append({'units': torch.zeros(length, 80), 'pitch': torch.zeros(length), 'note_midi': torch.from_numpy(midi),
                        'note_rest': torch.from_numpy(rest), 'note_dur': torch.from_numpy(dur), 'unit2note': torch.from_numpy(u2n)})
    b = data.collater(samples, cfg)
    B, T = b['units'].shape[:2]
    assert B == 2 and b['probs'].shape == (2, T, 128) and b['bounds'].shape == (2, T)
    short = samples[0]['units'].shape[0]
    assert (b['unit2note'][0, short:] == 0).all() and (b['probs'][0, short:] == 0).all()      # padding
    u2n0 = samples[1]['unit2note']
    starts = torch.nonzero(torch.diff(u2n0, prepend=u2n0.new_zeros(1)) > 0).reshape(-1)
    assert torch.equal(torch.nonzero(b['bounds'][1]).reshape(-1), starts)
    for t in (0, T // 2):
        n = int(b['unit2note'][1, t]) - 1
        row = b['probs'][1, t]
        if bool(samples[1]['note_rest'][n]):
            assert (row == 0).all()
        else:
            assert abs(float(row.argmax()) - float(samples[1]['note_midi'][n])) <= 0.5 and row.max() <= 1.0


def test_batch_planner():
    lengths = [100, 200, 300, 50, 60, 70, 400, 30]
    plan = data.batches(lengths, 700, 3, seed=1)
    assert sorted(i for b in plan for i in b) == list(range(8))
    for b in plan:
        assert len(b) <= 3 and max(lengths[i] for i in b) * len(b) <= 700
    r0, r1 = data.batches(lengths, 700, 3, 0, 2, seed=1), data.batches(lengths, 700, 3, 1, 2, seed=1)
    assert len(r0) == len(r1) and not set(map(tuple, r0)) & set(map(tuple, r1))


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ['REPO'])
import torch.distributed as dist
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + os.environ['PORT'], rank=int(os.environ['RANK']), world_size=2)
rank = dist.get_rank()
# the trainer's synchronisation protocol on plain tensors: broadcast of the flat parameters from rank 0, one summing
# all-reduce of the flat gradient, averaging folded into the optimiser's grad_scale = 1 / world
flat = torch.full((1000,), float(rank + 1))
dist.broadcast(flat, src=0)
assert torch.equal(flat, torch.ones(1000))
grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
dist.all_reduce(grad, op=dist.ReduceOp.SUM)
mean = grad * (1.0 / 2)
assert torch.allclose(mean, torch.arange(1000, dtype=torch.float32) * 1.5)
from some_amd.training import data
a = data.batches(list(range(10, 90, 10)), 200, 4, rank, 2, seed=3)
gathered = [None, None]
dist.all_gather_object(gathered, a)
assert len(gathered[0]) == len(gathered[1]) and not set(map(tuple, gathered[0])) & set(map(tuple, gathered[1]))
dist.barrier()
dist.destroy_process_group()
print('ok', rank)
'''


def test_two_rank_gloo_gradient_sync(tmp_path):
    import os
    import pathlib
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    root = pathlib.Path(__file__).resolve().parents[1]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), PORT=str(port), REPO=str(root)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
