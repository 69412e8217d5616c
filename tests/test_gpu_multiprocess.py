"""Multi-process paths on the ONE GPU of the test box ("RCCL has never seen N > 1 ranks" cannot be closed on
a one-GPU box, but everything short of a second device can run on every driver pass):

* the `nccl` (= RCCL) process group of ONE rank under torch.distributed.run: communicator creation, the weight-arena broadcast, the
  barrier and the max-over-ranks all-reduce of bench.py, and batch_infer.py's broadcast + gather_object;
* batch_infer.py with TWO ranks (gloo moving the CUDA tensors, both on this GPU) over real WAV files end to end: the CSV is
  byte-identical to the single-process run."""
import csv
import json
import os
import pathlib
import socket
import subprocess
import sys

import numpy as np
import pytest

from some_amd import synth
from some_amd.configs import get_config
from some_amd.utils.audio import save_wav

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def _port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _torchrun(nproc, script_args, env=None, timeout=900):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_port())] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, MASTER_ADDR='127.0.0.1', **(env or {})), timeout=timeout)


def _dataset(root: pathlib.Path, rows: int, seconds: float):
    (root / 'wavs').mkdir(parents=True)
    lines = []
    for i in range(rows):
        sec = seconds * (1.0 + 0.3 * (i % 3))                     # different lengths: the size-sorted round-robin deal matters
        save_wav(root / 'wavs' / f'clip_{i:03d}.wav', synth.synth_clip(900 + i, sec, silence_every=4.0), 44100)
        n_ph = 12
        lines.append({'name': f'clip_{i:03d}', 'ph_seq': ' '.join(['a'] * n_ph), 'ph_dur': ' '.join([f'{sec / n_ph:.6f}'] * n_ph),
                      'ph_num': ' '.join(['2'] * (n_ph // 2))})
    with open(root / 'transcriptions.csv', 'w', encoding='utf8', newline='') as f:
        w = csv.DictWriter(f, fieldnames=['name', 'ph_seq', 'ph_dur', 'ph_num'])
        w.writeheader()
        w.writerows(lines)
    synth.save_checkpoint(get_config('midi_conformer', lay=2), root / 'model' / 'model.ckpt', seed=5)


def test_bench_under_torchrun_with_one_nccl_rank():
    r = _torchrun(1, [str(ROOT / 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--lay', '1', '--batch', '4', '--seconds', '5',
                      '--no-cpu-baseline', '--no-f32-leg', '--no-latency', '--no-secondary', '--no-live-pmc'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['process_group'] == {'backend': 'nccl', 'world_size': 1} and res['n_gpus'] == 1
    assert res['notes_decoded_last_step'] > 0 and res['value'] > 0


def test_bench_with_two_ranks_sharing_the_gpu():
    """bench.py --gpus 2 as the driver's scaling run launches it, with gloo standing in for RCCL on this one-GPU box: every rank pins itself
    to its own cores (sharding.bind_rank_to_cores), takes the broadcast weight arena, runs its own 4 clips, and rank 0 reports the
    max-over-ranks time - value = the audio of BOTH ranks per second, n_gpus 2, weak scaling."""
    r = _torchrun(2, [str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--lay', '1', '--batch', '4', '--seconds', '5',
                      '--no-cpu-baseline', '--no-f32-leg', '--no-latency', '--no-secondary', '--no-live-pmc', '--no-kernel-profile'],
                  env={'SOME_AMD_DIST_BACKEND': 'gloo'})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    res = json.loads(lines[0])
    assert res['process_group'] == {'backend': 'gloo', 'world_size': 2} and res['n_gpus'] == 2 and res['scaling'] == 'weak'
    assert abs(res['value'] - 2 * 4 * 5.0 * 2 / (res['ms_per_step'] * 2 * 1e-3)) < 0.02 * res['value']
    assert res['notes_decoded_last_step'] > 0 and 'e2e_batch_infer' not in res and 'train_epoch' not in res


def test_batch_infer_one_nccl_rank_and_two_gloo_ranks_equal_the_plain_run(tmp_path):
    _dataset(tmp_path, rows=7, seconds=6.0)
    base = [str(ROOT / 'batch_infer.py'), '--dataset', str(tmp_path), '--model', str(tmp_path / 'model' / 'model.ckpt'), '--overwrite']
    r = subprocess.run([sys.executable] + base + ['--csv', str(tmp_path / 'plain.csv')], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    plain = (tmp_path / 'plain.csv').read_bytes()
    assert plain.count(b'\n') == 8 and b'note_seq' in plain
    r = _torchrun(1, base + ['--csv', str(tmp_path / 'nccl1.csv')])                       # RCCL: init, arena broadcast, gather_object, barrier
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert (tmp_path / 'nccl1.csv').read_bytes() == plain
    r = _torchrun(2, base + ['--csv', str(tmp_path / 'gloo2.csv')], env={'SOME_AMD_DIST_BACKEND': 'gloo'})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    # rows are dealt to the ranks by size and a rank packs ITS rows into device batches, so a row meets other neighbours than in the
    # plain run - and its result must not care (clip-aligned attention key tiles; tests/test_gpu_parity.py pins the forward bit for
    # bit): the sharded job's CSV is the single-process CSV, byte for byte, as the reference's per-chunk loop guarantees by construction
    assert (tmp_path / 'gloo2.csv').read_bytes() == plain
