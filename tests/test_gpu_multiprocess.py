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


def test_train_cli_under_torchrun_with_one_nccl_rank(tmp_path):
    """train.py's RCCL path on this one-GPU box: the process group with a HIGH-priority communication stream (train.py), the parameter
    broadcast, the bucketed all-reduce launched from inside backward (world size 1 keeps grad_sync off: the single-rank group still
    creates the communicator and runs the barrier), checkpoint written."""
    r = _torchrun(1, [str(ROOT / 'train.py'), '--config', 'two_head_model', '--exp_name', 'n1', '--work_dir', str(tmp_path), '--synthetic', '12',
                      '--max_updates', '3', '--log_interval', '1', '--val_clips', '2'], timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'step 3:' in r.stdout and (tmp_path / 'n1' / 'model_ckpt_steps_3.ckpt').exists()


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


# ---- world size 8: every distributed path at the node size BASELINE configs[3] / [4] name, before 8-GPU hardware runs it ----------------
# Eight rank processes share this box's one GPU with gloo carrying the collectives (device tensors staged through the host), so what runs
# is the real rank logic - partition(..., world=8), the arena broadcast to seven receivers, gather_to_rank0 with eight buckets, core
# binding with LOCAL_WORLD_SIZE = 8, DsBatchSampler(num_replicas=8), the bucketed all-reduce's fixed launch order on eight ranks - with
# only the transport differing from an 8-GPU node (reference: batch_infer.py:164-226, utils/training_utils.py:99-124, 307-319).

def test_batch_infer_with_eight_gloo_ranks_equals_the_plain_run(tmp_path):
    _dataset(tmp_path, rows=64, seconds=2.5)
    base = [str(ROOT / 'batch_infer.py'), '--dataset', str(tmp_path), '--model', str(tmp_path / 'model' / 'model.ckpt'), '--overwrite']
    r = subprocess.run([sys.executable] + base + ['--csv', str(tmp_path / 'plain.csv')], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    plain = (tmp_path / 'plain.csv').read_bytes()
    assert plain.count(b'\n') == 65 and b'note_seq' in plain
    r = _torchrun(8, base + ['--csv', str(tmp_path / 'gloo8.csv')], env={'SOME_AMD_DIST_BACKEND': 'gloo'}, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert (tmp_path / 'gloo8.csv').read_bytes() == plain          # 8 rows per rank, dealt by size; file order and every string restored


def test_bench_with_eight_ranks_sharing_the_gpu():
    """bench.py --gpus 8 exactly as the driver's scaling run launches it (one JSON line from rank 0, max-over-ranks time, whole-job value)."""
    r = _torchrun(8, [str(ROOT / 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--lay', '1', '--batch', '4', '--seconds', '5',
                      '--no-cpu-baseline', '--no-f32-leg', '--no-latency', '--no-secondary', '--no-live-pmc', '--no-kernel-profile'],
                  env={'SOME_AMD_DIST_BACKEND': 'gloo'}, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    res = json.loads(lines[0])
    assert res['process_group'] == {'backend': 'gloo', 'world_size': 8} and res['n_gpus'] == 8 and res['scaling'] == 'weak'
    assert abs(res['value'] - 8 * 4 * 5.0 * 2 / (res['ms_per_step'] * 2 * 1e-3)) < 0.02 * res['value']
    assert res['notes_decoded_last_step'] > 0 and 'e2e_batch_infer' not in res and 'train_epoch' not in res


_DDP8_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ['REPO'])
import torch.distributed as dist
W = int(os.environ['WORLD'])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + os.environ['PORT'], rank=int(os.environ['RANK']), world_size=W)
rank = dist.get_rank()
from some_amd import synth
from some_amd.configs import get_config
from some_amd.training.task import MIDIExtractionTrainer
cfg = get_config('two_head_model', lay=2)
for k in ('conv_drop', 'ffn_latent_drop', 'ffn_out_drop', 'attention_drop'):       # dropout streams are per process: off for the 1-rank comparison
    cfg['midi_extractor_args'][k] = 0.0
cfg = dict(cfg, pl_trainer_precision='bf16', some_amd_ddp_bucket_mb=4)                # what train.py runs: bf16 operands, two lanes,
tr = MIDIExtractionTrainer(cfg, device='cuda:0', seed=100 + rank)                     # weight-gradient side streams, bucketed overlap
assert tr.world == W and tr.grad_sync is not None and len(tr.grad_sync.bounds) >= 4
sample = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_train_batch(seed=21 + rank).items()}     # every rank its own batch
for _ in range(4):
    out = tr.training_step(sample, sync=False)
tr.flush()
assert tr.ops._lane_version[1] > 0                                                    # the bound stream's blocks ran on the second lane
assert tr.grad_sync.launch_order == list(reversed(range(len(tr.grad_sync.bounds))))   # fixed descending order on every rank
flat = tr.model.params.flat
every = [torch.empty_like(flat) for _ in range(W)]
dist.all_gather(every, flat)
assert all(torch.equal(every[0], q) for q in every), 'replicas diverged'
if rank == 0:
    torch.save({'flat': flat.cpu(), 'loss': float(out['total_loss'])}, os.environ['OUT'])
dist.barrier()
dist.destroy_process_group()
print('ok', rank)
'''


def test_eight_rank_data_parallel_training_equals_eight_micro_batches_on_one_rank(tmp_path):
    """Eight trainer processes (bf16, lanes + weight-gradient side streams + bucketed all-reduce overlapped with backward), four
    asynchronous updates, a different batch on every rank: the replicas end bit-identical, and equal - to fp32 summation order - to ONE
    rank that takes the same eight batches as eight micro-batches per update (configs/base.yaml:50 accumulate_grad_batches: the mean of
    eight gradients either way)."""
    import torch
    from some_amd.training.task import MIDIExtractionTrainer
    script = tmp_path / 'ddp8_worker.py'
    script.write_text(_DDP8_WORKER)
    env = dict(os.environ, PORT=str(_port()), REPO=str(ROOT), OUT=str(tmp_path / 'res.pt'), WORLD='8')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(8)]
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-1500:] for o in outs]
    got = torch.load(tmp_path / 'res.pt')['flat'].cuda()
    cfg = get_config('two_head_model', lay=2)
    for k in ('conv_drop', 'ffn_latent_drop', 'ffn_out_drop', 'attention_drop'):
        cfg['midi_extractor_args'][k] = 0.0
    one = MIDIExtractionTrainer(dict(cfg, pl_trainer_precision='bf16'), device='cuda:0', seed=100)        # rank 0's initial weights
    start = one.model.params.flat.clone()
    micro = [{k: torch.from_numpy(v).cuda() for k, v in synth.synth_train_batch(seed=21 + r).items()} for r in range(8)]
    for _ in range(4):
        one.training_step(micro, sync=False)
    one.flush()
    want = one.model.params.flat
    moved = float((want - start).norm())
    err = float((got - want).norm())
    print(f'8 ranks vs 8 micro-batches after 4 updates: |diff| = {err:.3e}, distance travelled {moved:.3e}, ratio {err / moved:.2e}')
    # AdamW's first updates are sign-like (|step| = lr wherever |g| >> eps), so a gradient that differs in its last bits moves a parameter
    # differently only where |g| ~ eps: the replicas' sum and the sequential sum agree to a small fraction of the distance travelled
    assert moved > 0 and err < 2e-3 * moved


def test_train_cli_with_eight_gloo_ranks(tmp_path):
    """train.py under torch.distributed.run with 8 ranks on this GPU: DsBatchSampler(num_replicas=8) columns, core binding, the loader
    threads of eight processes, checkpoint from rank 0 - and the checkpoint loads into the inference class."""
    r = _torchrun(8, [str(ROOT / 'train.py'), '--config', 'two_head_model', '--exp_name', 'w8', '--work_dir', str(tmp_path), '--synthetic', '96',
                      '--max_updates', '4', '--log_interval', '1', '--val_clips', '2'], env={'SOME_AMD_DIST_BACKEND': 'gloo'}, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'step 4:' in r.stdout and r.stdout.count('step 1:') == 1            # rank 0 alone reports
    ckpt = tmp_path / 'w8' / 'model_ckpt_steps_4.ckpt'
    assert ckpt.exists() and (tmp_path / 'w8' / 'config.yaml').exists()
    sys.path.insert(0, str(ROOT))
    from infer import load_inference
    ins, cfg = load_inference(ckpt)
    res = ins.infer([synth.synth_clip(5, 3.0)])[0]
    assert set(res) == {'note_midi', 'note_dur', 'note_rest'} and np.isfinite(res['note_midi']).all()
