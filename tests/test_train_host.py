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


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ['REPO'])
import torch.distributed as dist
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + os.environ['PORT'], rank=int(os.environ['RANK']), world_size=int(os.environ.get('WORLD', '2')))
W = dist.get_world_size()
rank = dist.get_rank()
# the trainer's synchronisation protocol on plain tensors: broadcast of the flat parameters from rank 0, one summing
# all-reduce of the flat gradient, averaging folded into the optimiser's grad_scale = 1 / world
flat = torch.full((1000,), float(rank + 1))
dist.broadcast(flat, src=0)
assert torch.equal(flat, torch.ones(1000))
grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
dist.all_reduce(grad, op=dist.ReduceOp.SUM)
mean = grad * (1.0 / W)
assert torch.allclose(mean, torch.arange(1000, dtype=torch.float32) * ((W + 1) / 2))
# every rank plans the whole epoch itself (no communication) and keeps its own column of the batch grid
import numpy as np
from some_amd.training.samplers import DsBatchSampler
class Lengths:
    _sizes = np.arange(10, 210, 10)
    def __len__(self): return len(self._sizes)
    def num_frames(self, i): return self._sizes[i]
sm = DsBatchSampler(Lengths(), 400, 4, num_replicas=W, rank=rank, shuffle_sample=True, seed=3)
sm.set_epoch(1)
gathered = [None] * W
dist.all_gather_object(gathered, [list(map(int, b)) for b in sm])
every = [p for g in gathered for p in g]
assert len({len(g) for g in gathered}) == 1 and {i for p in every for i in p} == set(range(20))      # same update count on every rank
assert sum(len(p) for p in every) <= 20 + 4 * (W - 1)  # the batch grid is filled up to a multiple of W by repeating at most W - 1 batches
# the columns of the grid are disjoint batches: no batch is trained twice in one update (utils/training_utils.py:99-124)
for step in range(len(gathered[0])):
    row = [tuple(g[step]) for g in gathered]
    assert len(set(row)) == W or sum(len(p) for p in every) > 20, row
dist.barrier()
dist.destroy_process_group()
print('ok', rank)
'''


@pytest.mark.parametrize('world', [2, 8])
def test_two_rank_gloo_gradient_sync(tmp_path, world):
    import os
    import pathlib
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    root = pathlib.Path(__file__).resolve().parents[1]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), PORT=str(port), REPO=str(root), WORLD=str(world), OMP_NUM_THREADS='1'),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


# ---- batch planning: the reference's samplers (utils/training_utils.py:45-176, utils/__init__.py:60-111) --------------
class _Lengths:
    def __init__(self, sizes):
        self._sizes = np.asarray(sizes)

    def __len__(self):
        return len(self._sizes)

    def num_frames(self, i):
        return self._sizes[i]


@pytest.mark.parametrize('case', ['single', 'ddp8', 'ddp2_accum4', 'ddp3_unsorted_drop', 'ddp4_few'])
def test_ds_batch_sampler_plans_equal_the_reference(golden_dir, case):
    """Same batches, in the same order, on every rank and epoch as the reference's DsBatchSampler - the numpy Generator
    call sequence is part of the contract."""
    from some_amd.training.samplers import DsBatchSampler
    g = json.loads((golden_dir / 'samplers.json').read_text())[case]
    ds = _Lengths(g['lengths'])
    seen = {}
    for key, want in g['plans'].items():
        rank, epoch = (int(p.lstrip('rankepoch')) for p in key.split('.'))
        sm = DsBatchSampler(ds, g['max_batch_frames'], g['max_batch_size'], num_replicas=g['num_replicas'], rank=rank,
                            frame_count_grid=g['grid'], required_batch_count_multiple=g['multiple'],
                            sort_by_similar_size=g['sort'], shuffle_sample=True, shuffle_batch=g['shuffle_batch'], seed=g['seed'],
                            drop_last=g['drop_last'])
        sm.set_epoch(epoch)
        got = [list(map(int, b)) for b in sm]
        assert got == want, key
        assert len(sm) == len(want) and len(want) % g['multiple'] == 0
        for b in got:                                         # the budget the plan exists for
            assert len(b) <= g['max_batch_size'] and len(b) * max(g['lengths'][i] for i in b) <= g['max_batch_frames']
        seen.setdefault(epoch, []).append(got)
    for epoch, per_rank in seen.items():                      # every rank takes the same number of steps
        assert len({len(p) for p in per_rank}) == 1
        if not g['drop_last']:
            assert {i for p in per_rank for b in p for i in b} == set(range(len(ds)))


def test_eval_sampler_and_batch_by_size_equal_the_reference(golden_dir):
    from some_amd.training.samplers import DsEvalBatchSampler, batch_by_size
    g = json.loads((golden_dir / 'samplers.json').read_text())
    ds = _Lengths(g['eval']['lengths'])
    assert list(DsEvalBatchSampler(ds, 20000, 4, rank=0, batch_by_size=False)) == g['eval']['rank0_fixed']
    assert [list(map(int, b)) for b in DsEvalBatchSampler(ds, 3000, 4, rank=0, batch_by_size=True)] == g['eval']['rank0_by_size']
    assert list(DsEvalBatchSampler(ds, 20000, 4, rank=1)) == g['eval']['rank1'] == [[0]]
    got = batch_by_size(list(range(50)), ds.num_frames, max_batch_frames=4000, max_batch_size=7, required_batch_size_multiple=2)
    assert [list(map(int, b)) for b in got] == g['batch_by_size_multiple']
    with pytest.raises(AssertionError, match='exceeds'):
        batch_by_size([0], lambda i: 5000, max_batch_frames=4000)
    # fewer batches than replicas: the reference's sampler divides by zero for the ranks left without one (utils/training_utils.py:115);
    # the same exception type here, naming the cause
    from some_amd.training.samplers import DsBatchSampler
    few = _Lengths([100] * 6)
    assert len(list(DsBatchSampler(few, 20000, 1, num_replicas=8, rank=5, shuffle_sample=True, seed=1))) == 1
    with pytest.raises(ZeroDivisionError, match='6 batch'):
        list(DsBatchSampler(few, 20000, 1, num_replicas=8, rank=6, shuffle_sample=True, seed=1))


_SYNC_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ['REPO'])
import torch.distributed as dist
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + os.environ['PORT'], rank=int(os.environ['RANK']), world_size=int(os.environ.get('WORLD', '2')))
W = dist.get_world_size()
rank = dist.get_rank()
from some_amd.training.grad_sync import BucketedGradSync
def same(got, want):
    """Two addends sum to the same bits in any order; with more ranks gloo's ring adds a bucket's elements in another rank order than
    the whole buffer's, so the comparison is to fp32 summation order - and the replicas must still hold IDENTICAL bits."""
    if W == 2:
        return torch.equal(got, want)
    every = [torch.empty_like(got) for _ in range(W)]
    dist.all_gather(every, got)
    return all(torch.equal(every[0], q) for q in every) and torch.allclose(got, want, rtol=2e-5, atol=2e-6 * float(want.abs().max()))
# a flat parameter / gradient buffer cut into views, as FlatParams does; "layers" used in order by the forward pass
sizes = [4096, 64] * 4
offs, pos = [], 0
for n in sizes:
    offs.append(pos); pos += n
flat = torch.linspace(-1, 1, pos).clone()
grad = torch.zeros(pos)
views = []
for n, off in zip(sizes, offs):
    p = flat[off:off + n]
    p.requires_grad_(True)
    p.grad = grad[off:off + n]
    views.append(p)
STATIC = os.environ.get('STATIC') == '1'
sync = BucketedGradSync(grad, [(p, off, n) for p, off, n in zip(views, offs, sizes)], None, bucket_bytes=5000 * 4, static_graph=STATIC)
assert sync.bounds == [(0, 4160), (4160, 8320), (8320, 12480), (12480, 16640)]
def loss_fn(skip_last=False):
    x = torch.full((64,), 0.5 + rank)
    for i in range(0, len(views) - (2 if skip_last else 0), 2):
        w, b = views[i].view(-1, 64), views[i + 1]
        x = torch.tanh(w @ x) + b * x
    return (x * x).sum()
for skip_last in (False, True):
    # reference: plain backward, one all-reduce of the whole buffer
    grad.zero_()
    loss_fn(skip_last).backward()
    want = grad.clone()
    dist.all_reduce(want)
    # micro-batch accumulation: the first backward is NOT reduced, the armed one is, bucket by bucket, during backward
    grad.zero_()
    loss_fn(skip_last).backward()
    assert not sync.armed
    first = grad.clone()
    sync.arm()
    loss_fn(skip_last).backward()
    launched_in_backward = list(sync.launch_order)
    sync.finish()
    local_two = first * 2
    want_two = local_two.clone()
    dist.all_reduce(want_two)
    assert same(grad, want_two), (grad - want_two).abs().max()
    # last layers first, in a FIXED descending order (collectives pair up across ranks by issue order): bucket k leaves only after
    # every bucket above it.  The first pass in which the top bucket's parameters get no gradient (skip_last) therefore sends
    # everything from finish(); with static_graph the parameters are then known to be absent and the next pass overlaps again,
    # without it every pass that leaves them out is sent from finish() (always correct, whatever the other rank does).
    assert launched_in_backward == sorted(launched_in_backward, reverse=True)
    assert len(launched_in_backward) == (0 if skip_last else len(sync.bounds))
    assert sync.launch_order == list(reversed(range(len(sync.bounds))))
    assert sync.unreported() == (['param6', 'param7'] if skip_last else [])
    grad.zero_()
    sync.arm()
    loss_fn(skip_last).backward()
    in_backward = list(sync.launch_order)
    sync.finish()
    assert same(grad, want)
    assert len(in_backward) >= (3 if STATIC or not skip_last else 0) and sync.launch_order == list(reversed(range(len(sync.bounds))))
# a parameter counted as absent that reports after all (skip_last -> full model): its gradient is in the buffer before its bucket
# leaves - correct result, and it is no longer absent afterwards
grad.zero_()
loss_fn(False).backward()
want = grad.clone()
dist.all_reduce(want)
assert sync.absent == ({6, 7} if STATIC else set())
grad.zero_()
sync.arm()
loss_fn(False).backward()
sync.finish()
assert same(grad, want) and sync.absent == set()
if not STATIC:
    # a gradient path that switches on and off from step to step, DIFFERENTLY on the two ranks: never an error, always the sum
    for step in range(6):
        skip = (step + rank) % 2 == 1
        grad.zero_()
        loss_fn(skip).backward()
        want = grad.clone()
        dist.all_reduce(want)
        grad.zero_()
        sync.arm()
        loss_fn(skip).backward()
        sync.finish()
        assert same(grad, want), step
        assert sync.launch_order == list(reversed(range(len(sync.bounds))))
# a parameter that reports twice in one armed pass raises instead of launching its bucket early
sync.arm()
sync.mark(views[7])
try:
    sync.mark(views[7])
    raise SystemExit('second report must raise')
except RuntimeError as e:
    assert 'twice' in str(e)
sync.finish()
try:
    sync.finish()
    raise SystemExit('finish() without arm() must raise')
except RuntimeError:
    pass
dist.barrier()
dist.destroy_process_group()
print('ok', rank)
'''


@pytest.mark.parametrize('static_graph,world', [(False, 2), (True, 2), (False, 8), (True, 8)])
def test_bucketed_gradient_sync_overlaps_backward_and_equals_one_all_reduce(tmp_path, static_graph, world):
    """grad_sync.BucketedGradSync on gloo, world size 2 (CPU): buckets are launched from inside the backward pass, last
    layers first; the reduced flat gradient equals a single all-reduce bit for bit, with gradient accumulation and with
    parameters that receive no gradient."""
    import os
    import pathlib
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = tmp_path / 'sync_worker.py'
    script.write_text(_SYNC_WORKER)
    root = pathlib.Path(__file__).resolve().parents[1]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), PORT=str(port), REPO=str(root), STATIC=str(int(static_graph)), WORLD=str(world), OMP_NUM_THREADS='1'),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


_MARK_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ['REPO'])
import torch.distributed as dist
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + os.environ['PORT'], rank=int(os.environ['RANK']), world_size=int(os.environ.get('WORLD', '2')))
W = dist.get_world_size()
rank = dist.get_rank()
from some_amd.training.grad_sync import BucketedGradSync
def same(got, want):
    """Two addends sum to the same bits in any order; with more ranks gloo's ring adds a bucket's elements in another rank order than
    the whole buffer's, so the comparison is to fp32 summation order - and the replicas must still hold IDENTICAL bits."""
    if W == 2:
        return torch.equal(got, want)
    every = [torch.empty_like(got) for _ in range(W)]
    dist.all_gather(every, got)
    return all(torch.equal(every[0], q) for q in every) and torch.allclose(got, want, rtol=2e-5, atol=2e-6 * float(want.abs().max()))
sizes = [4096, 64] * 4
offs, pos = [], 0
for n in sizes:
    offs.append(pos); pos += n
flat = torch.linspace(-1, 1, pos).clone()
grad = torch.zeros(pos)
views = []
for n, off in zip(sizes, offs):
    p = flat[off:off + n]
    p.requires_grad_(True)
    p.grad = grad[off:off + n]
    views.append(p)
sync = BucketedGradSync(grad, [(p, off, n) for p, off, n in zip(views, offs, sizes)], None, bucket_bytes=5000 * 4)

class SinkLinear(torch.autograd.Function):
    """y = tanh(W x) + b * x with the parameter gradients written in place + sync.mark (as TrainOps' gradient sinks do)."""
    @staticmethod
    def forward(ctx, x, w, b, use_sink):
        ctx.save_for_backward(x, w, b)
        ctx.wp, ctx.bp, ctx.use_sink = w, b, use_sink
        return torch.tanh(w.view(-1, 64) @ x) + b * x
    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        t = torch.tanh(w.view(-1, 64) @ x)
        dpre = dy * (1 - t * t)
        dw = torch.outer(dpre, x).reshape(-1)
        db = dy * x
        dx = w.view(-1, 64).t() @ dpre + dy * b
        if ctx.use_sink:
            with torch.no_grad():
                ctx.wp.grad += dw
                ctx.bp.grad += db
            active.mark(ctx.wp); active.mark(ctx.bp)
            return dx, None, None, None
        return dx, dw, db, None

def loss_fn(sink_layers):
    x = torch.full((64,), 0.5 + rank)
    for i in range(0, len(views), 2):
        x = SinkLinear.apply(x, views[i], views[i + 1], (i // 2) in sink_layers)
    return (x * x).sum()

# `wide`: two layers (four parameters) per bucket - the shape that exposed round 2's bug: autograd fires a parameter's
# post-accumulate hook even when the backward function returned None for it, so a marked parameter used to be counted twice and a
# bucket left when HALF of its gradients were in the buffer (layer 3 marked twice -> bucket {2, 3} sent before layer 2's backward)
wide = BucketedGradSync(grad, [(p, off, n) for p, off, n in zip(views, offs, sizes)], None, bucket_bytes=9000 * 4)
assert wide.bounds == [(0, 8320), (8320, 16640)]
for which in (sync, wide):
    active = which
    for sink_layers in [(), (0, 1, 2, 3), (0, 2), (1, 3), (3,)]:
        grad.zero_()
        loss_fn(()).backward()
        want = grad.clone(); dist.all_reduce(want)
        grad.zero_()
        which.arm()
        loss_fn(sink_layers).backward()
        order = list(which.launch_order)
        fired = list(which.fire_order)
        which.finish()
        assert same(grad, want), (sink_layers, float((grad - want).abs().max()))
        assert order == list(reversed(range(len(which.bounds)))), order              # every bucket launched from inside backward, last layers first
        assert sorted(fired) == list(range(8)) and which.unreported() == []          # every parameter counted exactly once
        assert sum(which.echoed) == 2 * len(sink_layers)                             # the hooks behind the marks were seen and ignored
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''


@pytest.mark.parametrize('world', [2, 8])
def test_gradient_sync_mark_stands_in_for_the_hook(tmp_path, world):
    """BucketedGradSync.mark(param): a backward function that writes a parameter's gradient into the flat buffer itself and returns
    None for it (TrainOps' gradient sinks) reports the parameter through mark() instead of autograd's post-accumulate hook.  gloo,
    world size 2, CPU: any mix of marked and autograd-accumulated parameters - also inside one bucket - gives the single all-reduce
    result bit for bit, buckets still go out during backward.  Pins the root cause of round 2's diverging replicas: autograd fires
    the post-accumulate hook of a parameter even when backward returned None for it, so mark() + hook used to count twice; with several
    layers per bucket (`wide`) the bucket then left before all its gradients were written."""
    import os
    import pathlib
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = tmp_path / 'mark_worker.py'
    script.write_text(_MARK_WORKER)
    root = pathlib.Path(__file__).resolve().parents[1]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), PORT=str(port), REPO=str(root), WORLD=str(world), OMP_NUM_THREADS='1'),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_prefetch_loader_batches_equal_the_collater(golden_dir):
    """training/loader.py (the reference's DataLoader role, training/base_task.py:374-380): batches built by worker threads (host
    padding) + the device half of the collater equal MIDIExtractionDataset.collater (training/me_task.py:26-52) tensor for tensor,
    in plan order, for any worker count - on the libhdf5-written fixture (tests/golden/binary)."""
    from some_amd.configs import get_config
    from some_amd.training import data
    from some_amd.training.loader import PrefetchLoader
    cfg = get_config('two_head_model')
    ds = data.MIDIExtractionDataset(cfg, golden_dir / 'binary', 'train')
    rng = np.random.default_rng(0)
    plan = [list(map(int, rng.choice(len(ds), size=int(rng.integers(1, 6)), replace=False))) for _ in range(11)]
    want = [ds.collater([ds[i] for i in idx]) for idx in plan]
    for workers, pf in [(0, 1), (1, 1), (3, 2)]:
        loader = PrefetchLoader(ds, cfg, 'cpu', workers=workers, prefetch_factor=pf)
        got = list(loader.batches(plan))
        loader.close()
        assert len(got) == len(want) and loader.stats['batches'] == len(plan)
        for a, b in zip(got, want):
            assert set(a) == set(b)
            for k in b:
                if torch.is_tensor(b[k]):
                    assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
                else:
                    assert a[k] == b[k]
    assert list(PrefetchLoader(ds, cfg, 'cpu').batches([])) == []


def test_ffn_dropout_generator_statistics():
    """The fused FFN epilogues' dropout bits (csrc/train_gemm16s.hip, restated in some_amd/training/dropout_bits.py): keep rate at the
    requested p, no visible correlation between neighbouring cells (columns, rows, the two halves of a word) or between call sites."""
    from some_amd.training.dropout_bits import ffn_keep_mask, key_words, threshold
    M, N = 512, 2048
    for p in (0.1, 0.5):
        a = ffn_keep_mask(12345, M, N, p).astype(np.float64)
        q = 1.0 - threshold(p) / 65536.0
        sd = (q * (1 - q) / a.size) ** 0.5
        assert abs(a.mean() - q) < 4 * sd
        c = a - q
        for u, v in ((c[:, :-1], c[:, 1:]), (c[:-1], c[1:]), (c[0::2], c[1::2]), (c[:-2], c[2:]), (c[:, :-32], c[:, 32:])):
            corr = float((u * v).mean()) / (q * (1 - q))
            assert abs(corr) < 5 / u.size ** 0.5, corr
        b = ffn_keep_mask(12346, M, N, p).astype(np.float64) - q                   # the next call site's seed
        assert abs(float((c * b).mean()) / (q * (1 - q))) < 5 / a.size ** 0.5
        assert abs(a.mean(0) - q).max() < 6 * (q * (1 - q) / M) ** 0.5             # per column and per row too
        assert abs(a.mean(1) - q).max() < 6 * (q * (1 - q) / N) ** 0.5
    assert ffn_keep_mask(1, 64, 64, 0.0).all()
    assert key_words(1) != key_words(2)


def test_quant_collater_equals_the_reference_collater(golden_dir):
    """some_amd.training.data.quant_collater against the batch the REFERENCE's own collater text produced from the same items
    (tests/golden/train_step_quant.npz, oracle/make_golden.py:_ref_quant_collater; training/me_quant_task.py:14-27): padding values,
    midi_idx (-1 on padding frames and behind padded notes), boundaries - exactly."""
    import numpy as np
    import torch
    from some_amd.configs import get_config
    from some_amd.training import data
    g = np.load(golden_dir / 'train_step_quant.npz')
    items = []
    i = 0
    while f'item{i}.units' in g.files:
        items.append({k: torch.from_numpy(g[f'item{i}.{k}']) for k in ('units', 'pitch', 'note_midi', 'note_dur', 'unit2note')})
        i += 1
    assert len(items) == 2
    batch = data.quant_collater(items, get_config('quant_two_head_model', lay=1))
    assert batch['size'] == 2
    for k in ('units', 'note_midi', 'note_dur', 'unit2note', 'midi_idx', 'bounds'):
        want = g['batch.' + k]
        assert batch[k].dtype == torch.from_numpy(want).dtype and np.array_equal(batch[k].numpy(), want), k
    assert (batch['midi_idx'][1, -26:] == -1).all() and (batch['midi_idx'] == 128).any()       # padding frames ignored, a rest class present
    # a continuous item -> quantised fields (me_quant_binarizer.py:25-32)
    q = data.quantize_item({'units': torch.zeros(4, 80), 'pitch': torch.zeros(4), 'note_midi': torch.tensor([60.4, 61.5, 62.5]),
                            'note_rest': torch.tensor([False, True, False]), 'note_dur': torch.tensor([1, 1, 2]), 'unit2note': torch.tensor([1, 2, 3, 3])})
    assert q['note_midi'].tolist() == [60, 128, 62] and 'note_rest' not in q                   # torch.round: half to even
