"""Which torch (ATen) operators the training step still launches, by the line of some_amd/ that issues them.

The step's own kernels go through the C ABI; everything torch launches besides them (element-wise temporaries, copies, fills) costs
the host 8 - 15 us per operator at the reference's batch shape, where the step is host-bound.  torch.profiler with Python stacks
over a few steps, grouped by (operator, innermost some_amd frame).

    python tools/train_eager_ops.py [--batch 8] [--frames 520] [--lay 3] [--steps 3]
"""
import argparse
import collections
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from some_amd.configs import get_config  # noqa: E402
from some_amd.training.task import MIDIExtractionTrainer  # noqa: E402

SKIP = ('aten::empty', 'aten::empty_like', 'aten::empty_strided', 'aten::view', 'aten::reshape', 'aten::as_strided', 'aten::slice', 'aten::select',
        'aten::t', 'aten::transpose', 'aten::contiguous', 'aten::detach', 'aten::alias', 'aten::_unsafe_view', 'aten::unsqueeze', 'aten::squeeze',
        'aten::expand', 'aten::permute', 'aten::result_type', 'aten::item', 'aten::_local_scalar_dense', 'aten::is_nonzero', 'aten::resolve_conj',
        'aten::resolve_neg', 'aten::lift_fresh', 'aten::narrow', 'aten::unbind', 'aten::split', 'aten::chunk', 'aten::flatten', 'aten::numel')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=520)
    ap.add_argument('--lay', type=int, default=3)
    ap.add_argument('--steps', type=int, default=3)
    args = ap.parse_args()
    cfg = get_config('two_head_model', lay=args.lay)
    cfg['pl_trainer_precision'] = 'bf16'
    tr = MIDIExtractionTrainer(cfg, device='cuda:0', seed=1)
    B, T = args.batch, args.frames
    rng = np.random.default_rng(0)
    u2n = np.repeat(np.arange(1, T // 40 + 2), 40)[:T][None].repeat(B, 0)
    sample = {
        'units': torch.from_numpy((rng.standard_normal((B, T, 80)) - 4).astype(np.float32)).cuda(),
        'unit2note': torch.from_numpy(u2n).cuda(),
        'probs': torch.rand(B, T, 128, device='cuda') * 0.1,
        'bounds': (torch.from_numpy(np.diff(u2n, axis=1, prepend=0)) > 0).float().cuda(),
    }
    for _ in range(3):
        tr.training_step(sample)
    torch.cuda.synchronize()
    # (stacks of CPU-only profiles need the verbose experimental config on this torch build)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True,
                                experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        for _ in range(args.steps):
            tr.training_step(sample)
        torch.cuda.synchronize()
    by = collections.Counter()
    host_us = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.name in SKIP or ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
            continue
        where = '?'
        for fr in ev.stack or ():
            if 'some_amd/' in fr:
                where = fr.split('some_amd/')[-1]
                break
        by[(ev.name, where)] += 1
        host_us[(ev.name, where)] += ev.cpu_time_total
    total = 0
    print(f'top-level ATen operators per step at {B} x {T} frames (bf16), by issuing line; host us per step')
    for k, n in sorted(by.items(), key=lambda kv: -host_us[kv[0]]):
        print(f'{n / args.steps:7.1f}  {host_us[k] / args.steps:8.1f} us  {k[0]:28s} {k[1]}')
        total += host_us[k]
    print(f'total {sum(by.values()) / args.steps:.0f} operators, {total / args.steps / 1e3:.2f} ms of host time per step')


if __name__ == '__main__':
    main()
