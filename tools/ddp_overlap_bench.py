#!/usr/bin/env python
"""How much of the data-parallel gradient all-reduce stays EXPOSED behind the four-stream backward pass?

One GPU, one process: the trainer's own BucketedGradSync (training/grad_sync.py: fixed descending launch order, lanes + weight-gradient
side streams joined in front of every bucket) with the collective replaced by a MODEL of it - a device-side wait on a communication
stream of its own, started behind an event of the issuing stream exactly as RCCL orders itself, lasting
``bytes x ring_ms_per_32MiB / 32 MiB + launch_us`` (default 0.7 ms per 32 MiB: one xGMI ring link at ~50 GB/s effective, 8 hops; the
figure grad_sync.py's bucket size was chosen for), one collective at a time (RCCL serialises on its stream).  The model holds no CUs
(RCCL's kernels take a few: not modelled) and no HBM bandwidth.

    python tools/ddp_overlap_bench.py [--frames 520 2584] [--buckets 4 8 16 32 64 256]

Reported per batch shape and bucket size: step time without any synchronisation, with it, the difference (= exposed all-reduce time per
step), the modelled ring time of the whole gradient, and how many buckets left from inside the backward pass."""
import argparse
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from some_amd.configs import get_config  # noqa: E402
from some_amd.training.grad_sync import BucketedGradSync  # noqa: E402
from some_amd.training.task import MIDIExtractionTrainer  # noqa: E402


class _Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def attach_model_sync(tr, bucket_mb, cycles_per_ms, ring_ms_per_32mib, launch_us, stats, comm_priority=0):
    P = tr.model.params
    order = [(P.views[k], P.offsets[k], (P.views[k].numel() + 63) // 64 * 64) for k in P.param_names]
    gs = BucketedGradSync(P.grad, order, None, int(bucket_mb * (1 << 20)), names=list(P.param_names))
    gs.before_launch = tr.ops.sync_other_lane
    tr.ops.register_grad_sinks(P.views.values(), gs.mark)
    comm = torch.cuda.Stream(priority=comm_priority)      # -1: a high-priority stream (hardware queues of its own in the HIP runtime)

    def launch(bucket):
        a, b = gs.bounds[bucket]
        if gs.before_launch is not None:
            gs.before_launch()
        ev0 = torch.cuda.Event()
        ev0.record()
        comm.wait_event(ev0)                       # the collective orders itself behind the issuing stream's work at this moment
        ms = (b - a) * 4 / (32 << 20) * ring_ms_per_32mib + launch_us * 1e-3
        with torch.cuda.stream(comm):
            torch.cuda._sleep(int(ms * cycles_per_ms))
            ev1 = torch.cuda.Event()
            ev1.record()
        gs.work[bucket] = _Work(ev1)
        gs.launched[bucket] = True
        gs.launch_order.append(bucket)
        stats['ring_ms'] += ms
        stats['in_backward'] += 1 if gs.armed else 0
    gs._launch = launch
    tr.grad_sync = gs
    return gs


def sample_of(B, T):
    rng = np.random.default_rng(0)
    u2n = np.repeat(np.arange(1, T // 40 + 2), 40)[:T][None].repeat(B, 0)
    return {'units': torch.from_numpy((rng.standard_normal((B, T, 80)) - 4).astype(np.float32)).cuda(),
            'unit2note': torch.from_numpy(u2n).cuda(),
            'probs': torch.rand(B, T, 128, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1000)) * 0.1,
            'bounds': (torch.from_numpy(np.diff(u2n, axis=1, prepend=0)) > 0).float().cuda()}


def time_steps(tr, sample, steps, warmup=3):
    for _ in range(warmup):
        tr.training_step(sample, sync=False)
    tr.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.training_step(sample, sync=False)
    tr.flush()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, nargs='+', default=[520, 2584])
    ap.add_argument('--buckets', type=float, nargs='+', default=[4, 8, 16, 32, 64, 256])
    ap.add_argument('--ring-ms-per-32mib', type=float, default=0.7)
    ap.add_argument('--launch-us', type=float, default=20.0)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--lay', type=int, default=3)
    ap.add_argument('--comm-priority', type=int, default=0, help='priority of the modelled communication stream (-1 = high, as ProcessGroupNCCL with is_high_priority_stream)')
    args = ap.parse_args()
    # calibrate torch.cuda._sleep: cycles per millisecond
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(50_000_000)
    e1.record()
    torch.cuda.synchronize()
    cycles_per_ms = 50_000_000 / e0.elapsed_time(e1)
    cfg = get_config('two_head_model', lay=args.lay)
    cfg['pl_trainer_precision'] = 'bf16'
    print(f'model of the collective: {args.ring_ms_per_32mib} ms per 32 MiB + {args.launch_us} us per launch, one at a time on its own stream; '
          f'two_head_model lay {args.lay} bf16, asynchronous updates, communication stream priority {args.comm_priority}; _sleep calibration {cycles_per_ms / 1e3:.0f} cycles/us')
    for T in args.frames:
        sample = sample_of(args.batch, T)
        base = MIDIExtractionTrainer(cfg, device='cuda', seed=1)
        t_base = time_steps(base, sample, args.steps)
        grad_mb = base.model.params.numel * 4 / (1 << 20)
        del base
        print(f'{args.batch} x {T} frames: step without synchronisation {t_base:.2f} ms; flat gradient {grad_mb:.0f} MiB')
        for mb in args.buckets:
            tr = MIDIExtractionTrainer(cfg, device='cuda', seed=1)
            stats = {'ring_ms': 0.0, 'in_backward': 0}
            gs = attach_model_sync(tr, mb, cycles_per_ms, args.ring_ms_per_32mib, args.launch_us, stats, args.comm_priority)
            t = time_steps(tr, sample, args.steps)
            n_steps = args.steps + 3
            print(f'    bucket {mb:5g} MiB: {len(gs.bounds):3d} buckets, {stats["in_backward"] / n_steps:5.1f} of them launched inside backward, modelled ring time '
                  f'{stats["ring_ms"] / n_steps:5.2f} ms per step | step {t:6.2f} ms | exposed {t - t_base:+5.2f} ms ({100.0 * (t - t_base) / max(stats["ring_ms"] / n_steps, 1e-9):4.0f} % of the ring time)')
            del tr, gs


if __name__ == '__main__':
    main()
