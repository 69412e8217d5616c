"""Probe (round 4): what would a captured HIP graph of the training step buy at a given batch shape?

Captures forward + losses + backward + gradient norm of ONE training step (torch.cuda.graph around the trainer's own tape; the learning rate,
dropout seeds and AdamW are left out - a timing probe, not a training mode) and times replays against the eager step body.

    python tools/train_graph_probe.py [--frames 520] [--batch 8] [--steps 30]
"""
import argparse
import ctypes as C
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from some_amd.configs import get_config  # noqa: E402
from some_amd.training import task as task_mod  # noqa: E402
from some_amd.training.ops import Tape  # noqa: E402
from some_amd.training.task import MIDIExtractionTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=520)
    ap.add_argument('--steps', type=int, default=30)
    args = ap.parse_args()
    cfg = get_config('two_head_model', pl_trainer_precision='bf16')
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
    # the batch descriptor uploads its offsets (a synchronous copy): build it once per shape
    cache = {}
    real = task_mod.ClipBatch

    def cached(frame_counts, device, sample_counts=None):
        key = tuple(frame_counts)
        if key not in cache:
            cache[key] = real(frame_counts, device, sample_counts)
        return cache[key]
    task_mod.ClipBatch = cached
    P, ops = tr.model.params, tr.ops

    def body():
        ops.pin_stream()
        try:
            P.zero_grad()
            tape = ops.tape = Tape(ops)
            with torch.no_grad():
                part = tr.run_model(sample)
            tape.backward([(v, 1.0) for v in part.values()])
            ops.tape = None
            sc = ops.scratch(1, 1)
            ops.check(ops.lib.some_train_sumsq(ops.h, C.c_void_p(P.grad.data_ptr()), P.numel, C.c_void_p(tr._sumsq.data_ptr()), C.c_void_p(sc.data_ptr()),
                                               sc.numel(), ops.stream()))
        finally:
            ops.unpin_stream()

    def timed(fn, n):
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(ts))

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    torch.cuda.synchronize()
    eager_ms = timed(body, args.steps)
    g = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(g, stream=side):
        body()
    torch.cuda.synchronize()
    capture_ms = 1e3 * (time.perf_counter() - t0)
    first = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    first_ms = 1e3 * (time.perf_counter() - first)
    replay_ms = timed(g.replay, args.steps)
    print(f'{B} x {T} frames, bf16: eager step body (forward + backward + gradient norm, one sync) {eager_ms:.2f} ms; captured graph replay {replay_ms:.2f} ms '
          f'(capture + instantiate {capture_ms:.0f} ms, first replay {first_ms:.1f} ms); sumsq {float(tr._sumsq.item()):.6e}')


if __name__ == '__main__':
    main()
