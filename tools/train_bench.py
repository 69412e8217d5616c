"""Training-step timing of the HIP training path (BASELINE config 5 shape: configs/two_head_model.yaml, synthetic batch).

    python tools/train_bench.py [--batch 8] [--frames 2584] [--lay 3] [--steps 5]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...   (data parallel)
"""
import argparse
import os
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from some_amd.configs import get_config  # noqa: E402
from some_amd.training.task import MIDIExtractionTrainer  # noqa: E402


def _crc(t):
    import zlib
    return zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=2584)
    ap.add_argument('--lay', type=int, default=3)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mixed', action='store_true', help='mixed precision: 16-bit operands on the matrix pipe (pl_trainer_precision 16-bit)')
    ap.add_argument('--operand', choices=['f16', 'bf16'], default='bf16', help="with --mixed: 'bf16' = pl_trainer_precision bf16 (the reference's configs), 'f16' = 16-mixed")
    ap.add_argument('--digest', action='store_true', help='print CRC-32 digests of the batch, of the gradient after every timed step and of the final parameters (run-to-run determinism probe)')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank, local = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        torch.distributed.init_process_group(os.environ.get('SOME_AMD_DIST_BACKEND', 'nccl'))
    cfg = get_config('two_head_model', lay=args.lay)
    if args.mixed:
        cfg['pl_trainer_precision'] = 'bf16' if args.operand == 'bf16' else '16-mixed'
    tr = MIDIExtractionTrainer(cfg, device=f'cuda:{local}', seed=1)
    B, T = args.batch, args.frames
    rng = np.random.default_rng(rank)
    u2n = np.repeat(np.arange(1, T // 40 + 2), 40)[:T][None].repeat(B, 0)
    sample = {
        'units': torch.from_numpy((rng.standard_normal((B, T, 80)) - 4).astype(np.float32)).cuda(),
        'unit2note': torch.from_numpy(u2n).cuda(),
        'probs': torch.rand(B, T, 128, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1000 + rank)) * 0.1,     # (seeded: the default CUDA generator's seed differs from process to process)
        'bounds': (torch.from_numpy(np.diff(u2n, axis=1, prepend=0)) > 0).float().cuda(),
    }
    for _ in range(args.warmup):
        tr.training_step(sample)
    torch.cuda.synchronize()
    host0 = tr.host_enqueue_s
    t0 = time.perf_counter()
    crcs = []
    for _ in range(args.steps):
        out = tr.training_step(sample)
        if args.digest:
            crcs.append(_crc(tr.model.params.grad))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if args.digest and rank == 0:
        print('digest: batch', ' '.join(f'{_crc(v):08x}' for v in sample.values()), '| grad per step', ' '.join(f'{c:08x}' for c in crcs),
              '| parameters', f'{_crc(tr.model.params.flat):08x}', '| loss', repr(float(out['total_loss'])))
    nb = 2 * args.lay + 2
    f_dense = nb * 12090368 + args.lay * 2097152 + 163840 + 1024 * 128 + 1024          # SURVEY.md section 8(d), per frame, forward
    flops = 3.0 * (f_dense + nb * 2048 * T) * B * T                                     # fwd + 2x bwd
    if rank == 0:
        print(f'two_head_model lay {args.lay} ({("mixed " + tr.mixed_operand) if tr.mixed else "fp32-equivalent"}): {B} x {T} frames/GPU x {world} GPU: {dt * 1e3:.1f} ms/step, '
              f'{world * B * T / dt:.0f} frames/s, {world * B * T * 512 / 44100 / dt:.0f} audio-s/s trained, '
              f'{flops / dt / 1e12:.1f} TFLOP/s/GPU (fwd+bwd model FLOPs), host enqueue {(tr.host_enqueue_s - host0) / args.steps * 1e3:.1f} ms/step, '
              f'loss {out["total_loss"].item():.4f}')
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
