#!/usr/bin/env python
"""BASELINE configs[4] at its own size on ONE GPU: ``train.py configs/two_head_model.yaml`` bf16 over a synthetic 3-hour binarised
dataset, ``max_batch_frames: 80000`` through DsBatchSampler (training/base_task.py:360-395), one epoch - the loop of train.py with a
clock around every update.

    python tools/make_train_dataset.py --dir /tmp/some_ds --hours 3
    python tools/train_epoch_bench.py --dir /tmp/some_ds [--precision bf16] [--max_batch_frames 80000] [--workers 4]

Prints one JSON object: audio-seconds trained per second, step-time percentiles, the fraction of wall time the training thread
waited for data.  Also importable (``run(...)``): bench.py --train adds the result to its JSON line.  Works under
``torch.distributed.run`` (every rank its own DsBatchSampler column; rank 0 reports)."""
import argparse
import json
import os
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def run(data_dir, precision='bf16', max_batch_frames=80000, max_batch_size=8, workers=4, prefetch_factor=2, max_steps=None, device='cuda:0',
        world=1, rank=0, sync=False):
    from some_amd.configs import get_config
    from some_amd.training import data
    from some_amd.training.loader import PrefetchLoader
    from some_amd.training.samplers import DsBatchSampler
    from some_amd.training.task import MIDIExtractionTrainer
    cfg = get_config('two_head_model', binary_data_dir=str(data_dir), pl_trainer_precision=precision, max_batch_frames=max_batch_frames,
                     max_batch_size=max_batch_size, ds_workers=workers, dataloader_prefetch_factor=prefetch_factor)
    trainer = MIDIExtractionTrainer(cfg, device=device, seed=cfg['seed'])
    train_set = data.MIDIExtractionDataset(cfg, cfg['binary_data_dir'], cfg['train_set_name'], allow_aug=True, device=trainer.ops.device)
    sampler = DsBatchSampler(train_set, max_batch_frames=cfg['max_batch_frames'], max_batch_size=cfg['max_batch_size'], num_replicas=world, rank=rank,
                             sort_by_similar_size=cfg['sort_by_len'], required_batch_count_multiple=1,
                             frame_count_grid=cfg['sampler_frame_count_grid'], shuffle_sample=True, shuffle_batch=False, seed=cfg['seed'])
    sampler.set_epoch(0)
    plan = list(sampler)
    if max_steps:
        plan = plan[:max_steps]
    loader = PrefetchLoader(train_set, cfg, trainer.ops.device, workers=workers, prefetch_factor=prefetch_factor)
    hop_s = cfg['hop_size'] / cfg['audio_sample_rate']
    frames_valid = frames_padded = 0
    step_ms, losses = [], []
    # warm-up outside the clock: first-use allocations, kernel attribute calls, pinned-buffer pool - on the two LARGEST batches of the
    # plan, so the caching allocator already holds blocks of every size the epoch asks for (what every epoch after the first sees;
    # `device_allocs_during_epoch` reports what is left)
    warm = PrefetchLoader(train_set, cfg, trainer.ops.device, workers=0)
    biggest = sorted(plan, key=lambda idx: len(idx) * max(int(train_set.sizes[i]) for i in idx))[-2:]
    for mb in warm.batches(biggest):
        trainer.training_step(mb)
    torch.cuda.synchronize()
    skipped = 0
    host0 = trainer.host_enqueue_s
    segs0 = torch.cuda.memory_stats().get('num_device_alloc', 0) if torch.cuda.is_available() else 0
    t0 = time.perf_counter()
    last = t0
    for mb, idx in zip(loader.batches(plan), plan):
        out = trainer.training_step(mb, sync=sync)            # as train.py runs it: no host synchronisation per update (sync=True: one)
        now = time.perf_counter()
        step_ms.append((now - last) * 1e3)
        last = now
        skipped += bool(out['skipped'])
        losses.append(out['total_loss'])                      # a device tensor: read after the epoch
        frames_valid += int(sum(train_set.sizes[i] for i in idx))
        frames_padded += int(mb['units'].shape[0] * mb['units'].shape[1])
    trainer.flush()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    losses = [float(v) for v in losses]
    loader.close()
    st = loader.stats
    ms = np.asarray(step_ms)
    if world > 1:
        t = torch.tensor([wall, float(frames_valid), float(frames_padded)], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t[:1], op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(t[1:], op=torch.distributed.ReduceOp.SUM)
        wall, frames_valid, frames_padded = float(t[0]), int(t[1]), int(t[2])
    return {
        'workload': f'train.py two_head_model (lay 3), pl_trainer_precision {precision}, synthetic binarised dataset {data_dir}: '
                    f'{len(train_set)} items, {int(train_set.sizes.sum()) * hop_s / 3600:.2f} h; DsBatchSampler max_batch_frames {max_batch_frames}, '
                    f'max_batch_size {max_batch_size}; one epoch{" (truncated)" if max_steps else ""}',
        'n_gpus': world, 'updates': len(step_ms), 'skipped_updates': skipped, 'epoch_wall_s': round(wall, 3),
        'host_sync_per_update': bool(sync),
        'lanes': trainer.ops.lanes, 'weight_gradient_lanes': bool(trainer.ops.wgrad_lanes), 'binding': 'fastcall' if getattr(trainer.ops.lib, 'fastcall', False) else 'ctypes',
        'audio_s_per_s_trained': round(frames_valid * hop_s / wall, 1), 'frames_per_s': round(frames_valid / wall, 1),
        'padding_overhead': round(frames_padded / max(frames_valid, 1), 4),
        'step_ms': {'mean': round(float(ms.mean()), 2), 'p10': round(float(np.percentile(ms, 10)), 2), 'p50': round(float(np.percentile(ms, 50)), 2),
                    'p90': round(float(np.percentile(ms, 90)), 2), 'max': round(float(ms.max()), 2)},
        'host_enqueue_ms_mean': round((trainer.host_enqueue_s - host0) / max(len(step_ms), 1) * 1e3, 2),
        'step_ms_first_quarter_mean': round(float(ms[:max(len(ms) // 4, 1)].mean()), 2), 'step_ms_last_quarter_mean': round(float(ms[-max(len(ms) // 4, 1):].mean()), 2),
        'device_allocs_during_epoch': int(torch.cuda.memory_stats().get('num_device_alloc', 0) - segs0),
        'data_wait_s': round(st['wait_s'], 3), 'data_wait_frac_of_wall': round(st['wait_s'] / wall, 4),
        'host_collate_s_in_workers': round(st['host_collate_s'], 3), 'loader_workers': workers, 'prefetch_factor': prefetch_factor,
        'loss_first_last': [round(float(np.mean(losses[:5])), 4), round(float(np.mean(losses[-5:])), 4)],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', required=True)
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--max_batch_frames', type=int, default=80000)
    ap.add_argument('--max_batch_size', type=int, default=8)
    ap.add_argument('--workers', type=int, default=4)
    ap.add_argument('--prefetch_factor', type=int, default=2)
    ap.add_argument('--max_steps', type=int, default=None)
    ap.add_argument('--sync', action='store_true', help='read the gradient norm back after every update (round-3 behaviour)')
    a = ap.parse_args()
    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        torch.distributed.init_process_group(os.environ.get('SOME_AMD_DIST_BACKEND', 'nccl'))
    res = run(a.dir, a.precision, a.max_batch_frames, a.max_batch_size, a.workers, a.prefetch_factor, a.max_steps, f'cuda:{local}', world, rank, a.sync)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
