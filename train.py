#!/usr/bin/env python
"""``python train.py --config CONFIG --exp_name EXP [--work_dir DIR]`` - the reference's training command
(train.py:27-110) on the HIP training path: MIDIExtractionTask losses, AdamW + WarmupLR, checkpoints in the
Lightning layout the inference classes load (``{'state_dict': {'model.<key>': ...}}`` + config.yaml beside it).

Data: the reference reads a binarised dataset through h5py, which this image lacks; ``--synthetic N`` trains on N
synthetic sung clips with known notes instead (units from the HIP log-mel front end).  Multi-GPU: launch with
``python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...`` - one process per GPU, rank 0's
initial weights broadcast once, one all-reduce of the flat gradient per step (RCCL), every rank its own batches."""
import os
import pathlib

import click
import torch
import yaml

from some_amd import configs


def _load_config(config: str) -> dict:
    if config in configs.config_names():
        return configs.get_config(config)
    p = pathlib.Path(config)
    stem = p.stem
    with open(p, 'r', encoding='utf8') as f:
        user = yaml.safe_load(f) or {}
    base = configs.get_config(stem) if stem in configs.config_names() else configs.get_config('two_head_model')
    base.update({k: v for k, v in user.items() if k != 'base_config'})
    return base


@click.command(help='Train a SOME model')
@click.option('--config', required=True, metavar='FILE', help='Path to the configuration file (or a built-in config name)')
@click.option('--exp_name', required=True, metavar='EXP', help='Name of the experiment')
@click.option('--work_dir', required=False, metavar='DIR', help='Directory to save the experiment')
@click.option('--synthetic', type=int, default=0, metavar='N', help='Train on N synthetic clips with known notes')
@click.option('--max_updates', type=int, default=None, help='Override max_updates')
@click.option('--log_interval', type=int, default=10)
@click.option('--val_clips', type=int, default=8, help='held-out synthetic clips for the validation pass')
def train(config, exp_name, work_dir, synthetic, max_updates, log_interval, val_clips):
    from some_amd.training import data
    from some_amd.training.task import MIDIExtractionTrainer
    cfg = _load_config(config)
    work = (pathlib.Path(work_dir) if work_dir else pathlib.Path(__file__).parent / 'experiments') / exp_name
    assert not work.exists() or work.is_dir(), f'Path \'{work}\' is not a directory.'
    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        torch.distributed.init_process_group(os.environ.get('SOME_AMD_DIST_BACKEND', 'nccl'))
    if rank == 0:
        work.mkdir(parents=True, exist_ok=True)
        with open(work / 'config.yaml', 'w', encoding='utf8') as f:
            yaml.safe_dump(cfg, f)
    if synthetic <= 0:
        raise click.UsageError('reading binary_data_dir needs h5py, which is not available here: pass --synthetic N')
    trainer = MIDIExtractionTrainer(cfg, device=f'cuda:{local}', seed=cfg.get('seed', 114514))
    timestep = cfg['hop_size'] / cfg['audio_sample_rate']
    rng_len = torch.Generator().manual_seed(1)
    items = [data.make_sample(trainer.engine, data.synth_note_clip(i, 4.0 + 8.0 * torch.rand((), generator=rng_len).item()), timestep)
             for i in range(synthetic)]
    lengths = [int(s['units'].shape[0]) for s in items]
    val_items = [data.make_sample(trainer.engine, data.synth_note_clip(10 ** 6 + i, 6.0), timestep) for i in range(val_clips)]

    def validate(step):
        trainer.sync_eval_engine()
        res = trainer.validation_step(data.collater(val_items, cfg))
        acc = float(res['midi_acc_correct']) / max(float(res['midi_acc_total']), 1.0)
        print(f'validation @ {step}: ' + ', '.join(f'{k}={float(v):.5f}' for k, v in res.items() if k.endswith('loss')) + f', midi_acc={acc:.4f}')

    total = max_updates if max_updates is not None else cfg.get('max_updates', 100000)
    # train.py:98-108 of the reference: continue from the newest checkpoint of the experiment directory, if any
    existing = sorted(work.glob('model_ckpt_steps_*.ckpt'), key=lambda p: int(p.stem.rsplit('_', 1)[1]))
    if existing:
        trainer.load_checkpoint(torch.load(existing[-1], map_location='cpu'))
        if rank == 0:
            print(f'resumed from {existing[-1].name} at step {trainer.global_step}')
    keep, interval = cfg.get('num_ckpt_keep', 5), cfg.get('val_check_interval', 1000)
    saved, epoch = list(existing), trainer.global_step // max(1, len(data.batches(lengths, cfg.get('max_batch_frames', 80000), cfg.get('max_batch_size', 8), rank, world)))
    while trainer.global_step < total:
        plan = data.batches(lengths, cfg.get('max_batch_frames', 80000), cfg.get('max_batch_size', 8), rank, world, seed=epoch)
        epoch += 1
        for idx in plan:
            out = trainer.training_step(data.collater([items[i] for i in idx], cfg))
            step = trainer.global_step
            if rank == 0 and (step % log_interval == 0 or step == total):
                print(f'step {step}: ' + ', '.join(f'{k}={float(v):.5f}' for k, v in out.items() if k.endswith('loss')) +
                      f', lr={out["lr"]:.3e}, loss_scale={trainer.loss_scale:g}')
            if rank == 0 and (step % interval == 0 or step == total) and not out['skipped']:
                path = work / f'model_ckpt_steps_{step}.ckpt'
                torch.save(trainer.checkpoint(), path)
                saved.append(path)
                if val_items:
                    validate(step)
                while len(saved) > keep:
                    saved.pop(0).unlink(missing_ok=True)
            if step >= total:
                break
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    train()
