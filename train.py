#!/usr/bin/env python
"""``python train.py --config CONFIG --exp_name EXP [--work_dir DIR]`` - the reference's training command
(train.py:27-110) on the HIP training path: MIDIExtractionTask losses, AdamW + WarmupLR, checkpoints in the
Lightning layout the inference classes load (``{'state_dict': {'model.<key>': ...}}`` + config.yaml beside it).

Data: ``binary_data_dir`` of the config (the reference's binarised dataset: ``train.data`` / ``valid.data`` HDF5
containers + ``.lengths``, read without h5py through some_amd/utils/hdf5_lite.py), batched by the reference's
DsBatchSampler / DsEvalBatchSampler plans; ``--synthetic N`` trains on N synthetic sung clips with known notes instead
(units from the HIP log-mel front end).  Multi-GPU: launch with
``python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...`` - one process per GPU, rank 0's
initial weights broadcast once, the flat gradient all-reduced in buckets overlapped with backward (RCCL), every rank its own
DsBatchSampler column."""
import copy
import os
import pathlib
import time

import click
import torch
import yaml

from some_amd import configs


# keys this command indexes directly (configs/base.yaml + the model YAML of the reference provide all of them)
_REQUIRED = ('task_cls', 'midi_extractor_args', 'midi_num_bins', 'hop_size', 'win_size', 'audio_sample_rate', 'units_dim', 'seed',
             'max_batch_frames', 'max_batch_size', 'sort_by_len', 'sampler_frame_count_grid', 'max_val_batch_frames', 'max_val_batch_size',
             'train_set_name', 'valid_set_name')


def _load_config(config: str) -> dict:
    """``--config``: a built-in name, or a YAML file read as the reference reads it (train.py:33-35 -> utils/config_utils.py:19-41:
    recursive ``base_config`` chains, deep-dict override, unknown keys kept).  No fallback to a default model: a file whose chain
    does not provide the trainer's keys is an error naming them."""
    from some_amd.utils.config_utils import read_full_config
    if config in configs.config_names():
        return configs.get_config(config)
    cfg = copy.deepcopy(read_full_config(pathlib.Path(config)))
    missing = [k for k in _REQUIRED if k not in cfg]
    if missing:
        raise click.UsageError(f"config '{config}' (with its base_config chain) does not define: {', '.join(missing)} - "
                               f"inherit from configs/base.yaml and a model config as the reference's configs do")
    return cfg


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
    from some_amd.training.loader import PrefetchLoader
    from some_amd.training.run_log import CheckpointKeeper, ScalarLog
    from some_amd.training.samplers import DsBatchSampler, DsEvalBatchSampler
    from some_amd.training.task import TRAINERS
    cfg = _load_config(config)
    if cfg['task_cls'] not in TRAINERS:           # training/__init__.py of the reference: MIDIExtractionTask, QuantizedMIDIExtractionTask
        raise click.UsageError(f"task_cls {cfg['task_cls']!r}: this command trains {sorted(TRAINERS)}")
    quantized = cfg['task_cls'] == 'training.QuantizedMIDIExtractionTask'
    work = (pathlib.Path(work_dir) if work_dir else pathlib.Path(__file__).parent / 'experiments') / exp_name
    assert not work.exists() or work.is_dir(), f'Path \'{work}\' is not a directory.'
    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    dev = local % max(1, torch.cuda.device_count())        # one GPU per rank; ranks sharing a GPU only in dry runs (SOME_AMD_DIST_BACKEND=gloo)
    torch.cuda.set_device(dev)
    # (under torch.distributed.run the group is initialised for ONE rank too, as bench.py / batch_infer.py do: a one-GPU box then runs the
    # RCCL communicator, the parameter broadcast and the barrier of the N > 1 path)
    if world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ):
        from some_amd import sharding
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        sharding.bind_rank_to_cores(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))       # loader threads stay next to this rank's GPU
        backend = os.environ.get('SOME_AMD_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            # RCCL's stream on HIGH priority: the HIP runtime keeps separate hardware queues per priority level, so the collectives of the
            # bucketed all-reduce cannot land on a queue one of the step's compute streams (two lanes + two weight-gradient side streams +
            # the loader's copy stream on 4 normal-priority queues) already occupies - modelled on one GPU, the all-reduce is 35 - 50 %
            # exposed that way and 90 - 130 % when it shares a queue (tools/ddp_overlap_bench.py)
            opts = torch.distributed.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            torch.distributed.init_process_group('nccl', pg_options=opts, device_id=torch.device('cuda', dev))
        else:
            torch.distributed.init_process_group(backend)
    if rank == 0:
        work.mkdir(parents=True, exist_ok=True)
        with open(work / 'config.yaml', 'w', encoding='utf8') as f:
            yaml.safe_dump(cfg, f)
    trainer = TRAINERS[cfg['task_cls']](cfg, device=f'cuda:{dev}', seed=cfg.get('seed', 114514))
    if synthetic > 0:
        rng_len = torch.Generator().manual_seed(1)
        seconds = [4.0 + 8.0 * torch.rand((), generator=rng_len).item() for _ in range(synthetic)]
        train_set = data.SyntheticNoteDataset(cfg, trainer.engine, range(synthetic), seconds, allow_aug=True, quantized=quantized)
        valid_set = data.SyntheticNoteDataset(cfg, trainer.engine, range(10 ** 6, 10 ** 6 + val_clips), [6.0] * val_clips, quantized=quantized) if val_clips else None
        max_val_batch_size = max(val_clips, 1)
    else:
        if not cfg.get('binary_data_dir'):
            raise click.UsageError('the config has no binary_data_dir: set it, or pass --synthetic N')
        # training/base_task.py:135-142
        ds_cls = data.QuantizedMIDIExtractionDataset if quantized else data.MIDIExtractionDataset           # task.dataset_cls
        train_set = ds_cls(cfg, cfg['binary_data_dir'], cfg['train_set_name'], allow_aug=True, device=trainer.ops.device)
        valid_set = ds_cls(cfg, cfg['binary_data_dir'], cfg['valid_set_name'], device=trainer.ops.device)
        max_val_batch_size = cfg['max_val_batch_size']
    accumulate = int(cfg.get('accumulate_grad_batches', 1))
    # training/base_task.py:360-395
    sampler = DsBatchSampler(train_set, max_batch_frames=cfg['max_batch_frames'], max_batch_size=cfg['max_batch_size'],
                             num_replicas=world, rank=rank, sort_by_similar_size=cfg['sort_by_len'],
                             required_batch_count_multiple=accumulate, frame_count_grid=cfg['sampler_frame_count_grid'],
                             shuffle_sample=True, shuffle_batch=False, seed=cfg['seed'])
    val_sampler = DsEvalBatchSampler(valid_set, max_batch_frames=cfg['max_val_batch_frames'], max_batch_size=max_val_batch_size,
                                     rank=rank, batch_by_size=False) if valid_set is not None and len(valid_set) else None

    def validate(step):
        trainer.sync_eval_engine()
        sums, n = {}, 0
        for idx in val_sampler:
            res = trainer.validation_step(valid_set.collater([valid_set[i] for i in idx]))
            n += 1
            for k, v in res.items():
                if k.endswith('loss') or k.startswith('midi_acc'):
                    sums[k] = sums.get(k, 0.0) + float(v)
        acc = sums['midi_acc_correct'] / max(sums['midi_acc_total'], 1.0)
        print(f'validation @ {step}: ' + ', '.join(f'{k}={v / n:.5f}' for k, v in sums.items() if k.endswith('loss')) + f', midi_acc={acc:.4f}')
        if scalar_log is not None:         # training/base_task.py:311-316
            val = {k: v / n for k, v in sums.items() if k.endswith('loss')}
            scalar_log.log_metrics({'validation/total_loss': sum(val.values()), **{f'validation/{k}': v for k, v in val.items()}}, step)
            scalar_log.log_metrics({'metrics/midi_acc': acc}, step)

    total = max_updates if max_updates is not None else cfg.get('max_updates', 100000)
    # train.py:98-108 of the reference: continue from the newest checkpoint of the experiment directory, if any
    existing = CheckpointKeeper.existing(work)
    if existing:
        # the learning rate is a pure function of (global_step, the CURRENT config's optimizer_args / lr_scheduler_args) - what the reference's
        # on_load_checkpoint re-simulation (training/base_task.py:412-456) arrives at: a changed lr / warmup in the config takes effect on resume
        trainer.load_checkpoint(torch.load(existing[-1], map_location='cpu'))
        if rank == 0:
            print(f'resumed from {existing[-1].name} at step {trainer.global_step}')
    interval = cfg.get('val_check_interval', 1000)
    # utils/training_utils.py:182-256 (DsModelCheckpoint): newest num_ckpt_keep + permanent checkpoints; train.py:83-87 (TensorBoardLogger)
    keeper = CheckpointKeeper(work, cfg.get('num_ckpt_keep', 5), cfg.get('permanent_ckpt_start', 0), cfg.get('permanent_ckpt_interval', 0))
    scalar_log = ScalarLog(work) if rank == 0 else None
    # training/base_task.py:374-380: DataLoader(num_workers=ds_workers, prefetch_factor=dataloader_prefetch_factor, pin_memory=True,
    # persistent_workers=True) - here worker threads collating into pinned buffers + uploads on a copy stream (training/loader.py)
    loader = PrefetchLoader(train_set, cfg, trainer.ops.device, workers=int(cfg.get('ds_workers', 4)),
                            prefetch_factor=int(cfg.get('dataloader_prefetch_factor', 2)))
    steps_per_epoch = max(1, -(-len(sampler) // accumulate))      # leftover micro-batches step the optimiser too (loop below): ceil
    epoch = trainer.global_step // steps_per_epoch
    skip = trainer.global_step % steps_per_epoch          # resumed inside an epoch: its first `skip` updates have been applied already
    t_train = time.perf_counter()

    def update(micro):
        # sync=False: no gradient-norm readback per update where no loss scaling is active (bf16 / exact-f32) - the host enqueues the next
        # step while this one's tail runs; the losses below are device tensors, read only on the steps that print or log them
        out = trainer.training_step(micro if len(micro) > 1 else micro[0], sync=False)
        step = trainer.global_step
        if rank == 0 and (step % log_interval == 0 or step == total):
            print(f'step {step}: ' + ', '.join(f'{k}={float(v):.5f}' for k, v in out.items() if k.endswith('loss')) +
                  f', lr={out["lr"]:.3e}, loss_scale={trainer.loss_scale:g}')
        if scalar_log is not None and step % int(cfg.get('log_interval', 100)) == 0 and not out['skipped']:       # training/base_task.py:254-260
            scalar_log.log_metrics({**{f'training/{k}': float(v) for k, v in out.items() if k.endswith('loss') and k != 'total_loss'},
                                    'training/batch_size': float(sum(int(m['units'].shape[0]) for m in micro)) / len(micro), 'training/lr': out['lr']}, step)
        if rank == 0 and (step % interval == 0 or step == total) and not out['skipped']:
            path = keeper.path_for(step)
            if keeper.wants(step):                       # num_ckpt_keep = 0: Lightning's save_top_k = 0 saves nothing (but permanent steps)
                torch.save(trainer.checkpoint(), path)
            if val_sampler is not None:
                validate(step)
            if path.exists():
                for line in keeper.saved(path):
                    print(line)

    while trainer.global_step < total:
        sampler.set_epoch(epoch)
        plan = list(sampler)[skip * accumulate:]
        if not plan:
            raise RuntimeError(f'epoch {epoch}: the batch sampler produced no batch for rank {rank} (dataset too small for this world size?)')
        epoch, skip = epoch + 1, 0
        micro = []
        for mb in loader.batches(plan):
            micro.append(mb)
            if len(micro) < accumulate:
                continue
            update(micro)
            micro = []
            if trainer.global_step >= total:
                break
        if micro and trainer.global_step < total:
            # fewer than accumulate_grad_batches micro-batches left at the end of the epoch: Lightning steps the optimiser on what has
            # accumulated (DsBatchSampler pads the batch count to a multiple, so this is for resumed / sliced plans)
            update(micro)
    trainer.flush()
    torch.cuda.synchronize()
    if rank == 0:
        wall = time.perf_counter() - t_train
        st = loader.stats
        print(f'loader: {st["batches"]} batches, waited {st["wait_s"]:.2f} s of {wall:.2f} s for data ({100.0 * st["wait_s"] / max(wall, 1e-9):.1f} %), '
              f'host collate {st["host_collate_s"]:.2f} s in {loader.workers} worker threads')
    loader.close()
    if scalar_log is not None:
        scalar_log.close()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    train()
