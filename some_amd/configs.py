"""Built-in copies of the hot-path keys of the reference's YAML configs.

The reference flattens ``configs/base.yaml`` + a model YAML into a ``config.yaml`` that is saved beside
the checkpoint (reference ``train.py:42-43``) and read back by ``infer.py:20-22``.  Only the keys the
inference path reads are kept here (list in SURVEY.md section 5, "Config / flags"):

* audio front end: ``configs/base.yaml:11-15``
* model dims:      ``configs/midi_conformer.yaml:22-33``, ``configs/quant_two_head_model.yaml:8-20``,
                   ``configs/two_head_model.yaml:24-35``
* decode:          ``configs/midi_conformer.yaml:16-19``, ``configs/base.yaml:23-24``

They are used by ``bench.py``, the tests and ``some_amd.synth`` to build synthetic checkpoints; a real
deployment reads the ``config.yaml`` written by the reference's trainer instead.
"""
import copy

_BASE = {
    'hop_size': 512,
    'win_size': 2048,
    'audio_sample_rate': 44100,
    'fmin': 40,
    'fmax': 8000,
    'units_encoder': 'mel',
    'pe': 'rmvpe',
    'midi_min': 0,
    'midi_max': 127,
    'units_dim': 80,
    'midi_num_bins': 128,
    'seed': 114514,
    'pl_trainer_precision': '32-true',      # configs/base.yaml:74
    # data / schedule keys of the trainer (configs/base.yaml:4-9, 36, 50-66)
    'binary_data_dir': None,
    'train_set_name': 'train',
    'valid_set_name': 'valid',
    'sort_by_len': True,
    'accumulate_grad_batches': 1,
    'sampler_frame_count_grid': 6,
    'max_batch_size': 8,
    'max_batch_frames': 80000,
    'max_val_batch_size': 1,
    'max_val_batch_frames': 10000,
    'val_check_interval': 1000,
    'num_ckpt_keep': 5,
    'max_updates': 100000,
    'ds_workers': 4,                        # configs/base.yaml:52-53 (training/base_task.py:374-380: DataLoader workers / prefetch)
    'dataloader_prefetch_factor': 2,
}


def _extractor_args(lay):
    return {
        'lay': lay,
        'dim': 512,
        'use_lay_skip': True,
        'kernel_size': 31,
        'conv_drop': 0.1,
        'ffn_latent_drop': 0.1,
        'ffn_out_drop': 0.1,
        'attention_drop': 0.1,
        'attention_heads': 8,
        'attention_heads_dim': 64,
    }


_CONFIGS = {
    'midi_conformer': dict(
        _BASE,
        pl_trainer_precision='bf16',            # configs/midi_conformer.yaml:35 (training only)
        model_cls='modules.model.Gmidi_conform.midi_conforms',
        task_cls='training.MIDIExtractionTask',
        midi_prob_deviation=1.0,
        rest_threshold=0.1,
        midi_extractor_args=_extractor_args(8),
    ),
    'two_head_model': dict(
        _BASE,
        model_cls='modules.model.Gmidi_conform.midi_conforms',
        task_cls='training.MIDIExtractionTask',
        midi_prob_deviation=1.0,
        rest_threshold=0.1,
        midi_extractor_args=_extractor_args(3),
        # training keys (configs/two_head_model.yaml:38-56)
        use_bound_loss=True,
        use_midi_loss=True,
        optimizer_args={'optimizer_cls': 'torch.optim.AdamW', 'lr': 0.0001, 'beta1': 0.9, 'beta2': 0.98, 'weight_decay': 0},
        lr_scheduler_args={'scheduler_cls': 'lr_scheduler.scheduler.WarmupLR', 'warmup_steps': 5000, 'min_lr': 0.00001},
        max_batch_size=8,
        max_batch_frames=80000,
        clip_grad_norm=1,                 # configs/base.yaml:49 (Lightning gradient_clip_val)
    ),
    'quant_two_head_model': dict(
        _BASE,
        midi_num_bins=129,
        model_cls='modules.model.Gmidi_conform.midi_conforms',
        task_cls='training.QuantizedMIDIExtractionTask',
        midi_extractor_args=_extractor_args(3),
        # training keys (configs/quant_two_head_model.yaml:24-45, configs/discrete.yaml:14-15)
        use_bound_loss=True,
        use_midi_loss=True,
        optimizer_args={'optimizer_cls': 'torch.optim.AdamW', 'lr': 0.0001, 'beta1': 0.9, 'beta2': 0.98, 'weight_decay': 0},
        lr_scheduler_args={'scheduler_cls': 'lr_scheduler.scheduler.WarmupLR', 'warmup_steps': 10000, 'min_lr': 0.00001},
        max_batch_size=8,
        max_batch_frames=80000,
        clip_grad_norm=1,
    ),
}


def get_config(name: str, **overrides) -> dict:
    """Return a deep copy of a built-in config; ``lay=...`` overrides ``midi_extractor_args.lay``."""
    cfg = copy.deepcopy(_CONFIGS[name])
    lay = overrides.pop('lay', None)
    if lay is not None:
        cfg['midi_extractor_args']['lay'] = int(lay)
    cfg.update(overrides)
    return cfg


def config_names():
    return sorted(_CONFIGS)


def builtin_yaml(path):
    """The built-in configs are addressable under the reference's file names: ``configs/base.yaml`` -> the shared keys,
    ``configs/<name>.yaml`` -> the flattened built-in config - so ``python train.py --config configs/two_head_model.yaml`` and a
    user's own file with ``base_config: configs/two_head_model.yaml`` work in a checkout that carries no YAML files.  Returns None
    for any other path (the caller raises: there is no silent fallback to some default model)."""
    import pathlib
    p = pathlib.Path(path)
    if p.suffix not in ('.yaml', '.yml') or p.parent.name != 'configs':
        return None
    if p.stem == 'base':
        return copy.deepcopy(_BASE)
    return get_config(p.stem) if p.stem in _CONFIGS else None
