"""End-to-end timing of the ``batch_infer.py`` command (BASELINE config 4 shape at a reduced clip count): WAV files on
disk + transcriptions.csv -> decode -> device ingest (RMS slicer, chunk cut) -> log-mel -> conformer -> decode ->
word alignment -> CSV on disk.

    python tools/batch_infer_bench.py [--clips 256] [--seconds 30] [--lay 8]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/batch_infer_bench.py ...
"""
import argparse
import csv
import os
import pathlib
import sys
import tempfile
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from some_amd import synth  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.utils.audio import save_wav  # noqa: E402


def build_dataset(root: pathlib.Path, clips: int, seconds: float):
    (root / 'wavs').mkdir(parents=True, exist_ok=True)
    base = [synth.synth_clip(700 + i, seconds, silence_every=8.0) for i in range(8)]
    rows = []
    for i in range(clips):
        save_wav(root / 'wavs' / f'clip_{i:05d}.wav', base[i % 8], 44100)
        n_ph = 60
        rows.append({'name': f'clip_{i:05d}', 'ph_seq': ' '.join(['a'] * n_ph), 'ph_dur': ' '.join([f'{seconds / n_ph:.6f}'] * n_ph),
                     'ph_num': ' '.join(['2'] * (n_ph // 2))})
    with open(root / 'transcriptions.csv', 'w', encoding='utf8', newline='') as f:
        w = csv.DictWriter(f, fieldnames=['name', 'ph_seq', 'ph_dur', 'ph_num'])
        w.writeheader()
        w.writerows(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=256)
    ap.add_argument('--seconds', type=float, default=30.0)
    ap.add_argument('--lay', type=int, default=8)
    ap.add_argument('--dir', default=None, help='dataset directory to (re)use; default: a temporary directory')
    ap.add_argument('--train_updates', type=int, default=0,
                    help='train the checkpoint for this many updates on synthetic sung clips first (train.py): random weights emit ~1500 '
                         'notes per clip, a trained model ~60, and the per-row word alignment scales with the note count')
    args = ap.parse_args()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    tmp = None
    if args.dir is None:
        if world > 1:
            raise SystemExit('pass --dir (a path every rank sees) for multi-process runs')
        tmp = tempfile.TemporaryDirectory()
        root = pathlib.Path(tmp.name)
    else:
        root = pathlib.Path(args.dir)
    if rank == 0 and not (root / 'transcriptions.csv').exists():
        t0 = time.perf_counter()
        build_dataset(root, args.clips, args.seconds)
        if args.train_updates > 0:
            import subprocess
            import yaml
            cfg = get_config('midi_conformer', lay=args.lay)
            cfg['lr_scheduler_args'] = {'scheduler_cls': 'lr_scheduler.scheduler.WarmupLR', 'warmup_steps': 40, 'min_lr': 1e-5}
            cfg['optimizer_args'] = {'optimizer_cls': 'torch.optim.AdamW', 'lr': 3e-4, 'beta1': 0.9, 'beta2': 0.98, 'weight_decay': 0}
            cfg.update(use_bound_loss=True, use_midi_loss=True, max_batch_size=8, max_batch_frames=80000, clip_grad_norm=1,
                       val_check_interval=args.train_updates)
            (root / 'train_cfg').mkdir(exist_ok=True)
            with open(root / 'train_cfg' / 'midi_conformer.yaml', 'w') as f:
                yaml.safe_dump(cfg, f)
            repo = pathlib.Path(__file__).resolve().parents[1]
            r = subprocess.run([sys.executable, str(repo / 'train.py'), '--config', str(root / 'train_cfg' / 'midi_conformer.yaml'), '--exp_name', 'model',
                                '--work_dir', str(root), '--synthetic', '64', '--max_updates', str(args.train_updates), '--log_interval', '100'],
                               capture_output=True, text=True, cwd=repo)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            print([ln for ln in r.stdout.splitlines() if ln.startswith('validation')][-1:])
            (root / 'model' / f'model_ckpt_steps_{args.train_updates}.ckpt').rename(root / 'model' / 'model.ckpt')
        else:
            synth.save_checkpoint(get_config('midi_conformer', lay=args.lay), root / 'model' / 'model.ckpt', seed=1)
        print(f'dataset: {args.clips} x {args.seconds:g} s int16 WAVs written in {time.perf_counter() - t0:.1f} s')
    import batch_infer as bi
    import torch
    import utils.config_utils
    utils.config_utils.print_config = lambda *_a, **_k: None          # keep the timing output readable
    if world > 1:
        time.sleep(0 if rank == 0 else 2)
    t0 = time.perf_counter()
    bi.batch_infer.callback(dataset=str(root), model=str(root / 'model' / 'model.ckpt'), round_midi=False, csv=str(root / f'out.csv'), overwrite=True)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        rows = list(csv.DictReader(open(root / 'out.csv', encoding='utf8')))
        filled = sum(1 for r in rows if r.get('note_seq'))
        print(f'batch_infer.py end to end ({world} process(es), model load + weight pack included): {args.clips} x {args.seconds:g} s in '
              f'{dt:.2f} s -> {args.clips * args.seconds / dt:.0f} audio-s/s; {filled}/{len(rows)} rows annotated')
    if tmp is not None:
        tmp.cleanup()


if __name__ == '__main__':
    main()
