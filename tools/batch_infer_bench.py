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


def build_dataset(root: pathlib.Path, clips: int, seconds: float, distinct: int = 0):
    """``clips`` rows; the first ``distinct`` (default: all) are files of their own, the rest hard links to them (BASELINE configs[3]:
    10 000 x 30 s = 26.5 GB as int16 - 1 250 distinct files keep the dataset at 3.3 GB while every row still opens, reads and
    decodes a file)."""
    (root / 'wavs').mkdir(parents=True, exist_ok=True)
    base = [synth.synth_clip(700 + i, seconds, silence_every=8.0) for i in range(8)]
    distinct = clips if distinct <= 0 else min(distinct, clips)
    rows = []
    for i in range(clips):
        path = root / 'wavs' / f'clip_{i:05d}.wav'
        if i < distinct:
            save_wav(path, base[i % 8], 44100)
        else:
            if path.exists():
                path.unlink()
            os.link(root / 'wavs' / f'clip_{i % distinct:05d}.wav', path)
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
    ap.add_argument('--distinct', type=int, default=0, help='number of distinct WAV files (the other rows are hard links); 0: all')
    ap.add_argument('--check', type=int, default=0, help='recompute this many sampled rows one by one (host Slicer + infer) and compare the CSV cells')
    ap.add_argument('--json', action='store_true', help='print one JSON object instead of the sentence')
    ap.add_argument('--limit', type=int, default=0, help='annotate only the first N rows of the dataset (a sub-dataset directory sharing the WAV files)')
    ap.add_argument('--out', default='out.csv', help='name of the CSV written into --dir')
    ap.add_argument('--compare', default=None, help='another CSV in --dir (e.g. the other arithmetic mode\'s): count rows / note boundaries that differ')
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
        build_dataset(root, args.clips, args.seconds, args.distinct)
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
            print([ln for ln in r.stdout.splitlines() if ln.startswith('validation')][-1:], file=sys.stderr)
            (root / 'model' / f'model_ckpt_steps_{args.train_updates}.ckpt').rename(root / 'model' / 'model.ckpt')
        else:
            synth.save_checkpoint(get_config('midi_conformer', lay=args.lay), root / 'model' / 'model.ckpt', seed=1)
        print(f'dataset: {args.clips} x {args.seconds:g} s int16 WAVs written in {time.perf_counter() - t0:.1f} s', file=sys.stderr)
    data_root, n_rows = root, args.clips
    if args.limit and args.limit < args.clips:
        # the first N rows as a dataset of their own: same WAV files (symlinked directory), truncated transcriptions.csv
        data_root, n_rows = root / f'first_{args.limit}', args.limit
        if rank == 0 and not (data_root / 'transcriptions.csv').exists():
            data_root.mkdir(exist_ok=True)
            if not (data_root / 'wavs').exists():
                os.symlink(root / 'wavs', data_root / 'wavs')
            with open(root / 'transcriptions.csv', encoding='utf8') as f:
                lines = f.readlines()
            (data_root / 'transcriptions.csv').write_text(''.join(lines[:1 + args.limit]), encoding='utf8')
    import batch_infer as bi
    import torch
    import utils.config_utils
    utils.config_utils.print_config = lambda *_a, **_k: None          # keep the timing output readable
    if world > 1:
        time.sleep(0 if rank == 0 else 2)
    cached = any((root / 'model').glob('*.arena'))
    t0 = time.perf_counter()
    bi.batch_infer.callback(dataset=str(data_root), model=str(root / 'model' / 'model.ckpt'), round_midi=False, csv=str(root / args.out), overwrite=True)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        rows = list(csv.DictReader(open(root / args.out, encoding='utf8')))
        filled = sum(1 for r in rows if r.get('note_seq'))
        res = {'workload': f'batch_infer.py --dataset DIR --model CKPT --csv: {n_rows} rows x {args.seconds:g} s int16 WAVs '
                           f'({args.distinct or args.clips} distinct files) + transcriptions.csv -> CSV, lay {args.lay}, {world} process(es), '
                           f'model load included ({"warm" if cached else "cold"} weight cache)',
               'rows': n_rows, 'rows_annotated': filled, 'wall_s': round(dt, 3), 'rows_per_s': round(n_rows / dt, 1),
               'audio_s_per_s': round(n_rows * args.seconds / dt, 1), 'host_stages_s_rank0': {k: round(float(v), 3) for k, v in bi.LAST_STAGES.items()}}
        if args.check > 0:
            # per-row recomputation through the reference's own granularity: Slicer.slice on the host + infer() per file
            import yaml
            import inference
            from some_amd.utils.audio import load_wav
            from some_amd import batch_logic
            from some_amd.utils.slicer2 import Slicer
            cfg = yaml.safe_load(open(root / 'model' / 'config.yaml'))
            ins = inference.task_inference_mapping  # noqa: F841  (registry import keeps the dotted class paths resolvable)
            from some_amd.inference.me_infer import MIDIExtractionInference
            one = MIDIExtractionInference(cfg, root / 'model' / 'model.ckpt', device='cuda')
            slicer = Slicer(sr=cfg['audio_sample_rate'], max_sil_kept=1000)
            rng = np.random.default_rng(0)
            picks = sorted(rng.choice(len(rows), size=min(args.check, len(rows)), replace=False).tolist())
            bad = notes = moved = 0
            for i in picks:
                wave, _ = load_wav(root / 'wavs' / f"{rows[i]['name']}.wav", cfg['audio_sample_rate'])
                chunks = slicer.slice(wave)
                midis = one.infer([c['waveform'] for c in chunks])
                seq, dur = batch_logic.align_job([c['offset'] for c in chunks], midis, rows[i]['ph_dur'], rows[i]['ph_num'], False)
                bad += (seq != rows[i]['note_seq']) or (dur != rows[i]['note_dur'])
                # note by note: boundaries in frames (a row string differs as soon as one duration's 6th decimal does)
                ta = np.round(np.cumsum([float(x) for x in dur.split()]) * 44100 / 512).astype(int)
                tb = np.round(np.cumsum([float(x) for x in rows[i]['note_dur'].split()]) * 44100 / 512).astype(int)
                notes += len(tb)
                moved += len(set(ta.tolist()) ^ set(tb.tolist()))
            res['rows_rechecked_one_by_one'] = len(picks)
            res['rows_rechecked_differing_as_strings'] = int(bad)
            res['rechecked_note_boundaries'] = int(notes)
            res['rechecked_note_boundaries_differing'] = int(moved)
            res['recheck_note'] = ('rows recomputed alone through host Slicer + infer(), the reference\'s own granularity (batch_infer.py:49-81): a '
                                   'clip\'s result does not depend on its batch (clip-local attention key tiles), so the CSV cells are expected to be '
                                   'identical STRINGS')
        if args.compare:
            # this run's CSV against another arithmetic mode's (same rows by name): rows differing as strings, note boundaries (frames) differing
            other = {r['name']: r for r in csv.DictReader(open(root / args.compare, encoding='utf8'))}
            n_cmp = bad = notes = moved = 0
            for r in rows:
                o = other.get(r['name'])
                if o is None or not r.get('note_dur') or not o.get('note_dur'):
                    continue
                n_cmp += 1
                bad += (r['note_seq'] != o['note_seq']) or (r['note_dur'] != o['note_dur'])
                ta = np.round(np.cumsum([float(x) for x in r['note_dur'].split()]) * 44100 / 512).astype(int)
                tb = np.round(np.cumsum([float(x) for x in o['note_dur'].split()]) * 44100 / 512).astype(int)
                notes += len(tb)
                moved += len(set(ta.tolist()) ^ set(tb.tolist()))
            res['compared_with'] = {'csv': args.compare, 'rows': n_cmp, 'rows_differing_as_strings': int(bad), 'note_boundaries': int(notes),
                                    'note_boundaries_differing': int(moved)}
        if args.json:
            import json
            print(json.dumps(res))
        else:
            print(f'batch_infer.py end to end ({world} process(es), model load + weight pack included): {n_rows} x {args.seconds:g} s in '
                  f'{dt:.2f} s -> {n_rows * args.seconds / dt:.0f} audio-s/s; {filled}/{len(rows)} rows annotated')
    if tmp is not None:
        tmp.cleanup()


if __name__ == '__main__':
    main()
