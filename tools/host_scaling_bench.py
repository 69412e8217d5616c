#!/usr/bin/env python
"""Host-side scaling evidence for BASELINE configs[3] (batch_infer.py over 10 000 x 30 s clips on 8 GPUs) WITHOUT an 8-GPU node.

The sharded job has no cross-GPU dependence, so what can keep 8 GPUs from scaling is the ONE host they share: WAV reads (8.5 GB/s of
int16 PCM at 8 x 400 rows/s), the pinned staging copies, each rank's Python main thread (silence decisions, note collection, job
hand-off) and the word-alignment worker pools.  This tool runs the REAL ``batch_infer.process_rows`` in N rank processes over a
dataset of DISTINCT files, with only the device stage replaced by a model of it:

* ``SimulatedInference.infer_files`` does the host half of ``MIDIExtractionInference.infer_files`` for real - the staging memcpy of every
  file into a per-slot buffer, ``Slicer.spans_from_rms`` on an RMS curve per file - and then WAITS until a simulated device, busy
  ``frames x (ms per frame measured on the MI355X)`` per batch and pipelined one batch ahead like the real one, would have delivered;
* it returns canned per-chunk note arrays recorded from a real run of the trained checkpoint on the same base clips, so the
  word-alignment work per row is the real one.

    python tools/host_scaling_bench.py --dir /tmp/some_amd_bench/hs --files 10000 --ranks 8 [--cold]

Prints a table per rank (rows/s, WAV-reader wait, simulated-device time, alignment hand-off / drain, CPU seconds) and one JSON line."""
import argparse
import csv
import json
import os
import pathlib
import pickle
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

N_BASE = 8


def _write_files(args):
    root, base_path, lo, hi = args
    from scipy.io import wavfile
    base = np.load(base_path)
    for i in range(lo, hi):
        rng = np.random.default_rng(10_000 + i)
        pcm = base[i % N_BASE]
        gain = 0.6 + 0.4 * rng.random()
        shift = int(rng.integers(0, 44100)) * 2 // 2
        out = np.roll((pcm.astype(np.float32) * gain).astype(np.int16), shift)          # distinct bytes in every file
        path = root / 'wavs' / f'clip_{i:05d}.wav'
        wavfile.write(str(path), 44100, out)
        fd = os.open(path, os.O_RDONLY)
        try:
            os.fsync(fd)
        finally:
            os.close(fd)
    return hi - lo


def build_dataset(root: pathlib.Path, files: int, seconds: float):
    from concurrent.futures import ProcessPoolExecutor
    from some_amd import synth
    (root / 'wavs').mkdir(parents=True, exist_ok=True)
    base = np.stack([np.clip(np.round(synth.synth_clip(700 + i, seconds, silence_every=8.0).astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
                     for i in range(N_BASE)])
    np.save(root / 'base.npy', base)
    per = 50
    jobs = [(root, root / 'base.npy', lo, min(files, lo + per)) for lo in range(0, files, per)]
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
        list(ex.map(_write_files, jobs))
    n_ph = 60
    with open(root / 'transcriptions.csv', 'w', encoding='utf8', newline='') as f:
        w = csv.DictWriter(f, fieldnames=['name', 'ph_seq', 'ph_dur', 'ph_num'])
        w.writeheader()
        for i in range(files):
            w.writerow({'name': f'clip_{i:05d}', 'ph_seq': ' '.join(['a'] * n_ph), 'ph_dur': ' '.join([f'{seconds / n_ph:.6f}'] * n_ph),
                        'ph_num': ' '.join(['2'] * (n_ph // 2))})


def record_canned(root: pathlib.Path, model_dir: pathlib.Path):
    """A real run of the trained checkpoint over the base clips: per-file chunk results, RMS curves, measured device ms per frame."""
    import torch
    import yaml
    from some_amd.inference.me_infer import MIDIExtractionInference
    from some_amd.utils.slicer2 import Slicer
    cfg = yaml.safe_load(open(model_dir / 'config.yaml'))
    ins = MIDIExtractionInference(cfg, model_dir / 'model.ckpt', device='cuda')
    slicer = Slicer(sr=cfg['audio_sample_rate'], max_sil_kept=1000)
    base = np.load(root / 'base.npy')
    clips = [np.ascontiguousarray(b) for b in base]
    per_file = ins.infer_files(clips, slicer)
    curves = []
    for c in clips:
        rms, _ = ins.engine.slicer_rms(torch.from_numpy(c).cuda(), np.asarray([len(c)]), slicer.win_size, slicer.hop_size)
        curves.append(rms.cpu().numpy())
    # device time per frame: 4 packed batches of 32 files through the whole device stage (upload, RMS, cut, log-mel, forward, decode)
    batch = [clips[i % N_BASE] for i in range(32)]
    ins.max_batch_frames = 131072
    ins.infer_files(batch, slicer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        ins.infer_files(batch, slicer)
    torch.cuda.synchronize()
    ms_per_frame = 1e3 * (time.perf_counter() - t0) / 4 / sum(1 + len(c) // cfg['hop_size'] for c in batch)
    canned = {'segments': [[(off, {k: np.asarray(v) for k, v in seg.items()}) for off, seg in segs] for segs in per_file], 'curves': curves,
              'ms_per_frame': ms_per_frame, 'config': cfg}
    with open(root / 'canned.pkl', 'wb') as f:
        pickle.dump(canned, f)
    return canned


class SimulatedInference:
    """The host half of MIDIExtractionInference.infer_files for real, the device as a timeline (see the module docstring)."""

    def __init__(self, canned, hop: int):
        self.canned, self.hop = canned, hop
        self.max_batch_frames = 131072
        self.ms_per_frame = canned['ms_per_frame']
        self._slots = {}
        self.device_free_at = 0.0
        self.sim_device_busy_s = 0.0
        self.host_stage_s = 0.0

    def _stage(self, clips, slicer, slot):
        t0 = time.perf_counter()
        lens = [int(c.shape[0]) for c in clips]
        total = sum(lens)
        buf = self._slots.get(slot)
        if buf is None or buf.shape[0] < total:
            buf = self._slots[slot] = np.empty(max(total, 1), dtype=np.int16)
        pos = 0
        for c, n in zip(clips, lens):
            buf[pos:pos + n] = c                                       # the pinned-buffer copy of _stage_files
            pos += n
        spans = []
        for c, n in zip(clips, lens):
            base = int(c[:64].astype(np.int64).sum()) % N_BASE          # any curve of the right length: the decision loop is what costs
            spans.append(slicer.spans_from_rms(self.canned['curves'][base], n) if slicer.needs_rms(n) else [(0, n)])
        self.host_stage_s += time.perf_counter() - t0
        return lens, spans

    def infer_files(self, clips, slicer):
        groups, group, frames = [], [], 0
        for i, c in enumerate(clips):
            t = 1 + int(c.shape[0]) // self.hop
            if group and frames + t > self.max_batch_frames:
                groups.append(group)
                group, frames = [], 0
            group.append(i)
            frames += t
        if group:
            groups.append(group)
        results = [None] * len(clips)
        sr = slicer.sr
        pending = None

        def finish(p):
            ready_at, idx = p
            wait = ready_at - time.perf_counter()
            if wait > 0:
                time.sleep(wait)                                         # the batch's one host synchronisation
            for i in idx:
                segs = self.canned['segments'][i % N_BASE]
                results[i] = [(off, {k: v.copy() for k, v in seg.items()}) for off, seg in segs]     # _collect's per-chunk array copies

        staged = self._stage([clips[i] for i in groups[0]], slicer, 0) if groups else None
        for g, idx in enumerate(groups):
            lens, _ = staged
            busy = 1e-3 * self.ms_per_frame * sum(1 + n // self.hop for n in lens)
            self.device_free_at = max(self.device_free_at, time.perf_counter()) + busy
            self.sim_device_busy_s += busy
            ready_at = self.device_free_at
            if g + 1 < len(groups):
                staged = self._stage([clips[i] for i in groups[g + 1]], slicer, (g + 1) & 1)
            if pending is not None:
                finish(pending)
            pending = (ready_at, idx)
        if pending is not None:
            finish(pending)
        return results


def worker(root: pathlib.Path, rank: int, world: int, start_at: float, opts):
    import threading
    import batch_infer as bi
    from some_amd import sharding
    if opts.bind:                      # what batch_infer.py does under torch.distributed.run (sharding.init_distributed)
        sharding.bind_rank_to_cores(rank, world)
    lat = {'n': 0, 's': 0.0}
    lock = threading.Lock()
    real_load = bi.load_pcm

    def timed_load(path, rate, *rest):
        t = time.perf_counter()
        out = real_load(path, rate, *rest)
        dt = time.perf_counter() - t
        with lock:
            lat['n'] += 1
            lat['s'] += dt
        return out
    bi.load_pcm = timed_load
    with open(root / 'canned.pkl', 'rb') as f:
        canned = pickle.load(f)
    cfg = canned['config']
    rows = list(csv.DictReader(open(root / 'transcriptions.csv', encoding='utf8'))) * max(1, opts.repeat)     # --repeat: every file k times per job
    sizes = [(root / 'wavs' / f"{r['name']}.wav").stat().st_size for r in rows]
    mine = sorted(sharding.partition(sizes, rank, world))
    io_threads, align_workers = sharding.host_workers(world)
    io_threads = opts.io_threads or io_threads
    align_workers = opts.align_workers if opts.align_workers is not None else align_workers
    kw = {}
    if opts.prefetch:
        kw['prefetch'] = opts.prefetch
    if opts.flush_batches:
        kw['flush_batches'] = opts.flush_batches
    ins = SimulatedInference(canned, cfg['hop_size'])
    time.sleep(max(0.0, start_at - time.time()))                        # all ranks start together
    c0, t0 = os.times(), time.perf_counter()
    done = bi.process_rows(rows, mine, root, ins, cfg, False, io_threads=io_threads, align_workers=align_workers, **kw)
    wall = time.perf_counter() - t0
    c1 = os.times()
    print(json.dumps({'rank': rank, 'rows': len(done), 'wall_s': round(wall, 3), 'rows_per_s': round(len(done) / wall, 1),
                      'sim_device_busy_s': round(ins.sim_device_busy_s, 3), 'host_stage_in_infer_files_s': round(ins.host_stage_s, 3),
                      'stages': {k: round(float(v), 3) for k, v in bi.LAST_STAGES.items()},
                      'cpu_s_main_process': round((c1.user - c0.user) + (c1.system - c0.system), 2), 'cpu_s_main_system': round(c1.system - c0.system, 2),
                      'load_pcm_ms_mean': round(1e3 * lat['s'] / max(lat['n'], 1), 2),
                      'cpu_s_children': round((c1.children_user - c0.children_user) + (c1.children_system - c0.children_system), 2),
                      'io_threads': io_threads, 'align_workers': align_workers}))


def drop_page_cache(root: pathlib.Path):
    """Evict the dataset's pages: drop_caches when permitted, else POSIX_FADV_DONTNEED per file (clean pages, fsync-ed at build time)."""
    try:
        subprocess.run(['sync'], check=False)
        with open('/proc/sys/vm/drop_caches', 'w') as f:
            f.write('3\n')
        return 'drop_caches'
    except OSError:
        for p in (root / 'wavs').iterdir():
            fd = os.open(p, os.O_RDONLY)
            try:
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            finally:
                os.close(fd)
        return 'posix_fadvise(DONTNEED)'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', required=True)
    ap.add_argument('--files', type=int, default=10000)
    ap.add_argument('--seconds', type=float, default=30.0)
    ap.add_argument('--ranks', type=int, default=8)
    ap.add_argument('--cold', action='store_true', help='evict the dataset from the page cache before the run')
    ap.add_argument('--model-dir', default=None, help='directory with model.ckpt + config.yaml (default: trains one through tools/batch_infer_bench.py)')
    ap.add_argument('--io-threads', type=int, default=0, help='WAV reader threads per rank (default: sharding.host_workers)')
    ap.add_argument('--align-workers', type=int, default=None)
    ap.add_argument('--prefetch', type=int, default=0, help='decoded files in flight per rank (default: process_rows default)')
    ap.add_argument('--flush-batches', type=int, default=0)
    ap.add_argument('--repeat', type=int, default=1, help='process the dataset this many times in one job (longer steady state, same distinct files)')
    ap.add_argument('--bind', action='store_true', help='pin every rank to its own slice of the cores')
    ap.add_argument('--worker', type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument('--start-at', type=float, default=0.0, help=argparse.SUPPRESS)
    a = ap.parse_args()
    root = pathlib.Path(a.dir)
    if a.worker is not None:
        worker(root, a.worker, a.ranks, a.start_at, a)
        return
    t0 = time.perf_counter()
    if not (root / 'transcriptions.csv').exists():
        build_dataset(root, a.files, a.seconds)
        print(f'dataset: {a.files} distinct x {a.seconds:g} s int16 WAVs ({sum(p.stat().st_size for p in (root / "wavs").iterdir()) / 2 ** 30:.1f} GiB) '
              f'written in {time.perf_counter() - t0:.1f} s', file=sys.stderr)
    if not (root / 'canned.pkl').exists():
        model_dir = pathlib.Path(a.model_dir) if a.model_dir else root / 'seed' / 'model'
        if not (model_dir / 'model.ckpt').exists():
            r = subprocess.run([sys.executable, str(ROOT / 'tools' / 'batch_infer_bench.py'), '--clips', '16', '--dir', str(root / 'seed'), '--train_updates', '300',
                                '--json'], capture_output=True, text=True, cwd=ROOT)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        canned = record_canned(root, model_dir)
        print(f'canned: {sum(len(s) for s in canned["segments"])} chunks of {N_BASE} base files, '
              f'{sum(len(seg["note_midi"]) for s in canned["segments"] for _, seg in s) / N_BASE:.0f} notes per file, device {canned["ms_per_frame"] * 1e3:.3f} us per frame '
              f'= {1e3 / (canned["ms_per_frame"] * 2584):.0f} files/s per GPU', file=sys.stderr)
    with open(root / 'canned.pkl', 'rb') as f:
        canned = pickle.load(f)
    how = drop_page_cache(root) if a.cold else None
    start_at = time.time() + 4.0
    env = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get('PYTHONPATH', ''))
    t_run = time.perf_counter()
    extra = (['--io-threads', str(a.io_threads)] if a.io_threads else []) + (['--align-workers', str(a.align_workers)] if a.align_workers is not None else []) + \
        (['--prefetch', str(a.prefetch)] if a.prefetch else []) + (['--flush-batches', str(a.flush_batches)] if a.flush_batches else []) + (['--bind'] if a.bind else []) + \
        (['--repeat', str(a.repeat)] if a.repeat > 1 else [])
    procs = [subprocess.Popen([sys.executable, __file__, '--dir', str(root), '--ranks', str(a.ranks), '--worker', str(r), '--start-at', str(start_at)] + extra,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT) for r in range(a.ranks)]
    outs = [p.communicate() for p in procs]
    total_wall = time.perf_counter() - t_run - max(0.0, start_at - time.time())
    res = []
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
        res.append(json.loads([ln for ln in o.splitlines() if ln.startswith('{')][-1]))
    res.sort(key=lambda r: r['rank'])
    wall = max(r['wall_s'] for r in res)
    rows = sum(r['rows'] for r in res)
    target = 1e3 / (canned['ms_per_frame'] * (1 + int(a.seconds * 44100) // 512))
    print(f'# {a.ranks} rank processes, {rows} rows over {a.files} distinct files, page cache {"COLD (" + how + ")" if a.cold else "warm"}; simulated device: '
          f'{canned["ms_per_frame"] * 1e3:.3f} us/frame = {target:.0f} files/s per GPU; host: {os.cpu_count()} cores')
    print('rank  rows  wall_s  rows/s  dev_sim_s  wav_wait_s  stage_s  align_submit_s  align_drain_s  cpu_main_s (sys)  cpu_children_s  load_pcm_ms')
    for r in res:
        s = r['stages']
        print(f"{r['rank']:4d} {r['rows']:5d} {r['wall_s']:7.2f} {r['rows_per_s']:7.1f} {r['sim_device_busy_s']:10.2f} {s['wav_wait']:11.2f} "
              f"{r['host_stage_in_infer_files_s']:8.2f} {s['align_submit']:15.2f} {s['align_drain']:14.2f} {r['cpu_s_main_process']:9.2f} ({r['cpu_s_main_system']:5.2f}) {r['cpu_s_children']:13.2f} {r['load_pcm_ms_mean']:11.2f}")
    summary = {'ranks': a.ranks, 'files': rows, 'cold': bool(a.cold), 'wall_s': round(wall, 2), 'rows_per_s_total': round(rows / wall, 1),
               'rows_per_s_per_gpu_device_bound': round(target, 1), 'fraction_of_device_bound_rate': round(rows / wall / (a.ranks * target), 4),
               'read_GBps': round(rows * (44 + 2 * int(a.seconds * 44100)) / wall / 1e9, 2), 'host_cores': os.cpu_count(),
               'io_threads_per_rank': res[0]['io_threads'], 'align_workers_per_rank': res[0]['align_workers'], 'prefetch': a.prefetch or 640,
               'flush_batches': a.flush_batches or 8, 'bind': bool(a.bind), 'load_pcm_ms_mean': round(float(np.mean([r['load_pcm_ms_mean'] for r in res])), 2),
               'cpu_core_seconds_per_wall_second': round(sum(r['cpu_s_main_process'] + r['cpu_s_children'] for r in res) / wall, 1)}
    print(json.dumps(summary))


if __name__ == '__main__':
    main()
