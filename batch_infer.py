#!/usr/bin/env python
"""``python batch_infer.py --dataset DIR --model CKPT [--round_midi] [--csv OUT] [--overwrite]`` - the
reference's DiffSinger-dataset command (batch_infer.py:137-226) on the HIP hot path.

Single process: same behaviour as the reference, but the chunks of MANY rows are packed into each device batch.
Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node 8 batch_infer.py ...``; rows are
sharded across the ranks (largest first, round-robin), rank 0 packs the weights and broadcasts them once with
RCCL, every rank works on its own rows with no cross-GPU dependence, rank 0 gathers and writes the CSV."""
import pathlib
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from csv import DictReader, DictWriter
from typing import Dict, List

import click
import numpy as np

from infer import load_inference
from some_amd import batch_logic, sharding
from some_amd.utils.audio import PcmPool, load_pcm, load_wav
from utils.slicer2 import Slicer

CSV_FIELDS = ['name', 'ph_seq', 'ph_dur', 'ph_num', 'note_seq', 'note_dur']


def model_init(model_path):
    """batch_infer.py:21-34."""
    return load_inference(pathlib.Path(model_path))


def load_and_slice(audio_path, config):
    """batch_infer.py:50-53 (host side of one row): decode + silence slicing."""
    waveform, _ = load_wav(audio_path, sr=config['audio_sample_rate'], mono=True)
    return Slicer(sr=config['audio_sample_rate'], max_sil_kept=1000).slice(waveform)


def infer(wav, infer_ins, config):
    """batch_infer.py:49-81 for one file: absolute-time note list."""
    chunks = load_and_slice(pathlib.Path(wav), config)
    midis = infer_ins.infer([c['waveform'] for c in chunks])
    return batch_logic.notes_from_segments([c['offset'] for c in chunks], midis)


LAST_STAGES: Dict[str, float] = {}        # host stage timers of the last process_rows call (tools/batch_infer_bench.py, bench.py --e2e)


def process_rows(rows: List[dict], indices: List[int], data_path: pathlib.Path, infer_ins, config,
                 round_midi: bool, max_batch_frames: int = 131072, io_threads: int = 8, prefetch: int = 640,
                 flush_batches: int = 8, align_workers: int = 8) -> Dict[int, tuple]:
    """Rows ``indices`` of the CSV -> {row index: (note_seq, note_dur)}.  WAV files are read by a thread pool with a
    bounded read-ahead; up to ``flush_batches`` device batches of rows go through ``infer_files`` at a time (upload +
    RMS of batch k + 1 overlap the forward of batch k); chunks of consecutive rows share packed device batches.
    ``prefetch`` (decoded files in flight, 2.6 MB each) exceeds the ~400 thirty-second files of a flush group, so the readers
    fill the NEXT group while this one is on the device (the window is only refilled between flushes)"""
    import os
    import time
    hop = config['hop_size']
    infer_ins.max_batch_frames = max_batch_frames
    out: Dict[int, tuple] = {}
    stage_s = {'wav_wait': 0.0, 'device': 0.0, 'align_submit': 0.0, 'align_drain': 0.0}     # SOME_AMD_PROFILE_HOST=1 prints them
    jobs = []
    for i in indices:
        audio_path = data_path / 'wavs' / f"{rows[i]['name']}.wav"
        if not audio_path.exists():
            print(f'WARNING: audio file does not exist: \'{audio_path}\'')          # batch_infer.py:166-168
            continue
        jobs.append((i, audio_path))

    slicer = Slicer(sr=config['audio_sample_rate'], max_sil_kept=1000)
    # the word alignment (batch_infer.py:172-219) is pure-Python float / string work, ~2.5 ms per row and more with many
    # notes: big jobs hand it to worker subprocesses (numpy + batch_logic only) so it overlaps the GPU
    align_pool = None
    if align_workers > 0 and len(jobs) >= 64:
        from some_amd.align_worker import AlignPool
        align_pool = AlignPool(align_workers)

    device_ingest = hasattr(infer_ins, 'infer_files')      # any other BaseInference: host Slicer + infer(), as the reference

    def flush(group):
        # Slicer.slice + infer of batch_infer.py:52-57 for many rows at once: files go up as stored (int16 PCM), the
        # RMS curve and the chunk cut run on the device, the silence decisions on the host
        t_dev = time.perf_counter()
        if device_ingest:
            per_file = infer_ins.infer_files([pcm for _, pcm in group], slicer)
        else:
            per_file = []
            for _, pcm in group:
                wave = pcm if pcm.dtype == np.float32 else pcm.astype(np.float32) / np.float32(32768.0)
                chunks = slicer.slice(wave)
                per_file.append(list(zip([c['offset'] for c in chunks], infer_ins.infer([c['waveform'] for c in chunks]))))
        stage_s['device'] += time.perf_counter() - t_dev
        for _, pcm in group:               # staged and consumed: the sample buffers go back to the readers
            pcm_pool.give(pcm)
        if align_pool is None:
            post([i for i, _ in group], per_file)
        else:
            # hand-off to the alignment workers (pickling ~0.2 ms per row) on a helper thread: the main thread goes straight on to the next
            # group, whose device wait releases the GIL for it (measured: 0.7 s of a 10.3 s rank otherwise spent with the device idle)
            post_queue.put(([i for i, _ in group], per_file))

    def post(indices_, per_file):
        t_al = time.perf_counter()
        for i, segments in zip(indices_, per_file):
            job = ([off for off, _ in segments], [seg for _, seg in segments], rows[i]['ph_dur'], rows[i]['ph_num'], round_midi)
            if align_pool is None:
                out[i] = batch_logic.align_job(*job)
            else:
                align_pool.submit(i, job)
        stage_s['align_submit'] += time.perf_counter() - t_al

    post_queue, post_thread, post_error = None, None, []
    if align_pool is not None:
        import queue
        import threading
        post_queue = queue.Queue()

        def post_loop():
            while True:
                item = post_queue.get()
                if item is None:
                    return
                try:
                    post(*item)
                except BaseException as e:  # noqa: BLE001  (re-raised on the main thread)
                    post_error.append(e)
        post_thread = threading.Thread(target=post_loop, name='some-align-submit', daemon=True)
        post_thread.start()

    rate = config['audio_sample_rate']
    pcm_pool = PcmPool()
    with ThreadPoolExecutor(max_workers=io_threads) as pool:
        window = deque()                                  # bounded read-ahead: at most `prefetch` decoded files in flight
        it = iter(jobs)

        def refill():
            while len(window) < prefetch:
                job = next(it, None)
                if job is None:
                    return
                window.append((job[0], pool.submit(load_pcm, job[1], rate, pcm_pool)))

        refill()
        group, frames, ramp = [], 0, 1
        while window:
            i, fut = window.popleft()
            refill()
            t_w = time.perf_counter()
            pcm, _ = fut.result()
            stage_s['wav_wait'] += time.perf_counter() - t_w
            t = 1 + pcm.shape[-1] // hop
            if group and frames + t > ramp * max_batch_frames:
                flush(group)
                group, frames = [], 0
                ramp = min(flush_batches, 2 * ramp)        # 1, 2, 4, ... device batches per flush: the first one starts after ~50 files
            group.append((i, pcm))
            frames += t
        if group:
            flush(group)
    if align_pool is not None:
        t_d = time.perf_counter()
        post_queue.put(None)
        post_thread.join()
        if post_error:
            align_pool.close()
            raise post_error[0]
        out.update(align_pool.close())
        stage_s['align_drain'] = time.perf_counter() - t_d
    LAST_STAGES.clear()
    LAST_STAGES.update(stage_s, files=len(jobs))
    if os.environ.get('SOME_AMD_PROFILE_HOST'):
        print('host stages [s]: ' + ', '.join(f'{k} {v:.2f}' for k, v in stage_s.items()) + f' ({len(jobs)} files)')
    return out


@click.command(help='Batch inference on existing DiffSinger dataset.')
@click.option(
    '--dataset', required=True, metavar='RAW_DATA_DIR',
    help='Path to the dataset directory. Equivalent to \'raw_data_dir\' in DiffSinger configuration files.'
)
@click.option('--model', required=True, metavar='CKPT_PATH', help='Path to the model checkpoint (*.ckpt)')
@click.option('--round_midi', is_flag=True, help='Round MIDI values to integers')
@click.option(
    '--csv', required=False, metavar='CSV_PATH',
    help='Path to the output transcriptions.csv file (default to the same file in the dataset)'
)
@click.option('--overwrite', is_flag=True, help='Overwrite the existing transcriptions.csv file')
def batch_infer(dataset, model, round_midi, csv, overwrite):
    data_path = pathlib.Path(dataset)
    model_path = pathlib.Path(model)
    csv_path = pathlib.Path(csv) if csv is not None else data_path / 'transcriptions.csv'
    if csv_path.exists() and not overwrite:
        raise FileExistsError(f'The CSV path \'{csv_path}\' already exists. Please re-try with --overwrite option.')
    dist = sharding.init_distributed()
    rank, _, world = sharding.dist_env()
    infer_ins, config = model_init(model_path)

    with open(data_path / 'transcriptions.csv', 'r', encoding='utf8', newline='') as f:
        csv_data: List[dict] = list(DictReader(f))

    sizes = []
    for row in csv_data:
        p = data_path / 'wavs' / f"{row['name']}.wav"
        sizes.append(p.stat().st_size if p.exists() else 0)
    mine = sorted(sharding.partition(sizes, rank, world))
    io_threads, align_workers = sharding.host_workers(world)
    done = process_rows(csv_data, mine, data_path, infer_ins, config, round_midi, io_threads=io_threads, align_workers=align_workers)
    merged = sharding.gather_to_rank0(sorted(done.items()))
    if rank == 0:
        for i, (seq, dur) in merged:
            csv_data[i]['note_seq'], csv_data[i]['note_dur'] = seq, dur
        with open(csv_path, 'w', encoding='utf8', newline='') as f:
            writer = DictWriter(f, fieldnames=CSV_FIELDS)
            writer.writeheader()
            writer.writerows(csv_data)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    batch_infer()
