#!/usr/bin/env python
"""How fast can P processes x T threads read a directory of int16 WAV files out of the page cache, by read path?

    python tools/wav_read_probe.py --dir DATASET/wavs --procs 8 --threads 8 --mode scipy|readinto|mmap|os_read

scipy: scipy.io.wavfile.read (what load_pcm does: a fresh bytes object per file); readinto: file.readinto a reusable per-thread buffer;
os_read: os.preadv into the reusable buffer; mmap: np.memmap + copy into the reusable buffer.  Prints aggregate GB/s."""
import argparse
import json
import os
import pathlib
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def read_one(path, mode, buf):
    if mode == 'scipy':
        from scipy.io import wavfile
        _, data = wavfile.read(str(path))
        return data.nbytes
    size = os.path.getsize(path)
    if mode == 'readinto':
        with open(path, 'rb', buffering=0) as f:
            return f.readinto(memoryview(buf)[:size])
    if mode == 'os_read':
        fd = os.open(path, os.O_RDONLY)
        try:
            return os.preadv(fd, [memoryview(buf)[:size]], 0)
        finally:
            os.close(fd)
    if mode == 'mmap':
        m = np.memmap(path, dtype=np.uint8, mode='r')
        buf[:size] = m
        del m
        return size
    raise ValueError(mode)


def worker(a):
    files = sorted(pathlib.Path(a.dir).iterdir())[a.worker::a.procs]
    if a.limit:
        files = files[:a.limit]
    tls = {}

    def job(p):
        import threading
        k = threading.get_ident()
        if k not in tls:
            tls[k] = np.empty(4 << 20, dtype=np.uint8)
        return read_one(p, a.mode, tls[k])
    time.sleep(max(0.0, a.start_at - time.time()))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(a.threads) as ex:
        n = sum(ex.map(job, files))
    print(json.dumps({'bytes': n, 'wall': time.perf_counter() - t0}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', required=True)
    ap.add_argument('--procs', type=int, default=8)
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--mode', default='scipy')
    ap.add_argument('--limit', type=int, default=0)
    ap.add_argument('--worker', type=int, default=None)
    ap.add_argument('--start-at', type=float, default=0.0)
    a = ap.parse_args()
    if a.worker is not None:
        return worker(a)
    start = time.time() + 2.0
    procs = [subprocess.Popen([sys.executable, __file__, '--dir', a.dir, '--procs', str(a.procs), '--threads', str(a.threads), '--mode', a.mode,
                               '--limit', str(a.limit), '--worker', str(w), '--start-at', str(start)], stdout=subprocess.PIPE, text=True) for w in range(a.procs)]
    res = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    wall = max(r['wall'] for r in res)
    print(f'{a.mode:9s} procs {a.procs:3d} x threads {a.threads:2d}: {sum(r["bytes"] for r in res) / wall / 1e9:6.2f} GB/s aggregate ({wall:.2f} s)')


if __name__ == '__main__':
    main()
