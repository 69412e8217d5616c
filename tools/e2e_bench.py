"""End-to-end timing of the drop-in inference class from HOST numpy waveforms (includes pinned staging, H2D over
PCIe, all kernels, D2H of the notes): the number DESIGN.md quotes next to the device-resident `bench.py` value.

    python tools/e2e_bench.py [--clips 96] [--seconds 30] [--lay 8]
"""
import argparse
import pathlib
import sys
import tempfile
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import inference  # noqa: E402
from some_amd import synth  # noqa: E402
from some_amd.configs import get_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=96)
    ap.add_argument('--seconds', type=float, default=30.0)
    ap.add_argument('--lay', type=int, default=8)
    args = ap.parse_args()
    cfg = get_config('midi_conformer', lay=args.lay)
    with tempfile.TemporaryDirectory() as d:
        ckpt = synth.save_checkpoint(cfg, pathlib.Path(d) / 'model.ckpt', seed=1)
        t0 = time.perf_counter()
        ins = inference.MIDIExtractionInference(config=cfg, model_path=ckpt)
        print(f'model load + pack + upload: {time.perf_counter() - t0:.2f} s')
        t0 = time.perf_counter()
        again = inference.MIDIExtractionInference(config=cfg, model_path=ckpt)
        print(f'second start (cached flat arena: {again.loaded_from_cache}): {time.perf_counter() - t0:.2f} s')
        del again
    base = [synth.synth_clip(i, args.seconds) for i in range(8)]
    waves = [base[i % 8] for i in range(args.clips)]
    ins.infer(waves[:32])                      # warm-up (allocations, pinned buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ins.infer(waves)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    notes = sum(len(r['note_midi']) for r in res)
    print(f'{args.clips} x {args.seconds:g} s clips from host memory: {dt * 1e3:.1f} ms  ->  '
          f'{args.clips * args.seconds / dt:.0f} audio-s/s end to end ({notes} notes)')

    # whole FILES (int16 PCM as stored, 0.6 s silences every 8 s so the slicer cuts): host Slicer + infer vs the
    # device ingest (upload int16, RMS + chunk cut on the GPU, silence state machine on the host)
    from some_amd.utils.slicer2 import Slicer
    slicer = Slicer(sr=44100, max_sil_kept=1000)
    fbase = [synth.synth_clip(100 + i, args.seconds, silence_every=8.0) for i in range(8)]
    pcm8 = [np.clip(np.round(w.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16) for w in fbase]
    files = [pcm8[i % 8] for i in range(args.clips)]

    def host_path():
        chunks = []
        for p in files:
            w = p.astype(np.float32) / np.float32(32768.0)                   # utils/audio.load_wav
            chunks.extend(c['waveform'] for c in slicer.slice(w))
        return ins.infer(chunks)

    for name, fn in (('host Slicer + infer', host_path), ('device ingest infer_files', lambda: ins.infer_files(files, slicer))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_seg = len(out) if name.startswith('host') else sum(len(f) for f in out)
        print(f'{args.clips} int16 files x {args.seconds:g} s, {name}: {dt * 1e3:.1f} ms  ->  '
              f'{args.clips * args.seconds / dt:.0f} audio-s/s ({n_seg} chunks)')


if __name__ == '__main__':
    main()
