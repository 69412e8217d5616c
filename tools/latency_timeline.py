#!/usr/bin/env python
"""Where the single-clip latency goes: one 30 s clip, B = 1, log-mel -> forward -> decode.

    run:      rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/latency_timeline.py run [--steps 12] [--graph]
    analyse:  python tools/latency_timeline.py analyse DIR > profiles/r06_latency_timeline.txt

``run`` prints the host-side wall time per step (synchronised) and marks nothing on the device - the analysis finds the steps in the
kernel trace by the idle gap the runner leaves between them.  ``analyse`` reports, for the steady-state steps: the span from the first kernel's
start to the last kernel's end, the time at least one kernel is running (union of the intervals), the time TWO kernels overlap (the midi /
bound model streams run on two HIP streams), the idle time inside the span (no kernel resident: launch / dependency gaps), the serial sum
per kernel family, and the largest idle gaps with the kernels either side."""
import argparse
import csv
import glob
import pathlib
import re
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def run(args):
    import numpy as np
    import torch
    from some_amd import _lib, synth
    from some_amd.configs import get_config
    from some_amd.engine import ClipBatch, Engine
    cfg = get_config(args.config)
    quant = cfg['task_cls'].endswith('QuantizedMIDIExtractionTask')
    eng = Engine(cfg, device='cuda')
    eng.attach_arena(eng.pack_state_dict(synth.synth_state_dict(cfg, seed=cfg.get('seed', 114514))).cuda())
    clip = synth.synth_clip(0, args.seconds, cfg['audio_sample_rate'])
    one = ClipBatch.from_sample_counts([len(clip)], eng.hop, 'cuda')
    audio = torch.from_numpy(clip).cuda()
    head = _lib.HEAD_SOFTMAX if quant else _lib.HEAD_SIGMOID

    def step():
        u = eng.logmel(audio, one)
        p, b = eng.forward(u, one, head_mode=head)
        return eng.decode(p, b, one, quantized=quant)
    runner = step
    if args.graph:
        runner = eng.graph_runner(audio, one, head_mode=head, quantized=quant)
    lat = []
    for _ in range(args.steps):
        torch.cuda.synchronize()
        time.sleep(0.004)                      # an idle gap in the kernel trace: how `analyse` tells the steps apart
        t = time.perf_counter()
        out = runner()
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t))
    print(f'B=1 {args.seconds:g} s clip, {"graph replay" if args.graph else "eager"}: host wall per step (ms) ' + ' '.join(f'{v:.3f}' for v in lat) +
          f' | p50 of the last {len(lat) - 2}: {float(np.median(lat[2:])):.3f}; notes {int(out["n_notes"][0])}')


def family(name: str) -> str:
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'hgemm3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)', name)
    if m:
        wm, wn, tm, tn, epi = map(int, m.groups())
        return f'gemm {wm * tm * 32}x{wn * tn * 32} epi{epi}'
    m = re.match(r'hgemm3p_kernel<(\d+)', name)
    if m:
        return f'gemm 256x256 persistent epi{m.group(1)}'
    return re.sub(r'<.*', '', name)[:32]


def analyse(args):
    rows = []
    for f in glob.glob(f'{args.dir}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    if not rows:
        raise SystemExit('no *kernel_trace.csv under ' + args.dir)
    # steps: the runner sleeps between them, so a step is a run of kernels with no idle gap longer than 1.5 ms
    steps, cur, end = [], [], None
    for r in rows:
        if cur and r[0] - end > 1_500_000:
            steps.append(cur)
            cur = []
        cur.append(r)
        end = r[1] if len(cur) == 1 else max(end, r[1])
    steps.append(cur)
    n_found = len(steps)
    common = max(set(len(s) for s in steps), key=lambda c: sum(1 for s in steps if len(s) == c))
    steps = [s for s in steps if len(s) == common][2:]                     # steady state: the usual launch count, warm-ups dropped
    if not steps:
        raise SystemExit(f'{len(rows)} kernel records in {n_found} runs, none repeated often enough; first names: ' + '; '.join(r[2][:40] for r in rows[:12]))
    print(f'{len(rows)} kernel records, {n_found} steps found, {len(steps)} steady-state steps of {common} launches analysed')
    agg = {'span': 0.0, 'busy': 0.0, 'overlap2': 0.0, 'idle': 0.0, 'serial': 0.0}
    fam = {}
    gaps = []
    for s in steps:
        t0, t1 = s[0][0], max(r[1] for r in s)
        ev = sorted([(r[0], 1) for r in s] + [(r[1], -1) for r in s])
        depth, last, busy, over = 0, t0, 0, 0
        for t, d in ev:
            if depth >= 1:
                busy += t - last
            if depth >= 2:
                over += t - last
            depth += d
            last = t
        agg['span'] += t1 - t0
        agg['busy'] += busy
        agg['overlap2'] += over
        agg['idle'] += (t1 - t0) - busy
        agg['serial'] += sum(r[1] - r[0] for r in s)
        for r in s:
            a = fam.setdefault(family(r[2]), [0, 0.0])
            a[0] += 1
            a[1] += r[1] - r[0]
        # idle gaps: between the running maximum of the end times and the next start
        end = s[0][1]
        prev = s[0]
        for r in s[1:]:
            if r[0] > end:
                gaps.append((r[0] - end, family(prev[2]), family(r[2])))
            if r[1] > end:
                end, prev = r[1], r
    n = len(steps)
    print(f'per step (mean of {n}): span {agg["span"] / n / 1e3:.1f} us | >= 1 kernel resident {agg["busy"] / n / 1e3:.1f} us | two kernels overlapping '
          f'{agg["overlap2"] / n / 1e3:.1f} us | idle inside the span {agg["idle"] / n / 1e3:.1f} us ({100.0 * agg["idle"] / agg["span"]:.1f} %) | '
          f'serial sum of kernel durations {agg["serial"] / n / 1e3:.1f} us')
    print('kernel families (launches per step, us per step, share of the serial sum):')
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f'  {k:34s} {c / n:6.1f}  {t / n / 1e3:8.1f}  {100.0 * t / agg["serial"]:5.1f} %')
    by = {}
    for g, a, b in gaps:
        e = by.setdefault((a, b), [0, 0.0])
        e[0] += 1
        e[1] += g
    print('idle gaps by (kernel before -> kernel after), per step:')
    for (a, b), (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f'  {a:30s} -> {b:30s} {c / n:5.1f} x  {t / n / 1e3:7.1f} us  (mean {t / c / 1e3:.1f} us)')


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest='cmd', required=True)
    r = sub.add_parser('run')
    r.add_argument('--steps', type=int, default=12)
    r.add_argument('--seconds', type=float, default=30.0)
    r.add_argument('--config', default='midi_conformer')
    r.add_argument('--graph', action='store_true', help='replay a captured hipGraph of the step (Engine.graph_runner)')
    a = sub.add_parser('analyse')
    a.add_argument('dir')
    args = ap.parse_args()
    run(args) if args.cmd == 'run' else analyse(args)


if __name__ == '__main__':
    main()
