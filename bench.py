#!/usr/bin/env python
"""Headline benchmark: audio-seconds per second (x real-time) of the SOME inference hot path on MI355X.

One "step" = one pass of the whole hot path (log-mel front end -> conformer forward -> note decode) over one
batch of synthetic 44.1 kHz clips that is already resident in HBM.  Workload = BASELINE.json configs[1]:
configs/midi_conformer.yaml (lay 8, 117.6 M params, fp32), batch of 32 x 30 s clips per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 5 --warmup 2

Multi-GPU: utterance sharding, one process per GPU, weights packed on rank 0 and broadcast once with RCCL,
no collective in the timed loop (weak scaling: every rank processes its own 32 clips).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

F32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)
F16_MATRIX_PEAK_TFLOPS = 2500.0     # dense f16 MFMA peak (v_mfma_f32_32x32x16_f16)
HBM_PEAK_GBS = 8000.0


def flops_per_frame(lay, outdim, T):
    """SURVEY.md section 8(d) 'Algorithmic FLOPs'."""
    nb = 2 * lay + 2
    dense = nb * 12_090_368 + lay * 2_097_152 + 163_840 + 1_024 * outdim + 1_024
    attn = nb * 2_048 * T
    return dense, attn


def _offline_evidence(kernel_name: str) -> dict:
    """Hardware-counter evidence for the roofline kernel.  Counters cannot be collected inside this process, so these
    figures are NOT measured by this run: they are read from the rocprofv3 --pmc passes committed under profiles/
    (same default workload) and reported under a separate `offline_evidence` key that says so.  HBM bytes per launch =
    FETCH_SIZE x 2 (the gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md "HBM") + WRITE_SIZE; MFMA-busy
    at the power-limited clock the kernel actually ran at."""
    prof = ROOT / 'profiles'
    key = {'attention': 'attention3'}.get(kernel_name)            # attention3i_kernel<DMA> (round 5) / attention3_kernel<...>
    out = {}
    if key is None:
        return out
    for hbm_file, busy_file in (('r06zz_pmc_hbm_traffic.json', 'r06zz_pmc_mfma_busy.json'),      # newest committed passes first
                                ('r06z_pmc_hbm_traffic.json', 'r06z_pmc_mfma_busy.json'),
                                ('r05zy_pmc_hbm_traffic.json', 'r05zy_pmc_mfma_busy.json'),
                                ('r05z_pmc_hbm_traffic.json', 'r05z_pmc_mfma_busy.json'),
                                ('r04n_pmc_hbm_traffic.json', 'r04n_pmc_mfma_busy.json'),
                                ('r03f_pmc_hbm_traffic.json', 'r03f_pmc_mfma_busy.json'),
                                ('r02f_pmc_hbm_traffic.json', 'r02f_pmc_mfma_busy.json'),
                                ('r02_pmc_hbm_traffic.json', 'r02_pmc_mfma_busy.json'),
                                ('r01_pmc_hbm_traffic_f16x3.json', 'r01_pmc_mfma_busy.json')):
        try:
            hbm = json.loads((prof / hbm_file).read_text())
            row = next(v for k, v in hbm.items() if k.startswith(key))
            out['traffic_bytes_per_launch'] = int((2 * row['FETCH_SIZE']['avg_KiB'] + row['WRITE_SIZE']['avg_KiB']) * 1024)
            out['traffic_source'] = f'profiles/{hbm_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x 2)'
        except (OSError, StopIteration, KeyError, ValueError):
            pass
        try:
            busy = json.loads((prof / busy_file).read_text())
            row = next(v for k, v in busy.items() if k.startswith(key))
            out['pmc_mfma_busy_frac'] = round(row['MfmaUtil_percent'] / 100.0, 4)
            out['pmc_effective_clock_mhz'] = round(row['effective_clock_MHz'])
            out['pmc_source'] = f'profiles/{busy_file} (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs))'
        except (OSError, StopIteration, KeyError, ValueError):
            pass
        if out:
            break
    if out:
        out['from_committed_profile'] = True
    return out


def _live_pmc(kernel_name: str, config_args) -> dict:
    """Hardware counters of the roofline kernel read IN THIS RUN: three one-step child runs of this script under
    ``rocprofv3 --kernel-trace --pmc ...`` (separate passes as MI355X_MICROARCH.md prescribes: MFMA-busy, FETCH_SIZE, WRITE_SIZE;
    serial grouped launches so that every dispatch is the kernel alone).  HBM bytes per launch = FETCH_SIZE x 2 (the gfx950
    correction for wide coalesced reads) + WRITE_SIZE, both in KiB; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs
    x 1024 SIMDs).  Returns {} when rocprofv3 is absent or a pass fails (the committed passes stay under offline_evidence)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    key = {'attention': 'attention3'}.get(kernel_name)            # attention3i_kernel<DMA> (round 5) / attention3_kernel<...>
    exe = shutil.which('rocprofv3')
    if key is None or exe is None or os.environ.get('SOME_AMD_BENCH_CHILD'):
        return {}
    child = [sys.executable, str(ROOT / 'bench.py'), '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-kernel-profile', '--no-latency',
             '--no-f32-leg', '--no-fast-leg', '--no-secondary', '--no-live-pmc', '--no-e2e', '--no-train', '--no-calibration'] + list(config_args)
    # (a child must not join the parent's process group: drop the torch.distributed.run variables)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'GROUP_RANK',
                                                              'LOCAL_WORLD_SIZE', 'ROLE_RANK', 'ROLE_WORLD_SIZE') and not k.startswith('TORCHELASTIC')}
    env.update(SOME_AMD_DUAL_STREAM='0', SOME_AMD_BENCH_CHILD='1', TMPDIR='/tmp')
    sums = {}
    with tempfile.TemporaryDirectory(dir='/tmp') as tmp:
        for tag, counters in (('mfma', ['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE']), ('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE'])):
            out_dir = os.path.join(tmp, tag)
            try:
                r = subprocess.run([exe, '--kernel-trace', '--pmc'] + counters + ['--output-format', 'csv', '-d', out_dir, '--'] + child,
                                   env=env, cwd='/tmp', capture_output=True, text=True, timeout=240)
            except (subprocess.TimeoutExpired, OSError):
                return {}
            if r.returncode != 0:
                return {}
            for f in glob.glob(out_dir + '/**/*counter_collection.csv', recursive=True):
                for row in csv.DictReader(open(f)):
                    if key not in row['Kernel_Name']:
                        continue
                    a = sums.setdefault(row['Counter_Name'], [0.0, 0, 0.0])
                    a[0] += float(row['Counter_Value'])
                    a[1] += 1
                    a[2] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    out = {}
    if 'FETCH_SIZE' in sums and 'WRITE_SIZE' in sums:
        out['traffic_bytes_per_launch'] = int((2.0 * sums['FETCH_SIZE'][0] / sums['FETCH_SIZE'][1] + sums['WRITE_SIZE'][0] / sums['WRITE_SIZE'][1]) * 1024)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in sums and sums.get('GRBM_GUI_ACTIVE', [0])[0] > 0:
        busy, gui = sums['SQ_VALU_MFMA_BUSY_CYCLES'], sums['GRBM_GUI_ACTIVE']
        out['mfma_busy_frac'] = round(busy[0] / (gui[0] / 8 * 1024), 4)
        out['effective_clock_mhz'] = round(gui[0] / 8 / gui[2] * 1e3)
        out['dispatches_sampled'] = busy[1]
    if out:
        out['source'] = 'live: rocprofv3 --kernel-trace --pmc child runs of this command (1 step each, serial grouped launches)'
    return out


def _tool_json(cmd, timeout, env=None):
    """Run a measurement tool as a child process (its own CUDA context) and return the JSON object on its last stdout line."""
    import subprocess
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT), env=dict(os.environ, **(env or {})))
    except subprocess.TimeoutExpired:
        return {'error': f'timed out after {timeout} s'}
    if r.returncode != 0:
        return {'error': (r.stdout[-600:] + r.stderr[-1200:]).strip()}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    return {'error': 'no JSON in the output'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='midi_conformer')
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU per step')
    ap.add_argument('--seconds', type=float, default=30.0, help='clip length')
    ap.add_argument('--lay', type=int, default=None, help='override lay (debug only; invalidates the metric)')
    ap.add_argument('--precision', default=None, choices=['f32', 'f16x3', 'f16x3_fast'], help='GEMM arithmetic (default: library default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-latency', action='store_true', help='skip the B=1 p50 latency leg')
    ap.add_argument('--no-f32-leg', action='store_true', help='skip the secondary exact-f32 measurement')
    ap.add_argument('--no-fast-leg', action='store_true', help='skip the opt-in f16x3_fast measurement')
    ap.add_argument('--no-secondary', action='store_true', help='skip the quant_two_head_model (BASELINE configs[2]) leg')
    ap.add_argument('--cpu-clips', type=int, default=10, help='clips in the bounded CPU-baseline sample (N-thread figure)')
    ap.add_argument('--cpu-single-thread-seconds', type=float, default=10.0, help='length of the one clip the 1-thread CPU figure runs')
    ap.add_argument('--no-live-pmc', action='store_true', help='skip the rocprofv3 --pmc child passes for roofline.traffic / MFMA-busy')
    ap.add_argument('--no-e2e', action='store_true', help='skip the whole-command leg of BASELINE configs[3]: batch_infer.py over --e2e-rows '
                    'synthetic 30 s WAVs on disk -> CSV (tools/batch_infer_bench.py), both arithmetic modes, rows re-checked one by one')
    ap.add_argument('--e2e-rows', type=int, default=10000, help='rows of the e2e leg (BASELINE configs[3] names 10 000 x 30 s clips)')
    ap.add_argument('--e2e-distinct', type=int, default=0, help='distinct WAV files of the e2e dataset (the other rows are hard links to them); 0 = rows / 8 '
                    '(3.3 GB for 10 000 rows, stays in the page cache); 10000 = every row a file of its own (26.5 GB, SURVEY 8(d))')
    ap.add_argument('--e2e-f32-rows', type=int, default=2000, help='rows the exact-f32 arm of the e2e leg annotates (the first N of the same dataset)')
    ap.add_argument('--no-train', action='store_true', help='skip the whole-command leg of BASELINE configs[4] on this GPU: one epoch of train.py '
                    'two_head_model bf16 over a synthetic 3 h binarised dataset (tools/train_epoch_bench.py)')
    ap.add_argument('--train-hours', type=float, default=3.0)
    ap.add_argument('--e2e', action='store_true', help=argparse.SUPPRESS)       # round-3 opt-in spellings: the legs are on by default now
    ap.add_argument('--train', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-calibration', action='store_true', help='skip the box calibration leg (some_box_calibrate, untimed, before the main loop)')
    ap.add_argument('--calibration-seconds', type=float, default=0.4)
    ap.add_argument('--scratch', default='/tmp/some_amd_bench', help='directory for the datasets of the --e2e / --train legs')
    args = ap.parse_args()

    import numpy as np
    import torch

    from some_amd import _lib, synth
    from some_amd.configs import get_config
    from some_amd.engine import ClipBatch, Engine

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f'--gpus {args.gpus} needs a torch.distributed.run launch with {args.gpus} processes')
    dev_index = local_rank % max(1, torch.cuda.device_count())
    if world > 1:                     # one process per GPU on a shared host: every rank on its own cores (some_amd/sharding.py)
        from some_amd import sharding
        sharding.bind_rank_to_cores(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dist = None
    # under torch.distributed.run the process group is initialised even for ONE rank, so that a single-GPU box exercises the RCCL
    # communicator, the arena broadcast, the barrier and the max-over-ranks all-reduce of the N > 1 path (tests/test_gpu_parity.py)
    launched = 'RANK' in os.environ and 'MASTER_PORT' in os.environ
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('SOME_AMD_DIST_BACKEND', 'nccl')   # "nccl" is RCCL on ROCm; gloo only for dry runs
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    cfg = get_config(args.config, lay=args.lay) if args.lay is not None else get_config(args.config)
    if args.precision:
        cfg['some_amd_precision'] = args.precision
    quant = cfg['task_cls'].endswith('QuantizedMIDIExtractionTask')
    lay, outdim = cfg['midi_extractor_args']['lay'], cfg['midi_num_bins']
    eng = Engine(cfg, device=device)
    precision_name = {v: k for k, v in __import__('some_amd.engine', fromlist=['PRECISIONS']).PRECISIONS.items()}[eng.c_config.precision]

    # ---- weights: rank 0 packs, RCCL broadcast of the flat fp32 arena over xGMI -------------------------
    sd = None
    arena = torch.empty(eng.arena_numel, dtype=torch.float32, device=device)
    if rank == 0:
        sd = synth.synth_state_dict(cfg, seed=cfg.get('seed', 114514))
        arena.copy_(eng.pack_state_dict(sd))
    if dist is not None:
        dist.broadcast(arena, src=0)
    eng.attach_arena(arena)

    # ---- synthetic clips (8 distinct per rank, tiled to the batch), resident in HBM ---------------------
    sr = cfg['audio_sample_rate']
    default_workload = (args.config == 'midi_conformer' and args.batch == 32 and args.seconds == 30.0 and
                        cfg['midi_extractor_args']['lay'] == 8)          # the configuration the committed PMC passes ran
    n_distinct = min(8, args.batch)
    clips = [synth.synth_clip(rank * 100 + i, args.seconds, sr) for i in range(n_distinct)]
    wave_list = [clips[i % n_distinct] for i in range(args.batch)]
    batch = ClipBatch.from_sample_counts([len(w) for w in wave_list], eng.hop, device)
    audio = torch.from_numpy(np.concatenate(wave_list)).to(device)
    T = int(batch.frame_counts[0])
    head = _lib.HEAD_SOFTMAX if quant else _lib.HEAD_SIGMOID

    def step():
        units = eng.logmel(audio, batch)
        probs, bounds = eng.forward(units, batch, head_mode=head)
        return eng.decode(probs, bounds, batch, quantized=quant)

    def sync_all():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    # ---- box calibration (untimed, before the main loop): what THIS GPU sustains on a pure f16 MFMA stream under its power limit and
    # on a float4 copy - boxes of this pool have run the same binary at 81.5 - 85.4 ms per step (power-limited clocks differ per package)
    box = None
    if not args.no_calibration:
        import ctypes
        lib = _lib.load()
        tf, mhz, gbs = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        rc = lib.some_box_calibrate(args.calibration_seconds, ctypes.addressof(tf), ctypes.addressof(mhz), ctypes.addressof(gbs),
                                    torch.cuda.current_stream(device).cuda_stream)
        if rc == 0:
            box = {'mfma_tf': round(tf.value, 1), 'mfma_effective_mhz': round(mhz.value), 'copy_gbs': round(gbs.value, 1),
                   'seconds': args.calibration_seconds,
                   'what': 'pure v_mfma_f32_32x32x16_f16 stream on random operands (issued TFLOP/s under the package power limit; clock = '
                           'TF / (1024 SIMDs x 1024 FLOP per SIMD-cycle)) and a 1 GiB float4 copy (read + written), some_box_calibrate'}
        else:
            box = {'error': f'some_box_calibrate returned {rc}'}

    for _ in range(args.warmup):
        out = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    n_notes = int(out['n_notes'].sum())
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    audio_seconds = world * args.batch * args.seconds * args.steps
    value = audio_seconds / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    dense, attn = flops_per_frame(lay, outdim, T)
    step_flops = (dense + attn) * batch.total_frames
    result = {
        'metric': 'audio-seconds/s (x real-time), SOME inference hot path (log-mel + conformer + decode)',
        'value': round(value, 2), 'unit': 'audio-s/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if precision_name == 'f32' else 'f32-equivalent (3-term split f16 MFMA, f32 accumulate)',
        'data': 'synthetic',
        'config': {'workload': f'configs/{args.config}.yaml inference, batch of {args.batch} x {args.seconds:g} s '
                               f'44.1 kHz mono clips per GPU (lay {lay}, {outdim} bins, T={T} frames/clip), random-init weights',
                   'clips_per_gpu': args.batch, 'clip_seconds': args.seconds, 'frames_per_clip': T,
                   'gemm_precision': precision_name,
                   'parallelism': f'utterance-sharded x{world}, RCCL weight broadcast, no data-path collective',
                   'execution': 'per GPU: the midi and bound model streams of each layer run on two HIP streams (fork / join per '
                                'layer); the per-kernel leg below is measured with serial grouped launches (HIP events need it), so '
                                'its kernel times sum to slightly more than ms_per_step'},
        'model_tflops': round(step_flops * world / (ms_per_step * 1e-3) / 1e12, 2),
        'notes_decoded_last_step': n_notes,
    }
    if dist is not None:
        result['process_group'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size()}
    if box is not None:
        result['box'] = box

    if rank == 0:
        # ---- per-kernel leg: HIP events around every launch on the launch stream (some_profile_*) -------
        if not args.no_kernel_profile:
            eng.profile_enable(True)
            prof_steps = max(1, min(args.steps, 3))
            for _ in range(prof_steps):
                step()
            torch.cuda.synchronize(device)
            stats = eng.profile_collect()
            eng.profile_enable(False)
            tot = sum(s['total_ms'] for s in stats) or 1.0
            kernels = []
            for s in sorted(stats, key=lambda s: -s['total_ms']):
                avg_ms = s['total_ms'] / s['launches']
                k = {'name': s['name'], 'launches_per_step': s['launches'] // prof_steps, 'avg_ms': round(avg_ms, 4),
                     'share': round(s['total_ms'] / tot, 4)}
                if s['flops'] > 0:
                    k['tflops'] = round(s['flops'] / s['launches'] / (avg_ms * 1e-3) / 1e12, 2)
                if s['bytes'] > 0:
                    k['gbs'] = round(s['bytes'] / s['launches'] / (avg_ms * 1e-3) / 1e9, 1)
                kernels.append(k)
            for k in kernels:           # front end: algorithmic bytes = samples in + log-mel out
                if k['name'] == 'logmel':
                    k['gbs'] = round((4.0 * audio.numel() + 4.0 * 80 * batch.total_frames) / (k['avg_ms'] * 1e-3) / 1e9, 1)
                if 'gbs' in k:
                    k['hbm_frac'] = round(k['gbs'] / HBM_PEAK_GBS, 4)
            result['kernels'] = kernels
            dom = next((k for k in kernels if 'tflops' in k), None)
            if dom is not None:
                flops_per_launch = dom['tflops'] * 1e12 * dom['avg_ms'] * 1e-3
                if precision_name in ('f16x3', 'f16x3_fast') and 'in->512' not in dom['name']:
                    # achieved = ALGORITHMIC FLOPs per launch / average launch time, against the
                    # dense f16 MFMA peak of the pipe the kernel runs on.  The 3-term split executes three f16 MFMA
                    # products per logical fp32 multiply-add; that issued-work figure is reported separately, as is the
                    # fraction of the exact-fp32 matrix peak (157.3 TF) the same logical work corresponds to.
                    result['roofline'] = {
                        'kernel': dom['name'], 'bound': 'mfma', 'achieved': dom['tflops'], 'peak': F16_MATRIX_PEAK_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': round(dom['tflops'] / F16_MATRIX_PEAK_TFLOPS, 4), 'traffic': None,
                        'avg_launch_ms': dom['avg_ms'], 'algorithmic_flops_per_launch': round(flops_per_launch),
                        'issued_f16_tflops': round(3.0 * dom['tflops'], 1),
                        'issued_frac': round(3.0 * dom['tflops'] / F16_MATRIX_PEAK_TFLOPS, 4),
                        'frac_of_fp32_matrix_peak': round(dom['tflops'] / F32_MATRIX_PEAK_TFLOPS, 3),
                        'note': 'frac = algorithmic fp32 FLOPs / time / 2.5 PF dense f16 peak; the matrix pipe issues 3 f16 MFMA '
                                'products per logical multiply-add (x = hi + lo split; ah*bh + ah*bl + al*bh) -> issued_frac',
                    }
                    if default_workload:
                        ev = _offline_evidence(dom['name'])
                        if ev:
                            result['offline_evidence'] = ev
                    if world == 1 and not args.no_live_pmc:
                        cfg_args = ['--config', args.config, '--batch', str(args.batch), '--seconds', str(args.seconds)]
                        if args.lay is not None:
                            cfg_args += ['--lay', str(args.lay)]
                        if args.precision:
                            cfg_args += ['--precision', args.precision]
                        live = _live_pmc(dom['name'], cfg_args)
                        if live:
                            result['roofline']['traffic'] = live.get('traffic_bytes_per_launch')
                            if dom['name'] == 'attention' and live.get('traffic_bytes_per_launch'):
                                # Q | K SPLIT32 planes + V^T f16 planes read, SPLIT32 output written: 4 x 2048 B per frame and model stream
                                algo = 2 * batch.total_frames * 8192
                                result['roofline']['algorithmic_bytes_per_launch'] = algo
                                result['roofline']['traffic_over_algorithmic_bytes'] = round(live['traffic_bytes_per_launch'] / algo, 3)
                            result['live_pmc'] = live
                else:
                    result['roofline'] = {
                        'kernel': dom['name'], 'bound': 'mfma', 'achieved': dom['tflops'], 'peak': F32_MATRIX_PEAK_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': round(dom['tflops'] / F32_MATRIX_PEAK_TFLOPS, 4), 'traffic': None,
                        'avg_launch_ms': dom['avg_ms'], 'algorithmic_flops_per_launch': round(flops_per_launch),
                    }
        # ---- value normalised to the reference box (profiles/box_reference.json: the median calibration of the boxes measured so far) ----
        # model: the step's matrix-bound share scales with the box's sustained MFMA rate, its HBM-bound share with the copy rate; shares
        # from this run's own per-kernel leg (kernels with a FLOP count vs kernels with a byte count)
        if box is not None and 'mfma_tf' in box:
            try:
                ref = json.loads((ROOT / 'profiles' / 'box_reference.json').read_text())
                ks = result.get('kernels') or []
                w_m = sum(k['share'] for k in ks if 'tflops' in k) if ks else ref['default_matrix_share']
                w_h = 1.0 - w_m
                scale = w_m * box['mfma_tf'] / ref['mfma_tf'] + w_h * box['copy_gbs'] / ref['copy_gbs']      # > 1: a faster box than the reference
                result['box'].update(reference=ref, matrix_share=round(w_m, 4), speed_vs_reference=round(scale, 4))
                result['value_normalised'] = round(value / scale, 2)
                result['ms_per_step_normalised'] = round(ms_per_step * scale, 3)
            except (OSError, KeyError, ValueError, ZeroDivisionError):
                pass
        # ---- p50 single-clip latency (B = 1, the reference's own granularity) ---------------------------
        one = ClipBatch.from_sample_counts([len(clips[0])], eng.hop, device)
        a1 = torch.from_numpy(clips[0]).to(device)
        lat = []
        for i in range(0 if args.no_latency else 8):
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            u = eng.logmel(a1, one)
            p, b = eng.forward(u, one, head_mode=head)
            eng.decode(p, b, one, quantized=quant)
            torch.cuda.synchronize(device)
            lat.append(time.perf_counter() - t1)
        if lat:
            result['p50_clip_latency_ms'] = round(1e3 * float(np.median(lat[2:])), 3)
            # the same clip through ONE hipGraph launch (Engine.graph_runner: repeated batch shapes; bit-identical outputs)
            try:
                runner = eng.graph_runner(a1, one, head_mode=head, quantized=quant)
                glat = []
                for i in range(10):
                    torch.cuda.synchronize(device)
                    t1 = time.perf_counter()
                    runner(a1)
                    torch.cuda.synchronize(device)
                    glat.append(time.perf_counter() - t1)
                result['p50_clip_latency_graph_replay_ms'] = round(1e3 * float(np.median(glat[2:])), 3)
                del runner
            except RuntimeError as e:
                result['p50_clip_latency_graph_replay_ms'] = f'capture failed: {e}'[:200]

        # ---- the same workload in the exact-f32 arithmetic mode (bit-conservative option), for reference ---------
        if world == 1 and precision_name != 'f32' and not args.no_f32_leg:
            cfg32 = dict(cfg, some_amd_precision='f32')
            eng32 = Engine(cfg32, device=device)
            arena32 = eng32.pack_state_dict(sd).to(device)
            eng32.attach_arena(arena32)

            def step32():
                u = eng32.logmel(audio, batch)
                p, b = eng32.forward(u, batch, head_mode=head)
                return eng32.decode(p, b, batch, quantized=quant)
            step32()
            torch.cuda.synchronize(device)
            t32 = time.perf_counter()
            for _ in range(2):
                step32()
            torch.cuda.synchronize(device)
            dt32 = (time.perf_counter() - t32) / 2
            result['exact_f32_mode'] = {'value': round(args.batch * args.seconds / dt32, 2), 'unit': 'audio-s/s',
                                        'ms_per_step': round(dt32 * 1e3, 3), 'dtype': 'f32 (v_mfma_f32_32x32x2_f32)'}
            del eng32, arena32
        # ---- the opt-in fast mode (f16x3_fast: attention P V on two terms; never the headline) --------------------------------------
        if world == 1 and precision_name == 'f16x3' and not args.no_fast_leg:
            cfgf = dict(cfg, some_amd_precision='f16x3_fast')
            engf = Engine(cfgf, device=device)
            engf.attach_arena(arena)                      # same packed weights (the GEMM operands are the f16x3 ones)

            def stepf():
                u = engf.logmel(audio, batch)
                p, b = engf.forward(u, batch, head_mode=head)
                return p, b, engf.decode(p, b, batch, quantized=quant)
            stepf()
            torch.cuda.synchronize(device)
            tf0 = time.perf_counter()
            for _ in range(args.steps):
                pf, bf, outf = stepf()
            torch.cuda.synchronize(device)
            dtf = (time.perf_counter() - tf0) / args.steps
            u0 = eng.logmel(audio, batch)
            p0, b0 = eng.forward(u0, batch, head_mode=head)
            out0 = eng.decode(p0, b0, batch, quantized=quant)
            result['fast_mode'] = {'value': round(args.batch * args.seconds / dtf, 2), 'unit': 'audio-s/s', 'ms_per_step': round(dtf * 1e3, 3),
                                   'dtype': 'f16x3 with the attention product P V on two terms (SOME_PRECISION_F16X3_FAST, opt-in)',
                                   'max_abs_dprob_vs_default': float((pf - p0).abs().max()), 'max_abs_dbound_vs_default': float((bf - b0).abs().max()),
                                   'notes_decoded': int(outf['n_notes'].sum()), 'notes_decoded_default': int(out0['n_notes'].sum())}
            del engf
        # ---- BASELINE.json configs[2]: quant_two_head_model (lay 3, 129 bins, softmax head + argmax decode), same batch ----
        if world == 1 and args.config == 'midi_conformer' and not args.no_secondary:
            cfg_q = get_config('quant_two_head_model')
            if args.precision:
                cfg_q['some_amd_precision'] = args.precision
            eng_q = Engine(cfg_q, device=device)
            eng_q.attach_arena(eng_q.pack_state_dict(synth.synth_state_dict(cfg_q, seed=cfg_q.get('seed', 114514))).to(device))

            def step_q():
                u = eng_q.logmel(audio, batch)
                p, b = eng_q.forward(u, batch, head_mode=_lib.HEAD_SOFTMAX)
                return eng_q.decode(p, b, batch, quantized=True)
            for _ in range(2):
                step_q()
            torch.cuda.synchronize(device)
            tq = time.perf_counter()
            for _ in range(args.steps):
                step_q()
            torch.cuda.synchronize(device)
            dtq = (time.perf_counter() - tq) / args.steps
            dq, aq = flops_per_frame(cfg_q['midi_extractor_args']['lay'], cfg_q['midi_num_bins'], T)
            result['secondary'] = {
                'workload': f'configs/quant_two_head_model.yaml inference (quantised pitch head: softmax + argmax decode), batch of '
                            f'{args.batch} x {args.seconds:g} s clips (lay 3, 129 bins, T={T})',
                'value': round(args.batch * args.seconds / dtq, 2), 'unit': 'audio-s/s', 'ms_per_step': round(dtq * 1e3, 3),
                'steps': args.steps, 'model_tflops': round((dq + aq) * batch.total_frames / dtq / 1e12, 2)}
            del eng_q
        # ---- CPU baseline: the oracle (port of the reference CPU path), bounded sample, rank 0, N = 1 ---
        # SURVEY 8(d) protocol: B = 1 per clip exactly as BaseInference.infer (inference/base_infer.py:46-53), an N-thread AND a 1-thread
        # figure, the stages (mel / forward / decode) timed separately, >= 10 clips for the N-thread figure
        if world == 1 and not args.no_cpu_baseline:
            from oracle import restate
            # B = 1 conformer inference scales poorly past ~32 threads on a 2-socket host (measured on the MI355X
            # box, 2 x EPYC 9575F: 8 -> 8.2, 16 -> 9.2, 32 -> 10.7, 64 -> 6.9, 128 -> 3.0 audio-s/s): use the best
            threads = min(32, torch.get_num_threads())
            sd_t = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}

            def cpu_leg(n_threads, n_clips, seconds):
                torch.set_num_threads(n_threads)
                n_samp = int(round(seconds * sr))
                restate.infer_clip(sd_t, cfg, clips[0][: sr * 2], quantized=quant)            # warm-up (2 s)
                stages = {'mel': 0.0, 'forward': 0.0, 'decode': 0.0}
                tc = time.perf_counter()
                for i in range(n_clips):
                    r = restate.infer_clip(sd_t, cfg, clips[i % n_distinct][:n_samp], quantized=quant)      # B = 1 per clip, as the reference
                    for k2, v2 in r['_stage_s'].items():
                        stages[k2] += v2
                dt = time.perf_counter() - tc
                return {'value': round(n_clips * seconds / dt, 2), 'unit': 'audio-s/s', 'threads': n_threads, 'clips': n_clips,
                        'clip_seconds': seconds, 'wall_s': round(dt, 2), 'stage_s_per_clip': {k2: round(v2 / n_clips, 4) for k2, v2 in stages.items()}}
            many = cpu_leg(threads, args.cpu_clips, args.seconds)
            single = cpu_leg(1, 1, min(args.seconds, args.cpu_single_thread_seconds))
            torch.set_num_threads(threads)
            result['cpu_baseline'] = {
                'value': many['value'], 'unit': 'audio-s/s', 'cores': threads,
                'threads': threads, 'host_cores': os.cpu_count(),      # cores = threads used (the bench contract); the box has host_cores
                'host_cores_available_to_this_process': len(os.sched_getaffinity(0)),
                'kind': 'port',
                'sample': f'{args.cpu_clips} x {args.seconds:g} s clips of the same workload, B=1 per clip '
                          f'(log-mel + forward + decode), torch-CPU fp32 oracle, {threads} threads; one_thread: 1 x '
                          f'{single["clip_seconds"]:g} s clip on 1 thread',
                'stage_s_per_clip': many['stage_s_per_clip'], 'wall_s': many['wall_s'],
                'one_thread': single,
            }
        whole_command_legs = world == 1 and default_workload and not os.environ.get('SOME_AMD_BENCH_CHILD')
        if whole_command_legs and not args.no_e2e:
            # BASELINE configs[3], the WHOLE command: batch_infer.py from WAV files on disk to the CSV on disk (reference batch_infer.py:149-226)
            # with a briefly trained checkpoint (realistic note counts), dataset cached under --scratch.  Three child runs: cold (torch.load +
            # weight pack), warm (cached weight arena) with sampled rows recomputed ONE BY ONE through host Slicer + infer() and compared as
            # CSV strings, and the same in the exact-f32 arithmetic mode.
            distinct = args.e2e_distinct if args.e2e_distinct > 0 else max(8, args.e2e_rows // 8)
            ds = os.path.join(args.scratch, f'e2e_{args.e2e_rows}' + (f'_d{distinct}' if args.e2e_distinct > 0 else ''))
            tool = [sys.executable, str(ROOT / 'tools' / 'batch_infer_bench.py'), '--clips', str(args.e2e_rows), '--distinct',
                    str(distinct), '--seconds', '30', '--lay', str(lay), '--dir', ds, '--train_updates', '300', '--json']
            t_leg = time.perf_counter()
            cold = _tool_json(tool, 1500)
            warm = _tool_json(tool + ['--check', '24'], 1500)
            # exact-f32 arm: the first --e2e-f32-rows rows of the same dataset, its CSV compared with the f16x3 CSV (rows / note boundaries differing)
            f32 = _tool_json(tool + ['--check', '24', '--limit', str(min(args.e2e_f32_rows, args.e2e_rows)), '--out', 'out_f32.csv', '--compare', 'out.csv'],
                             1500, env={'SOME_AMD_PRECISION': 'f32'})
            result['e2e_batch_infer'] = dict(warm, gemm_precision='f16x3', cold_start=cold, exact_f32_mode=dict(f32, gemm_precision='f32'),
                                             leg_wall_s=round(time.perf_counter() - t_leg, 1))
        if whole_command_legs and not args.no_train:
            # BASELINE configs[4] at its own size on this ONE GPU (reference train.py:57-98): one epoch over a synthetic 3 h binarised dataset
            ds = os.path.join(args.scratch, f'train_{args.train_hours:g}h')
            t_leg = time.perf_counter()
            if not os.path.exists(os.path.join(ds, 'train.lengths')):
                made = _tool_json([sys.executable, str(ROOT / 'tools' / 'make_train_dataset.py'), '--dir', ds, '--hours', str(args.train_hours)], 1500)
                if 'error' in made and 'no JSON' not in made['error']:
                    result['train_epoch'] = made
            if 'train_epoch' not in result:
                result['train_epoch'] = _tool_json([sys.executable, str(ROOT / 'tools' / 'train_epoch_bench.py'), '--dir', ds], 1500)
                result['train_epoch']['leg_wall_s'] = round(time.perf_counter() - t_leg, 1)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
