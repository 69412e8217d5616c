"""Repeat-run bit-equality gate at FULL size (32 x 30 s = 82 688 frames, BASELINE configs[1]).

Why it exists: twice this code base met results that depended on timing and passed every tolerance test - round 1's attention
K-ring race (rare large errors on clips whose first key tile needs no masking) and round 2's SGPR-wave-index FFN1 epilogue (a
gfx950 store-data hazard hipcc leaves unprotected, ~1 % of rows off by a lo half; profiles/r03_sgpr_epilogue_hazard.md).  Both
showed up only at full size and only as run-to-run differences; the second one was caught by bench.py's note-count fingerprint,
not by a test.  So: every kernel family and the whole forward, 5 repeats, {dual-stream, grouped launches} x {f16x3, f32},
bit-equality, plus the bench fingerprint (44 604 notes for the default bench workload) as an assertion.

Through the C ABI like every GPU test (Engine -> ctypes -> libsome_amd.so)."""
import ctypes as C

import numpy as np
import pytest
import torch

from some_amd import synth
from some_amd.configs import get_config

pytestmark = pytest.mark.gpu

REPEATS = 5
FULL_ROWS = 32 * 2584
BENCH_FINGERPRINT = 44604          # notes decoded in one default `python bench.py` step (profiles/r02_experiments.md, r03a_*)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bench_workload(cfg):
    """bench.py's rank-0 step: 8 distinct 30 s clips tiled to 32, seed 114514 weights."""
    from some_amd.engine import ClipBatch
    clips = [synth.synth_clip(i, 30.0, cfg['audio_sample_rate']) for i in range(8)]
    waves = [clips[i % 8] for i in range(32)]
    batch = ClipBatch.from_sample_counts([len(w) for w in waves], cfg['hop_size'], 'cuda')
    audio = torch.from_numpy(np.concatenate(waves)).cuda()
    return audio, batch


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
def test_full_size_forward_is_bit_stable_across_repeats_and_launch_groupings(precision, monkeypatch):
    from some_amd import _lib
    from some_amd.engine import Engine
    cfg = get_config('midi_conformer', some_amd_precision=precision)
    sd = synth.synth_state_dict(cfg, seed=cfg.get('seed', 114514))
    audio, batch = _bench_workload(cfg)
    assert batch.total_frames == FULL_ROWS
    first = {}
    for dual in ('1', '0'):
        monkeypatch.setenv('SOME_AMD_DUAL_STREAM', dual)
        eng = Engine(cfg, device='cuda')
        eng.load_state_dict(sd)
        ref = None
        for rep in range(REPEATS):
            units = eng.logmel(audio, batch)
            probs, bounds = eng.forward(units, batch, head_mode=_lib.HEAD_SIGMOID)
            dec = eng.decode(probs, bounds, batch, quantized=False)
            torch.cuda.synchronize()
            # note arrays are packed per clip at its frame offset; entries behind a clip's n_notes are never written
            n = dec['n_notes'].to(torch.int64)
            pos = torch.arange(batch.total_frames, device='cuda')
            clip = torch.repeat_interleave(torch.arange(batch.B, device='cuda'), torch.from_numpy(np.asarray(batch.frame_counts)).cuda().to(torch.int64))
            valid = pos - batch.frame_offsets_dev.to(torch.int64)[clip] < n[clip]
            got = (units.clone(), probs.clone(), bounds.clone(), dec['n_notes'].clone(), dec['note_dur'][valid].clone(),
                   dec['note_midi'][valid].clone(), dec['note_rest'][valid].clone())
            if ref is None:
                ref = got
            else:
                for name, a, b in zip(('units', 'probs', 'bounds', 'n_notes', 'note_dur', 'note_midi', 'note_rest'), ref, got):
                    assert torch.equal(a, b), f'{precision} dual={dual}: {name} differs between run 0 and run {rep}'
        first[dual] = ref
        del eng
    for name, a, b in zip(('units', 'probs', 'bounds', 'n_notes'), first['1'], first['0']):
        assert torch.equal(a, b), f'{precision}: {name} differs between the dual-stream and the grouped-launch forward'
    n_notes = int(first['1'][3].sum())
    print(f'{precision}: {n_notes} notes in the bench step')
    if precision == 'f16x3':
        assert n_notes == BENCH_FINGERPRINT


def _repeat_equal(fn, what):
    ref = fn()
    torch.cuda.synchronize()
    ref = [t.clone() for t in ref]
    for rep in range(1, REPEATS):
        out = fn()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(ref, out)):
            assert torch.equal(a, b), f'{what}: output {i} differs between run 0 and run {rep} ({int((a != b).sum())} elements)'


@pytest.fixture(scope='module')
def eng():
    from some_amd.engine import Engine
    return Engine(get_config('midi_conformer', lay=1), device='cuda')


@pytest.fixture(scope='module')
def operands(eng):
    g = torch.Generator(device='cuda').manual_seed(77)
    M = FULL_ROWS
    x512 = torch.randn(M, 512, device='cuda', generator=g)
    x2048 = torch.randn(M, 2048, device='cuda', generator=g) * 0.3
    return {'x512': x512, 'x2048': x2048, 'g': g}


def _split(eng, x):
    from some_amd import _lib
    out = torch.empty_like(x)
    _lib.check(eng.handle, eng.lib.some_op_split_rows(eng.handle, _p(x), _p(out), x.shape[0], x.shape[1], _stream()))
    return out


GEMM_CASES = [
    # name, epilogue, K, N, n_out, split output, residual, mask
    ('ffn1 bias+silu -> SPLIT32', 'EPI_BIAS_SILU', 512, 2048, 2048, True, False, False),
    ('ffn2 bias+res', 'EPI_BIAS_RES', 2048, 512, 512, False, True, False),
    ('out-proj bias+res', 'EPI_BIAS_RES', 512, 512, 512, False, True, False),
    ('pw1 glu', 'EPI_GLU', 512, 1024, 512, False, False, False),
    ('gate glu+res+mask', 'EPI_GLU_RES', 512, 1024, 512, False, True, True),
    ('head bias', 'EPI_BIAS', 512, 128, 128, False, False, False),
]


@pytest.mark.parametrize('case', GEMM_CASES, ids=[c[0] for c in GEMM_CASES])
@pytest.mark.parametrize('mode', ['f16x3', 'f32'])
def test_full_size_gemm_epilogues_bit_stable(eng, operands, case, mode):
    from some_amd import _lib as L
    name, epi, K, N, n_out, out_split, use_res, use_mask = case
    if mode == 'f32' and out_split:
        pytest.skip('SPLIT32 outputs exist in the split-f16 mode only')
    g = operands['g']
    M = FULL_ROWS
    A = operands['x512'] if K == 512 else operands['x2048']
    W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    res = torch.randn(M, n_out, device='cuda', generator=g) if use_res else None
    mask = (torch.rand(M, device='cuda', generator=g) > 0.1).to(torch.uint8) if use_mask else None
    flags = 0
    if mode == 'f16x3':
        A, W = _split(eng, A), _split(eng, W.contiguous())
        flags = L.GEMM_SPLIT_IN | (2 << 8) | (L.GEMM_SPLIT_OUT if out_split else 0)
    out = torch.empty(M, n_out, device='cuda')

    def run():
        out.view(torch.int32).fill_(0x7FC12345)
        L.check(eng.handle, eng.lib.some_op_gemm(eng.handle, getattr(L, epi), _p(A), A.stride(0), _p(W), _p(b), _p(res), 0 if res is None else n_out,
                                                 _p(out), n_out, M, N, K, 0.5, 0, _p(mask), flags, _stream()))
        return [out]

    _repeat_equal(run, f'{name} [{mode}]')
    assert not (out.view(torch.int32) == 0x7FC12345).any(), 'an output element was never written'


def test_full_size_qkv_attention_f16x3_bit_stable(eng, operands):
    from some_amd import _lib
    from some_amd.engine import ClipBatch
    lens = [2584] * 32                       # every clip starts on a 64-frame boundary only if 2584 * b % 64 == 0: b = 0, 8, 16, 24
    batch = ClipBatch(lens, 'cuda')
    M = batch.total_frames
    g = operands['g']
    W = torch.randn(1536, 512, device='cuda', generator=g) / 512 ** 0.5
    W[:512] *= 3.0
    hs, Ws = _split(eng, operands['x512']), _split(eng, W)
    ws = torch.empty(eng.lib.some_op_qkv_attention_f16x3_bytes(M, batch.B), dtype=torch.uint8, device='cuda')
    out = torch.empty(M, 512, device='cuda')

    def run():
        out.view(torch.int32).fill_(0x7FC12345)
        _lib.check(eng.handle, eng.lib.some_op_qkv_attention_f16x3(eng.handle, _p(hs), _p(Ws), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M,
                                                                   _p(out), _p(ws), ws.numel(), _stream()))
        return [out]

    _repeat_equal(run, 'qkv + attention f16x3')


def test_full_size_attention_f32_bit_stable(eng, operands):
    from some_amd import _lib
    from some_amd.engine import ClipBatch
    batch = ClipBatch([2584] * 8 + [64, 1, 700], 'cuda')
    M = batch.total_frames
    g = operands['g']
    qkv = torch.randn(M, 1536, device='cuda', generator=g)
    out = torch.empty(M, 512, device='cuda')

    def run():
        _lib.check(eng.handle, eng.lib.some_op_attention(eng.handle, _p(qkv), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, _p(out), 0, _stream()))
        return [out]

    _repeat_equal(run, 'attention f32')


def test_full_size_layernorm_and_dwconv_bit_stable(eng, operands):
    from some_amd import _lib
    from some_amd.engine import ClipBatch
    M = FULL_ROWS
    g = operands['g']
    x = operands['x512']
    gamma, beta = torch.randn(512, device='cuda', generator=g), torch.randn(512, device='cuda', generator=g)
    y32, ysp = torch.empty_like(x), torch.empty_like(x)

    def run_ln():
        _lib.check(eng.handle, eng.lib.some_op_layernorm(eng.handle, _p(x), _p(gamma), _p(beta), _p(y32), _p(ysp), M, _stream()))
        return [y32, ysp]

    _repeat_equal(run_ln, 'layernorm')
    batch = ClipBatch([2584] * 32, 'cuda')
    taps = torch.randn(31, 512, device='cuda', generator=g) / 31 ** 0.5
    shift = torch.randn(512, device='cuda', generator=g) * 0.1
    out = torch.empty_like(x)

    def run_dw():
        _lib.check(eng.handle, eng.lib.some_op_dwconv_silu(eng.handle, _p(x), _p(taps), _p(shift), _p(batch.frame_offsets_dev), batch.B, batch.max_frames,
                                                           _p(out), 1, _stream()))
        return [out]

    _repeat_equal(run_dw, 'dwconv + bn + silu')


@pytest.mark.parametrize('mode', ['bf16', 'f16x3'])
def test_training_step_is_bit_stable_across_fresh_trainers(mode):
    """Two fresh trainers, same seed, same 8 x 2584-frame batch, two updates each: identical parameters (no atomics anywhere in the
    training kernels; split-K / column reductions sum partial planes in a fixed order)."""
    from some_amd.training.task import MIDIExtractionTrainer
    cfg = dict(get_config('two_head_model'), pl_trainer_precision='bf16' if mode == 'bf16' else '32-true')
    sample = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_train_batch(B=8, T=2584, seed=5).items()}
    flats = []
    for _ in range(2):
        tr = MIDIExtractionTrainer(cfg, device='cuda', seed=3)
        for _ in range(2):
            out = tr.training_step(sample)
            assert not out['skipped']
        flats.append(tr.model.params.flat.clone())
        del tr
    assert torch.equal(flats[0], flats[1])
