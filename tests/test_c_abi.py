"""The C-ABI shared library loads and exports every symbol include/some_amd.h declares; host-only entry
points (weight packing, mel basis, argument validation) behave.  No compute calls: runs without a GPU."""
import ctypes as C
import pathlib
import re

import numpy as np
import pytest

from some_amd import _lib, synth
from some_amd.configs import get_config

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_every_declared_symbol_is_exported_and_bound():
    header = (ROOT / 'include' / 'some_amd.h').read_text()
    declared = set(re.findall(r'\b(some_[a-z_0-9]+)\s*\(', header))
    lib = _lib.load()
    assert declared, 'no declarations parsed'
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/some_amd.h but not exported'
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert b'gfx950' in lib.some_version()


def test_create_validates_configuration():
    lib = _lib.load()
    from some_amd.engine import make_some_config
    cfg = get_config('midi_conformer')
    cfg['midi_extractor_args']['dim'] = 256
    h = C.c_void_p()
    rc = lib.some_create(C.byref(make_some_config(cfg)), C.byref(h))
    assert rc == _lib.SOME_EINVAL and b'dim=256' in lib.some_last_error(None)


def test_pack_weights_layout_and_strictness():
    from some_amd.engine import Engine
    import torch
    cfg = get_config('midi_conformer', lay=1, some_amd_precision='f32')
    eng = Engine(cfg, host_only=True)
    sd = synth.synth_state_dict(cfg, 3)
    arena = eng.pack_state_dict(sd).numpy()
    # f16x3 mode: same arena size; GEMM weights become SPLIT32 (per 32-element k-block: 32 f16 hi | 32 f16 lo)
    eng3 = Engine(get_config('midi_conformer', lay=1, some_amd_precision='f16x3'), host_only=True)
    arena3 = eng3.pack_state_dict(sd).numpy()
    assert arena3.shape == arena.shape
    np.testing.assert_array_equal(arena3[:512 * 80], arena[:512 * 80])            # input projection stays fp32
    w = sd['model.outln.weight']                                                  # first split tensor in the arena
    off = int(np.flatnonzero(arena == w.reshape(-1)[0])[0])
    blk = arena3[off:off + 32].view(np.float16)
    hi, lo = blk[:32].astype(np.float32), blk[32:].astype(np.float32)
    np.testing.assert_array_equal(hi, w.reshape(-1)[:32].astype(np.float16).astype(np.float32))
    # |x - hi - lo| <= 2^-22 |x|, floored by the f16 subnormal quantum (2^-24) / 2 for the small lo halves
    assert (np.abs(hi + lo - w.reshape(-1)[:32]) <= np.abs(w.reshape(-1)[:32]) * 2.0 ** -21 + 2.0 ** -25).all()
    # the whole tensor, against numpy's own round-to-nearest-even fp16 conversions (the host side converts with F16C where
    # the CPU has it): hi = f16(x), lo = f16(x - hi), bit for bit, tiny values (f16 subnormals) included
    wf = (w.reshape(-1) * np.where(np.arange(w.size) % 7 == 0, 1e-4, 1.0)).astype(np.float32).reshape(w.shape)
    sd2 = dict(sd, **{'model.outln.weight': wf})
    a3 = eng3.pack_state_dict(sd2).numpy()
    blocks = a3[off:off + wf.size].view(np.float16).reshape(-1, 2, 32)
    x = wf.reshape(-1, 32)
    want_hi = x.astype(np.float16)
    want_lo = (x - want_hi.astype(np.float32)).astype(np.float16)
    np.testing.assert_array_equal(blocks[:, 0].view(np.uint16), want_hi.view(np.uint16))
    np.testing.assert_array_equal(blocks[:, 1].view(np.uint16), want_lo.view(np.uint16))
    assert arena.shape[0] == eng.arena_numel
    # first tensor of the arena is inln.weight verbatim
    np.testing.assert_array_equal(arena[:512 * 80].reshape(512, 80), sd['model.inln.weight'])
    total = sum(int(np.prod(v.shape)) for k, v in sd.items() if not k.endswith('num_batches_tracked'))
    # BN (4 x 512 per block) is folded away; padding only adds zeros
    assert np.count_nonzero(arena) <= total and np.count_nonzero(arena) >= total - 4 * 4 * 512 - 100
    # the folded depthwise taps: w' = w * gamma / sqrt(var + eps), stored [31, 512]
    p = 'model.cf_lay.0.att1.conv.'
    scale = sd[p + 'norm.weight'].astype(np.float64) / np.sqrt(sd[p + 'norm.running_var'].astype(np.float64) + 1e-5)
    taps = (sd[p + 'depthwise_conv.weight'][:, 0, :].astype(np.float64) * scale[:, None]).T.astype(np.float32)
    flat = taps.reshape(-1)
    idx = np.flatnonzero(arena == flat[0])
    assert any(np.array_equal(arena[i:i + flat.size], flat) for i in idx)
    for mutate, pattern in [(lambda d: d.pop('model.att2.norm5.weight'), 'Missing key'),
                            (lambda d: d.update({'model.bogus': np.zeros(2, np.float32)}), 'Unexpected key'),
                            (lambda d: d.update({'model.cutheard.weight': np.zeros((2, 512), np.float32)}), 'size mismatch')]:
        bad = dict(sd)
        mutate(bad)
        with pytest.raises(_lib.SomeError, match=pattern):
            eng.pack_state_dict(bad)


def test_forward_requires_weights_and_gpu_pointers():
    lib = _lib.load()
    from some_amd.engine import Engine
    eng = Engine(get_config('midi_conformer', lay=1), host_only=True)
    rc = lib.some_forward(eng.handle, None, None, 1, 10, 10, None, 0, None, None, None, 0, None)
    assert rc == _lib.SOME_ESTATE and b'attach' in lib.some_last_error(eng.handle)
    assert lib.some_workspace_bytes(eng.handle, 1000, 1) >= 2 * 1000 * (512 * 3 + 2048) * 4
    # a batch of very short clips: the clip-aligned attention operands (every clip padded to a multiple of 16 rows) outgrow the FFN rows
    mc = (1000 + 15 * 1000 + 126) // 64 * 64
    assert lib.some_workspace_bytes(eng.handle, 1000, 1000) >= 2 * (1000 * 512 * 3 * 4 + mc * 4096 + 2048 * mc) + 4 * mc
    assert lib.some_op_qkv_attention_f16x3_bytes(2584, 1) >= 2688 * 4096 + 2048 * 2688 + 4 * 2688
    # argument checks that need no GPU: with an (unused) arena pointer attached, oversized batches are refused before anything is launched
    import ctypes as C
    fake = C.c_void_p(1 << 20)
    assert lib.some_attach_arena(eng.handle, fake, lib.some_arena_bytes(eng.handle)) == _lib.SOME_OK
    assert lib.some_forward(eng.handle, fake, fake, 1, 262144, 262144, None, 0, fake, fake, fake, 1 << 40, None) == _lib.SOME_EINVAL
    assert b'total_frames too large' in lib.some_last_error(eng.handle)
    assert lib.some_forward(eng.handle, fake, fake, 70000, 70000, 1, None, 0, fake, fake, fake, 1 << 40, None) == _lib.SOME_EINVAL
    assert b'too many short clips' in lib.some_last_error(eng.handle)
    assert lib.some_decode_scratch_bytes(eng.handle, 1000) >= 1000 * 13


def test_train_gemm16_validates_arguments_before_any_launch():
    """some_train_gemm16 (training GEMMs on fp32 operands as they lie): layout / alignment / size rules are checked on the host
    and refused with SOME_EINVAL - in particular a partial-plane buffer smaller than the MODE'S OWN slice count needs (the
    one-product modes cut the contraction finer than the split mode: sizing by the wrong mode once wrote past the buffer)."""
    from some_amd.engine import Engine
    lib = _lib.load()
    eng = Engine(get_config('two_head_model', lay=1), host_only=True)
    h = eng.handle
    fake = C.c_void_p(4096)                  # never dereferenced: every call below fails validation first

    def call(lda=512, ta=0, ldb=512, tb=0, bias=None, ldc=2048, M=256, N=2048, K=512, operand=2, sum_col=-1, partial=None, pbytes=0):
        return lib.some_train_gemm16(h, fake, lda, ta, fake, ldb, tb, bias, fake, ldc, M, N, K, operand, sum_col, partial, pbytes, None)

    for kw, msg in [(dict(operand=0), b'operand'), (dict(operand=4), b'operand'), (dict(ta=1, tb=0), b'layouts'), (dict(K=512, lda=514, ldb=514), b'% 4'),
                    (dict(K=48, lda=48, ldb=48), b'K % 32'), (dict(lda=510), b'leading'), (dict(sum_col=2048), b'sum_col'),
                    (dict(ldc=1024), b'ldc'), (dict(M=1 << 20, lda=2048, K=2048, ldb=2048), b'2 GiB')]:
        assert call(**kw) == _lib.SOME_EINVAL, kw
        assert msg in lib.some_last_error(h), (kw, lib.some_last_error(h))
    # weight-gradient layout: dW [2048, 512 (+4)] = dY [20672, 2048]^T x [20672, 512]
    wg = dict(lda=2048, ta=1, ldb=512, tb=1, ldc=516, M=2048, N=512, K=20672, sum_col=512)
    plane = 2048 * 516 * 4
    need = lib.some_train_gemm16_bytes(h, 2048, 512, 20672, 516)
    assert need >= 8 * plane                                           # at least 8 slices for a 32 / 64-tile output
    for operand in (1, 2, 3):
        assert call(operand=operand, partial=fake, pbytes=plane, **wg) == _lib.SOME_EINVAL
        assert b'partial buffer too small' in lib.some_last_error(h)
    assert call(bias=fake, partial=fake, pbytes=need, **wg) == _lib.SOME_EINVAL and b'no bias' in lib.some_last_error(h)


_SWEEP = r"""
import ctypes as C, json, sys
sys.path.insert(0, sys.argv[1])
from some_amd import _lib, fastcall_gen
lib, fast = _lib.load(), _lib.fast()
assert getattr(fast, 'fastcall', False)
out = {}
for name, (res, args) in _lib.SYMBOLS.items():
    if not fastcall_gen.wrappable(res, args):
        assert getattr(fast, name) is getattr(lib, name) or getattr(fast, name).__name__ == getattr(lib, name).__name__
        continue
    assert type(getattr(fast, name)).__name__ == 'builtin_function_or_method', name
    vals = [None if a is C.c_void_p else (0.5 if a in (C.c_float, C.c_double) else 1) for a in args]
    out[name] = [int(getattr(lib, name)(*vals)), int(getattr(fast, name)(*vals))]
print(json.dumps(out))
"""


def test_generated_fast_binding_covers_the_abi_and_agrees_with_ctypes():
    """some_amd/_fastcall.so (generated from _lib.SYMBOLS against the header, some_amd/fastcall_gen.py): every entry point with a plain
    pointer / integer / float signature has a wrapper bound to the loaded library, the others fall through to ctypes; each wrapper,
    called with a NULL handle and dummy arguments (every entry point refuses a NULL handle or is a pure sizing function - no GPU work),
    returns what the ctypes call returns.  In a subprocess: a wrong wrapper would crash, not fail."""
    import json
    import subprocess
    import sys
    from some_amd import build
    build.build_fastcall()
    r = subprocess.run([sys.executable, '-c', _SWEEP, str(ROOT)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(got) >= 68
    for name, (a, b) in got.items():
        assert a == b, (name, a, b)
    assert got['some_train_gemm16s'] == [_lib.SOME_EINVAL, _lib.SOME_EINVAL] and got['some_train_scratch_bytes'][0] > 0


def test_fast_binding_argument_conversions():
    """Pointers as None / address / ctypes object, 64-bit seeds masked as ctypes masks them, floats from ints, and the errors a wrong call
    raises (arity, type) - the conversions the training operators rely on."""
    fast = _lib.fast()
    if not getattr(fast, 'fastcall', False):
        pytest.skip('SOME_AMD_FASTCALL=0')
    p = C.c_void_p(4096)
    args = [None, 0, p, 512, 4096, 512, None, C.c_void_p(0), 512, None, 0, 0, 4160, 512, 512, 2, 0.1, 2 ** 64 + 5, 1, p]
    assert fast.some_train_gemm16s(*args) == _lib.SOME_EINVAL                      # reaches the library: a NULL handle is refused
    assert fast.some_train_scratch_bytes(None, np.int64(4160), True + 511) == _lib.load().some_train_scratch_bytes(None, 4160, 512)
    with pytest.raises(TypeError, match='takes 3 arguments'):
        fast.some_train_scratch_bytes(None, 512)
    with pytest.raises(TypeError):
        fast.some_train_scratch_bytes(None, 'x', 512)
    with pytest.raises(TypeError, match='pointer argument'):
        fast.some_train_scratch_bytes(object(), 1, 512)
    with pytest.raises(TypeError):
        fast.some_train_gemm16s(*(args[:16] + ['0.1'] + args[17:]))
    with pytest.raises(OverflowError):
        fast.some_train_scratch_bytes(None, 2 ** 70, 512)


def test_sizing_calls_do_not_divide_by_zero_on_degenerate_shapes():
    """some_train_gemm_splitk_bytes(K < 32) raised SIGFPE before round 5 (a slice count of 0 k-blocks); the GEMM itself refuses such a K."""
    lib = _lib.load()
    assert lib.some_train_gemm_splitk_bytes(None, 1, 1, 1) >= 256
    assert lib.some_train_gemm16_bytes(None, 1, 1, 1, 4) >= 256
