"""Kernel-level parity on a real MI355X: each HIP operator against a plain PyTorch fp32 reference of the same
op (the ATen sequence the reference model runs), through the C ABI (`some_op_*`)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from some_amd.configs import get_config
    from some_amd.engine import Engine
    return Engine(get_config('midi_conformer', lay=1), device='cuda')


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _split(eng, x):
    """fp32 [R, K] -> SPLIT32 bytes (same shape / dtype container)."""
    from some_amd import _lib
    out = torch.empty_like(x)
    _lib.check(eng.handle, eng.lib.some_op_split_rows(eng.handle, _p(x), _p(out), x.shape[0], x.shape[1], _stream()))
    return out


def _unsplit(t):
    """SPLIT32 container [R, K] -> fp32 values hi + lo (host-side decode for checking split outputs)."""
    r, k = t.shape
    h = t.contiguous().view(torch.float16).view(r, k // 32, 2, 32).float()
    return (h[:, :, 0] + h[:, :, 1]).reshape(r, k)


def _gemm(eng, epi, A, W, bias=None, res=None, alpha=1.0, act=0, mask=None, n_out=None, split=False, tile=0, out_split=False):
    from some_amd import _lib
    M, K = A.shape
    N = W.shape[0]
    n_out = N if n_out is None else n_out
    Cm = torch.full((M, n_out), float('nan'), device='cuda')
    flags = 0
    if split:
        A, W = _split(eng, A.contiguous()), _split(eng, W.contiguous())
        flags = _lib.GEMM_SPLIT_IN | (tile << 8) | (_lib.GEMM_SPLIT_OUT if out_split else 0)
    _lib.check(eng.handle, eng.lib.some_op_gemm(
        eng.handle, epi, _p(A), A.stride(0), _p(W), _p(bias), _p(res), 0 if res is None else res.stride(0),
        _p(Cm), n_out, M, N, K, alpha, act, _p(mask), flags, _stream()))
    torch.cuda.synchronize()
    return _unsplit(Cm) if out_split else Cm


def _interleave_glu(w):
    """row p of the packed matrix <- a-row / gate-row, 32 + 32 per 64 (api.hip copy_glu)."""
    d = w.shape[0] // 2
    p = torch.arange(2 * d, device=w.device)
    c64, wi = p // 64, p % 64
    src = torch.where(wi < 32, c64 * 32 + wi, d + c64 * 32 + (wi - 32))
    return w[src].contiguous()


def _ref_mm(A, W):
    return (A.double() @ W.double().t()).float()


@pytest.mark.parametrize('M,N,K', [(128, 128, 32), (300, 512, 512), (1, 512, 80), (257, 1536, 512), (130, 512, 2048), (77, 129, 512), (64, 1, 512)])
def test_gemm_plain_and_bias(eng, M, N, K):
    from some_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N)
    A = torch.randn(M, K, device='cuda', generator=g)
    # asymmetric, non-square data: catches transposed C/D layouts
    W = torch.randn(N, K, device='cuda', generator=g) * (1 + torch.arange(N, device='cuda')[:, None] % 5)
    b = torch.randn(N, device='cuda', generator=g)
    ref = _ref_mm(A, W)
    scale = ref.abs().max().item() + 1.0
    out = _gemm(eng, _lib.EPI_NONE, A, W)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 2e-6 * scale * max(1, K // 256)
    out = _gemm(eng, _lib.EPI_BIAS, A, W, bias=b)
    assert (out - (ref + b)).abs().max().item() < 2e-6 * scale * max(1, K // 256)
    out = _gemm(eng, _lib.EPI_BIAS, A, W, bias=b, act=1)
    # |d sigmoid| <= |d x| / 4
    assert (out - torch.sigmoid(ref + b)).abs().max().item() < 0.25 * 2e-6 * scale * max(1, K // 256) + 2e-6


def test_gemm_epilogues(eng):
    from some_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(5)
    M, K = 333, 512
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(2048, K, device='cuda', generator=g) / 20
    b = torch.randn(2048, device='cuda', generator=g)
    ref = F.silu(_ref_mm(A, W) + b)
    out = _gemm(eng, _lib.EPI_BIAS_SILU, A, W, bias=b)
    assert (out - ref).abs().max().item() < 2e-5
    # residual, alpha = 0.5, in place (C aliases res) as the FFN second GEMM runs it
    W2 = torch.randn(512, 2048, device='cuda', generator=g) / 40
    b2 = torch.randn(512, device='cuda', generator=g)
    X = torch.randn(M, 512, device='cuda', generator=g)
    want = X + 0.5 * (_ref_mm(out, W2) + b2)
    Xc = X.clone()
    from some_amd import _lib as L
    L.check(eng.handle, eng.lib.some_op_gemm(eng.handle, L.EPI_BIAS_RES, _p(out), 2048, _p(W2), _p(b2), _p(Xc), 512,
                                             _p(Xc), 512, M, 512, 2048, 0.5, 0, None, 0, _stream()))
    torch.cuda.synchronize()
    assert (Xc - want).abs().max().item() < 3e-5
    # GLU (conv pointwise 1) and GLU + residual + mask (cross gate)
    Wg = torch.randn(1024, K, device='cuda', generator=g) / 20
    bg = torch.randn(1024, device='cuda', generator=g)
    y = _ref_mm(A, Wg) + bg
    glu = y[:, :512] * torch.sigmoid(y[:, 512:])
    out = _gemm(eng, L.EPI_GLU, A, _interleave_glu(Wg), bias=_interleave_glu(bg[:, None])[:, 0].contiguous(), n_out=512)
    assert (out - glu).abs().max().item() < 2e-5
    R = torch.randn(M, 512, device='cuda', generator=g)
    mask = (torch.rand(M, device='cuda', generator=g) > 0.2).to(torch.uint8)
    out = _gemm(eng, L.EPI_GLU_RES, A, _interleave_glu(Wg), bias=_interleave_glu(bg[:, None])[:, 0].contiguous(), res=R, mask=mask, n_out=512)
    want = (R + glu) * mask[:, None]
    assert (out - want).abs().max().item() < 2e-5


@pytest.mark.parametrize('M', [1, 5, 1000])
def test_layernorm(eng, M):
    from some_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(M)
    x = torch.randn(M, 512, device='cuda', generator=g) * 3 + 1.5
    gamma = torch.randn(512, device='cuda', generator=g)
    beta = torch.randn(512, device='cuda', generator=g)
    y = torch.empty_like(x)
    ys = torch.empty_like(x)
    _lib.check(eng.handle, eng.lib.some_op_layernorm(eng.handle, _p(x), _p(gamma), _p(beta), _p(y), _p(ys), M, _stream()))
    torch.cuda.synchronize()
    ref = F.layer_norm(x.double(), (512,), gamma.double(), beta.double(), eps=1e-5).float()
    assert (y - ref).abs().max().item() < 5e-6
    # SPLIT32 twin: hi + lo reproduces the fp32 output to 2^-21 relative
    assert ((_unsplit(ys) - y).abs() <= y.abs() * 2.0 ** -20 + 1e-7).all()


@pytest.mark.parametrize('lens', [[64], [1], [33], [130, 257], [128, 1, 300, 65], [862]])
def test_attention(eng, lens):
    from some_amd import _lib
    from some_amd.engine import ClipBatch
    g = torch.Generator(device='cuda').manual_seed(sum(lens))
    batch = ClipBatch(lens, 'cuda')
    M = batch.total_frames
    qkv = torch.randn(M, 1536, device='cuda', generator=g)
    qkv[:, :512] *= 2.0        # sharper softmax
    out = torch.full((M, 512), float('nan'), device='cuda')
    _lib.check(eng.handle, eng.lib.some_op_attention(eng.handle, _p(qkv), _p(batch.frame_offsets_dev), batch.B,
                                                     batch.max_frames, _p(out), 0, _stream()))
    outs = torch.empty_like(out)
    _lib.check(eng.handle, eng.lib.some_op_attention(eng.handle, _p(qkv), _p(batch.frame_offsets_dev), batch.B,
                                                     batch.max_frames, _p(outs), 1, _stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert ((_unsplit(outs) - out).abs() <= out.abs() * 2.0 ** -20 + 1e-7).all()
    for b, t in enumerate(lens):
        s = int(batch.frame_offsets[b])
        x = qkv[s:s + t].double()
        q, k, v = (x[:, i * 512:(i + 1) * 512].reshape(t, 8, 64).transpose(0, 1) for i in range(3))
        p = torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1)
        ref = (p @ v).transpose(0, 1).reshape(t, 512).float()
        assert (out[s:s + t] - ref).abs().max().item() < 5e-6, (b, t)


def test_attention_online_softmax_rescale(eng):
    """Force the running max to jump at a late key tile (cdna guide rule 26: a rare data-dependent branch)."""
    from some_amd import _lib
    from some_amd.engine import ClipBatch
    t = 300
    g = torch.Generator(device='cuda').manual_seed(3)
    qkv = torch.randn(t, 1536, device='cuda', generator=g) * 0.5
    qkv[7, :64] = 3.0
    qkv[250, 512:576] = 4.0      # key 250 (4th tile) dominates query 7 of head 0
    batch = ClipBatch([t], 'cuda')
    out = torch.empty(t, 512, device='cuda')
    _lib.check(eng.handle, eng.lib.some_op_attention(eng.handle, _p(qkv), _p(batch.frame_offsets_dev), 1, t, _p(out), 0, _stream()))
    torch.cuda.synchronize()
    x = qkv.double()
    q, k, v = (x[:, i * 512:(i + 1) * 512].reshape(t, 8, 64).transpose(0, 1) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1) @ v).transpose(0, 1).reshape(t, 512).float()
    assert (out - ref).abs().max().item() < 5e-6


@pytest.mark.parametrize('lens', [[1], [10], [31, 200], [128, 129, 7]])
def test_dwconv_silu(eng, lens):
    from some_amd import _lib
    from some_amd.engine import ClipBatch
    g = torch.Generator(device='cuda').manual_seed(11 + sum(lens))
    batch = ClipBatch(lens, 'cuda')
    M = batch.total_frames
    x = torch.randn(M, 512, device='cuda', generator=g)
    w = torch.randn(512, 1, 31, device='cuda', generator=g) / 5
    bias = torch.randn(512, device='cuda', generator=g)
    taps = w[:, 0, :].t().contiguous()          # [31, 512]
    y = torch.full((M, 512), float('nan'), device='cuda')
    _lib.check(eng.handle, eng.lib.some_op_dwconv_silu(eng.handle, _p(x), _p(taps), _p(bias), _p(batch.frame_offsets_dev),
                                                       batch.B, batch.max_frames, _p(y), 0, _stream()))
    ys = torch.empty_like(y)
    _lib.check(eng.handle, eng.lib.some_op_dwconv_silu(eng.handle, _p(x), _p(taps), _p(bias), _p(batch.frame_offsets_dev),
                                                       batch.B, batch.max_frames, _p(ys), 1, _stream()))
    torch.cuda.synchronize()
    assert ((_unsplit(ys) - y).abs() <= y.abs() * 2.0 ** -20 + 1e-7).all()
    for b, t in enumerate(lens):
        s = int(batch.frame_offsets[b])
        ref = F.silu(F.conv1d(x[s:s + t].t()[None].double(), w.double(), bias.double(), padding=15, groups=512))[0].t().float()
        assert (y[s:s + t] - ref).abs().max().item() < 1e-5, (b, t)


# ---- 3-term split-f16 GEMM (gemm_f16x3.hip) ----------------------------------------------------------
@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize('M,N,K', [(128, 128, 32), (300, 512, 512), (257, 1536, 512), (130, 512, 2048), (77, 129, 512), (64, 1, 512), (515, 2048, 512)])
def test_gemm_f16x3_matches_fp64(eng, M, N, K, tile):
    from some_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(M * 3 + N + tile)
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(N, K, device='cuda', generator=g) * (1 + torch.arange(N, device='cuda')[:, None] % 5) / 10
    b = torch.randn(N, device='cuda', generator=g)
    ref = _ref_mm(A, W)
    scale = ref.abs().max().item() + 1.0
    out = _gemm(eng, _lib.EPI_NONE, A, W, split=True, tile=tile)
    assert torch.isfinite(out).all()
    err = (out - ref).abs().max().item()
    assert err < 4e-6 * scale * max(1, K // 256), err            # fp32-class accuracy from f16 MFMAs
    out = _gemm(eng, _lib.EPI_BIAS, A, W, bias=b, split=True, tile=tile)
    assert (out - (ref + b)).abs().max().item() < 4e-6 * scale * max(1, K // 256)


def test_gemm_f16x3_small_magnitudes_need_f16_subnormals(eng):
    """|x| ~ 1e-3: the lo halves are f16 subnormals; a pipe that flushed them would lose ~11 bits."""
    from some_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(9)
    A = torch.randn(200, 512, device='cuda', generator=g) * 1e-3
    W = torch.randn(256, 512, device='cuda', generator=g) * 1e-2
    ref = _ref_mm(A, W)
    out = _gemm(eng, _lib.EPI_NONE, A, W, split=True, tile=0)
    rel = ((out - ref).abs().max() / ref.abs().max()).item()
    assert rel < 2e-5, rel


@pytest.mark.parametrize('tile', [0, 2, 3, 4, 5])
def test_gemm_f16x3_epilogues(eng, tile):
    from some_amd import _lib as L
    g = torch.Generator(device='cuda').manual_seed(50 + tile)
    M, K = 333, 512
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(2048, K, device='cuda', generator=g) / 20
    b = torch.randn(2048, device='cuda', generator=g)
    ref = F.silu(_ref_mm(A, W) + b)
    out = _gemm(eng, L.EPI_BIAS_SILU, A, W, bias=b, split=True, tile=tile)
    assert (out - ref).abs().max().item() < 2e-5
    out_s = _gemm(eng, L.EPI_BIAS_SILU, A, W, bias=b, split=True, tile=tile, out_split=True)     # SPLIT32 output
    assert (out_s - ref).abs().max().item() < 2e-5
    W2 = torch.randn(512, 2048, device='cuda', generator=g) / 40
    b2 = torch.randn(512, device='cuda', generator=g)
    X = torch.randn(M, 512, device='cuda', generator=g)
    want = X + 0.5 * (_ref_mm(out, W2) + b2)
    got = _gemm(eng, L.EPI_BIAS_RES, out, W2, bias=b2, res=X, alpha=0.5, split=True, tile=tile)
    assert (got - want).abs().max().item() < 3e-5
    Wg = torch.randn(1024, K, device='cuda', generator=g) / 20
    bg = torch.randn(1024, device='cuda', generator=g)
    y = _ref_mm(A, Wg) + bg
    glu = y[:, :512] * torch.sigmoid(y[:, 512:])
    bgi = _interleave_glu(bg[:, None])[:, 0].contiguous()
    out = _gemm(eng, L.EPI_GLU, A, _interleave_glu(Wg), bias=bgi, n_out=512, split=True, tile=tile)
    assert (out - glu).abs().max().item() < 2e-5
    R = torch.randn(M, 512, device='cuda', generator=g)
    mask = (torch.rand(M, device='cuda', generator=g) > 0.2).to(torch.uint8)
    out = _gemm(eng, L.EPI_GLU_RES, A, _interleave_glu(Wg), bias=bgi, res=R, mask=mask, n_out=512, split=True, tile=tile)
    assert (out - (R + glu) * mask[:, None]).abs().max().item() < 2e-5


def test_gemm_single_stage_tile_equals_the_double_buffered_tile_bit_for_bit(eng):
    """Tile 5 (round 6: 128 x 256, ONE LDS stage, two independent workgroups per CU) keeps the 256 x 256 kernel's wave tile, fragment
    schedule and product order per output element - so every epilogue's output is the big tile's, bit for bit, on ragged shapes too
    (the packing-invariance gates of the model rest on all tile variants agreeing)."""
    from some_amd import _lib as L
    g = torch.Generator(device='cuda').manual_seed(77)
    M, K = 1000, 512
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(2048, K, device='cuda', generator=g) / 20
    b = torch.randn(2048, device='cuda', generator=g)
    X = torch.randn(M, 512, device='cuda', generator=g)
    W2 = torch.randn(512, 2048, device='cuda', generator=g) / 40
    b2 = torch.randn(512, device='cuda', generator=g)
    Wg = _interleave_glu(torch.randn(1024, K, device='cuda', generator=g) / 20)
    bg = torch.randn(1024, device='cuda', generator=g)
    mask = (torch.rand(M, device='cuda', generator=g) > 0.2).to(torch.uint8)
    outs = {}
    for tile in (2, 5):
        h = _gemm(eng, L.EPI_BIAS_SILU, A, W, bias=b, split=True, tile=tile)
        outs[tile] = [h, _gemm(eng, L.EPI_BIAS_SILU, A, W, bias=b, split=True, tile=tile, out_split=True),
                      _gemm(eng, L.EPI_BIAS_RES, h, W2, bias=b2, res=X, alpha=0.5, split=True, tile=tile),
                      _gemm(eng, L.EPI_NONE, A, W, split=True, tile=tile), _gemm(eng, L.EPI_BIAS, A, W, bias=b, split=True, tile=tile),
                      _gemm(eng, L.EPI_GLU, A, Wg, bias=bg, n_out=512, split=True, tile=tile),
                      _gemm(eng, L.EPI_GLU_RES, A, Wg, bias=bg, res=X, mask=mask, n_out=512, split=True, tile=tile)]
    for a, c in zip(outs[2], outs[5]):
        assert torch.equal(a, c)


def _engine_with_gemm_flags(flags, lay=1):
    """An Engine whose handle read SOME_AMD_GEMM_FLAGS = flags at creation (bit 0 row-per-lane epilogues, bit 1 the persistent stream
    kernel for the 256 x 256 launches)."""
    import os
    from some_amd.configs import get_config
    from some_amd.engine import Engine
    old = os.environ.get('SOME_AMD_GEMM_FLAGS')
    os.environ['SOME_AMD_GEMM_FLAGS'] = str(flags)
    try:
        return Engine(get_config('midi_conformer', lay=lay), device='cuda')
    finally:
        if old is None:
            del os.environ['SOME_AMD_GEMM_FLAGS']
        else:
            os.environ['SOME_AMD_GEMM_FLAGS'] = old


@pytest.mark.parametrize('M', [40000, 10007, 65536 + 300])
def test_persistent_stream_gemm_equals_the_one_tile_per_workgroup_kernel_bit_for_bit(M):
    """hgemm3p_kernel (round 6): one resident workgroup per CU walks its tiles with the k-blocks of consecutive tiles as ONE stream (the
    last two iterations of a tile load / store the first two k-blocks of the next).  Same products in the same order per output
    element: every epilogue's output equals the non-persistent 256 x 256 kernel's bit for bit - with several tiles per workgroup
    (M = 40 000: 320 virtual blocks for N = 512; 1 280 for N = 2048), ragged last row tiles, the wave-quantisation tail, K = 512 and
    K = 2048."""
    from some_amd import _lib as L
    base, pers, lines = _engine_with_gemm_flags(1), _engine_with_gemm_flags(3), _engine_with_gemm_flags(7)     # 7: + whole-line stores in FFN1's SPLIT32 epilogue
    g = torch.Generator(device='cuda').manual_seed(M)
    K = 512
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(2048, K, device='cuda', generator=g) / 20
    b = torch.randn(2048, device='cuda', generator=g)
    X = torch.randn(M, 512, device='cuda', generator=g)
    W2 = torch.randn(512, 2048, device='cuda', generator=g) / 40
    b2 = torch.randn(512, device='cuda', generator=g)
    Wg = _interleave_glu(torch.randn(1024, K, device='cuda', generator=g) / 20)
    bg = torch.randn(1024, device='cuda', generator=g)
    mask = (torch.rand(M, device='cuda', generator=g) > 0.2).to(torch.uint8)
    outs = []
    for e in (base, pers, lines):
        h = _gemm(e, L.EPI_BIAS_SILU, A, W, bias=b, split=True, tile=2)
        outs.append([h, _gemm(e, L.EPI_BIAS_SILU, A, W, bias=b, split=True, tile=2, out_split=True),
                     _gemm(e, L.EPI_BIAS_RES, h, W2, bias=b2, res=X, alpha=0.5, split=True, tile=2),
                     _gemm(e, L.EPI_NONE, A, W, split=True, tile=2), _gemm(e, L.EPI_BIAS, A, W2[:, :K].contiguous(), bias=b2, split=True, tile=2),
                     _gemm(e, L.EPI_GLU, A, Wg, bias=bg, n_out=512, split=True, tile=2),
                     _gemm(e, L.EPI_GLU_RES, A, Wg, bias=bg, res=X, mask=mask, n_out=512, split=True, tile=2)])
    for other in outs[1:]:
        for i, (a, c) in enumerate(zip(outs[0], other)):
            assert torch.isfinite(c).all(), i
            assert torch.equal(a, c), (i, float((a - c).abs().max()))


def test_forward_with_the_persistent_gemms_equals_the_default_bit_for_bit():
    """The whole forward (QKV projection with its row gather and V^T patch included) at a size where every 256 x 256 launch has several
    tiles per workgroup: 16 x 30 s + ragged clips, lay 2."""
    from some_amd import _lib, synth
    from some_amd.engine import ClipBatch
    engines = [_engine_with_gemm_flags(f, lay=2) for f in (1, 3)]
    sd = synth.synth_state_dict(engines[0].config, 11)
    counts = [2584] * 15 + [1000, 333, 77]
    gen = torch.Generator(device='cuda').manual_seed(5)
    res = []
    for e in engines:
        e.load_state_dict(sd)
        batch = ClipBatch(counts, 'cuda')
        units = torch.randn(batch.total_frames, 80, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)) * 2 - 4
        res.append(e.forward(units, batch, head_mode=_lib.HEAD_SIGMOID))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.isfinite(res[1][0]).all()


@pytest.mark.parametrize('lens', [[64], [1], [33], [130, 257], [128, 1, 300, 65], [862], [2584, 100], [2584] * 9 + [64, 200]])
def test_qkv_attention_f16x3(eng, lens):
    """Split-f16 QKV projection (Q | K SPLIT32 planes + transposed V) + split-f16 flash attention vs fp64."""
    from some_amd import _lib
    from some_amd.engine import ClipBatch
    g = torch.Generator(device='cuda').manual_seed(100 + sum(lens))
    batch = ClipBatch(lens, 'cuda')
    M = batch.total_frames
    h = torch.randn(M, 512, device='cuda', generator=g)
    W = torch.randn(1536, 512, device='cuda', generator=g) / 512 ** 0.5
    W[:512] *= 3.0                                           # sharper softmax
    out = torch.full((M, 512), float('nan'), device='cuda')
    ws = torch.empty(eng.lib.some_op_qkv_attention_f16x3_bytes(M, batch.B), dtype=torch.uint8, device='cuda')
    hs, Ws = _split(eng, h), _split(eng, W)          # keep alive: the library only borrows the pointers
    _lib.check(eng.handle, eng.lib.some_op_qkv_attention_f16x3(
        eng.handle, _p(hs), _p(Ws), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M,
        _p(out), _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    got = _unsplit(out)
    assert torch.isfinite(got).all()
    # bit-identical on a second run (round 2 found a K-ring race here: clips starting on a 64-frame boundary - clips 0 and
    # 8 of the last case - raced between QK(0)'s fragment reads and the first in-loop K store; rare, large, run-dependent)
    out2 = torch.full((M, 512), float('nan'), device='cuda')
    _lib.check(eng.handle, eng.lib.some_op_qkv_attention_f16x3(
        eng.handle, _p(hs), _p(Ws), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M,
        _p(out2), _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    qkv = h.double() @ W.double().t()
    for b, t in enumerate(lens):
        s = int(batch.frame_offsets[b])
        x = qkv[s:s + t]
        q, k, v = (x[:, i * 512:(i + 1) * 512].reshape(t, 8, 64).transpose(0, 1) for i in range(3))
        ref = (torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1) @ v).transpose(0, 1).reshape(t, 512).float()
        err = (got[s:s + t] - ref).abs().max().item()
        assert err < 1.2e-5, (b, t, err)            # measured 6e-6 .. 1.0e-5 over 2584 keys with the x3 sharpened scores


@pytest.mark.parametrize('fast', [1, 2])
@pytest.mark.parametrize('lens', [[64], [33], [130, 257], [862], [2584, 100]])
def test_qkv_attention_fast_mode_error_bound(lens, fast):
    """SOME_PRECISION_F16X3_FAST (opt-in): the attention product P V with ph = rn_f16(2^11 p) alone, normalised by the sum of the ROUNDED
    values.  Same sharpened-score inputs as test_qkv_attention_f16x3 (3 x the trained scale, up to 2584 keys), against fp64: the stated
    bound is 2^-12 of the largest |V| a query can pick up (relative rounding of one weight; the shipped three-term kernel sits at
    6e-6 .. 1e-5 here), measured ~1e-4.  fast = 2 (SOME_AMD_ATTN_FAST=2, Q K^T without kh * ql as well) is measured here to document
    why it is NOT offered as a mode: the scores themselves move by ~2^-12 |s| and the bound is 10 x looser.  Results are deterministic
    and a one-key softmax (one score far above the rest) returns that key's V row to the three-term accuracy."""
    import os
    from some_amd import _lib
    from some_amd.configs import get_config
    from some_amd.engine import ClipBatch, Engine
    old = os.environ.get('SOME_AMD_ATTN_FAST')
    os.environ['SOME_AMD_ATTN_FAST'] = str(fast)
    try:
        e = Engine(get_config('midi_conformer', lay=1, some_amd_precision='f16x3_fast'), device='cuda')
    finally:
        if old is None:
            del os.environ['SOME_AMD_ATTN_FAST']
        else:
            os.environ['SOME_AMD_ATTN_FAST'] = old
    g = torch.Generator(device='cuda').manual_seed(100 + sum(lens))
    batch = ClipBatch(lens, 'cuda')
    M = batch.total_frames
    h = torch.randn(M, 512, device='cuda', generator=g)
    W = torch.randn(1536, 512, device='cuda', generator=g) / 512 ** 0.5
    W[:512] *= 3.0
    hs, Ws = _split(e, h), _split(e, W)
    ws = torch.empty(e.lib.some_op_qkv_attention_f16x3_bytes(M, batch.B), dtype=torch.uint8, device='cuda')
    outs = []
    for _ in range(2):
        out = torch.full((M, 512), float('nan'), device='cuda')
        _lib.check(e.handle, e.lib.some_op_qkv_attention_f16x3(e.handle, _p(hs), _p(Ws), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M,
                                                               _p(out), _p(ws), ws.numel(), _stream()))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    got = _unsplit(outs[0])
    qkv = h.double() @ W.double().t()
    worst = 0.0
    for b, t in enumerate(lens):
        s0 = int(batch.frame_offsets[b])
        x = qkv[s0:s0 + t]
        q, k, v = (x[:, i * 512:(i + 1) * 512].reshape(t, 8, 64).transpose(0, 1) for i in range(3))
        ref = (torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1) @ v).transpose(0, 1).reshape(t, 512).float()
        scale = float(v.abs().max())
        err = (got[s0:s0 + t] - ref).abs().max().item()
        worst = max(worst, err / scale)
        assert err < (2.0 ** -12 if fast == 1 else 10 * 2.0 ** -12) * scale, (b, t, err, scale)
    print(f'fast = {fast}, lens {lens}: max error / max|V| = {worst:.2e} (bound {2.0 ** -12 if fast == 1 else 10 * 2.0 ** -12:.2e})')


def test_qkv_attention_f16x3_is_packing_independent(eng):
    """A clip's attention output is the same BITS alone, at any position of a packed batch and beside any neighbours: operand rows
    are clip-aligned (row gather in the QKV projection) and key tiles are counted from the clip's first frame.  Round 3 aligned the
    tiles in global packed coordinates: last-bit differences that moved note boundaries between 1-rank and N-rank jobs."""
    from some_amd import _lib
    from some_amd.engine import ClipBatch
    g = torch.Generator(device='cuda').manual_seed(4242)
    lens = [130, 257, 64, 1, 33, 862, 16, 2584]
    W = torch.randn(1536, 512, device='cuda', generator=g) / 512 ** 0.5
    W[:512] *= 3.0
    Ws = _split(eng, W)
    hs = [torch.randn(t, 512, device='cuda', generator=g) for t in lens]

    def run(order):
        batch = ClipBatch([lens[i] for i in order], 'cuda')
        M = batch.total_frames
        h_split = _split(eng, torch.cat([hs[i] for i in order]))
        out = torch.full((M, 512), float('nan'), device='cuda')
        ws = torch.empty(eng.lib.some_op_qkv_attention_f16x3_bytes(M, batch.B), dtype=torch.uint8, device='cuda')
        ws.fill_(0xFF)                                  # NaN bit patterns in every unwritten operand slot
        _lib.check(eng.handle, eng.lib.some_op_qkv_attention_f16x3(
            eng.handle, _p(h_split), _p(Ws), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M,
            _p(out), _p(ws), ws.numel(), _stream()))
        torch.cuda.synchronize()
        assert torch.isfinite(_unsplit(out)).all()
        return batch, out

    alone = [run([i])[1] for i in range(len(lens))]
    for order in (list(range(len(lens))), [7, 3, 5, 0, 6, 2, 4, 1], [2, 2, 5, 3, 3, 7, 7]):
        batch, out = run(order)
        for pos, i in enumerate(order):
            s = int(batch.frame_offsets[pos])
            assert torch.equal(out[s:s + lens[i]], alone[i]), (order, pos)


@pytest.mark.parametrize('tile', [0, 1, 2, 4, 5])
@pytest.mark.parametrize('epi', ['bias_res', 'bias_silu_split', 'glu_res'])
def test_gemm_f16x3_never_writes_past_row_m(eng, tile, epi):
    """Partial row tiles (M not a multiple of the tile height): the epilogues rely on the buffer descriptors' range check
    to drop rows >= M - make sure nothing lands behind the output (or the residual) buffer."""
    from some_amd import _lib as L
    g = torch.Generator(device='cuda').manual_seed(3)
    M, K = 300, 512
    guard = 64                                   # sentinel rows behind the M valid ones
    A = _split(eng, torch.randn(M, K, device='cuda', generator=g))
    if epi == 'glu_res':
        N, n_out = 1024, 512
    elif epi == 'bias_silu_split':
        N, n_out = 2048, 2048
    else:
        N, n_out = 512, 512
    W = _split(eng, (torch.randn(N, K, device='cuda', generator=g) / 20).contiguous())
    b = torch.randn(N, device='cuda', generator=g)
    big = torch.full((M + guard, n_out), 12345.0, device='cuda')
    Cm = big[:M]
    res = torch.randn(M, n_out, device='cuda', generator=g)
    code = {'bias_res': L.EPI_BIAS_RES, 'bias_silu_split': L.EPI_BIAS_SILU, 'glu_res': L.EPI_GLU_RES}[epi]
    flags = L.GEMM_SPLIT_IN | (tile << 8) | (L.GEMM_SPLIT_OUT if epi == 'bias_silu_split' else 0)
    L.check(eng.handle, eng.lib.some_op_gemm(eng.handle, code, _p(A), K, _p(W), _p(b), _p(res) if 'res' in epi else None, n_out,
                                             _p(Cm), n_out, M, N, K, 1.0, 0, None, flags, _stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(Cm).all()
    assert (big[M:] == 12345.0).all()
