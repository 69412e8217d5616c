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


def _gemm(eng, epi, A, W, bias=None, res=None, alpha=1.0, act=0, mask=None, n_out=None):
    from some_amd import _lib
    M, K = A.shape
    N = W.shape[0]
    n_out = N if n_out is None else n_out
    Cm = torch.full((M, n_out), float('nan'), device='cuda')
    _lib.check(eng.handle, eng.lib.some_op_gemm(
        eng.handle, epi, _p(A), A.stride(0), _p(W), _p(bias), _p(res), 0 if res is None else res.stride(0),
        _p(Cm), n_out, M, N, K, alpha, act, _p(mask), _stream()))
    torch.cuda.synchronize()
    return Cm


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
                                             _p(Xc), 512, M, 512, 2048, 0.5, 0, None, _stream()))
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
    _lib.check(eng.handle, eng.lib.some_op_layernorm(eng.handle, _p(x), _p(gamma), _p(beta), _p(y), M, _stream()))
    torch.cuda.synchronize()
    ref = F.layer_norm(x.double(), (512,), gamma.double(), beta.double(), eps=1e-5).float()
    assert (y - ref).abs().max().item() < 5e-6


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
                                                     batch.max_frames, _p(out), _stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
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
    _lib.check(eng.handle, eng.lib.some_op_attention(eng.handle, _p(qkv), _p(batch.frame_offsets_dev), 1, t, _p(out), _stream()))
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
                                                       batch.B, batch.max_frames, _p(y), _stream()))
    torch.cuda.synchronize()
    for b, t in enumerate(lens):
        s = int(batch.frame_offsets[b])
        ref = F.silu(F.conv1d(x[s:s + t].t()[None].double(), w.double(), bias.double(), padding=15, groups=512))[0].t().float()
        assert (y[s:s + t] - ref).abs().max().item() < 1e-5, (b, t)
