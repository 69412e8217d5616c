"""The 16-bit-storage training kernels (csrc/train_gemm16s.hip, include/some_amd.h "16-bit STORED operands") against PyTorch restatements
of the same arithmetic: fp64 products of the stored operands, 16-bit roundings where the kernels round, the kernels' own dropout mask
(recovered from the forward output) in the backward check."""
import ctypes as C

import numpy as np
import pytest
import torch

from some_amd import _lib
from some_amd.configs import get_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from some_amd.engine import Engine
    from some_amd.training.ops import TrainOps
    return TrainOps(Engine(get_config('two_head_model', lay=1), device='cuda'))


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return torch.randn(*shape, device='cuda', generator=g) * scale


def _silu(x):
    return x * torch.sigmoid(x)


def _dsilu(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


@pytest.mark.parametrize('operand', ['bf16', 'f16'])
@pytest.mark.parametrize('M,N,K', [(700, 2048, 512), (129, 96, 64), (64, 512, 2048)])
def test_gemm16s_plain_matches_products_of_the_stored_operands(ops, operand, M, N, K):
    ops.set_mixed_precision(True, operand)
    try:
        x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
        x16 = ops.cast16(x)
        w16, w16t = ops.shadow16(w)
        assert torch.equal(x16, x.to(ops.dtype16)) and torch.equal(w16, w.to(ops.dtype16)) and torch.equal(w16t, w.to(ops.dtype16).t().contiguous())
        y = ops.new(M, N)
        ops.gemm16s(0, x16, w16, b, y, N, M, N, K)
        want = x16.double() @ w16.double().t() + b.double()
        assert float((y.double() - want).abs().max()) < 2e-6 * float(want.abs().max())
        dy16 = ops.cast16(_rand(M, N, seed=4))
        dx = ops.new(M, K)
        ops.gemm16s(0, dy16, w16t, None, dx, K, M, K, N)                 # data gradient: the transposed image is the [K, N] operand
        want = dy16.double() @ w16.double()
        assert float((dx.double() - want).abs().max()) < 2e-6 * float(want.abs().max())
        dw, db = ops.new(N, K), ops.new(N)
        ops.wgrad16(dy16, x16, dw, db, accumulate=False)
        want = dy16.double().t() @ x16.double()
        assert float((dw.double() - want).abs().max()) < 2e-6 * float(want.abs().max())
        assert float((db.double() - dy16.double().sum(0)).abs().max()) < 2e-6 * float(dy16.double().sum(0).abs().max()) + 1e-6
        ops.wgrad16(dy16, x16, dw, db, accumulate=True)                  # (+)=
        assert float((dw.double() - 2 * want).abs().max()) < 4e-6 * float(want.abs().max())
    finally:
        ops.set_mixed_precision(False)


@pytest.mark.parametrize('M', [129, 700, 777])
@pytest.mark.parametrize('epi', [0, 3])
def test_gemm16s_partial_row_tiles_stay_inside_the_output(ops, epi, M):
    """Canary rows behind the M valid ones: the fp32 (epilogue 0) and residual + dropout (epilogue 3) epilogues of a partial
    128-row tile must neither write rows >= M nor read the residual there - M % 128 != 0 is nearly every training batch."""
    ops.set_mixed_precision(True, 'bf16')
    try:
        N, K, guard = 512, 512, 160
        x16 = ops.cast16(_rand(M, K, seed=11))
        w16, _ = ops.shadow16(_rand(N, K, seed=12, scale=K ** -0.5))
        b = _rand(N, seed=13)
        big = torch.full((M + guard, N), 12345.0, device='cuda')
        res_big = torch.full((M + guard, N), float('nan'), device='cuda')        # a read past row M would poison nothing visible, but must not fault
        res_big[:M] = _rand(M, N, seed=14)
        if epi == 0:
            ops.gemm16s(0, x16, w16, b, big[:M], N, M, N, K)
        else:
            ops.gemm16s(3, x16, w16, b, big[:M], N, M, N, K, h16=res_big[:M], p=0.1, seed=77, alpha=0.5)
        torch.cuda.synchronize()
        assert torch.isfinite(big[:M]).all()
        assert (big[M:] == 12345.0).all()
    finally:
        ops.set_mixed_precision(False)


@pytest.mark.parametrize('operand', ['bf16', 'f16'])
@pytest.mark.parametrize('M,p', [(700, 0.0), (700, 0.1), (131, 0.5)])
def test_ffn16_forward_and_backward_match_the_restated_arithmetic(ops, operand, M, p):
    K, H, N = 512, 2048, 512
    dt = torch.bfloat16 if operand == 'bf16' else torch.float16
    ops.set_mixed_precision(True, operand)
    try:
        x = _rand(M, K, seed=5).requires_grad_()
        w1, b1 = _rand(H, K, seed=6, scale=K ** -0.5).requires_grad_(), _rand(H, seed=7, scale=0.3).requires_grad_()
        w2, b2 = _rand(N, H, seed=8, scale=H ** -0.5).requires_grad_(), _rand(N, seed=9, scale=0.3).requires_grad_()
        ops.weights_version += 1
        y = ops.ffn(x, w1, b1, w2, b2, p, seed=1234567)
        assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith('_Ffn16')
        x16, ha = y.grad_fn.saved_tensors
        h16, a16 = ha[0], ha[1]
        # forward, stage by stage
        hw = x.detach().to(dt).double() @ w1.detach().to(dt).double().t() + b1.detach().double()
        assert torch.equal(x16, x.detach().to(dt))
        err_h = (h16.double() - hw).abs().max() / hw.abs().max()
        assert float(err_h) < (5e-3 if operand == 'bf16' else 6e-4)                   # one 16-bit rounding of h
        kept = a16 != 0
        act = _silu(h16.float())
        unscaled = act.abs() > 1e-3                                                   # where a zero can only mean "dropped"
        rate = 1.0 - float((kept & unscaled).sum()) / float(unscaled.sum())
        thr = round(p * 65536)
        keep = 65536.0 / (65536.0 - thr)
        assert abs(rate - p) < 4 * (p * (1 - p) / float(unscaled.sum())) ** 0.5 + 1e-9
        want_a = (act * keep).to(dt)
        assert torch.equal(a16[kept], want_a[kept]) or float((a16[kept].float() - want_a[kept].float()).abs().max()) <= 2 * float(torch.finfo(dt).eps) * float(want_a.float().abs().max())
        yw = a16.double() @ w2.detach().to(dt).double().t() + b2.detach().double()
        assert float((y.detach().double() - yw).abs().max()) < 2e-6 * float(yw.abs().max())
        # backward with the kernel's own mask
        dy = _rand(M, N, seed=10)
        y.backward(dy)
        mask = torch.where(a16 != 0, keep, 0.0).double()
        mask = torch.where(unscaled, mask, torch.full_like(mask, keep) if p == 0 else mask)
        dy16 = dy.to(dt).double()
        dh = (dy16 @ w2.detach().to(dt).double()) * mask * _dsilu(h16.double())
        dh16 = dh.to(dt).double()
        sure = unscaled | (p == 0)                                                    # cells whose mask bit is known
        dxw = dh16 @ w1.detach().to(dt).double()
        tol = 3e-2 if p > 0 else 2e-2                                                 # |silu| < 1e-3 cells: mask unknown, gradient tiny
        assert float((x.grad.double() - dxw).abs().max()) < tol * float(dxw.abs().max())
        assert float((w2.grad.double() - dy16.t() @ a16.double()).abs().max()) < 2e-6 * float((dy16.t() @ a16.double()).abs().max())
        assert float((b2.grad.double() - dy16.sum(0)).abs().max()) < 2e-6 * float(dy16.sum(0).abs().max())
        dw1 = dh16.t() @ x16.double()
        assert float((w1.grad.double() - dw1).abs().max()) < tol * float(dw1.abs().max())
        assert float((b1.grad.double() - dh16.sum(0)).abs().max()) < tol * float(dh16.sum(0).abs().max())
        assert bool(sure.any())
    finally:
        ops.set_mixed_precision(False)


def test_ffn16_equals_the_fp32_composition_within_16_bit_roundings(ops):
    """Same weights, same input, dropout off: the 16-bit-intermediate FFN against linear / silu / linear on fp32 arrays in the same
    mixed-precision mode (which already rounds every GEMM operand to bf16)."""
    M, K, H, N = 1000, 512, 2048, 512
    ops.set_mixed_precision(True, 'bf16')
    try:
        x = _rand(M, K, seed=11)
        w1, b1, w2, b2 = _rand(H, K, seed=12, scale=K ** -0.5), _rand(H, seed=13, scale=0.3), _rand(N, H, seed=14, scale=H ** -0.5), _rand(N, seed=15)
        outs = []
        for on in (True, False):
            ops.ffn16 = on
            ops.weights_version += 1
            leaves = [t.clone().requires_grad_() for t in (x, w1, b1, w2, b2)]
            y = ops.ffn(*leaves, 0.0, 0)
            y.backward(_rand(M, N, seed=16))
            outs.append([y.detach()] + [t.grad for t in leaves])
        for a, b in zip(*outs):
            assert float((a - b).abs().max()) < 2e-2 * float(b.abs().max())
        assert float((outs[0][0] - outs[1][0]).abs().max()) > 0                       # and it really is another path
    finally:
        ops.ffn16 = True
        ops.set_mixed_precision(False)


def test_dropout_bits_match_the_host_restatement(ops):
    """The mask of the FFN epilogue is the documented pure function of (seed, row, column) that some_amd/training/dropout_bits.py restates
    in numpy; here the kernel's zeros are compared with it cell by cell."""
    from some_amd.training.dropout_bits import ffn_keep_mask
    M, K, H, N, p, seed = 300, 512, 2048, 512, 0.25, 987654321
    ops.set_mixed_precision(True, 'bf16')
    try:
        x = _rand(M, K, seed=17).requires_grad_()
        w1, b1, w2, b2 = _rand(H, K, seed=18, scale=K ** -0.5), _rand(H, seed=19, scale=0.3) + 2.0, _rand(N, H, seed=20, scale=H ** -0.5), _rand(N, seed=21)
        ops.weights_version += 1
        y = ops.ffn(x, w1, b1, w2, b2, p, seed)
        _, ha = y.grad_fn.saved_tensors
        act = _silu(ha[0].float()).cpu().numpy()
        kept = (ha[1] != 0).cpu().numpy()
        want = ffn_keep_mask(seed, M, H, p)
        clear = np.abs(act) > 1e-3
        assert np.array_equal(kept[clear], want[clear])
    finally:
        ops.set_mixed_precision(False)


@pytest.mark.parametrize('operand', ['bf16', 'f16'])
def test_ffn_block_equals_the_unfused_composition(ops, operand):
    """x + 0.5 * ffn(LayerNorm(x)) as ONE operator (LayerNorm writing the 16-bit GEMM operand, residual in FFN2's epilogue, the residual
    gradient summed inside LayerNorm-backward) against layernorm -> ffn -> axpy on the same kernels otherwise; latent dropout on (both
    paths draw the same mask), output dropout off."""
    M, K, H = 777, 512, 2048
    ops.set_mixed_precision(True, operand)
    try:
        x = _rand(M, K, seed=31)
        g, b = _rand(K, seed=32, scale=0.1) + 1.0, _rand(K, seed=33, scale=0.1)
        w1, b1, w2, b2 = _rand(H, K, seed=34, scale=K ** -0.5), _rand(H, seed=35, scale=0.3), _rand(K, H, seed=36, scale=H ** -0.5), _rand(K, seed=37)
        d = _rand(M, K, seed=38)
        outs = []
        for fused in (True, False):
            ops.weights_version += 1
            leaves = [t.clone().requires_grad_() for t in (x, g, b, w1, b1, w2, b2)]
            if fused:
                y = ops.ffn_block(*leaves, 0.5, 0.1, 4242, 0.0, 0)
                assert type(y.grad_fn).__name__.startswith('_FfnBlock16')
            else:
                xx, gg, bb = leaves[:3]
                y = ops.axpy_dropout(0.5, ops.ffn(ops.layernorm(xx, gg, bb), *leaves[3:], 0.1, 4242), xx, 0.0, 0)
            y.backward(d)
            outs.append([y.detach()] + [t.grad for t in leaves])
        names = ['y', 'dx', 'dgamma', 'dbeta', 'dw1', 'db1', 'dw2', 'db2']
        for n, a, c in zip(names, *outs):
            assert float((a - c).abs().max()) <= 2e-6 * float(c.abs().max()), n
    finally:
        ops.set_mixed_precision(False)


def test_ffn_block_output_dropout_and_its_gradient(ops):
    """Output dropout folded into FFN2's epilogue: dropped cells keep exactly the residual, the rate is the requested one, the mask is the
    documented function of (seed, row, column), and the backward pass applies the SAME mask (checked on the bias gradient of FFN2, the
    column sums of rn16(0.5 * mask / (1 - p) * d))."""
    from some_amd.training.dropout_bits import ffn_keep_mask, threshold
    M, K, H, p, seed = 500, 512, 2048, 0.3, 99
    ops.set_mixed_precision(True, 'bf16')
    try:
        x = _rand(M, K, seed=41)
        g, b = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
        w1, b1, w2, b2 = _rand(H, K, seed=44, scale=K ** -0.5), _rand(H, seed=45, scale=0.3), _rand(K, H, seed=46, scale=H ** -0.5), _rand(K, seed=47) + 3.0
        ops.weights_version += 1
        leaves = [t.clone().requires_grad_() for t in (x, g, b, w1, b1, w2, b2)]
        y = ops.ffn_block(*leaves, 0.5, 0.0, 0, p, seed)
        kept = (y.detach() != x).cpu().numpy()                      # b2 = 3 + noise: a kept cell always moves the residual
        want = ffn_keep_mask(seed, M, K, p)
        assert np.array_equal(kept, want)
        assert abs(kept.mean() - (1 - threshold(p) / 65536.0)) < 4 * (p * (1 - p) / kept.size) ** 0.5
        d = _rand(M, K, seed=48)
        y.backward(d)
        keep = 65536.0 / (65536.0 - threshold(p))
        dy16 = (torch.from_numpy(want).cuda() * (0.5 * keep) * d).bfloat16().double()
        assert float((leaves[6].grad.double() - dy16.sum(0)).abs().max()) < 2e-6 * float(dy16.sum(0).abs().max())
        # the residual branch reaches dx unmasked: dx - d is the LayerNorm gradient, zero-mean over each row (gamma = 1)
        assert float((leaves[0].grad - d).sum(1).abs().max()) < 1e-3 * float((leaves[0].grad - d).abs().sum(1).max())
    finally:
        ops.set_mixed_precision(False)


def _flat_params(ops, shapes, seed):
    """Parameters laid out back to back in ONE buffer with a gradient buffer beside it, registered as gradient sinks (what FlatParams gives
    the trainer): returns (dict of leaf views, flat gradient)."""
    n = sum(int(np.prod(s)) for _, s, _ in shapes)
    flat = torch.empty(n, device='cuda')
    grad = torch.zeros(n, device='cuda')
    views, off = {}, 0
    for i, (name, shape, scale) in enumerate(shapes):
        k = int(np.prod(shape))
        flat[off:off + k] = _rand(k, seed=seed + i, scale=scale) + (1.0 if name.endswith('gamma') else 0.0)
        v = flat[off:off + k].view(shape).requires_grad_()
        v.grad = grad[off:off + k].view(shape)
        views[name] = v
        off += k
    ops.register_grad_sinks(views.values())
    return views, grad


def _clone_leaves(views):
    return {k: v.detach().clone().requires_grad_() for k, v in views.items()}


def test_attention_block_equals_the_unfused_composition(ops):
    from some_amd.engine import ClipBatch
    lens = [300, 257, 64]
    batch, M = ClipBatch(lens, 'cuda'), sum(lens)
    ops.set_mixed_precision(True, 'bf16')
    try:
        P, grad = _flat_params(ops, [('gamma', (512,), 0.1), ('beta', (512,), 0.1), ('wq', (512, 512), 512 ** -0.5), ('wkv', (1024, 512), 512 ** -0.5),
                                     ('wo', (512, 512), 512 ** -0.5), ('bo', (512,), 0.3)], seed=60)
        x, d = _rand(M, 512, seed=70), _rand(M, 512, seed=71)
        ops.weights_version += 1
        xf = x.clone().requires_grad_()
        y = ops.attention_block(xf, P['gamma'], P['beta'], P['wq'], P['wkv'], P['wo'], P['bo'], batch, 0.0, 0)
        assert type(y.grad_fn).__name__.startswith('_AttnBlock16')
        y.backward(d)
        ops.register_grad_sinks([])                                   # no sinks: attention_block falls back to the composition
        Q = _clone_leaves(P)
        xu = x.clone().requires_grad_()
        yu = ops.attention_block(xu, Q['gamma'], Q['beta'], Q['wq'], Q['wkv'], Q['wo'], Q['bo'], batch, 0.0, 0)
        assert not type(yu.grad_fn).__name__.startswith('_AttnBlock16')
        yu.backward(d)
        assert float((y.detach() - yu.detach()).abs().max()) <= 1e-5 * float(yu.detach().abs().max())
        assert float((xf.grad - xu.grad).abs().max()) <= 1e-5 * float(xu.grad.abs().max())
        for k in P:
            # the fused path sums the STORED 16-bit gradient for a Linear's bias, the unfused one the fp32 gradient: 2^-9 per element
            tol = 5e-3 if k == 'bo' else 1e-5
            assert float((P[k].grad - Q[k].grad).abs().max()) <= tol * float(Q[k].grad.abs().max()), k
    finally:
        ops.register_grad_sinks([])
        ops.set_mixed_precision(False)


def test_conv_block_equals_the_unfused_composition(ops):
    from some_amd.engine import ClipBatch
    lens = [300, 257, 64]
    batch, M = ClipBatch(lens, 'cuda'), sum(lens)
    ops.set_mixed_precision(True, 'bf16')
    try:
        P, grad = _flat_params(ops, [('gamma', (512,), 0.1), ('beta', (512,), 0.1), ('pw1_w', (1024, 512, 1), 512 ** -0.5), ('pw1_b', (1024,), 0.3),
                                     ('dw_w', (512, 1, 31), 0.2), ('dw_b', (512,), 0.1), ('bn_gamma', (512,), 0.1), ('bn_b', (512,), 0.1),
                                     ('pw2_w', (512, 512, 1), 512 ** -0.5), ('pw2_b', (512,), 0.3)], seed=80)
        x, d = _rand(M, 512, seed=90), _rand(M, 512, seed=91)
        order = ('gamma', 'beta', 'pw1_w', 'pw1_b', 'dw_w', 'dw_b', 'bn_gamma', 'bn_b')
        outs = []
        for fused in (True, False):
            if not fused:
                ops.register_grad_sinks([])
            R = P if fused else _clone_leaves(P)
            rm, rv = torch.zeros(512, device='cuda'), torch.ones(512, device='cuda')
            ops.weights_version += 1
            xx = x.clone().requires_grad_()
            y = ops.conv_block(xx, *[R[k] for k in order], rm, rv, R['pw2_w'], R['pw2_b'], batch, 0.0, 0)
            assert type(y.grad_fn).__name__.startswith('_ConvBlock16') == fused
            y.backward(d)
            outs.append((y.detach(), xx.grad, {k: R[k].grad.clone() for k in R}, rm, rv))
        (y0, dx0, g0, rm0, rv0), (y1, dx1, g1, rm1, rv1) = outs
        assert float((y0 - y1).abs().max()) <= 1e-5 * float(y1.abs().max())
        assert float((dx0 - dx1).abs().max()) <= 1e-5 * float(dx1.abs().max())
        assert torch.equal(rm0, rm1) and torch.equal(rv0, rv1)
        for k in g0:
            tol = 5e-3 if k in ('pw1_b', 'pw2_b') else 2e-5         # bias gradients: sums of the stored 16-bit gradient vs of the fp32 one
            assert float((g0[k] - g1[k]).abs().max()) <= tol * float(g1[k].abs().max()) + 1e-7, k
    finally:
        ops.register_grad_sinks([])
        ops.set_mixed_precision(False)


@pytest.mark.parametrize('operand', ['bf16', 'f16'])
def test_weight_images_of_many_weights_in_one_launch(ops, operand):
    """some_train_transpose16_table: W16 / W16T of weights of different shapes (a k = 1 Conv1d weight among them) from one launch equal
    the fp32 -> 16-bit cast and its transpose; ops.shadow16 then serves them without another launch until weights_version moves."""
    ops.set_mixed_precision(True, operand)
    try:
        ws = [_rand(2048, 512, seed=101), _rand(512, 2048, seed=102), _rand(1024, 512, 1, seed=103), _rand(96, 64, seed=104), _rand(1536, 512, seed=105)]
        ops.weights_version += 1
        ops.prepare_shadows(ws)
        for w in ws:
            w2 = w.reshape(w.shape[0], -1)
            hit = ops._shadows[id(w)]
            assert hit[0][0] == ops.weights_version
            a, t = ops.shadow16(w)
            assert a is hit[1] and t is hit[2]
            assert torch.equal(a, w2.to(ops.dtype16)) and torch.equal(t, w2.to(ops.dtype16).t().contiguous())
        ws[3].mul_(2.0)
        ops.weights_version += 1
        ops.prepare_shadows(ws)                                        # same table, refreshed contents
        assert torch.equal(ops.shadow16(ws[3])[0], ws[3].to(ops.dtype16))
    finally:
        ops.set_mixed_precision(False)


@pytest.mark.parametrize('epi', [0, 3])
def test_gemm16s_tiles_are_bit_identical(ops, epi, monkeypatch):
    """The 64 x 128 tile (small grids, round 5) against the 128 x 256 tile on the same operands: every output element accumulates its
    k-steps in the same order, so the results must be the same BITS (SOME_AMD_G16S_TILE forces the tile per call)."""
    ops.set_mixed_precision(True, 'bf16')
    gen = torch.Generator(device='cuda').manual_seed(9)
    M, N, K = 1000, 512, 2048
    a = torch.randn(M, K, device='cuda', generator=gen).to(ops.dtype16)
    w = torch.randn(N, K, device='cuda', generator=gen).to(ops.dtype16) / 30
    b = torch.randn(N, device='cuda', generator=gen)
    res = torch.randn(M, N, device='cuda', generator=gen)
    outs = []
    for tile in ('0', '1'):
        monkeypatch.setenv('SOME_AMD_G16S_TILE', tile)
        out = torch.full((M, N), float('nan'), device='cuda')
        if epi == 0:
            ops.gemm16s(0, a, w, b, out, N, M, N, K)
        else:
            ops.gemm16s(3, a, w, b, out, N, M, N, K, h16=res, p=0.1, seed=77, alpha=0.5)
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('operand', ['bf16', 'f16'])
@pytest.mark.parametrize('M,N', [(1000, 1536), (4160, 512), (33, 512)])
def test_split_transpose_equals_split_rows_and_transpose(ops, operand, M, N):
    """One pass over x writes the bits the two separate passes write (row-major SPLIT32 and transposed SPLIT32 with zero padding)."""
    ops.set_mixed_precision(True, operand)
    try:
        x = _rand(M, N, seed=5)
        rows, t = ops.split_transpose(x, pad_to=64)
        assert torch.equal(rows.view(torch.int32), ops.split_rows(x).view(torch.int32))
        assert torch.equal(t.view(torch.int32), ops.transpose(x, pad_to=64, split=True).view(torch.int32))
    finally:
        ops.set_mixed_precision(False)


@pytest.mark.parametrize('operand', ['bf16', 'f16'])
@pytest.mark.parametrize('magnitude', [1.0, 3e-6, 2.0 ** -7, 0.0])
def test_attention_backward_with_the_device_side_factor_equals_the_host_formula(ops, operand, magnitude):
    """some_train_attention_bwd_f16x3_auto16 (factor, split layouts, row sums and 1 / factor made inside the call) against the same kernels
    driven by torch's exp2(floor(10 - log2(max|dO|))): identical 16-bit gradients.  2^-7: max|dO| an exact power of two."""
    from some_amd.engine import ClipBatch
    from some_amd.training.ops import _p
    lens = [300, 257, 64]
    batch, M = ClipBatch(lens, 'cuda'), sum(lens)
    ops.set_mixed_precision(True, operand)
    try:
        qkv = _rand(M, 1536, seed=11)
        R, Rt = ops.split_transpose(qkv, pad_to=64)
        Mp = Rt.shape[1]
        out, lse = ops.new(M, 512), ops.new(8, M)
        ops.check(ops.lib.some_train_attention_fwd_f16x3(ops.h, _p(R), _p(Rt), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M, Mp, ops._hi_mode,
                                                         _p(out), _p(lse), ops.stream()))
        dout = _rand(M, 512, seed=12).clamp_(-4, 4) / 4 * magnitude
        if magnitude:
            dout[5, 7] = magnitude                                   # the largest element, exactly
        got = torch.empty((M, 1536), dtype=ops.dtype16, device='cuda')
        work = torch.empty(int(ops.lib.some_train_attention_bwd16_work_bytes(ops.h, M, Mp)), dtype=torch.uint8, device='cuda')
        ops.check(ops.lib.some_train_attention_bwd_f16x3_auto16(ops.h, _p(R), _p(Rt), _p(out), _p(dout), _p(lse), _p(batch.frame_offsets_dev), batch.B,
                                                                batch.max_frames, M, Mp, ops._hi_mode, _p(got), _p(work), work.numel(), ops.stream()))
        scale = torch.exp2(torch.floor(10.0 - torch.log2(dout.abs().amax().clamp_min(1e-30))))
        ds = dout * scale
        inv = torch.reciprocal(scale).reshape(1)
        D, Dt = ops.split_rows(ds), ops.transpose(ds, pad_to=64, split=True)
        want, dsum = torch.empty_like(got), ops.new(8, M)
        ops.check(ops.lib.some_train_attention_bwd_f16x3_out16(ops.h, _p(R), _p(Rt), _p(D), _p(Dt), _p(out), _p(ds), _p(lse), _p(batch.frame_offsets_dev),
                                                               batch.B, batch.max_frames, M, Mp, ops._hi_mode, _p(want), _p(inv), _p(dsum), ops.stream()))
        torch.cuda.synchronize()
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
        if magnitude:
            assert float(got.float().abs().max()) > 0
    finally:
        ops.set_mixed_precision(False)


def test_attention_backward_poisons_the_gradient_when_dO_is_not_finite(ops):
    from some_amd.engine import ClipBatch
    from some_amd.training.ops import _p
    batch, M = ClipBatch([96], 'cuda'), 96
    ops.set_mixed_precision(True, 'bf16')
    try:
        qkv = _rand(M, 1536, seed=11)
        R, Rt = ops.split_transpose(qkv, pad_to=64)
        Mp = Rt.shape[1]
        out, lse = ops.new(M, 512), ops.new(8, M)
        ops.check(ops.lib.some_train_attention_fwd_f16x3(ops.h, _p(R), _p(Rt), _p(batch.frame_offsets_dev), 1, M, M, Mp, ops._hi_mode, _p(out), _p(lse),
                                                         ops.stream()))
        dout = _rand(M, 512, seed=12)
        dout[3, 3] = float('inf')
        got = torch.empty((M, 1536), dtype=ops.dtype16, device='cuda')
        work = torch.empty(int(ops.lib.some_train_attention_bwd16_work_bytes(ops.h, M, Mp)), dtype=torch.uint8, device='cuda')
        ops.check(ops.lib.some_train_attention_bwd_f16x3_auto16(ops.h, _p(R), _p(Rt), _p(out), _p(dout), _p(lse), _p(batch.frame_offsets_dev), 1, M, M, Mp,
                                                                ops._hi_mode, _p(got), _p(work), work.numel(), ops.stream()))
        assert not torch.isfinite(got.float()).all()                  # the trainer's non-finite check sees it (task.py)
    finally:
        ops.set_mixed_precision(False)


def test_ffn_block_library_calls_equal_the_call_by_call_operator_bit_for_bit():
    """some_train_ffn_block_fwd / _bwd (one library call per direction for the FFN sub-block; round 5) are entry points of the C ABI the
    shipped trainer no longer takes (no measured gain: profiles/r05aa_train_ab.txt; the caller is kept in tools/patches/r06_ffn_block_calls.patch).
    They stay pinned here: same launches in the same order as _FfnBlock16's call-by-call body - output, dx and all six parameter
    gradients (into preallocated gradient arrays) bit for bit, dropout on."""
    import ctypes as C
    from some_amd.configs import get_config
    from some_amd.engine import Engine
    from some_amd.training.ops import TrainOps, _FfnBlock16, _Ctx
    ops = TrainOps(Engine(get_config('two_head_model', lay=1), device='cuda'))
    ops.set_mixed_precision(True, 'bf16')
    g = torch.Generator(device='cuda').manual_seed(12)
    M, K, H = 700, 512, 2048
    x = torch.randn(M, K, device='cuda', generator=g)
    gamma, beta = torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g) * 0.1
    w1, b1 = torch.randn(H, K, device='cuda', generator=g) / 22, torch.randn(H, device='cuda', generator=g) * 0.1
    w2, b2 = torch.randn(K, H, device='cuda', generator=g) / 45, torch.randn(K, device='cuda', generator=g) * 0.1
    d = torch.randn(M, K, device='cuda', generator=g)
    params = [gamma, beta, w1, b1, w2, b2]
    for t in params:
        t.grad = torch.zeros_like(t)
    ops.register_grad_sinks(params)
    alpha, p_lat, s_lat, p_out, s_out = 0.5, 0.1, 1234, 0.1, 5678
    # the operator's own (call-by-call) body
    c = _Ctx()
    c.needs_input_grad = (False, True, True, True, True, True, True, True, False, False, False, False, False)
    ops.pin_stream()
    try:
        want = _FfnBlock16.forward(c, ops, x, gamma, beta, w1, b1, w2, b2, alpha, p_lat, s_lat, p_out, s_out)
        dx_want = _FfnBlock16.backward(c, d)[1]
        torch.cuda.synchronize()
        grads_want = [t.grad.clone() for t in params]
        for t in params:
            t.grad.zero_()
        # the two library calls
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        lib, h = ops.lib, ops.h
        save = torch.empty(int(lib.some_train_ffn_block_save_bytes(h, M, K, H)), dtype=torch.uint8, device='cuda')
        out = torch.empty(M, K, device='cuda')
        ops.check(lib.some_train_ffn_block_fwd(h, p(x), p(gamma), p(beta), p(ops.shadow16(w1)[0]), p(b1), p(ops.shadow16(w2)[0]), p(b2), M, K, H, K,
                                               ops._hi_mode, alpha, p_lat, s_lat, p_out, s_out, p(save), save.numel(), p(out), ops.stream()))
        dx = torch.empty_like(x)
        scr = torch.empty(int(lib.some_train_ffn_block_scratch_bytes(h, M, K, H, K)), dtype=torch.uint8, device='cuda')
        sc = ops.scratch(M, 512)
        n1, n2 = ops._bytes('some_train_gemm16_bytes', H, K, M, K + 4), ops._bytes('some_train_gemm16_bytes', K, H, M, H + 4)
        part = torch.empty(max(n1, n2), dtype=torch.uint8, device='cuda')
        ops.check(lib.some_train_ffn_block_bwd(h, p(d), p(x), p(gamma), p(save), p(ops.shadow16(w1)[1]), p(ops.shadow16(w2)[1]), M, K, H, K, ops._hi_mode,
                                               alpha, p_lat, s_lat, p_out, s_out, p(w1.grad), p(b1.grad), p(w2.grad), p(b2.grad), p(gamma.grad), p(beta.grad),
                                               1, p(dx), p(scr), scr.numel(), p(sc), sc.numel(), part.data_ptr(), part.numel(), ops.stream()))
        torch.cuda.synchronize()
    finally:
        ops.unpin_stream()
    assert torch.equal(out, want) and torch.equal(dx, dx_want)
    for t, gw in zip(params, grads_want):
        assert torch.equal(t.grad, gw) and float(gw.abs().sum()) > 0
