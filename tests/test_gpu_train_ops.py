"""Training operators (forward AND backward) against plain PyTorch fp32 references of the same ops with autograd, on the
GPU.  Tolerances: fp32 kernels vs fp32 torch, different summation orders -> 2e-5 of the
tensor's scale unless stated.  Needs a real MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from some_amd.configs import get_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from some_amd.engine import Engine
    from some_amd.training.ops import TrainOps
    return TrainOps(Engine(get_config('two_head_model', lay=0), device='cuda'))


def _close(a, b, tol=2e-5):
    a, b = a.detach().double(), b.detach().double()
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item()
    assert err <= tol * scale, (err, scale)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return (torch.randn(*shape, device='cuda', generator=g) * scale).requires_grad_(True)


def _pair(fn_mine, fn_ref, inputs, tol=2e-5, seed=99):
    """Run both on clones of ``inputs``; compare outputs and the gradients of a random cotangent."""
    a = [t.detach().clone().requires_grad_(t.requires_grad) if torch.is_tensor(t) and t.is_floating_point() else t for t in inputs]
    b = [t.detach().clone().requires_grad_(t.requires_grad) if torch.is_tensor(t) and t.is_floating_point() else t for t in inputs]
    ya, yb = fn_mine(*a), fn_ref(*b)
    _close(ya, yb, tol)
    g = torch.Generator(device='cuda').manual_seed(seed)
    cot = torch.randn(yb.shape, device='cuda', generator=g)
    ya.backward(cot)
    yb.backward(cot)
    for ta, tb in zip(a, b):
        if torch.is_tensor(ta) and ta.requires_grad:
            assert ta.grad is not None
            _close(ta.grad, tb.grad, tol)


@pytest.mark.parametrize('M,K,N,bias', [(300, 512, 2048, True), (70, 80, 512, True), (129, 512, 1, True), (33, 2048, 512, True),
                                         (257, 512, 1536, False), (1, 512, 128, True),
                                         (300, 512, 130, True), (300, 512, 129, True)])      # N % 4 != 0 (e.g. 130 / 129 bins): transpose + split-K weight-gradient path
def test_linear(ops, M, K, N, bias):
    x, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    b = _rand(N, seed=3) if bias else None
    if bias:
        _pair(lambda x, w, b: ops.linear(x, w, b), lambda x, w, b: F.linear(x, w, b), [x, w, b], tol=3e-5)
    else:
        _pair(lambda x, w: ops.linear(x, w), lambda x, w: F.linear(x, w), [x, w], tol=3e-5)


def test_linear_conv1d_weight_shape(ops):
    x, w, b = _rand(50, 512, seed=1), _rand(1024, 512, 1, seed=2, scale=0.05), _rand(1024, seed=3)
    _pair(lambda x, w, b: ops.linear(x, w, b), lambda x, w, b: F.conv1d(x.t()[None], w, b)[0].t(), [x, w, b], tol=3e-5)


@pytest.mark.parametrize('M', [1, 5, 1000])
def test_layernorm(ops, M):
    x, g, b = _rand(M, 512, seed=4, scale=3.0), _rand(512, seed=5), _rand(512, seed=6)
    with torch.no_grad():
        x += 1.5
    _pair(lambda x, g, b: ops.layernorm(x, g, b), lambda x, g, b: F.layer_norm(x, (512,), g, b, 1e-5), [x, g, b])


def test_elementwise(ops):
    x = _rand(77, 640, seed=7, scale=2.0)
    _pair(lambda x: ops.silu(x), lambda x: F.silu(x), [x])
    _pair(lambda x: ops.sigmoid(x), lambda x: torch.sigmoid(x), [x])
    _pair(lambda x: ops.glu(x), lambda x: F.glu(x, dim=-1), [x])
    y = _rand(77, 640, seed=8)
    _pair(lambda y, x: ops.axpy(0.5, y, x), lambda y, x: y * 0.5 + x, [y, x])
    _pair(lambda y, x: ops.axpy(1.0, y, x), lambda y, x: y + x, [y, x])
    mask = (torch.arange(77, device='cuda') % 3 != 0)
    _pair(lambda x: ops.mask_rows(x, mask.to(torch.uint8)), lambda x: x.masked_fill(~mask[:, None], 0), [x])


def test_dropout(ops):
    x = _rand(400, 512, seed=9)
    y = ops.dropout(x, 0.1, seed=1234)
    keep = (y != 0)
    assert abs(keep.float().mean().item() - 0.9) < 0.01
    torch.testing.assert_close(y[keep], (x / 0.9)[keep])
    assert torch.equal(ops.dropout(x, 0.1, seed=1234), y) and not torch.equal(ops.dropout(x, 0.1, seed=1235), y)
    y.backward(torch.ones_like(y))
    torch.testing.assert_close(x.grad, keep.float() / 0.9)
    assert ops.dropout(x, 0.0, seed=1) is x
    # the fused passes are the same arithmetic as their compositions - forward and backward, same (seed, index) mask
    a = x.detach().clone().requires_grad_(True)
    b = x.detach().clone().requires_grad_(True)
    res = _rand(400, 512, seed=10).detach()
    f1 = ops.axpy_dropout(0.5, ops.silu_dropout(a, 0.1, 77), res, 0.2, 78)
    f2 = ops.axpy(0.5, ops.dropout(ops.dropout(ops.silu(b), 0.1, 77), 0.2, 78), res)
    assert torch.equal(f1, f2)
    cot = torch.randn_like(f1)
    f1.backward(cot)
    f2.backward(cot)
    assert torch.equal(a.grad, b.grad)


def test_dwconv(ops):
    from some_amd.engine import ClipBatch
    lens = [40, 1, 17, 300]
    batch = ClipBatch(lens, 'cuda')
    x, w, b = _rand(sum(lens), 512, seed=10), _rand(512, 1, 31, seed=11, scale=0.2), _rand(512, seed=12)

    def ref(x, w, b):
        outs, pos = [], 0
        for t in lens:
            outs.append(F.conv1d(x[pos:pos + t].t()[None], w, b, padding=15, groups=512)[0].t())
            pos += t
        return torch.cat(outs)

    _pair(lambda x, w, b: ops.dwconv(x, w, b, batch), ref, [x, w, b])


def test_batchnorm_train(ops):
    M = 1300
    x, g, b = _rand(M, 512, seed=13, scale=2.0), _rand(512, seed=14), _rand(512, seed=15)
    with torch.no_grad():
        x += 0.7
    rm1, rv1 = torch.zeros(512, device='cuda'), torch.ones(512, device='cuda')
    rm2, rv2 = rm1.clone(), rv1.clone()
    _pair(lambda x, g, b: ops.batchnorm(x, g, b, rm1, rv1, 0.1, 1e-5),
          lambda x, g, b: F.batch_norm(x, rm2, rv2, g, b, True, 0.1, 1e-5), [x, g, b])
    _close(rm1, rm2)
    _close(rv1, rv2)


def test_bce_with_logits(ops):
    x = _rand(200, 128, seed=16, scale=4.0)
    t = torch.rand(200, 128, device='cuda')
    _pair(lambda x: ops.bce_with_logits(x, t), lambda x: F.binary_cross_entropy_with_logits(x, t), [x], tol=1e-5)


@pytest.mark.parametrize('B,T', [(1, 5), (3, 1000), (8, 2500)])
def test_binary_emd(ops, B, T):
    pred = torch.sigmoid(_rand(B, T, seed=17).detach() - 2.0).requires_grad_(True)
    gt = (torch.rand(B, T, device='cuda') < 0.05).float()

    def ref(p):                                                 # modules/losses/bound_loss.py:12-16
        scale = T ** 0.5
        return F.l1_loss(p.cumsum(dim=1) / scale, gt.cumsum(dim=1) / scale)

    _pair(lambda p: ops.binary_emd(p.reshape(-1), gt.reshape(-1), B, T), ref, [pred], tol=2e-5)


def test_adamw_matches_torch(ops):
    import ctypes as C
    n = 10007
    p0 = torch.randn(n, device='cuda')
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p_ref], lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for step in range(1, 6):
        g = torch.randn(n, device='cuda', generator=torch.Generator(device='cuda').manual_seed(step))
        p_ref.grad = g.clone()
        opt.step()
        ops.check(ops.lib.some_train_adamw(ops.h, C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                           C.c_void_p(v.data_ptr()), n, 1e-3, 0.9, 0.98, 1e-8, 0.01, step, 1.0, ops.stream()))
        _close(p, p_ref, tol=2e-6)


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('lens', [[64], [1], [33, 130], [257, 5, 128], [700], [31, 1, 2, 64, 95]])
def test_attention_fwd_bwd(ops, lens, precision):
    """Flash attention forward (with saved log-sum-exp) and its two-kernel backward - split-f16 and exact-f32 MFMA
    variants - vs torch SDPA autograd."""
    from some_amd.engine import ClipBatch
    ops.attention_precision = precision
    batch = ClipBatch(lens, 'cuda')
    M = sum(lens)
    qkv = _rand(M, 1536, seed=20 + M)
    with torch.no_grad():
        qkv[:, :512] *= 2.0                                     # sharper softmax

    def ref(qkv):
        outs, pos = [], 0
        for t in lens:
            x = qkv[pos:pos + t]
            q, k, v = (x[:, i * 512:(i + 1) * 512].reshape(t, 8, 64).transpose(0, 1) for i in range(3))
            p = torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1)
            outs.append((p @ v).transpose(0, 1).reshape(t, 512))
            pos += t
        return torch.cat(outs)

    try:
        _pair(lambda qkv: ops.attention(qkv, batch), ref, [qkv], tol=3e-5)
    finally:
        ops.attention_precision = ops.gemm_precision


# ---- some_train_gemm16: fp32 operands rounded (and transposed) in the staging path ---------------------------------------
def _round16(t, operand):
    if operand == 'f16x3':                 # hi + lo carries 22 bits: the fp32 value to 2^-22
        return t.double()
    return (t.bfloat16() if operand == 'bf16' else t.half()).double()


_OP16 = {'f16': 1, 'bf16': 2, 'f16x3': 3}


@pytest.mark.parametrize('operand', ['bf16', 'f16', 'f16x3'])
@pytest.mark.parametrize('M,K,N,bias', [(300, 512, 2048, True), (128, 32, 256, False), (129, 2048, 512, True), (2584, 512, 1536, False),
                                         (64, 512, 132, True), (1000, 64, 64, True)])
def test_gemm16_forward_layout(ops, operand, M, K, N, bias):
    """(ta, tb) = (0, 0): x [M, K] @ w [N, K]^T + b equals the fp64 product of the 16-bit roundings (products of 16-bit values are
    exact in fp32, only the accumulation order differs) - and is NOT the fp32 product."""
    from some_amd.training.ops import _p
    x, w = _rand(M, K, seed=11).detach() * 3, _rand(N, K, seed=12).detach()
    b = _rand(N, seed=13).detach() if bias else None
    out = torch.full((M, N), float('nan'), device='cuda')
    ops.check(ops.lib.some_train_gemm16(ops.h, _p(x), K, 0, _p(w), K, 0, _p(b), _p(out), N, M, N, K, _OP16[operand], -1, None, 0,
                                        ops.stream()))
    want = _round16(x, operand) @ _round16(w, operand).t() + (b.double() if bias else 0)
    scale = float(want.abs().max())
    assert float((out.double() - want).abs().max()) < 2e-6 * scale * max(1.0, (K / 512) ** 0.5)
    if K >= 512 and operand != 'f16x3':
        assert float((out.double() - (x.double() @ w.double().t() + (b.double() if bias else 0))).abs().max()) > 1e-4 * scale


@pytest.mark.parametrize('operand', ['bf16', 'f16', 'f16x3'])
@pytest.mark.parametrize('M,N,K', [(300, 2048, 512), (129, 512, 2048), (2584, 1536, 512), (70, 64, 132), (257, 128, 516)])
def test_gemm16_data_gradient_layout(ops, operand, M, N, K):
    """(0, 1): dx [M, K] = dy [M, N] @ w [N, K] with w read as it lies (its rows are the contraction index)."""
    from some_amd.training.ops import _p
    dy, w = _rand(M, N, seed=21).detach(), _rand(N, K, seed=22).detach() * 2
    out = torch.full((M, K), float('nan'), device='cuda')
    ops.check(ops.lib.some_train_gemm16(ops.h, _p(dy), N, 0, _p(w), K, 1, None, _p(out), K, M, K, N, _OP16[operand], -1, None, 0,
                                        ops.stream()))
    want = _round16(dy, operand) @ _round16(w, operand)
    assert float((out.double() - want).abs().max()) < 2e-6 * float(want.abs().max()) * max(1.0, (N / 512) ** 0.5)


@pytest.mark.parametrize('operand', ['bf16', 'f16', 'f16x3'])
@pytest.mark.parametrize('M,N,K,bias', [(300, 2048, 512, True), (2584 * 2 + 7, 512, 2048, True), (1001, 1536, 512, False), (95, 32, 80, True),
                                         (20672, 132, 36, True), (31, 512, 512, True)])
def test_gemm16_weight_gradient_layout(ops, operand, M, N, K, bias):
    """(1, 1): dW [N, K] = dy [M, N]^T @ x [M, K] over ALL M frames (any M: no padding copies), split across workgroups and
    summed in slice order; column K of the output = the fp32 column sums of dy (bias gradient), from the unrounded values."""
    from some_amd.training.ops import _p
    dy, x = _rand(M, N, seed=31).detach(), _rand(M, K, seed=32).detach() + 0.3
    ldc = K + (4 if bias else 0)
    out = torch.full((N, ldc), float('nan'), device='cuda')
    need = int(ops.lib.some_train_gemm16_bytes(ops.h, N, K, M, ldc))
    part = torch.empty(need, dtype=torch.uint8, device='cuda')
    args = (ops.h, _p(dy), N, 1, _p(x), K, 1, None, _p(out), ldc, N, K, M, _OP16[operand], K if bias else -1, _p(part), need, ops.stream())
    ops.check(ops.lib.some_train_gemm16(*args))
    want = _round16(dy, operand).t() @ _round16(x, operand)
    tol = 3e-6 * max(1.0, (M / 512) ** 0.5)
    assert float((out[:, :K].double() - want).abs().max()) < tol * float(want.abs().max())
    if bias:
        db = dy.double().sum(0)
        assert float((out[:, K].double() - db).abs().max()) < tol * float(db.abs().max() + dy.abs().max())
    first = out.clone()
    ops.check(ops.lib.some_train_gemm16(*args))                        # deterministic: no atomics anywhere
    assert torch.equal(first[:, :K + (1 if bias else 0)], out[:, :K + (1 if bias else 0)])


def test_gemm16_argument_checks(ops):
    from some_amd.training.ops import _p
    a, b, c = torch.zeros(64, 64, device='cuda'), torch.zeros(64, 64, device='cuda'), torch.zeros(64, 64, device='cuda')
    bad = [dict(ta=1, tb=0), dict(operand=0), dict(operand=4), dict(K=48), dict(lda=66), dict(sum_col=64)]
    for kw in bad:
        d = dict(lda=64, ta=0, ldb=64, tb=0, ldc=64, M=64, N=64, K=64, operand=2, sum_col=-1)
        d.update(kw)
        rc = ops.lib.some_train_gemm16(ops.h, _p(a), d['lda'], d['ta'], _p(b), d['ldb'], d['tb'], None, _p(c), d['ldc'], d['M'], d['N'], d['K'],
                                       d['operand'], d['sum_col'], None, 0, ops.stream())
        assert rc != 0, kw


@pytest.mark.parametrize('operand', ['bf16', 'f16', None])
def test_mixed_linear_matches_the_split_kernels(ops, operand):
    """ops.linear in the one-product modes: the fused-staging GEMMs (default) against the split_rows / transpose + SPLIT32 kernels
    they replace - same operand roundings, so outputs and all three gradients agree to accumulation-order noise."""
    x, w, b = _rand(700, 512, seed=41), _rand(2048, 512, seed=42, scale=512 ** -0.5), _rand(2048, seed=43)
    res = {}
    try:
        if operand:
            ops.set_mixed_precision(True, operand)
        for g16 in (True, False):
            ops.gemm16 = g16
            xa, wa, ba = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
            y = ops.linear(xa, wa, ba)
            y.backward(torch.ones_like(y) * 0.01 + y.detach() * 1e-3)
            res[g16] = (y.detach(), xa.grad, wa.grad, ba.grad)
    finally:
        ops.gemm16 = True
        ops.set_mixed_precision(False)
    for p, q in zip(res[True], res[False]):
        _close(p, q, {'f16': 5e-5, 'bf16': 3e-4, None: 5e-6}[operand])     # (bias gradient: fp32 sums here, 16-bit-rounded dy through the ones row there)


@pytest.mark.parametrize('operand', [None, 'bf16'])
def test_gradient_sinks_accumulate_in_place(ops, operand):
    """Parameters registered as gradient sinks (the trainer's views of the flat gradient buffer) get their gradients written by
    the backward kernels themselves - nn.Linear weight / bias through some_train_gemm16_wgrad, LayerNorm gamma / beta through
    the accumulate flag of the column reductions: backward returns None for them (autograd launches nothing), two micro-batches add up
    in the preallocated arrays, and the ready callback fires once per parameter and backward pass.  Checked against plain PyTorch
    (fp32-equivalent mode) and against the same operators without sinks (autograd accumulation) in both modes."""
    def make():
        w, b = _rand(256, 512, seed=1, scale=512 ** -0.5), _rand(256, seed=2)
        g, be = (_rand(512, seed=3).detach() * 0.1 + 1).requires_grad_(True), _rand(512, seed=4)
        return [w, b, g, be]

    def run(params):
        w, b, g, be = params
        for step in range(2):
            x = _rand(300, 512, seed=10 + step).detach()
            cot = _rand(300, 256, seed=20 + step).detach()
            ops.linear(ops.layernorm(x, g, be), w, b).backward(cot)

    sunk, plain, ref = make(), make(), make()
    arrays = []
    for p in sunk:
        p.grad = torch.zeros_like(p)
        arrays.append(p.grad.data_ptr())
    ready = []
    try:
        if operand:
            ops.set_mixed_precision(True, operand)
        ops.register_grad_sinks(sunk, ready.append)
        run(sunk)
        ops.register_grad_sinks([], None)
        run(plain)                                                     # same kernels for dx, autograd accumulation for the parameters
    finally:
        ops.set_mixed_precision(False)
        ops.register_grad_sinks([], None)
    assert [id(p) for p in ready].count(id(sunk[0])) == 2 and len(ready) == 8
    for p, q, ptr in zip(sunk, plain, arrays):
        assert p.grad.data_ptr() == ptr                                # still the preallocated array: nothing was swapped in
        _close(p.grad, q.grad, 1e-5 if not operand else 2e-3)          # (bias gradient: fp32 sums vs the 16-bit ones-row)
    if not operand:
        for step in range(2):
            x = _rand(300, 512, seed=10 + step).detach()
            cot = _rand(300, 256, seed=20 + step).detach()
            F.linear(F.layer_norm(x, (512,), ref[2], ref[3]), ref[0], ref[1]).backward(cot)
        for p, r in zip(sunk, ref):
            _close(p.grad, r.grad, 3e-5)
