"""Fused 16-bit operators of the mixed-precision training path (split out of ops.py in round 6): the FFN with its [M, 2048] intermediates
in 16-bit storage (_Ffn16) and the three sub-blocks of a conformer block as one operator each (_FfnBlock16, _AttnBlock16, _ConvBlock16:
``x + alpha * dropout(f(LayerNorm(x)))``, Gconform.py:57-60) - DESIGN.md section 6b, "the FFN in 16-bit storage" / "the three sub-blocks".
Every operator here is a composition of library calls through ``TrainOps``; the unfused compositions in ops.py are what the tests compare
them with (tests/test_gpu_train_ffn16.py).

Round 5's block-level library calls for the FFN sub-block (one C call per direction, some_train_ffn_block_fwd / _bwd) measured no gain
(profiles/r05aa_train_ab.txt) and are no longer a path of the trainer: tools/patches/r06_ffn_block_calls.patch holds the caller."""
import torch

from . import ops as _ops
from .ops import TrainOps, _Ctx, _p

class _Ffn16(torch.autograd.Function):
    """The FFN of a conformer block in mixed precision with 16-bit intermediates: x [M, K] fp32 -> y [M, N] fp32.
    forward:  x16 = rn16(x);  (h16 | a16) = epilogue(x16 W1_16^T + b1) [h16 = rn16(.), a16 = rn16(dropout(silu(h16)))];  y = a16 W2_16^T + b2
    backward: dy16 = rn16(dy);  dh16 = rn16((dy16 W2_16) * mask / (1 - p) * silu'(h16));  dx = dh16 W1_16;
              dW2 += dy16^T a16, db2 += 1^T dy16, dW1 += dh16^T x16, db1 += 1^T dh16
    - what nn.Linear / SiLU / Dropout compute under the reference's bf16 / fp16 autocast (16-bit linear outputs, fp32 accumulation)."""

    @staticmethod
    def forward(ctx, ops: TrainOps, x, w1, b1, w2, b2, p, seed):
        M, K = x.shape
        H, N = w1.shape[0], w2.shape[0]
        x16 = ops.cast16(x.contiguous())
        w1_16, _ = ops.shadow16(w1)
        w2_16, _ = ops.shadow16(w2)
        ha = torch.empty((2, M, H), dtype=ops.dtype16, device=ops.device)            # h16 plane, a16 plane
        ops.gemm16s(1, x16, w1_16, b1, ha, H, M, H, K, plane=M * H, p=p, seed=seed)
        y = ops.new(M, N)
        ops.gemm16s(0, ha[1], w2_16, b2, y, N, M, N, H)
        ctx.ops, ctx.p, ctx.seed = ops, p, seed
        ctx.save_for_backward(x16, ha)
        ctx.params = (w1, b1, w2, b2)                                                # identities: shadows and gradient sinks
        return y

    @staticmethod
    def backward(ctx, dy):
        ops: TrainOps = ctx.ops
        x16, ha = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        M, K = x16.shape
        H, N = w1.shape[0], w2.shape[0]
        dy16 = ops.cast16(dy.contiguous())
        dh16 = torch.empty((M, H), dtype=ops.dtype16, device=ops.device)
        ops.gemm16s(2, dy16, ops.shadow16(w2)[1], None, dh16, H, M, H, N, h16=ha[0], p=ctx.p, seed=ctx.seed)
        dx = None
        if ctx.needs_input_grad[1]:
            dx = ops.new(M, K)
            ops.gemm16s(0, dh16, ops.shadow16(w1)[1], None, dx, K, M, K, H)
        grads = []
        for (w, b, g16, in16, iw, ib) in ((w1, b1, dh16, x16, 2, 3), (w2, b2, dy16, ha[1], 4, 5)):
            dw = db = None
            want_w, want_b = ctx.needs_input_grad[iw], b is not None and ctx.needs_input_grad[ib]
            sw = ops.sink(w) if want_w else None
            sb = ops.sink(b) if want_b else None
            if want_w and sw is not None and (sb is not None or not want_b):
                ops.wgrad16(g16, in16, sw, sb, accumulate=True)                      # into the parameters' gradient arrays
                ops.deposited(w)
                if sb is not None:
                    ops.deposited(b)
            elif want_w or want_b:
                dw = ops.new(*w.shape)
                db = ops.new(w.shape[0]) if want_b else None
                ops.wgrad16(g16, in16, dw, db, accumulate=False)
                if not want_w:
                    dw = None
            grads += [dw, db]
        return None, dx, grads[0], grads[1], grads[2], grads[3], None, None


class _FfnBlock16(torch.autograd.Function):
    """x + alpha * dropout(ffn(LayerNorm(x))) in mixed precision, every intermediate written once and in 16 bits where a GEMM reads it:
    forward:  n16 = rn16(LayerNorm(x)) (one kernel, no cast pass);  (h16 | a16) = FFN1 epilogue;  out = x + alpha * dropout(a16 W2^T + b2)
              in FFN2's epilogue (no separate residual pass)
    backward: dy16 = rn16(alpha * mask / (1 - p) * d) (one kernel);  dh16, weight gradients as in _Ffn16;  dn = dh16 W1;
              dx = d + LayerNorm'(dn) with the addition inside the LayerNorm-backward kernel."""

    @staticmethod
    def forward(ctx, ops: TrainOps, x, gamma, beta, w1, b1, w2, b2, alpha, p_latent, seed_latent, p_out, seed_out):
        x = x.contiguous()
        M, K = x.shape
        H, N = w1.shape[0], w2.shape[0]
        ctx.ops = ops
        ctx.drop = (alpha, p_latent, seed_latent, p_out, seed_out)
        ctx.params = (gamma, beta, w1, b1, w2, b2)                                   # identities: shadows and gradient sinks
        n16 = torch.empty((M, K), dtype=ops.dtype16, device=ops.device)
        mean, rstd = ops.new(M), ops.new(M)
        ops.check(ops.lib.some_train_layernorm_fwd16(ops.h, _p(x), _p(gamma), _p(beta), _p(n16), _p(mean), _p(rstd), M, ops._hi_mode, ops.stream()))
        ha = torch.empty((2, M, H), dtype=ops.dtype16, device=ops.device)            # h16 plane, a16 plane
        ops.gemm16s(1, n16, ops.shadow16(w1)[0], b1, ha, H, M, H, K, plane=M * H, p=p_latent, seed=seed_latent)
        out = ops.new(M, N)
        ops.gemm16s(3, ha[1], ops.shadow16(w2)[0], b2, out, N, M, N, H, h16=x, p=p_out, seed=seed_out, alpha=alpha)
        ctx.save_for_backward(x, gamma, mean, rstd, n16, ha)
        return out

    @staticmethod
    def backward(ctx, d):
        ops: TrainOps = ctx.ops
        gamma, beta, w1, b1, w2, b2 = ctx.params
        alpha, p_latent, seed_latent, p_out, seed_out = ctx.drop
        d = d.contiguous()
        x, gamma_t, mean, rstd, n16, ha = ctx.saved_tensors
        M, K = x.shape
        H, N = w1.shape[0], w2.shape[0]
        dy16 = torch.empty((M, N), dtype=ops.dtype16, device=ops.device)
        ops.check(ops.lib.some_train_dropcast16(ops.h, _p(d), _p(dy16), M, N, float(alpha), float(p_out), seed_out, ops._hi_mode, ops.stream()))
        dh16 = torch.empty((M, H), dtype=ops.dtype16, device=ops.device)
        ops.gemm16s(2, dy16, ops.shadow16(w2)[1], None, dh16, H, M, H, N, h16=ha[0], p=p_latent, seed=seed_latent)
        dn = ops.new(M, K)
        ops.gemm16s(0, dh16, ops.shadow16(w1)[1], None, dn, K, M, K, H)
        grads = []
        for (w, b, g16, in16, iw, ib) in ((w1, b1, dh16, n16, 4, 5), (w2, b2, dy16, ha[1], 6, 7)):
            dw = db = None
            want_w, want_b = ctx.needs_input_grad[iw], b is not None and ctx.needs_input_grad[ib]
            sw = ops.sink(w) if want_w else None
            sb = ops.sink(b) if want_b else None
            if want_w and sw is not None and (sb is not None or not want_b):
                ops.wgrad16(g16, in16, sw, sb, accumulate=True)                      # into the parameters' gradient arrays
                ops.deposited(w)
                if sb is not None:
                    ops.deposited(b)
            elif want_w or want_b:
                dw = ops.new(*w.shape)
                db = ops.new(w.shape[0]) if want_b else None
                ops.wgrad16(g16, in16, dw, db, accumulate=False)
                if not want_w:
                    dw = None
            grads += [dw, db]
        # dx = d (the residual branch) + LayerNorm'(dn); gamma / beta gradients as in _ops._LayerNorm
        dx = torch.empty_like(x)
        sc = ops.scratch(M, 512)
        sg, sbeta = ops.sink(gamma), ops.sink(beta)
        add = d if ctx.needs_input_grad[1] else None
        if sg is not None and sbeta is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]:
            ops.check(ops.lib.some_train_layernorm_bwd_add(ops.h, _p(dn), _p(x), _p(gamma_t), _p(mean), _p(rstd), _p(add), _p(dx), _p(sg), _p(sbeta), 1, M,
                                                           _p(sc), sc.numel(), ops.stream()))
            ops.deposited(gamma)
            ops.deposited(beta)
            dg = dbeta = None
        else:
            dg, dbeta = torch.empty_like(gamma_t), torch.empty_like(gamma_t)
            ops.check(ops.lib.some_train_layernorm_bwd_add(ops.h, _p(dn), _p(x), _p(gamma_t), _p(mean), _p(rstd), _p(add), _p(dx), _p(dg), _p(dbeta), 0, M,
                                                           _p(sc), sc.numel(), ops.stream()))
        return None, dx, dg, dbeta, grads[0], grads[1], grads[2], grads[3], None, None, None, None, None


def _sub(fn, *args):
    """Run the body of another operator inside a fused one: (output, its context for the backward body)."""
    c = _Ctx()
    c.needs_input_grad = tuple(isinstance(a, torch.Tensor) for a in args)
    return fn.forward(c, *args), c


def _ln16(ops: TrainOps, x, gamma, beta):
    M = x.shape[0]
    n16 = torch.empty(x.shape, dtype=ops.dtype16, device=ops.device)
    mean, rstd = ops.new(M), ops.new(M)
    ops.check(ops.lib.some_train_layernorm_fwd16(ops.h, _p(x), _p(gamma), _p(beta), _p(n16), _p(mean), _p(rstd), M, ops._hi_mode, ops.stream()))
    return n16, mean, rstd


def _ln_bwd_add_into_sinks(ops: TrainOps, dn, x, gamma_t, mean, rstd, add, gamma, beta):
    """dx = add + LayerNorm'(dn); dgamma / dbeta accumulate in the parameters' gradient arrays."""
    M = x.shape[0]
    dx = torch.empty_like(x)
    sc = ops.scratch(M, 512)
    ops.check(ops.lib.some_train_layernorm_bwd_add(ops.h, _p(dn), _p(x), _p(gamma_t), _p(mean), _p(rstd), _p(add), _p(dx), _p(ops.sink(gamma)),
                                                   _p(ops.sink(beta)), 1, M, _p(sc), sc.numel(), ops.stream()))
    ops.deposited(gamma)
    ops.deposited(beta)
    return dx


def _wgrad16_into_sinks(ops: TrainOps, g16, in16, w, b):
    ops.wgrad16(g16, in16, ops.sink(w), ops.sink(b) if b is not None else None, accumulate=True)
    ops.deposited(w)
    if b is not None:
        ops.deposited(b)


class _AttnBlock16(torch.autograd.Function):
    """x + dropout(Wo attention(Wqkv LayerNorm(x)) + bo) in mixed precision (ops.can_block16: every parameter has a gradient sink):
    LayerNorm writes the 16-bit GEMM operand, to_q | to_kv are one [1536, 512] matrix (adjacent in the flat buffer: no concatenation),
    residual + dropout sit in the output projection's epilogue, their gradient and the 16-bit cast in one kernel, the residual gradient is
    summed inside LayerNorm-backward; the attention core is the _ops._Attention operator's own forward / backward."""

    @staticmethod
    def forward(ctx, ops: TrainOps, x, gamma, beta, wq, wkv, wo, bo, batch, p, seed):
        x = x.contiguous()
        M = x.shape[0]
        wqkv, _ = ops.joined(wq, wkv)
        n16, mean, rstd = _ln16(ops, x, gamma, beta)
        qkv = ops.new(M, 1536)
        ops.gemm16s(0, n16, ops.shadow16(wqkv)[0], None, qkv, 1536, M, 1536, 512)
        att, actx = _sub(_ops._Attention, ops, qkv, batch)
        att16 = ops.cast16(att)
        out = ops.new(M, 512)
        ops.gemm16s(3, att16, ops.shadow16(wo)[0], bo, out, 512, M, 512, 512, h16=x, p=p, seed=seed, alpha=1.0)
        ctx.ops, ctx.actx, ctx.drop = ops, actx, (p, seed)
        ctx.save_for_backward(x, gamma, mean, rstd, n16, att16)
        ctx.params = (gamma, beta, wq, wkv, wo, bo)
        return out

    @staticmethod
    def backward(ctx, d):
        ops: TrainOps = ctx.ops
        x, gamma_t, mean, rstd, n16, att16 = ctx.saved_tensors
        gamma, beta, wq, wkv, wo, bo = ctx.params
        d = d.contiguous()
        M = x.shape[0]
        dy16 = ops.dropcast16(d, 1.0, *ctx.drop)
        datt = ops.new(M, 512)
        ops.gemm16s(0, dy16, ops.shadow16(wo)[1], None, datt, 512, M, 512, 512)
        _wgrad16_into_sinks(ops, dy16, att16, wo, bo)
        dqkv16 = _ops._attention_bwd16(ctx.actx, datt) if ctx.actx.prec == 'f16x3' else ops.cast16(_ops._Attention.backward(ctx.actx, datt)[1])
        wqkv, gqkv = ops.joined(wq, wkv)
        dn = ops.new(M, 512)
        ops.gemm16s(0, dqkv16, ops.shadow16(wqkv)[1], None, dn, 512, M, 512, 1536)
        ops.wgrad16(dqkv16, n16, gqkv, None, accumulate=True)
        ops.deposited(wq)
        ops.deposited(wkv)
        dx = _ln_bwd_add_into_sinks(ops, dn, x, gamma_t, mean, rstd, d, gamma, beta)
        return None, dx, None, None, None, None, None, None, None, None, None


class _ConvBlock16(torch.autograd.Function):
    """x + dropout(pw2 silu(BatchNorm(dwconv(GLU(pw1 LayerNorm(x)))))) in mixed precision: LayerNorm and SiLU write the 16-bit GEMM operands,
    residual + dropout sit in pointwise_conv2's epilogue; GLU, the depthwise convolution and BatchNorm are their operators' own bodies."""

    @staticmethod
    def forward(ctx, ops: TrainOps, x, gamma, beta, pw1_w, pw1_b, dw_w, dw_b, bn_g, bn_b, bn_rm, bn_rv, pw2_w, pw2_b, batch, p, seed):
        x = x.contiguous()
        M = x.shape[0]
        n16, mean, rstd = _ln16(ops, x, gamma, beta)
        p1 = ops.new(M, 1024)
        ops.gemm16s(0, n16, ops.shadow16(pw1_w)[0], pw1_b, p1, 1024, M, 1024, 512)
        g, gctx = _sub(_ops._Glu, ops, p1)
        c, cctx = _sub(_ops._DwConv, ops, g, dw_w, dw_b, batch)
        bn, bctx = _sub(_ops._BatchNorm, ops, c, bn_g, bn_b, bn_rm, bn_rv, 0.1, 1e-5)
        s16 = ops.silu16(bn)
        out = ops.new(M, 512)
        ops.gemm16s(3, s16, ops.shadow16(pw2_w)[0], pw2_b, out, 512, M, 512, 512, h16=x, p=p, seed=seed, alpha=1.0)
        ctx.ops, ctx.sub, ctx.drop = ops, (gctx, cctx, bctx), (p, seed)
        ctx.save_for_backward(x, gamma, mean, rstd, n16, bn, s16)
        ctx.params = (gamma, beta, pw1_w, pw1_b, pw2_w, pw2_b)
        return out

    @staticmethod
    def backward(ctx, d):
        ops: TrainOps = ctx.ops
        x, gamma_t, mean, rstd, n16, bn, s16 = ctx.saved_tensors
        gamma, beta, pw1_w, pw1_b, pw2_w, pw2_b = ctx.params
        gctx, cctx, bctx = ctx.sub
        d = d.contiguous()
        M = x.shape[0]
        dy16 = ops.dropcast16(d, 1.0, *ctx.drop)
        ds = ops.new(M, 512)
        ops.gemm16s(0, dy16, ops.shadow16(pw2_w)[1], None, ds, 512, M, 512, 512)
        _wgrad16_into_sinks(ops, dy16, s16, pw2_w, pw2_b)
        dbn = ops.eltwise(_ops._lib.ELT_SILU_BWD, ds, bn)
        _, dc, dbn_g, dbn_b = _ops._BatchNorm.backward(bctx, dbn)[:4]
        _, dg, ddw_w, ddw_b = _ops._DwConv.backward(cctx, dc)[:4]
        dp1_16 = ops.cast16(_ops._Glu.backward(gctx, dg)[1])
        dn = ops.new(M, 512)
        ops.gemm16s(0, dp1_16, ops.shadow16(pw1_w)[1], None, dn, 512, M, 512, 1024)
        _wgrad16_into_sinks(ops, dp1_16, n16, pw1_w, pw1_b)
        dx = _ln_bwd_add_into_sinks(ops, dn, x, gamma_t, mean, rstd, d, gamma, beta)
        return None, dx, None, None, None, None, ddw_w, ddw_b, dbn_g, dbn_b, None, None, None, None, None, None, None
