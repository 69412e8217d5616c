#!/usr/bin/env python
"""CPU emulation of operand-format variants of the split-f16 contraction ("attack the 3x precision tax").

Not a test (pytest does not collect it): run it by hand in the build container,

    python tests/emulate_split_precision.py [--lay 8] [--seconds 30]

It runs the CPU oracle's model (oracle/restate.py - the reference's ATen sequence) on one full-size clip with every dense
contraction (nn.Linear, k = 1 Conv1d, Q K^T and P V of the attention) replaced by an EMULATED product:

    x = xh + xl, w = wh + wl   (f16 halves, as csrc/split.h)      shipped:  xh wh + xh wl + xl wh     (3 f16 MFMA products)

and variants that move the two cross terms - each carries 2^-11 of the magnitude - to a cheaper format: FP8 (e4m3 / e5m2, with an
MX-style power-of-two scale per 32-element block along the contraction, what v_mfma_scale_f32_*_f8f6f4 consumes; 2x the f16 MFMA
rate on gfx950, so a cross term costs half a product), or drop one.  Every operand rounding is emulated exactly (torch's own
float16 / bfloat16 / float8 conversions, round to nearest even); products are accumulated in fp64 and rounded to fp32 once, so the
numbers isolate the OPERAND FORMAT error (the fp32 accumulation error of the real kernels is common to all variants: the `f32` row).
The yardstick is the fp64 run of the same model.  Output: a table of max |d logit|, |d prob|, |d bound| per variant with its issue
cost in f16-product units - profiles/r03_precision_variants.md quotes it."""
import argparse
import pathlib
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import restate  # noqa: E402
from some_amd import synth  # noqa: E402
from some_amd.configs import get_config  # noqa: E402


def f16(x):
    return x.to(torch.float16).to(torch.float64)


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float64)


def fp8_mx(x, fmt, axis=-1, block=32):
    """MX-style FP8: per `block` consecutive elements along `axis` a shared power-of-two scale that puts the block maximum just
    under the format's largest finite value, elements rounded to nearest in the FP8 format (torch conversion), scaled back."""
    dt, fmax = (torch.float8_e4m3fn, 448.0) if fmt == 'e4m3' else (torch.float8_e5m2, 57344.0)
    x = x.movedim(axis, -1)
    shp = x.shape
    pad = (-shp[-1]) % block
    if pad:
        x = F.pad(x, (0, pad))
    xb = x.reshape(*x.shape[:-1], -1, block)
    amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.floor(torch.log2(fmax / amax)))                 # power of two: exact scaling
    q = (xb * scale).to(torch.float32).to(dt).to(torch.float64) / scale
    q = q.reshape(*x.shape[:-1], -1)[..., :shp[-1]]
    return q.movedim(-1, axis)


def _fp6_table(fmt):
    """Non-negative representable values of the OCP MX FP6 element formats (no inf / nan): e2m3 (bias 1, max 7.5), e3m2 (bias 3, max 28)."""
    if fmt == 'e2m3':
        vals = [m * 0.125 for m in range(8)] + [2.0 ** (e - 1) * (1 + m / 8) for e in (1, 2, 3) for m in range(8)]
    else:
        vals = [m * 0.0625 for m in range(4)] + [2.0 ** (e - 3) * (1 + m / 4) for e in range(1, 8) for m in range(4)]
    return torch.tensor(sorted(set(vals)), dtype=torch.float64)


def fp6_mx(x, fmt, axis=-1, block=32, norm='max'):
    """MX FP6: shared power-of-two scale per 32-element block (block maximum just under the format's largest value), elements
    rounded to the nearest representable FP6 value - what v_mfma_scale_f32_*_f8f6f4 consumes at the FP4 rate (4x f16) on gfx950."""
    table = _fp6_table(fmt)
    fmax = float(table[-1])
    x = x.movedim(axis, -1)
    shp = x.shape
    pad = (-shp[-1]) % block
    if pad:
        x = F.pad(x, (0, pad))
    xb = x.reshape(*x.shape[:-1], -1, block)
    amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    if norm == 'l2':            # a cheap upper bound of the block maximum: its Euclidean norm (16 v_dot2 instead of a 31-step max tree)
        amax = xb.pow(2).sum(dim=-1, keepdim=True).sqrt().clamp_min(1e-300)
    scale = torch.exp2(torch.floor(torch.log2(fmax / amax)))
    y = (xb * scale).abs()
    mid = (table[1:] + table[:-1]) / 2
    q = table[torch.bucketize(y, mid)] * torch.sign(xb) / scale
    q = q.reshape(*x.shape[:-1], -1)[..., :shp[-1]]
    return q.movedim(-1, axis)


def fp6_global(x, fmt, slack=1.0, reduce_dims=(-2, -1)):
    """FP6 with ONE power-of-two scale per matrix (per head in the attention products): the maximum over all rows (times `slack`,
    a deliberately loose bound) sits under the format's largest value; smaller elements fall into the subnormal range / flush."""
    table = _fp6_table(fmt)
    fmax = float(table[-1])
    amax = x.abs().amax(dim=reduce_dims, keepdim=True).clamp_min(1e-300) * slack
    scale = torch.exp2(torch.floor(torch.log2(fmax / amax)))
    y = (x * scale).abs().clamp_max(fmax)
    mid = (table[1:] + table[:-1]) / 2
    return table[torch.bucketize(y.contiguous(), mid)] * torch.sign(x) / scale


def _lo_from_hi(h, l, fmt, slack):
    """Quantise the lo half with the scale implied by the hi half's maximum: |lo| <= 2^-11 * 2^ceil(log2 max|hi|)."""
    table = _fp6_table(fmt)
    fmax = float(table[-1])
    amax = h.abs().amax(dim=(-2, -1), keepdim=True).clamp_min(1e-300) * slack
    bound = torch.exp2(torch.floor(torch.log2(amax)) + 1 - 11)
    scale = torch.exp2(torch.floor(torch.log2(fmax / bound)))
    y = (l * scale).abs().clamp_max(fmax)
    mid = (table[1:] + table[:-1]) / 2
    return table[torch.bucketize(y.contiguous(), mid)] * torch.sign(l) / scale


def split(x, kind='f16'):
    r = f16 if kind == 'f16' else bf16
    x = x.to(torch.float64)
    h = r(x.to(torch.float32))
    l = r((x - h).to(torch.float32))
    return h, l


# variant -> (function (a, b) -> a b^T contraction over the last axis of both, cost in f16 products)
def make_product(variant):
    def prod(a, b):                       # a [..., M, K], b [..., N, K]  ->  [..., M, N] fp32
        a, b = a.to(torch.float64), b.to(torch.float64)
        mm = lambda x, y: x @ y.transpose(-1, -2)        # noqa: E731  (fp64: exact products of the rounded operands)
        if variant == 'f32':
            return (a.to(torch.float32) @ b.to(torch.float32).transpose(-1, -2))
        if variant == 'f64':
            return mm(a, b)
        if variant == 'bf16x3':
            ah, al = split(a, 'bf16')
            bh, bl = split(b, 'bf16')
            return (mm(ah, bh) + mm(ah, bl) + mm(al, bh)).to(torch.float32)
        ah, al = split(a)
        bh, bl = split(b)
        if variant == 'f16x3':
            out = mm(ah, bh) + mm(ah, bl) + mm(al, bh)
        elif variant == 'f16x2_drop_a_lo':            # x lo dropped
            out = mm(ah, bh) + mm(ah, bl)
        elif variant == 'f16x2_drop_b_lo':            # w lo dropped
            out = mm(ah, bh) + mm(al, bh)
        elif variant == 'f16x1':
            out = mm(ah, bh)
        elif variant in ('cross_e4m3', 'cross_e5m2'):
            f = variant[-4:]
            out = mm(ah, bh) + mm(fp8_mx(ah, f), fp8_mx(bl, f)) + mm(fp8_mx(al, f), fp8_mx(bh, f))
        elif variant in ('cross_fp6_e2m3', 'cross_fp6_e3m2'):
            f = variant[-4:]
            out = mm(ah, bh) + mm(fp6_mx(ah, f), fp6_mx(bl, f)) + mm(fp6_mx(al, f), fp6_mx(bh, f))
        elif variant in ('cross_fp6l2_e2m3', 'cross_fp6l2_e3m2'):      # MX blocks, scale from the block's L2 norm, lo scale from the hi scale
            f = variant[-4:]
            out = mm(ah, bh) + mm(fp6_mx(ah, f, norm='l2'), fp6_mx(bl, f, norm='l2')) + mm(fp6_mx(al, f, norm='l2'), fp6_mx(bh, f, norm='l2'))
        elif variant.startswith('cross_fp6g_'):        # cross_fp6g_<fmt>_<slack>: one scale per matrix / head, loosened by `slack`
            _, _, f, slack = variant.split('_')
            slack = float(slack)
            # lo halves: scale derived from the hi scale (|lo| <= 2^-11 max|hi|), not from their own maximum
            q = lambda h, l: (fp6_global(h, f, slack), fp6_global(l, f, slack) if False else _lo_from_hi(h, l, f, slack))     # noqa: E731
            a6h, a6l = q(ah, al)
            b6h, b6l = q(bh, bl)
            out = mm(ah, bh) + mm(a6h, b6l) + mm(a6l, b6h)
        elif variant == 'a_lo_e4m3':                  # activations' lo term on the FP8 pipe, weights' lo term stays f16
            out = mm(ah, bh) + mm(ah, bl) + mm(fp8_mx(al, 'e4m3'), fp8_mx(bh, 'e4m3'))
        elif variant == 'b_lo_e4m3':
            out = mm(ah, bh) + mm(fp8_mx(ah, 'e4m3'), fp8_mx(bl, 'e4m3')) + mm(al, bh)
        elif variant == 'cross_bf16':                 # cross terms with bf16 operands (same cost as f16: control for mantissa width)
            out = mm(ah, bh) + mm(bf16(ah), bf16(bl)) + mm(bf16(al), bf16(bh))
        else:
            raise ValueError(variant)
        return out.to(torch.float32)
    return prod


COST = {'cross_fp6l2_e2m3': 1.5, 'cross_fp6l2_e3m2': 1.5,
        'cross_fp6g_e2m3_1': 1.5, 'cross_fp6g_e2m3_4': 1.5, 'cross_fp6g_e2m3_16': 1.5, 'cross_fp6g_e2m3_64': 1.5, 'cross_fp6g_e3m2_1': 1.5, 'cross_fp6g_e3m2_16': 1.5, 'cross_fp6g_e3m2_64': 1.5,
        'f32': None, 'bf16x3': 3.0, 'f16x3': 3.0, 'f16x2_drop_a_lo': 2.0, 'f16x2_drop_b_lo': 2.0, 'f16x1': 1.0, 'cross_e4m3': 2.0, 'cross_e5m2': 2.0,
        'cross_fp6_e2m3': 1.5, 'cross_fp6_e3m2': 1.5, 'a_lo_e4m3': 2.5, 'b_lo_e4m3': 2.5, 'cross_bf16': 3.0}


class Shim:
    """Stands in for torch.nn.functional inside oracle/restate.py: dense contractions go through `prod`, the rest through F."""

    def __init__(self, prod, gemms=True, attention=True, dtype=torch.float32, qk=True, pv=True):
        self.prod, self.gemms, self.attention, self.dtype = prod, gemms, attention, dtype
        base = make_product('f16x3')
        self.prod_qk, self.prod_pv = (prod if qk else base), (prod if pv else base)      # the other product stays the shipped 3-term form

    def __getattr__(self, name):
        return getattr(F, name)

    def linear(self, x, w, b=None):
        if not self.gemms or w.shape[1] % 32:                   # K = 80 input projection: exact-f32 kernel in the product too
            return F.linear(x, w, b)
        y = self.prod(x, w).to(self.dtype)
        return y if b is None else y + b

    def conv1d(self, x, w, b=None, **kw):
        if not self.gemms or w.shape[-1] != 1 or kw.get('groups', 1) != 1:
            return F.conv1d(x, w, b, **kw)
        y = self.prod(x.transpose(1, 2), w[:, :, 0]).to(self.dtype)          # [B, T, C] x [N, C]
        if b is not None:
            y = y + b
        return y.transpose(1, 2)

    def scaled_dot_product_attention(self, q, k, v):
        if not self.attention:
            return F.scaled_dot_product_attention(q, k, v)
        s = self.prod_qk(q, k).to(torch.float64) * q.shape[-1] ** -0.5
        p = torch.softmax(s, dim=-1)
        # the kernel carries P scaled by a power of two (<= 2^14) to keep its lo half out of the f16 subnormals; emulate the same
        o = self.prod_pv(p * 16384.0, v.transpose(-1, -2)).to(torch.float64) / 16384.0
        return o.to(self.dtype)


def run(sd, cfg, units, shim):
    old = restate.F
    restate.F = shim
    try:
        return restate.model_forward(sd, cfg, units, sig=True)
    finally:
        restate.F = old


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lay', type=int, default=8)
    ap.add_argument('--seconds', type=float, default=30.0)
    ap.add_argument('--where', default='all', help="all | attention")
    ap.add_argument('--variants', default='f32,f16x3,bf16x3,cross_bf16,cross_e4m3,cross_e5m2,a_lo_e4m3,b_lo_e4m3,f16x2_drop_a_lo,f16x2_drop_b_lo')
    a = ap.parse_args()
    torch.set_num_threads(8)
    cfg = get_config('midi_conformer', lay=a.lay)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state_dict(cfg, seed=cfg['seed']).items()}
    wave = synth.synth_clip(0, a.seconds)
    units = torch.from_numpy(restate.logmel(wave, cfg))[None]
    sd64 = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
    t0 = time.time()
    ref = run(sd64, cfg, units.double(), Shim(make_product('f64'), dtype=torch.float64))
    ref = [r.double() for r in ref]
    print(f'# lay {a.lay}, T = {units.shape[1]} frames; fp64 yardstick in {time.time() - t0:.0f} s', flush=True)
    print('| variant | where | issue cost (f16 products) | max abs d prob | max abs d bound | max abs d logit(midi) |')
    print('|---|---|---|---|---|---|')
    for v in a.variants.split(','):
        for where, g, at in (('GEMMs + attention', True, True), ('GEMMs only', True, False), ('attention only', False, True),
                             ('attention P V only', False, 'pv'), ('attention Q K^T only', False, 'qk')):
            if v == 'f32' and where != 'GEMMs + attention':
                continue
            if a.where == 'attention' and not where.startswith('attention'):
                continue
            if a.where == 'all' and at in ('pv', 'qk'):
                continue
            t0 = time.time()
            out = run(sd, cfg, units, Shim(make_product(v), g, bool(at), qk=at != 'pv', pv=at != 'qk'))
            dp = float((out[0].double() - ref[0]).abs().max())
            db = float((out[1].double() - ref[1]).abs().max())
            lg = torch.logit(out[0].double().clamp(1e-12, 1 - 1e-12)) - torch.logit(ref[0].clamp(1e-12, 1 - 1e-12))
            big = ref[0] > 1e-4                                              # logits where the probability is not vanishing
            dl = float(lg[big].abs().max())
            print(f'| {v} | {where} | {COST[v] if COST[v] else "-"} | {dp:.2e} | {db:.2e} | {dl:.2e} |   ({time.time() - t0:.0f} s)', flush=True)


if __name__ == '__main__':
    main()
