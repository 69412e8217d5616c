#!/usr/bin/env python
"""Diagnosis of the MX-FP6 attention path: some_op_qkv_attention_f16x3 with SOME_AMD_ATTN_MX = 1 vs 0 vs fp64, per shape; where the
errors sit; and the FP6 V blocks the QKV epilogue wrote, decoded on the host against the V the same GEMM put into the f16 hi plane."""
import ctypes as C
import os
import sys
import pathlib

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from some_amd import _lib  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import ClipBatch, Engine  # noqa: E402

p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def engine(mx):
    os.environ['SOME_AMD_ATTN_MX'] = '1' if mx else '0'
    return Engine(get_config('midi_conformer', lay=1), device='cuda')


def split(eng, x):
    out = torch.empty_like(x)
    _lib.check(eng.handle, eng.lib.some_op_split_rows(eng.handle, p(x), p(out), x.shape[0], x.shape[1], st()))
    return out


def unsplit(t):
    r, k = t.shape
    h = t.contiguous().view(torch.float16).view(r, k // 32, 2, 32).float()
    return (h[:, :, 0] + h[:, :, 1]).reshape(r, k)


E2M3 = np.array([m * 0.125 for m in range(8)] + [2.0 ** (e - 1) * (1 + m / 8) for e in (1, 2, 3) for m in range(8)])


def decode_fp6(words):          # [..., 6] uint32 -> [..., 32] floats (element j = bits 6 j .. 6 j + 5 of the little-endian stream)
    w = words.astype(np.uint64)
    bits = np.zeros(words.shape[:-1] + (192,), dtype=np.uint8)
    for i in range(6):
        for b in range(32):
            bits[..., 32 * i + b] = (w[..., i] >> np.uint64(b)) & np.uint64(1)
    out = np.zeros(words.shape[:-1] + (32,))
    for j in range(32):
        v = sum(bits[..., 6 * j + b].astype(np.int64) << b for b in range(6))
        out[..., j] = np.where(v & 32, -1.0, 1.0) * E2M3[v & 31]
    # v_cvt_scalef32_2xpk16_fp6_f32 interleaves its two source tuples: field 2 j = a[j], 2 j + 1 = b[j] (tools/mx_probe2.hip)
    return np.concatenate([out[..., 0::2], out[..., 1::2]], axis=-1)


def main():
    engs = {1: engine(True), 0: engine(False)}
    for lens in ([64], [128], [200], [33], [700], [2584], [2584, 100]):
        g = torch.Generator(device='cuda').manual_seed(100 + sum(lens))
        batch = ClipBatch(lens, 'cuda')
        M = batch.total_frames
        h = torch.randn(M, 512, device='cuda', generator=g)
        W = torch.randn(1536, 512, device='cuda', generator=g) / 512 ** 0.5
        W[:512] *= 3.0
        ldv = (M + 255) // 256 * 256
        outs, wss = {}, {}
        for mx, eng in engs.items():
            hs, Ws = split(eng, h), split(eng, W)
            ws = torch.zeros(M * 4096 + 2048 * ldv, dtype=torch.uint8, device='cuda')
            out = torch.full((M, 512), float('nan'), device='cuda')
            runs = []
            for _ in range(3):
                _lib.check(eng.handle, eng.lib.some_op_qkv_attention_f16x3(eng.handle, p(hs), p(Ws), p(batch.frame_offsets_dev), batch.B, batch.max_frames, M,
                                                                           p(out), p(ws), ws.numel(), st()))
                torch.cuda.synchronize()
                runs.append(unsplit(out).clone())
            outs[mx] = runs
            wss[mx] = ws
        qkv = h.double() @ W.double().t()
        ref = torch.empty(M, 512, dtype=torch.float64, device='cuda')
        for b, t in enumerate(lens):
            s = int(batch.frame_offsets[b])
            x = qkv[s:s + t]
            q, k, v = (x[:, i * 512:(i + 1) * 512].reshape(t, 8, 64).transpose(0, 1) for i in range(3))
            ref[s:s + t] = (torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1) @ v).transpose(0, 1).reshape(t, 512)
        e1 = (outs[1][0].double() - ref).abs()
        e0 = (outs[0][0].double() - ref).abs()
        same = all(torch.equal(outs[1][0], r) for r in outs[1][1:])
        print(f'lens {lens}: max err mx {e1.max().item():.2e} (row {int(e1.max(dim=1).values.argmax())}, col {int(e1.max(dim=0).values.argmax())}), '
              f'f16x3 {e0.max().item():.2e}; mx runs identical: {same}; rows with err > 2e-5: {int((e1.max(dim=1).values > 2e-5).sum())} of {M}')
        # the FP6 blocks: decode (head 0..7, tile 0) and compare with V from the fp64 product
        raw = wss[1].cpu().numpy()
        base = M * 4096 + 512 * ldv * 2
        V = qkv[:, 1024:].cpu().numpy()                     # [M, 512]
        worst_h = worst_l = 0.0
        n_gt = ldv // 64
        for head in (0, 5):
            for gt in range(min(n_gt, (M + 63) // 64)):
                blk = raw[base + (head * n_gt + gt) * 6272: base + (head * n_gt + gt + 1) * 6272]
                units = blk[:6144].view(np.uint32).reshape(2, 64, 12)
                scales = blk[6144:].reshape(2, 64).astype(np.int64)
                for kg in (0, 1):
                    frames = np.array([32 * sub + 8 * i + 4 * kg + e for sub in (0, 1) for i in range(4) for e in range(4)]) + 64 * gt
                    ok = frames < M
                    vv = np.zeros((64, 32))
                    vv[:, ok] = V[frames[ok]][:, head * 64:(head + 1) * 64].T
                    hi = decode_fp6(units[kg, :, :6]) * 2.0 ** (scales[kg][:, None] - 127)
                    vh = vv.astype(np.float16).astype(np.float64)
                    lo = decode_fp6(units[kg, :, 6:]) * 2.0 ** (scales[kg][:, None] - 127 - 11)
                    amax = np.abs(vv).max(axis=1, keepdims=True) + 1e-30
                    worst_h = max(worst_h, float((np.abs(hi - vv) / amax).max()))
                    worst_l = max(worst_l, float((np.abs(lo - (vv - vh)) / (amax * 2.0 ** -11)).max()))
        print(f'    FP6 V blocks vs fp64 V: hi block error / block max {worst_h:.3f} (e2m3 step: <= 0.07), lo block error / (2^-11 block max) {worst_l:.3f}')


if __name__ == '__main__' and not os.environ.get('CROSS'):
    main()


def cross_term_check():
    """Is the v_lo * p cross term present and correctly scaled?  o_A = softmax * f16(V) (no v_lo term), o_B = softmax * V: the MX output
    minus o_A, projected on o_B - o_A, should have coefficient 1 (0: term missing; 2^k: a scale is off)."""
    eng = engine(True)
    eng0 = engine(False)
    for lens in ([64], [700]):
        g = torch.Generator(device='cuda').manual_seed(100 + sum(lens))
        batch = ClipBatch(lens, 'cuda')
        M = batch.total_frames
        h = torch.randn(M, 512, device='cuda', generator=g)
        W = torch.randn(1536, 512, device='cuda', generator=g) / 512 ** 0.5
        W[:512] *= 3.0
        ldv = (M + 255) // 256 * 256
        res = {}
        for name, e in (('mx', eng), ('f16x3', eng0)):
            hs, Ws = split(e, h), split(e, W)
            ws = torch.zeros(M * 4096 + 2048 * ldv, dtype=torch.uint8, device='cuda')
            out = torch.full((M, 512), float('nan'), device='cuda')
            _lib.check(e.handle, e.lib.some_op_qkv_attention_f16x3(e.handle, p(hs), p(Ws), p(batch.frame_offsets_dev), batch.B, batch.max_frames, M,
                                                                   p(out), p(ws), ws.numel(), st()))
            torch.cuda.synchronize()
            res[name] = unsplit(out).double()
        qkv = (h.double() @ W.double().t())
        t = lens[0]
        q, k, v = (qkv[:t, i * 512:(i + 1) * 512].reshape(t, 8, 64).transpose(0, 1) for i in range(3))
        P = torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1)
        vh = v.float().half().double()
        oB = (P @ v).transpose(0, 1).reshape(t, 512)
        oA = (P @ vh).transpose(0, 1).reshape(t, 512)
        d = oB - oA
        for name in ('mx', 'f16x3'):
            coef = float(((res[name][:t] - oA) * d).sum() / (d * d).sum())
            print(f'lens {lens} {name}: coefficient of the v_lo * p term {coef:.3f}; |o - o_B| max {float((res[name][:t] - oB).abs().max()):.2e}; '
                  f'|o_B - o_A| max {float(d.abs().max()):.2e}')


if __name__ == '__main__' and os.environ.get('CROSS'):
    cross_term_check()
