#!/usr/bin/env python
"""Round-3 diagnosis of the round-2 open finding: with the wavefront index in an SGPR the row-per-lane SiLU epilogue of the split-f16
GEMM (FFN1: [M, 512] x [2048, 512]^T -> SPLIT32 [M, 2048]) gave run-to-run different results at 32 x 30 s.

Runs ONE FFN1-shaped GEMM `repeats` times with the library selected by SOME_AMD_LIBRARY into a buffer poisoned with a sentinel
before every run, and compares every run bit for bit with a reference result (`--save` writes it, `--ref` reads it: the stock
library in a first process, the variants in later ones).  For mismatching runs it says WHERE the wrong halves are and WHAT they
hold: sentinel (store missing), the expected hi / lo half of another place in the same row (store misdirected), or something else.

    python tools/diag_sgpr_epilogue.py --save /tmp/ffn1_ref.pt
    SOME_AMD_LIBRARY=tools/_bin/variants/sgpr/libsome_amd.so python tools/diag_sgpr_epilogue.py --ref /tmp/ffn1_ref.pt
"""
import argparse
import ctypes as C
import json
import os
import sys
import pathlib

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from some_amd import _lib  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import Engine  # noqa: E402

SENTINEL = 0x7E55          # an f16 NaN pattern no result half can be


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=32 * 2584)
    ap.add_argument('--repeats', type=int, default=12)
    ap.add_argument('--tile', type=int, default=2)
    ap.add_argument('--save')
    ap.add_argument('--ref')
    ap.add_argument('--tag', default=os.environ.get('SOME_AMD_LIBRARY', 'stock'))
    a = ap.parse_args()
    eng = Engine(get_config('midi_conformer', lay=1), device='cuda')
    g = torch.Generator(device='cuda').manual_seed(11)
    M, K, N = a.rows, 512, 2048
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(N, K, device='cuda', generator=g) / 20
    b = torch.randn(N, device='cuda', generator=g)
    As, Ws = torch.empty_like(A), torch.empty_like(W)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(eng.handle, eng.lib.some_op_split_rows(eng.handle, p(A), p(As), M, K, st))
    _lib.check(eng.handle, eng.lib.some_op_split_rows(eng.handle, p(W), p(Ws), N, K, st))
    flags = _lib.GEMM_SPLIT_IN | (a.tile << 8) | _lib.GEMM_SPLIT_OUT
    out = torch.empty(M, N, device='cuda')
    o16 = out.view(torch.int16)

    def run():
        o16.fill_(SENTINEL)
        _lib.check(eng.handle, eng.lib.some_op_gemm(eng.handle, _lib.EPI_BIAS_SILU, p(As), K, p(Ws), p(b), None, 0, p(out), N, M, N, K,
                                                    1.0, 0, None, flags, st))
        torch.cuda.synchronize()
        return out.clone()

    first = run()
    if a.save:
        same = all(torch.equal(first, run()) for _ in range(a.repeats - 1))
        torch.save(first.cpu(), a.save)
        print(json.dumps({'tag': a.tag, 'saved': a.save, 'repeat_runs_identical': same}))
        return
    ref = torch.load(a.ref).cuda()
    # [M, 64 k-blocks, 2 planes (hi, lo), 32]
    rv = ref.view(torch.int16).view(M, N // 32, 2, 32)
    report = {'tag': a.tag, 'rows': M, 'runs': []}
    res = first
    for it in range(a.repeats):
        if it:
            res = run()
        gv = res.view(torch.int16).view(M, N // 32, 2, 32)
        bad = gv != rv
        nbad = int(bad.sum())
        entry = {'run': it, 'wrong_halves': nbad}
        if nbad:
            entry['wrong_hi'] = int(bad[:, :, 0].sum())
            entry['wrong_lo'] = int(bad[:, :, 1].sum())
            entry['sentinel'] = int((gv[bad] == SENTINEL).sum())
            rows = bad.any(dim=3).any(dim=2).any(dim=1).nonzero()[:, 0]
            entry['rows_hit'] = int(rows.numel())
            entry['row_tiles_hit_256'] = sorted(set((rows // 256).tolist()))[:40]
            blk = bad.any(dim=3).any(dim=0)          # [64, 2]
            entry['kblocks_hit_hi'] = blk[:, 0].nonzero()[:, 0].tolist()
            entry['kblocks_hit_lo'] = blk[:, 1].nonzero()[:, 0].tolist()
            # what do the wrong halves hold?  look at up to 8 wrong (row, block, plane) runs on the host
            idx = bad.any(dim=3).nonzero()[:8].tolist()
            samples = []
            for (r, kb, pl) in idx:
                got = gv[r, kb, pl].cpu()
                want = rv[r, kb, pl].cpu()
                rowref = rv[r].cpu()                  # [64, 2, 32]
                where = None
                for kb2 in range(N // 32):
                    for pl2 in range(2):
                        w8 = rowref[kb2, pl2]
                        for off in (0, 8, 16, 24):
                            for off2 in (0, 8, 16, 24):
                                if torch.equal(got[off:off + 8], w8[off2:off2 + 8]) and not torch.equal(got[off:off + 8], want[off:off + 8]):
                                    where = {'got_cols': [off, off + 8], 'equals_block': kb2, 'plane': 'hi' if pl2 == 0 else 'lo', 'cols': [off2, off2 + 8]}
                samples.append({'row': r, 'kblock': kb, 'plane': 'hi' if pl == 0 else 'lo',
                                'wrong_positions': (got != want).nonzero()[:, 0].tolist(),
                                'got_hex': [f'{v & 0xffff:04x}' for v in got.tolist()][:32],
                                'want_hex': [f'{v & 0xffff:04x}' for v in want.tolist()][:32],
                                'matches_elsewhere_in_row': where})
            entry['samples'] = samples
        report['runs'].append(entry)
    report['runs_wrong'] = sum(1 for e in report['runs'] if e['wrong_halves'])
    print(json.dumps(report))


if __name__ == '__main__':
    main()
