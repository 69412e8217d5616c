"""Micro-benchmark of the GEMM kernels through the C ABI (tuning aid; prints TFLOP/s per shape / tile).

    python tools/gemm_bench.py [--split 1] [--tile 2] [--iters 20]
"""
import argparse
import ctypes as C
import sys
import pathlib

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from some_amd import _lib  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--split', type=int, default=1)
    ap.add_argument('--tile', type=int, default=2)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--M', type=int, default=82688)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    eng = Engine(get_config('midi_conformer', lay=0), device='cuda')
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = args.M
    shapes = [('ffn1 silu', _lib.EPI_BIAS_SILU, 2048, 512), ('ffn2 res', _lib.EPI_BIAS_RES, 512, 2048),
              ('proj res', _lib.EPI_BIAS_RES, 512, 512), ('qkv none', _lib.EPI_NONE, 1536, 512), ('glu', _lib.EPI_GLU, 1024, 512)]
    for name, epi, N, K in shapes:
        if args.only and args.only not in name:
            continue
        A = torch.randn(M, K, device='cuda')
        W = torch.randn(N, K, device='cuda') / K ** 0.5
        bias = torch.randn(N, device='cuda')
        n_out = N // 2 if epi == _lib.EPI_GLU else N
        res = torch.randn(M, n_out, device='cuda')
        Cm = torch.empty(M, n_out, device='cuda')
        flags = 0
        if args.split:
            As, Ws = torch.empty_like(A), torch.empty_like(W)
            _lib.check(eng.handle, eng.lib.some_op_split_rows(eng.handle, p(A), p(As), M, K, st))
            _lib.check(eng.handle, eng.lib.some_op_split_rows(eng.handle, p(W), p(Ws), N, K, st))
            A, W = As, Ws
            flags = _lib.GEMM_SPLIT_IN | (args.tile << 8) | (_lib.GEMM_SPLIT_OUT if epi == _lib.EPI_BIAS_SILU else 0)     # FFN1 as the model runs it: SPLIT32 output, row-per-lane epilogue

        def run():
            _lib.check(eng.handle, eng.lib.some_op_gemm(eng.handle, epi, p(A), K, p(W), p(bias), p(res), n_out, p(Cm), n_out,
                                                        M, N, K, 1.0, 0, None, flags, st))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        print(f'{name:10s} M={M} N={N} K={K} split={args.split} tile={args.tile}: {ms:.4f} ms  {tf:.1f} TF logical' +
              (f'  ({3 * tf:.0f} TF f16 issued)' if args.split else ''))


if __name__ == '__main__':
    main()
