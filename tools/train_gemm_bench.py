"""Micro-benchmark of the training GEMM (some_train_gemm16 / some_train_gemm16_wgrad) per layer shape and layout.

    python tools/train_gemm_bench.py [--M 20672] [--operand 2] [--iters 20]

Prints time, model TFLOP/s and the HBM bytes a launch has to move at least (operands once + output once), so that each shape can be
read against both roofs."""
import argparse
import ctypes as C
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from some_amd import _lib  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import Engine  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--M', type=int, default=20672)
    ap.add_argument('--operand', type=int, default=2)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--stored16', action='store_true')
    a = ap.parse_args()
    if a.stored16:
        return bench16s(a.M, a.operand, a.iters)
    eng = Engine(get_config('midi_conformer', lay=0), device='cuda')
    lib, h = eng.lib, eng.handle
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = a.M
    shapes = [('ffn1', 2048, 512), ('ffn2', 512, 2048), ('qkv', 1536, 512), ('proj', 512, 512), ('pw1', 1024, 512)]
    for name, N, K in shapes:
        x = torch.randn(M, K, device='cuda')
        w = torch.randn(N, K, device='cuda') / K ** 0.5
        b = torch.randn(N, device='cuda')
        y = torch.empty(M, N, device='cuda')
        dy = torch.randn(M, N, device='cuda')
        dx = torch.empty(M, K, device='cuda')
        dw = torch.zeros(N, K, device='cuda')
        db = torch.zeros(N, device='cuda')
        need = int(lib.some_train_gemm16_bytes(h, N, K, M, K + 4))
        part = torch.empty(need, dtype=torch.uint8, device='cuda')
        flops = 2.0 * M * N * K

        def fwd():
            _lib.check(h, lib.some_train_gemm16(h, p(x), K, 0, p(w), K, 0, p(b), p(y), N, M, N, K, a.operand, -1, None, 0, st))

        def dgrad():
            _lib.check(h, lib.some_train_gemm16(h, p(dy), N, 0, p(w), K, 1, None, p(dx), K, M, K, N, a.operand, -1, None, 0, st))

        def wgrad():
            _lib.check(h, lib.some_train_gemm16_wgrad(h, p(dy), N, p(x), K, p(dw), p(db), N, K, M, a.operand, 1, p(part), part.numel(), st))

        io = 4.0 * (M * K + N * K + M * N)
        for lab, fn in (('fwd', fwd), ('dgrad', dgrad), ('wgrad', wgrad)):
            ms = timed(fn, a.iters)
            print(f'{name:5s} {lab:6s} M={M} N={N} K={K} operand={a.operand}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF   min HBM {io / 1e6:6.1f} MB '
                  f'-> {io / ms / 1e9:5.2f} TB/s', flush=True)



def bench16s(M=20672, operand=2, iters=20):
    """The same shapes through the 16-bit-storage kernels (some_train_gemm16s / some_train_gemm16_wgrad16)."""
    eng = Engine(get_config('midi_conformer', lay=0), device='cuda')
    lib, h = eng.lib, eng.handle
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dt = torch.bfloat16 if operand == 2 else torch.float16
    shapes = [('ffn1', 2048, 512), ('ffn2', 512, 2048), ('qkv', 1536, 512), ('proj', 512, 512), ('pw1', 1024, 512)]
    for name, N, K in shapes:
        x = torch.randn(M, K, device='cuda')
        w = torch.randn(N, K, device='cuda') / K ** 0.5
        b = torch.randn(N, device='cuda')
        dy = torch.randn(M, N, device='cuda')
        x16, dy16 = torch.empty(M, K, dtype=dt, device='cuda'), torch.empty(M, N, dtype=dt, device='cuda')
        w16, w16t = torch.empty(N, K, dtype=dt, device='cuda'), torch.empty(K, N, dtype=dt, device='cuda')
        _lib.check(h, lib.some_train_cast16(h, p(x), p(x16), x.numel(), operand, st))
        _lib.check(h, lib.some_train_cast16(h, p(dy), p(dy16), dy.numel(), operand, st))
        _lib.check(h, lib.some_train_transpose16(h, p(w), p(w16), p(w16t), N, K, operand, st))
        assert torch.equal(x16, x.to(dt)) and torch.equal(w16, w.to(dt)) and torch.equal(w16t, w.to(dt).t().contiguous())
        y = torch.empty(M, N, device='cuda')
        dx = torch.empty(M, K, device='cuda')
        dw = torch.zeros(N, K, device='cuda')
        db = torch.zeros(N, device='cuda')
        need = int(lib.some_train_gemm16_bytes(h, N, K, M, K + 4))
        part = torch.empty(need, dtype=torch.uint8, device='cuda')
        flops = 2.0 * M * N * K

        def fwd():
            _lib.check(h, lib.some_train_gemm16s(h, 0, p(x16), K, p(w16), K, p(b), p(y), N, None, 0, 0, M, N, K, operand, 0.0, 0, 1.0, st))

        def dgrad():
            _lib.check(h, lib.some_train_gemm16s(h, 0, p(dy16), N, p(w16t), N, None, p(dx), K, None, 0, 0, M, K, N, operand, 0.0, 0, 1.0, st))

        def wgrad():
            _lib.check(h, lib.some_train_gemm16_wgrad16(h, p(dy16), N, p(x16), K, p(dw), p(db), N, K, M, operand, 0, p(part), part.numel(), st))

        def cast():
            _lib.check(h, lib.some_train_cast16(h, p(dy), p(dy16), dy.numel(), operand, st))

        fwd(); dgrad(); wgrad()
        torch.cuda.synchronize()
        xr, wr, dyr = x16.double(), w16.double(), dy16.double()
        e_f = float((y.double() - (xr @ wr.t() + b.double())).abs().max() / (xr @ wr.t()).abs().max())
        e_d = float((dx.double() - dyr @ wr).abs().max() / (dyr @ wr).abs().max())
        e_w = float((dw.double() - dyr.t() @ xr).abs().max() / (dyr.t() @ xr).abs().max())
        e_b = float((db.double() - dyr.sum(0)).abs().max() / dyr.sum(0).abs().max())
        io = 2.0 * (M * K + N * K) + 4.0 * M * N
        for lab, fn in (('fwd', fwd), ('dgrad', dgrad), ('wgrad', wgrad), ('cast dy', cast)):
            ms = timed(fn, iters)
            print(f'16s {name:5s} {lab:7s} M={M} N={N} K={K} operand={operand}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF', flush=True)
        print(f'    relative errors vs fp64 products of the stored operands: fwd {e_f:.2e} dgrad {e_d:.2e} wgrad {e_w:.2e} db {e_b:.2e}', flush=True)
        if name == 'ffn1':                                           # the fused epilogues on their layer shapes
            ha = torch.empty(2, M, N, dtype=dt, device='cuda')
            dh16 = torch.empty(M, N, dtype=dt, device='cuda')
            g16 = torch.empty(M, K, dtype=dt, device='cuda').normal_()        # dy of the FFN's second linear: [M, 512]
            wt2 = torch.empty(N, K, dtype=dt, device='cuda').normal_()        # W2^T image [2048, 512]
            for pdrop in (0.0, 0.1):
                def ffn1():
                    _lib.check(h, lib.some_train_gemm16s(h, 1, p(x16), K, p(w16), K, p(b), p(ha), N, None, 0, M * N, M, N, K, operand, pdrop, 77, 1.0, st))

                def dsilu():
                    _lib.check(h, lib.some_train_gemm16s(h, 2, p(g16), K, p(wt2), K, None, p(dh16), N, p(ha[0]), N, 0, M, N, K, operand, pdrop, 77, 1.0, st))
                for lab, fn in ((f'ffn1 epilogue p={pdrop}', ffn1), (f'dsilu epilogue p={pdrop}', dsilu)):
                    ms = timed(fn, iters)
                    print(f'16s {lab:26s} M={M} N={N} K={K}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF', flush=True)


if __name__ == '__main__':
    main()
