"""Host-side cost per training operator call (tiny tensors, so the GPU is never the limit): where the ~17 us per launch go.

    python tools/host_overhead_probe.py"""
import ctypes as C
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from some_amd import _lib  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import Engine  # noqa: E402
from some_amd.training.ops import TrainOps  # noqa: E402


def per_call(fn, n=3000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


def main():
    ops = TrainOps(Engine(get_config('two_head_model', lay=1), device='cuda'))
    ops.set_mixed_precision(True, 'bf16')
    a = torch.randn(64, 512, device='cuda')
    b = torch.randn(64, 512, device='cuda')
    w = torch.randn(512, 512, device='cuda')
    bias = torch.randn(512, device='cuda')
    out = torch.empty_like(a)
    lib, h = ops.lib, ops.h
    st = ops.stream()
    pa, pb, po = C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr())
    print('raw ctypes call, prebuilt args          %6.2f us' % per_call(lambda: lib.some_train_eltwise(h, _lib.ELT_AXPY, pa, pb, po, a.numel(), 0.5, 0.0, C.c_uint64(0), st)))
    print('  + int pointers instead of c_void_p    %6.2f us' % per_call(lambda: lib.some_train_eltwise(h, _lib.ELT_AXPY, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), 0.5, 0.0, 0, st)))
    print('torch.empty_like                        %6.2f us' % per_call(lambda: torch.empty_like(a)))
    print('ops.stream()                            %6.2f us' % per_call(lambda: ops.stream()))
    print('ops.eltwise (alloc + stream + call)     %6.2f us' % per_call(lambda: ops.eltwise(_lib.ELT_AXPY, a, b, alpha=0.5)))
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    print('ops.axpy (autograd Function.apply)      %6.2f us' % per_call(lambda: ops.axpy(0.5, ar, br)))
    wr = w.clone().requires_grad_()
    print('ops.linear forward (autograd)           %6.2f us' % per_call(lambda: ops.linear(ar, wr, bias)))
    print('torch a + b (ATen, for scale)           %6.2f us' % per_call(lambda: a + b))

    def fb():
        y = ops.axpy(0.5, ops.axpy(0.5, ops.axpy(0.5, ar, br), br), br)
        y.backward(a)
    print('3 x axpy forward + backward             %6.2f us' % per_call(fb, 1000))


if __name__ == '__main__':
    main()
