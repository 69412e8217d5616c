"""Package power / shader clock under a looped kernel (tuning aid): is a matrix kernel schedule-bound or power-bound?

    python tools/power_probe.py --op attention|ffn1 [--seconds 4]

Runs the op back to back for ``--seconds`` on random operands and again on all-zero operands while a thread polls
``rocm-smi -P -c``; prints ms/launch and the sampled power / sclk lines for both.
"""
import argparse
import ctypes as C
import pathlib
import subprocess
import sys
import threading
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from some_amd import _lib  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import ClipBatch, Engine  # noqa: E402


def poll(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(['rocm-smi', '-P', '-c'], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:  # noqa: BLE001
            txt = repr(e)
        keep = [ln.strip() for ln in txt.splitlines() if 'ower' in ln or 'sclk' in ln]
        out.append(' | '.join(keep))
        time.sleep(0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--op', default='attention')
    ap.add_argument('--seconds', type=float, default=4.0)
    ap.add_argument('--tile', type=int, default=2)
    args = ap.parse_args()
    eng = Engine(get_config('midi_conformer', lay=0), device='cuda')
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def split(x):
        y = torch.empty_like(x)
        _lib.check(eng.handle, eng.lib.some_op_split_rows(eng.handle, p(x), p(y), x.shape[0], x.shape[1], st))
        return y

    for data in ('random', 'zeros'):
        gen = (lambda *s: torch.randn(*s, device='cuda')) if data == 'random' else (lambda *s: torch.zeros(*s, device='cuda'))
        if args.op == 'attention':
            batch = ClipBatch([2584] * 32, 'cuda')
            M = batch.total_frames
            hs, Ws = split(gen(M, 512)), split(gen(1536, 512) / 512 ** 0.5)
            out = torch.empty(M, 512, device='cuda')
            ws = torch.empty(eng.lib.some_op_qkv_attention_f16x3_bytes(M, batch.B), dtype=torch.uint8, device='cuda')

            def run():
                _lib.check(eng.handle, eng.lib.some_op_qkv_attention_f16x3(
                    eng.handle, p(hs), p(Ws), p(batch.frame_offsets_dev), batch.B, batch.max_frames, M, p(out), p(ws),
                    ws.numel(), st))
        else:
            M, N, K = 82688, 2048, 512
            A, W = split(gen(M, K)), split(gen(N, K) / K ** 0.5)
            bias, Cm = gen(N), torch.empty(M, N, device='cuda')
            flags = _lib.GEMM_SPLIT_IN | (args.tile << 8)

            def run():
                _lib.check(eng.handle, eng.lib.some_op_gemm(eng.handle, _lib.EPI_BIAS_SILU, p(A), K, p(W), p(bias), None, N,
                                                            p(Cm), N, M, N, K, 1.0, 0, None, flags, st))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        stop, samples = threading.Event(), []
        th = threading.Thread(target=poll, args=(stop, samples))
        th.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < args.seconds:
            for _ in range(50):
                run()
            torch.cuda.synchronize()
            n += 50
        dt = time.perf_counter() - t0
        stop.set()
        th.join()
        print(f'{args.op} operands={data}: {dt / n * 1e3:.4f} ms/launch over {n} launches')
        for s in samples:
            print('   ', s)


if __name__ == '__main__':
    main()
