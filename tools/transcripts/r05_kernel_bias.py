"""round 5: which kernel carries a SIGNED bias?  Each operator against its fp64 reference: mean error, error correlated with the
sign of the result (a multiplicative shrink / growth), rms error.  (A zero-mean error of rms r moves a 2584-frame cumsum by ~50 r,
a bias b by 2584 b: profiles/r05_experiments.md, "the bound stream's bias".)"""
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
import test_gpu_kernels as tk  # noqa: E402
from some_amd import _lib  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.engine import Engine  # noqa: E402

eng = Engine(get_config('midi_conformer', lay=1), device='cuda')
g = torch.Generator(device='cuda').manual_seed(1)


def report(name, out, ref64):
    e = out.double() - ref64
    rms = e.pow(2).mean().sqrt().item()
    scale = ref64.abs().mean().item()
    print(f'{name:58s} rms err {rms:.2e}  mean err {e.mean().item():+.2e} ({e.mean().item() / rms:+.3f} rms)  '
          f'mean err*sign(ref) {(e * ref64.sign()).mean().item():+.2e} ({(e * ref64.sign()).mean().item() / rms:+.3f} rms)  mean|ref| {scale:.2e}')


for (M, N, K) in ((4096, 512, 512), (4096, 512, 2048), (4096, 2048, 512)):
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    ref = A.double() @ W.double().t()
    report(f'gemm f32 mode   [{M}x{N}x{K}]', tk._gemm(eng, _lib.EPI_NONE, A, W), ref)
    report(f'gemm f16x3 mode [{M}x{N}x{K}]', tk._gemm(eng, _lib.EPI_NONE, A, W, split=True, tile=2), ref)
    # the same product with POSITIVE operands (every partial sum has one sign: a truncating accumulator shows as a bias)
    Ap, Wp = A.abs(), W.abs()
    refp = Ap.double() @ Wp.double().t()
    report(f'gemm f32 mode   [{M}x{N}x{K}] positive operands', tk._gemm(eng, _lib.EPI_NONE, Ap, Wp), refp)
    report(f'gemm f16x3 mode [{M}x{N}x{K}] positive operands', tk._gemm(eng, _lib.EPI_NONE, Ap, Wp, split=True, tile=2), refp)
# split / unsplit round trip of activations
x = torch.randn(4096, 512, device='cuda', generator=g) * 3
report('split_rows (hi + lo) round trip', tk._unsplit(tk._split(eng, x)), x.double())
# SiLU epilogue, SPLIT32 output
A = torch.randn(4096, 512, device='cuda', generator=g)
W = torch.randn(2048, 512, device='cuda', generator=g) / 20
b = torch.randn(2048, device='cuda', generator=g)
y = A.double() @ W.double().t() + b.double()
report('gemm_bias_silu f32 mode', tk._gemm(eng, _lib.EPI_BIAS_SILU, A, W, bias=b), torch.nn.functional.silu(y))
report('gemm_bias_silu f16x3 mode (SPLIT32 out)', tk._gemm(eng, _lib.EPI_BIAS_SILU, A, W, bias=b, split=True, tile=2, out_split=True), torch.nn.functional.silu(y))

# ---- the other operators -------------------------------------------------------------------------------------------------
import ctypes as C  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from some_amd.engine import ClipBatch  # noqa: E402

_p, _stream = tk._p, tk._stream
M = 4096
x = torch.randn(M, 512, device='cuda', generator=g) * 3 + 1.5
gamma, beta = torch.randn(512, device='cuda', generator=g), torch.randn(512, device='cuda', generator=g)
y, ys = torch.empty_like(x), torch.empty_like(x)
_lib.check(eng.handle, eng.lib.some_op_layernorm(eng.handle, _p(x), _p(gamma), _p(beta), _p(y), _p(ys), M, _stream()))
ref = F.layer_norm(x.double(), (512,), gamma.double(), beta.double(), eps=1e-5)
report('layernorm fp32 out', y, ref)
report('layernorm SPLIT32 out', tk._unsplit(ys), ref)
# GLU epilogue and sigmoid head
A = torch.randn(M, 512, device='cuda', generator=g)
Wg = torch.randn(1024, 512, device='cuda', generator=g) / 20
bg = torch.randn(1024, device='cuda', generator=g)
yy = A.double() @ Wg.double().t() + bg.double()
glu = yy[:, :512] * torch.sigmoid(yy[:, 512:])
bgi = tk._interleave_glu(bg[:, None])[:, 0].contiguous()
report('gemm_glu f32 mode', tk._gemm(eng, _lib.EPI_GLU, A, tk._interleave_glu(Wg), bias=bgi, n_out=512), glu)
report('gemm_glu f16x3 mode', tk._gemm(eng, _lib.EPI_GLU, A, tk._interleave_glu(Wg), bias=bgi, n_out=512, split=True, tile=2), glu)
W1 = torch.randn(1, 512, device='cuda', generator=g) / 10
b1 = torch.randn(1, device='cuda', generator=g)
z = torch.sigmoid(A.double() @ W1.double().t() + b1.double())
report('gemm_bias sigmoid head [512->1] f32 mode', tk._gemm(eng, _lib.EPI_BIAS, A, W1, bias=b1, act=1), z)
report('gemm_bias sigmoid head [512->1] f16x3 mode', tk._gemm(eng, _lib.EPI_BIAS, A, W1, bias=b1, act=1, split=True, tile=4), z)
# attention: exact-f32 op on a qkv array vs the split-f16 QKV projection + attention
lens = [2584]
batch = ClipBatch(lens, 'cuda')
T = 2584
h = torch.randn(T, 512, device='cuda', generator=g)
W = torch.randn(1536, 512, device='cuda', generator=g) / 512 ** 0.5
qkv64 = h.double() @ W.double().t()
q, k, v = (qkv64[:, i * 512:(i + 1) * 512].reshape(T, 8, 64).transpose(0, 1) for i in range(3))
ref = (torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1) @ v).transpose(0, 1).reshape(T, 512)
out = torch.empty(T, 512, device='cuda')
qkv32 = qkv64.float().contiguous()
_lib.check(eng.handle, eng.lib.some_op_attention(eng.handle, _p(qkv32), _p(batch.frame_offsets_dev), 1, T, _p(out), 0, _stream()))
report('attention f32 mode (fp32-rounded qkv in)', out, ref)
ws = torch.empty(eng.lib.some_op_qkv_attention_f16x3_bytes(T, 1), dtype=torch.uint8, device='cuda')
hs, Ws = tk._split(eng, h), tk._split(eng, W)
o3 = torch.empty(T, 512, device='cuda')
_lib.check(eng.handle, eng.lib.some_op_qkv_attention_f16x3(eng.handle, _p(hs), _p(Ws), _p(batch.frame_offsets_dev), 1, T, T, _p(o3), _p(ws), ws.numel(), _stream()))
report('qkv + attention f16x3 mode', tk._unsplit(o3), ref)
# dwconv + folded BN + SiLU
xx = torch.randn(T, 512, device='cuda', generator=g)
w = torch.randn(512, 1, 31, device='cuda', generator=g) / 5
bias = torch.randn(512, device='cuda', generator=g)
taps = w[:, 0, :].t().contiguous()
yd, yds = torch.empty(T, 512, device='cuda'), torch.empty(T, 512, device='cuda')
_lib.check(eng.handle, eng.lib.some_op_dwconv_silu(eng.handle, _p(xx), _p(taps), _p(bias), _p(batch.frame_offsets_dev), 1, T, _p(yd), 0, _stream()))
_lib.check(eng.handle, eng.lib.some_op_dwconv_silu(eng.handle, _p(xx), _p(taps), _p(bias), _p(batch.frame_offsets_dev), 1, T, _p(yds), 1, _stream()))
refd = F.silu(F.conv1d(xx.t()[None].double(), w.double(), bias.double(), padding=15, groups=512))[0].t()
report('dwconv_silu fp32 out', yd, refd)
report('dwconv_silu SPLIT32 out', tk._unsplit(yds), refd)
