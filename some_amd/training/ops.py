"""Autograd functions over the C-ABI training operators (include/some_amd.h, "training operators").

Activations are packed [M, C] fp32 row-major tensors on the GPU (M = B * T_max frames of the padded training batch).
PyTorch only records the tape and owns the memory; forward and backward math run in libsome_amd.so.  Each function
names the reference op it stands for."""
import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib
from ..engine import Engine


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()          # ctypes converts the int at the call (argtypes are c_void_p): no object per argument


class TrainOps:
    """Holds the library handle + a scratch buffer; every method is differentiable."""

    def __init__(self, engine: Engine):
        if engine.host_only or engine.device.type != 'cuda':
            raise RuntimeError('the training operators need an AMD GPU (no CPU fallback)')
        self.engine = engine
        # the library through the generated METH_FASTCALL wrappers (_lib.fast(): ~0.2 us of binding per call instead of ctypes' ~2 us at 20
        # arguments; ~510 calls per step), the handle and every stream as plain addresses.  Measured (r05al_train_ab.txt): the step does
        # not move (8.2 vs 8.2 ms in order, 7.7 vs 7.8 with the weight-gradient lanes; host enqueue 6.2 - 6.6 ms either way) - the
        # enqueue path's time is inside hipLaunchKernel and the allocator, and the step follows the device's dependent chain
        self.lib, self.device = _lib.fast(), engine.device
        self.h = getattr(engine.handle, 'value', engine.handle)
        self._scratch: Dict[int, torch.Tensor] = {}         # per lane: two lanes reduce through their scratch at the same time
        self._pinned_stream = None
        self.tape: Optional['Tape'] = None        # set by the trainer for the duration of a forward + backward pass
        self._partials: Dict[int, torch.Tensor] = {}
        self._size_cache: Dict[tuple, int] = {}
        self.device_prep = os.environ.get('SOME_AMD_TRAIN_DEVICE_PREP', '1') != '0'    # 0: round 4's torch-side operand preparation (A / B runs)
        # Two LANES (trainer's tape only): the midi and the bound stream of a Gcf layer are independent between the cross gates
        # (Gconform.py:82-87), forward and backward, so the model issues the bound stream's block on lane 1 = a second HIP stream.
        # At the reference's batch shape (8 phrases, ~4 100 frames) a step is ~860 launches of 5 - 35 us, most of them far from filling
        # 256 CUs (a GEMM is 66 workgroups): the device is busy the whole step although hardly used (profiles/r04_train_graph_probe.txt).
        # Tensors carry the lane that produced them on the tape; a consumer on the other lane waits on an event of the producer's
        # stream and tells the caching allocator (record_stream).  SOME_AMD_TRAIN_LANES=1: everything on one stream (A/B runs).
        self.lanes = 2 if os.environ.get('SOME_AMD_TRAIN_LANES', '2') != '1' else 1
        # Weight-gradient LANES (round 5, trainer's tape only): nothing downstream in a backward pass reads a weight gradient, so the
        # split-K weight-gradient GEMMs and their reductions (64 + 73 launches of the ~830 per step) need not sit in the dependent chain of
        # the data gradients.  Inside Tape.backward every lane is paired with a side stream (some_train_set_wgrad_stream): the library
        # issues the weight-gradient launches there behind an event of the lane; the operands stay referenced until the lanes have been
        # joined (end of the pass, or before a gradient bucket's all-reduce).  Same kernels and summation order: bit-identical gradients.
        # Measured at 8 x 520 frames, interleaved repetitions: 8.8 / 8.2 -> 7.7 / 8.1 ms per step (profiles/r05ai_train_ab.txt),
        # 8.0 / 8.2 / 8.3 -> 7.9 / 7.6 / 7.6 (r05al_train_ab.txt); parameters after 48 updates identical (r05ak_train_ab.txt digests).
        # SOME_AMD_TRAIN_WGRAD_LANES=0: weight gradients in the lanes' own order (A/B runs).
        self.wgrad_lanes = os.environ.get('SOME_AMD_TRAIN_WGRAD_LANES', '1') != '0'
        # ... =2: the reductions behind those GEMMs (planes -> the gradient arrays) are deferred too and go out in ONE table-driven launch
        # per side stream and join (some_train_wgrad_flush; 73 launches per step become 3): every weight gradient of a pass then gets
        # planes of its own out of one arena per side stream instead of one reused buffer.  Bit-identical as well, 694 launches per step
        # instead of 764 - and no faster (8.1 / 8.0 ms): the table launches read ~0.8 GB of planes back from HBM long after they were
        # written (156 us each; reduced one by one behind their GEMM they come out of the MALL), and the last one sits in front of the
        # gradient norm.  Off by default.
        self.wgrad_defer = os.environ.get('SOME_AMD_TRAIN_WGRAD_LANES', '1') == '2'
        self._wg_arena = [None, None]              # planes of the deferred reductions: [tensor, base address, bytes used]
        # The depthwise convolution's weight / bias gradients written by the kernels into the gradient arrays themselves (and with them on the
        # weight-gradient side stream) instead of tensors the tape adds.  SOME_AMD_TRAIN_DWCONV_SINKS=0: the tape's additions (A/B runs).
        # Measured at 8 x 520 frames with the lanes on (profiles/r05am_train_ab.txt, interleaved): 7.6 / 7.8 -> 7.4 / 7.3 ms per step.
        self.dwconv_sinks = os.environ.get('SOME_AMD_TRAIN_DWCONV_SINKS', '1') != '0'
        self._wg_streams = [None, None]            # side stream of lane i
        self._wg_active = False                    # inside Tape.backward with the pairs registered
        self._wg_pending = [False, False]          # lane i's side stream has work nobody has waited for
        self._wg_keep: list = []                   # operands of the weight-gradient launches in flight on the side streams
        # ... released as the side streams get through them, not at the end of the pass: every _WG_MARK_EVERY calls an event goes onto each
        # busy side stream, and the operands issued before a mark are dropped once its events have completed (peak backward memory
        # would otherwise grow with depth: ~1 GB at 8 x 520 frames, > 10 GB at max_batch_frames = 80 000 - ADVICE r05)
        self._wg_marks: list = []                  # [(entries of _wg_keep covered, [events])] in issue order
        self._wg_dropped = 0                       # entries of this pass already released (marks count from the start of the pass)
        self._wg_since_mark = 0
        self._wg_event_pool: list = []
        self._wg_events = [None, None]
        self._lane = 0
        self._lane_streams = [None, None]          # torch streams of the pinned step: [the caller's, the helper]
        self._lane_ptrs = [None, None]
        self._side_stream = None
        self._lane_version = [0, 0]                # operators enqueued per lane
        self._lane_seen = {(0, 1): -1, (1, 0): -1} # (producer, consumer) -> the producer's version the consumer has waited for
        self._fork_cover = None                    # inside lane(i, after=fork): (the fork's lane, its version at the fork point)
        self.gemm_precision = 'f32' if engine.c_config.precision == _lib.PRECISION_F32 else 'f16x3'     # (f16x3_fast only changes the INFERENCE attention kernel)
        self.attention_precision = self.gemm_precision      # forward + backward: split-f16 or exact-f32 MFMA kernels
        self._hi = 0                                        # GEMM flag bits of the one-product modes (set_mixed_precision)
        self._hi_mode = 0                                   # the `hi_only` argument: 0 three products, 1 f16, 2 bf16
        self.operand = 'f16x2'
        # GEMMs on the fp32 arrays as they lie, rounded / split (and transposed) in the kernel's staging path (some_train_gemm16)
        # instead of split_rows / transpose passes + the SPLIT32 kernels; SOME_AMD_TRAIN_GEMM16=0: A/B runs
        self.gemm16 = os.environ.get('SOME_AMD_TRAIN_GEMM16', '1') != '0'
        # Gradient sinks: parameters whose gradient ARRAYS the backward kernels write themselves (FlatParams' views of the flat
        # gradient buffer), accumulating in place - nn.Linear weights / biases (some_train_gemm16_wgrad) and LayerNorm gamma / beta.
        # Their backward returns None for the parameter, so autograd launches no copy and no accumulation kernel for it
        # (291 + 126 launches per step before); ``on_grad_ready(param)`` stands in for the post-accumulate hook (gradient sync).
        self._sinks: Dict[int, torch.Tensor] = {}
        self.on_grad_ready = None
        self.use_sinks = os.environ.get('SOME_AMD_TRAIN_SINKS', '1') != '0'
        # Mixed precision only: the FFN's [M, 2048] intermediates live in memory as 16-bit arrays written / read by GEMM epilogues
        # (csrc/train_gemm16s.hip) and its four matrix products read 16-bit operands through the DMA-ring kernel; the weights' 16-bit
        # images are re-derived when ``weights_version`` moves (the model bumps it at the start of every forward pass).
        # SOME_AMD_TRAIN_FFN16=0: the composition of linear / silu_dropout / linear on fp32 arrays (A/B runs).
        self.ffn16 = os.environ.get('SOME_AMD_TRAIN_FFN16', '1') != '0'
        self.weights_version = 0
        self._shadows: Dict[int, tuple] = {}
        self._joined: Dict[tuple, tuple] = {}
        self._shadow_tables: Dict[tuple, tuple] = {}

    def set_mixed_precision(self, on: bool, operand: str = 'f16'):
        """Mixed-precision training (the reference's pl_trainer_precision '16-mixed' / 'bf16'): the matrix products read one
        16-bit value per element - ``operand`` 'f16' (the hi halves of the split operands; 11-bit significand, needs loss
        scaling) or 'bf16' (the reference's bf16 autocast arithmetic: 8-bit significand, fp32 range) - with fp32 accumulation, one
        MFMA product instead of three, while parameters, activations, reductions, softmax statistics and the optimiser stay fp32."""
        if on and self.gemm_precision != 'f16x3':
            raise ValueError('mixed precision runs on the split-f16 kernels: use some_amd_precision f16x3')
        if operand not in ('f16', 'bf16'):
            raise ValueError(f"mixed-precision operand must be 'f16' or 'bf16', got {operand!r}")
        bf16 = on and operand == 'bf16'
        self._hi = (_lib.GEMM_HI_ONLY | (_lib.GEMM_HI_BF16 if bf16 else 0)) if on else 0
        self._hi_mode = (2 if bf16 else 1) if on else 0
        self.operand = 'bf16' if bf16 else 'f16x2'

    # ---- gradient sinks ---------------------------------------------------------------------------------------
    def register_grad_sinks(self, params, on_grad_ready=None):
        """params: iterable of leaf tensors whose ``.grad`` is preallocated (a contiguous fp32 view that is zeroed every step)."""
        self._sinks = {id(p): p.grad for p in params}
        self.on_grad_ready = on_grad_ready

    def sink(self, param) -> Optional[torch.Tensor]:
        if param is None or not self.use_sinks:
            return None
        return self._sinks.get(id(param))

    def deposited(self, param):
        if self.on_grad_ready is not None:
            self.on_grad_ready(param)

    def can_wgrad_into(self, N: int, K: int) -> bool:
        # N % 4: dY is read with ld = N and the C entry points require ld % 4 == 0 (16-byte rows) - N = 130 bins takes the transpose path
        return self.gemm16 and self.gemm_precision == 'f16x3' and N >= 32 and N % 4 == 0 and K % 4 == 0 and K >= 32

    def gemm_dw_into(self, dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor]):
        """dw [N, K] += dy^T x, db [N] += column sums of dy, straight into the gradient arrays (some_train_gemm16_wgrad)."""
        M, N = dy.shape
        K = x.shape[1]
        part, part_bytes = self.wgrad_planes(self._bytes('some_train_gemm16_bytes', N, K, M, K + 4))
        self.check(self.lib.some_train_gemm16_wgrad(self.h, _p(dy), N, _p(x), K, _p(dw), _p(db), N, K, M, self._op16, 1, part,
                                                    part_bytes, self.stream()))
        self.wgrad_issued(dy, x)

    # ---- 16-bit stored operands (mixed precision) ---------------------------------------------------------------------
    @property
    def dtype16(self):
        return torch.bfloat16 if self._hi_mode == 2 else torch.float16

    def cast16(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 -> the mode's 16-bit format, round to nearest even (some_train_cast16)."""
        out = torch.empty(x.shape, dtype=self.dtype16, device=self.device)
        if x.numel() % 8 or x.data_ptr() % 16:
            return out.copy_(x)
        self.check(self.lib.some_train_cast16(self.h, _p(x), _p(out), x.numel(), self._hi_mode, self.stream()))
        return out

    def shadow16(self, w: torch.Tensor):
        """(W16 [N, K], W16T [K, N]) of the fp32 weight ``w`` [N, K], cached until ``weights_version`` moves."""
        key = id(w)
        hit = self._shadows.get(key)
        if hit is not None and hit[0] == (self.weights_version, self._hi_mode, w.data_ptr()):
            return hit[1], hit[2]
        N, K = w.shape[0], w[0].numel()                                  # k = 1 Conv1d weights are [N, K, 1]
        if hit is not None and hit[1].dtype == self.dtype16 and hit[1].shape == (N, K):
            w16, w16t = hit[1], hit[2]                                   # refreshed in place: same buffers every step
        else:
            w16 = torch.empty((N, K), dtype=self.dtype16, device=self.device)
            w16t = torch.empty((K, N), dtype=self.dtype16, device=self.device)
        self.check(self.lib.some_train_transpose16(self.h, _p(w), _p(w16), _p(w16t), N, K, self._hi_mode, self.stream()))
        self._shadows[key] = ((self.weights_version, self._hi_mode, w.data_ptr()), w16, w16t, w)   # keeps ``w`` alive: id() stays unique
        return w16, w16t

    def prepare_shadows(self, weights):
        """Refresh the 16-bit images of ``weights`` (fp32 [N, K] or [N, K, 1] tensors whose storage does not move) in ONE launch
        (some_train_transpose16_table) and mark them current for ``weights_version``."""
        if not weights or self._hi_mode not in (1, 2):
            return
        key = (tuple(id(w) for w in weights), self._hi_mode)
        hit = self._shadow_tables.get(key)
        if hit is None or any(w.data_ptr() != q for w, q in zip(weights, hit[3])):
            rows, max_n, max_k = [], 0, 0
            for w in weights:
                N, K = w.shape[0], w[0].numel()
                w16 = torch.empty((N, K), dtype=self.dtype16, device=self.device)
                w16t = torch.empty((K, N), dtype=self.dtype16, device=self.device)
                self._shadows[id(w)] = [None, w16, w16t, w]
                rows.append([w.data_ptr(), w16.data_ptr(), w16t.data_ptr(), N, K])
                max_n, max_k = max(max_n, N), max(max_k, K)
            hit = (torch.tensor(rows, dtype=torch.int64, device=self.device), max_n, max_k, [w.data_ptr() for w in weights])
            self._shadow_tables = {key: hit}                                          # one model, one table
        self.check(self.lib.some_train_transpose16_table(self.h, _p(hit[0]), len(weights), hit[1], hit[2], self._hi_mode, self.stream()))
        stamp = (self.weights_version, self._hi_mode)
        for w in weights:
            e = self._shadows[id(w)]
            self._shadows[id(w)] = ((stamp[0], stamp[1], w.data_ptr()), e[1], e[2], w)

    def gemm16s(self, epi: int, a16: torch.Tensor, b16: torch.Tensor, bias, out: torch.Tensor, ldc: int, M: int, N: int, K: int,
                h16: Optional[torch.Tensor] = None, plane: int = 0, p: float = 0.0, seed: int = 0, alpha: float = 1.0):
        """out = epilogue(a16 [M, K] @ b16 [N, K]^T): some_train_gemm16s (0 fp32 (+ bias), 1 FFN first linear, 2 SiLU / dropout gradient,
        3 residual ``h16`` (fp32) + alpha * dropout(. + bias))."""
        self.check(self.lib.some_train_gemm16s(self.h, epi, _p(a16), a16.stride(0), _p(b16), b16.stride(0), _p(bias), _p(out), ldc, _p(h16),
                                               h16.stride(0) if h16 is not None else 0, plane, M, N, K, self._hi_mode, float(p), seed, float(alpha),
                                               self.stream()))

    def wgrad16(self, dy16: torch.Tensor, x16: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor], accumulate: bool):
        """dw [N, K] (+)= dy16^T x16, db [N] (+)= column sums of dy16 (some_train_gemm16_wgrad16)."""
        M, N = dy16.shape
        K = x16.shape[1]
        part, part_bytes = self.wgrad_planes(self._bytes('some_train_gemm16_bytes', N, K, M, K + 4))
        self.check(self.lib.some_train_gemm16_wgrad16(self.h, _p(dy16), dy16.stride(0), _p(x16), x16.stride(0), _p(dw), _p(db), N, K, M, self._hi_mode, int(accumulate),
                                                      part, part_bytes, self.stream()))
        self.wgrad_issued(dy16, x16)
        if not accumulate:
            self.join_wgrad()                # a fresh array the tape adds on this lane: no run-ahead (not the default path)

    def silu16(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty(x.shape, dtype=self.dtype16, device=self.device)
        self.check(self.lib.some_train_silu16(self.h, _p(x), _p(out), x.numel(), self._hi_mode, self.stream()))
        return out

    def dropcast16(self, d: torch.Tensor, alpha: float, p: float, seed: int) -> torch.Tensor:
        """rn16(alpha * mask / (1 - p) * d): the gradient through ``alpha * dropout(y) + x`` (gemm16s epilogue 3) w.r.t. y."""
        M, N = d.shape
        out = torch.empty((M, N), dtype=self.dtype16, device=self.device)
        self.check(self.lib.some_train_dropcast16(self.h, _p(d), _p(out), M, N, float(alpha), float(p), seed, self._hi_mode, self.stream()))
        return out

    def joined(self, a: torch.Tensor, b: torch.Tensor):
        """(weights, gradient sink) of two parameters that lie back to back in the flat buffers - to_q | to_kv of an attention module - as ONE
        [Na + Nb, K] matrix, or None when they are not adjacent / have no sinks."""
        key = (id(a), id(b))
        hit = self._joined.get(key)
        if hit is None:
            sa, sb = self.sink(a), self.sink(b)
            ok = (sa is not None and sb is not None and a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1] and a.is_contiguous() and b.is_contiguous()
                  and b.data_ptr() == a.data_ptr() + a.numel() * 4 and sb.data_ptr() == sa.data_ptr() + sa.numel() * 4)
            if ok:
                shape, stride = (a.shape[0] + b.shape[0], a.shape[1]), (a.shape[1], 1)
                hit = (torch.as_strided(a.detach(), shape, stride), torch.as_strided(sa, shape, stride), a, b)
            else:
                hit = (None, None, a, b)
            self._joined[key] = hit
        return hit[0], hit[1]

    def can_block16(self, x: torch.Tensor, params) -> bool:
        """The fused attention / conv sub-blocks need mixed precision, 512 channels and every Linear / LayerNorm parameter's gradient sink."""
        return (self.ffn16 and self._hi_mode in (1, 2) and self.gemm16 and x.shape[0] >= 64 and x.shape[1] == 512 and 4 * x.shape[0] * 2048 < 0x7fffffff
                and all(self.sink(q) is not None for q in params))

    def can_ffn16(self, M: int, K: int, H: int, N: int) -> bool:
        return (self.ffn16 and self._hi_mode in (1, 2) and self.gemm16 and M >= 64 and K % 32 == 0 and H % 32 == 0 and N % 32 == 0
                and 4 * M * H < 0x7fffffff)

    # ---- plumbing -------------------------------------------------------------------------------------------
    def stream(self):
        # looking the current stream up costs 3.4 us - a third of an operator call's host time (tools/host_overhead_probe.py); a training
        # step runs on the caller's stream (+ the helper stream of lane 1), so the trainer pins them for the duration of the step
        if self._pinned_stream is not None:
            return self._lane_ptrs[self._lane]
        return torch.cuda.current_stream(self.device).cuda_stream

    def pin_stream(self):
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(self.device)
        self._lane_streams = [main, self._side_stream]
        self._lane_ptrs = [main.cuda_stream, self._side_stream.cuda_stream]           # addresses (0: the NULL stream)
        self._pinned_stream = self._lane_ptrs[0]
        self._lane = 0
        self._lane_version = [0, 0]
        self._lane_seen = {(0, 1): -1, (1, 0): -1}

    def unpin_stream(self):
        if self._pinned_stream is not None and self._lane_streams[0] is not None:
            torch.cuda.set_stream(self._lane_streams[0])           # (an exception inside a lane must not leave torch on the helper stream)
        self._pinned_stream = None
        self._lane = 0
        self._fork_cover = None

    # ---- lanes ---------------------------------------------------------------------------------------------------
    def two_lanes(self) -> bool:
        return self.lanes == 2 and self.tape is not None and self._pinned_stream is not None

    def fork_point(self):
        """An event on the CURRENT lane's stream at this point of the enqueue order (None when lanes are off): ``lane(i, after=...)``
        starts lane i behind it - and not behind what is enqueued on this lane later."""
        if not self.two_lanes():
            return None
        ev = torch.cuda.Event()
        ev.record(self._lane_streams[self._lane])
        return ev, self._lane, self._lane_version[self._lane]

    def lane(self, index: int, after=None):
        """Context: the operators issued inside run on lane ``index`` (its HIP stream, its allocator pool, its scratch)."""
        return _Lane(self, index if self.two_lanes() else self._lane, after)

    def wait_lane(self, producer: int, force: bool = False):
        """The current lane waits for everything enqueued on ``producer`` so far (skipped when it already has)."""
        me = self._lane
        if producer == me:
            return
        if not force and self._lane_seen[(producer, me)] == self._lane_version[producer]:
            return
        ev = torch.cuda.Event()
        ev.record(self._lane_streams[producer])
        self._lane_streams[me].wait_event(ev)
        self._lane_seen[(producer, me)] = self._lane_version[producer]

    def adopt(self, t: torch.Tensor, producer: int, version: int = -1):
        """Make ``t`` (produced on lane ``producer`` as its ``version``-th operator) safe to use on the current lane: order the streams -
        unless the lane was started behind a fork point that already covers the producer - and tell the caching allocator that the
        tensor's memory is in use on this lane's stream too."""
        if producer != self._lane:
            cover = self._fork_cover
            if not (cover is not None and cover[0] == producer and 0 <= version <= cover[1]):
                self.wait_lane(producer)
            t.record_stream(self._lane_streams[self._lane])

    def sync_other_lane(self):
        """The current lane waits for the other one (before a collective that reads what both have written)."""
        if self._pinned_stream is not None and self.lanes == 2 and self._lane_streams[1] is not None:
            self.wait_lane(1 - self._lane, force=True)
        self.join_wgrad()

    def join_lanes(self):
        """Lane 0 waits for lane 1 (end of the backward pass: the gradient norm and the optimiser run on the caller's stream)."""
        if self._pinned_stream is not None and self._lane_version[1] > 0:
            keep, self._lane = self._lane, 0
            self.wait_lane(1, force=True)
            self._lane = keep

    def _enter_lane(self, index: int):
        """Switch the lane operators are issued on; torch's own kernels and allocations follow (current stream)."""
        self._lane = index
        torch.cuda.set_stream(self._lane_streams[index])

    def _bytes(self, fn, *key) -> int:
        """A sizing entry point of the library, memoised per argument tuple (140 of these calls per step at 2 us each otherwise)."""
        k = (fn,) + key
        v = self._size_cache.get(k)
        if v is None:
            v = self._size_cache[k] = int(getattr(self.lib, fn)(self.h, *key))
        return v

    def scratch(self, M: int, N: int) -> torch.Tensor:
        need = self._bytes('some_train_scratch_bytes', M, N)
        buf = self._scratch.get(self._lane)
        if buf is None or buf.numel() < need:
            buf = self._scratch[self._lane] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return buf

    def partial(self, need: int, wgrad: bool = False) -> torch.Tensor:
        """Partial-plane buffer of the split-K GEMMs, one per lane; the weight-gradient entry points (which may run on the lane's side
        stream, see ``wgrad_lanes``) have their own, so that a call that stays on the lane never shares planes with one that left it."""
        key = (self._lane, wgrad)
        buf = self._partials.get(key)
        if buf is None or buf.numel() < need:
            if buf is not None and self._wg_active:
                self._wg_keep.append(buf)          # the side stream may still be reducing out of the old planes
            buf = self._partials[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return buf

    def wgrad_scratch(self, need: int) -> torch.Tensor:
        """Reduction scratch of a parameter-gradient call that may run on the lane's side stream (the depthwise convolution's tap and bias
        sums): one block per lane that no lane-resident call touches - the calls that use it serialise on the side stream."""
        key = (self._lane, 'scratch')
        buf = self._partials.get(key)
        if buf is None or buf.numel() < need:
            if buf is not None and self._wg_active:
                self._wg_keep.append(buf)
            buf = self._partials[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return buf

    # ---- weight-gradient lanes ---------------------------------------------------------------------------------------
    def wgrad_planes(self, need: int):
        """(pointer, bytes) of the partial planes of ONE weight-gradient call on the current lane: the lane's reused weight-gradient
        buffer - or, while reductions are deferred, a slice of the side stream's arena that nothing else touches until the flush.  A full
        arena is flushed and reused from its start (the side stream runs the flush before the next GEMM writes there)."""
        if not (self._wg_active and self.wgrad_defer):
            buf = self.partial(need, wgrad=True)
            return buf.data_ptr(), buf.numel()
        need = (need + 255) // 256 * 256
        lane = self._lane
        arena = self._wg_arena[lane]
        if arena is None or arena[2] + need > arena[0].numel():
            if arena is not None and need <= arena[0].numel():
                self.check(self.lib.some_train_wgrad_flush(self.h, self._lane_ptrs[lane]))
                arena[2] = 0
            else:
                if arena is not None:
                    self._wg_keep.append(arena[0])     # waiting reductions still read it
                buf = torch.empty(max(need, 1 << 30), dtype=torch.uint8, device=self.device)     # (~0.8 GB of planes per lane and pass at lay 3)
                arena = self._wg_arena[lane] = [buf, buf.data_ptr(), 0]
        off = arena[2]
        arena[2] = off + need
        return arena[1] + off, need

    def begin_wgrad_lanes(self):
        """Tape.backward: pair every lane's stream with its side stream for the duration of the pass."""
        if not self.wgrad_lanes or self._pinned_stream is None or self._wg_active:
            return
        try:
            for i in range(self.lanes):
                if self._wg_streams[i] is None:
                    self._wg_streams[i] = torch.cuda.Stream(self.device)
                    self._wg_events[i] = torch.cuda.Event()
                self.check(self.lib.some_train_set_wgrad_stream(self.h, self._lane_ptrs[i], self._wg_streams[i].cuda_stream,
                                                                1 if self.wgrad_defer else 0))
                if self._wg_arena[i] is not None:
                    self._wg_arena[i][2] = 0        # the previous pass's planes were reduced on the same side stream: reuse in order
        except Exception:
            for i in range(self.lanes):             # a pairing without its join would let weight gradients run ahead of their readers
                self.lib.some_train_set_wgrad_stream(self.h, self._lane_ptrs[i], None, 0)
            raise
        self._wg_active = True

    def wgrad_issued(self, *operands):
        """Bookkeeping behind a weight-gradient call on the current lane: its operands must outlive the side stream's use of them (they
        were allocated on the lane's stream: freed now, the caching allocator would hand them to the lane's next allocation)."""
        if self._wg_active:
            self._wg_pending[self._lane] = True
            self._wg_keep.append(operands)
            self._wg_since_mark += 1
            if self._wg_since_mark >= self._WG_MARK_EVERY:
                self._wg_mark()

    _WG_MARK_EVERY = int(os.environ.get('SOME_AMD_TRAIN_WG_MARK_EVERY', '16'))     # calls between two release marks (A/B runs: a huge value = release at the end of the pass)

    def _wg_mark(self):
        """An event on every side stream that has work; everything kept so far may go once those events have completed (the side
        streams run in order, so a completed mark covers every call issued before it).  Completed marks at the head are retired here."""
        self._wg_since_mark = 0
        evs = []
        for i in range(self.lanes):
            if self._wg_pending[i]:
                ev = self._wg_event_pool.pop() if self._wg_event_pool else torch.cuda.Event()
                ev.record(self._wg_streams[i])
                evs.append(ev)
        self._wg_marks.append((self._wg_dropped + len(self._wg_keep), evs))
        while self._wg_marks and all(e.query() for e in self._wg_marks[0][1]):
            covered, done = self._wg_marks.pop(0)
            del self._wg_keep[:covered - self._wg_dropped]
            self._wg_dropped = covered
            self._wg_event_pool.extend(done)

    def join_wgrad(self):
        """The current lane waits for the weight-gradient launches issued so far on every lane's side stream."""
        if not self._wg_active:
            return
        me = self._lane_streams[self._lane]
        for i in range(self.lanes):
            if self._wg_pending[i]:
                if self.wgrad_defer:
                    self.check(self.lib.some_train_wgrad_flush(self.h, self._lane_ptrs[i]))
                ev = self._wg_events[i]
                ev.record(self._wg_streams[i])
                me.wait_event(ev)

    def end_wgrad_lanes(self):
        """End of Tape.backward (lane 0 current): lane 0 waits for the side streams - the gradient norm and the optimiser read what they
        wrote -, the pairs are removed and the operands released (every later use of their memory is enqueued behind lane 0 from here on)."""
        if not self._wg_active:
            return
        try:
            self.join_wgrad()
        finally:
            for i in range(self.lanes):
                self.lib.some_train_set_wgrad_stream(self.h, self._lane_ptrs[i], None, 0)
            self._wg_active = False
            self._wg_pending = [False, False]
            self._wg_keep.clear()
            for _, evs in self._wg_marks:
                self._wg_event_pool.extend(evs)
            self._wg_marks.clear()
            self._wg_dropped = self._wg_since_mark = 0

    def check(self, rc):
        if rc:
            _lib.check(self.h, rc)

    def new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    # ---- raw (non-differentiable) helpers -------------------------------------------------------------------------
    def split_rows(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 [R, K] -> SPLIT32 rows (csrc/split.h: per 32-element k-block 32 f16 hi | 32 f16 lo; same byte size)."""
        out = torch.empty_like(x)
        fmt = _lib.OPERAND_BF16 if self._hi_mode == 2 else _lib.OPERAND_F16X2          # bf16 mode: hi slot = bf16(x), lo = 0
        self.check(self.lib.some_op_split_rows_fmt(self.h, _p(x), _p(out), x.shape[0], x.shape[1], fmt, self.stream()))
        return out

    def split_transpose(self, x: torch.Tensor, pad_to: int = 64):
        """(split_rows(x), transpose(x, pad_to, split=True)) from one pass over x [M, N] - the two operand layouts of the split attention kernels."""
        M, N = x.shape
        Mp = (M + pad_to - 1) // pad_to * pad_to
        rows, t = torch.empty_like(x), self.new(N, Mp)
        fmt = _lib.OPERAND_BF16 if self._hi_mode == 2 else _lib.OPERAND_F16X2
        self.check(self.lib.some_train_split_transpose(self.h, _p(x), M, N, _p(rows), _p(t), Mp, fmt, self.stream()))
        return rows, t

    @staticmethod
    def _tile(M: int, N: int) -> int:
        """f16x3 GEMM tile selector (SOME_GEMM_TILE)."""
        # measured at 8 x 2584 frames: 128 x 128 tiles for the N = 512 GEMMs change nothing (92.3 vs 90.9 ms/step) - keep 256 x 256
        return 2

    def _use_split(self, M: int, N: int, K: int) -> bool:
        return self.gemm_precision == 'f16x3' and K % 32 == 0 and N >= 64 and M >= 64

    def _use16(self, M: int, N: int, K: int) -> bool:
        return self.gemm16 and self._use_split(M, N, K) and N % 4 == 0

    @property
    def _op16(self) -> int:
        """some_train_gemm16 operand code: 1 f16 / 2 bf16 (one product), 3 split f16 (three products, fp32-equivalent)."""
        return self._hi_mode or 3

    def gemm(self, a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """a [M, K] @ w[N, K]^T (+ bias).  ``gemm_precision`` 'f16x3' (default): both operands are split into f16
        hi + lo halves and contracted with three f16 MFMA products, fp32 accumulation (gemm_f16x3.hip; fp32-equivalent,
        see DESIGN.md section 4) whenever K % 32 == 0 and the problem is big enough to matter; otherwise - and with
        'f32' - the exact fp32 MFMA kernel (gemm.hip), K padded to a multiple of 4."""
        M, K = a.shape
        N = w.shape[0]
        assert w.shape[1] == K and a.is_contiguous() and w.is_contiguous()
        out = self.new(M, N)
        if M == 0:
            return out
        epi = _lib.EPI_BIAS if bias is not None else _lib.EPI_NONE
        if self._use16(M, N, K):
            self.check(self.lib.some_train_gemm16(self.h, _p(a), K, 0, _p(w), K, 0, _p(bias), _p(out), N, M, N, K, self._op16, -1, None, 0,
                                                  self.stream()))
            return out
        if self._use_split(M, N, K):
            a3 = self.split_rows(a)
            w3 = self.split_rows(w)
            self.check(self.lib.some_op_gemm(self.h, epi, _p(a3), K, _p(w3), _p(bias), None, N, _p(out), N, M, N, K, 1.0, 0, None,
                                             _lib.GEMM_SPLIT_IN | self._hi | (self._tile(M, N) << 8), self.stream()))
            return out
        if K % 4:
            pad = 4 - K % 4
            a = torch.nn.functional.pad(a, (0, pad))
            w = torch.nn.functional.pad(w, (0, pad))
            K += pad
        self.check(self.lib.some_op_gemm(self.h, epi, _p(a), K, _p(w), _p(bias), None, N, _p(out), N, M, N, K, 1.0, 0, None, 0,
                                         self.stream()))
        return out

    def gemm_dx(self, dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """dy [M, N] @ w [N, K]: the data gradient of ``linear`` (the weight is transposed - and split - on the fly)."""
        M, N = dy.shape
        K = w.shape[1]
        if self._use16(M, K, N):                                        # w [N, K] is the contraction-major operand as it lies
            out = self.new(M, K)
            self.check(self.lib.some_train_gemm16(self.h, _p(dy), N, 0, _p(w), K, 1, None, _p(out), K, M, K, N, self._op16, -1, None, 0,
                                                  self.stream()))
            return out
        if self._use_split(M, K, N):
            wt3 = self.transpose(w, pad_to=32, split=True)
            a3 = self.split_rows(dy)
            out = self.new(M, K)
            self.check(self.lib.some_op_gemm(self.h, _lib.EPI_NONE, _p(a3), N, _p(wt3), None, None, K, _p(out), K, M, K, N, 1.0, 0, None,
                                             _lib.GEMM_SPLIT_IN | self._hi | (self._tile(M, K) << 8), self.stream()))
            return out
        return self.gemm(dy, self.transpose(w, pad_to=1))

    def gemm_dw(self, dy: torch.Tensor, x: torch.Tensor, with_bias: bool):
        """Weight (and bias) gradient of ``linear``: dW [N, K] = dy^T x contracted over the M frames, db = dy^T 1 obtained
        from the SAME GEMM by appending a row of ones to x^T.  Both operands come out of the transpose kernel already in
        SPLIT32 form in f16x3 mode; the long contraction is cut across workgroups (split-K)."""
        M, N = dy.shape
        K = x.shape[1]
        if N == 1:                                                      # one output (the bound head): a weighted column sum of x
            out, ws = self.new(1, K), self.new(K)
            sc = self.scratch(M, K)
            self.check(self.lib.some_train_weighted_colsum(self.h, _p(dy), 1, _p(x), M, K, K, _p(out), _p(ws), _p(sc), sc.numel(), self.stream()))
            return out, (ws[:1].clone() if with_bias else None)
        if self.can_wgrad_into(N, K):
            # dy [M, N] and x [M, K] are both contraction-major as they lie; the bias gradient is the fp32 column sum of dy,
            # accumulated in the kernel's staging registers into column K of the output
            ldc = K + (4 if with_bias else 0)
            out = self.new(N, ldc)
            part = self.partial(self._bytes('some_train_gemm16_bytes', N, K, M, ldc))
            self.check(self.lib.some_train_gemm16(self.h, _p(dy), N, 1, _p(x), K, 1, None, _p(out), ldc, N, K, M, self._op16,
                                                  K if with_bias else -1, _p(part), part.numel(), self.stream()))
            if not with_bias:
                return out, None
            return out[:, :K].contiguous(), out[:, K].contiguous()
        Mp = (M + 31) // 32 * 32
        extra = 4 if with_bias else 0                                   # ones row + 3 zero rows keep (K + extra) % 4 == 0
        use3 = self.gemm_precision == 'f16x3' and N >= 32 and K % 4 == 0
        xt = torch.empty((K + extra, Mp), dtype=torch.float32, device=self.device)
        split = (2 if self._hi_mode == 2 else 1) if use3 else 0         # SPLIT32 slots: f16 hi + lo, or bf16 hi
        self.check(self.lib.some_train_transpose(self.h, _p(x), M, K, K, _p(xt), Mp, split, self.stream()))
        if with_bias:
            tail = xt[K:]
            tail.zero_()
            if use3:                                                    # 1.0 in the hi halves (f16 0x3C00 / bf16 0x3F80), lo halves 0
                tail[0].view(torch.int32).view(-1, 32)[:, :16] = 0x3F803F80 if split == 2 else 0x3C003C00
            else:
                tail[0].fill_(1.0)
        dyt = torch.empty((N, Mp), dtype=torch.float32, device=self.device)
        self.check(self.lib.some_train_transpose(self.h, _p(dy), M, N, N, _p(dyt), Mp, split, self.stream()))
        Kx = K + extra
        out = self.new(N, Kx)
        if use3:
            part = self.partial(self._bytes('some_train_gemm_splitk_bytes', N, Kx, Mp))
            self.check(self.lib.some_train_gemm_splitk(self.h, _p(dyt), Mp, _p(xt), _p(out), N, Kx, Mp, self._hi_mode, _p(part), part.numel(),
                                                       self.stream()))
        else:
            self.check(self.lib.some_op_gemm(self.h, _lib.EPI_NONE, _p(dyt), Mp, _p(xt), None, None, Kx, _p(out), Kx, N, Kx, Mp, 1.0, 0, None, 0,
                                             self.stream()))
        if not with_bias:
            return out, None
        return out[:, :K].contiguous(), out[:, K].contiguous()

    def transpose(self, x: torch.Tensor, pad_to: int = 32, split: bool = False) -> torch.Tensor:
        """[M, N] -> [N, Mp] with Mp = M rounded up to ``pad_to`` (zero padded): a contraction operand over M;
        ``split``: written directly in SPLIT32 form."""
        M, N = x.shape
        Mp = (M + pad_to - 1) // pad_to * pad_to
        out = self.new(N, Mp)
        self.check(self.lib.some_train_transpose(self.h, _p(x), M, N, N, _p(out), Mp, (2 if self._hi_mode == 2 else 1) if split else 0, self.stream()))
        return out

    def colsum(self, x: torch.Tensor) -> torch.Tensor:
        M, N = x.shape
        out = self.new(N)
        sc = self.scratch(M, N)
        self.check(self.lib.some_train_colsum(self.h, _p(x), M, N, N, _p(out), 0, _p(sc), sc.numel(), self.stream()))
        return out

    def eltwise(self, op: int, a: torch.Tensor, b: Optional[torch.Tensor] = None, alpha: float = 0.0, seed: int = 0,
                p: float = 0.0) -> torch.Tensor:
        out = torch.empty_like(a)
        self.check(self.lib.some_train_eltwise(self.h, op, _p(a), _p(b), _p(out), a.numel(), float(alpha), float(p), seed,
                                               self.stream()))
        return out

    # ---- differentiable operators ---------------------------------------------------------------------------------------
    def _apply(self, fn, *args):
        """Run a differentiable operator: on the active tape (trainer, see ``Tape``) or through torch.autograd."""
        tape = self.tape
        return tape.apply(fn, self, *args) if tape is not None else fn.apply(self, *args)

    def cat_rows(self, a, b):
        """torch.cat([a, b], dim=0) of two weight matrices (the fused q | k | v projection)."""
        return self._apply(_CatRows, a, b)

    def reshape(self, x, *shape):
        return self._apply(_Reshape, x, shape)

    def linear(self, x, weight, bias=None):
        """nn.Linear / k = 1 Conv1d: x [M, K], weight [N, K] -> [M, N]."""
        return self._apply(_Linear, x, weight, bias)

    def ffn(self, x, w1, b1, w2, b2, p: float, seed: int):
        """conform_ffn.forward up to its output dropout (Gconform.py:29-33): ln2(drop1(silu(ln1(x))))."""
        if self.can_ffn16(x.shape[0], x.shape[1], w1.shape[0], w2.shape[0]):
            return self._apply(_Ffn16, x, w1, b1, w2, b2, p, seed)
        return self.linear(self.silu_dropout(self.linear(x, w1, b1), p, seed), w2, b2)

    def ffn_block(self, x, gamma, beta, w1, b1, w2, b2, alpha: float, p_latent: float, seed_latent: int, p_out: float, seed_out: int):
        """One FFN sub-block of conform_blocke.forward (Gconform.py:57,60): ``x + alpha * dropout(ffn(LayerNorm(x)))``."""
        if self.can_ffn16(x.shape[0], x.shape[1], w1.shape[0], w2.shape[0]) and x.shape[1] == 512 and w2.shape[0] == 512:
            return self._apply(_FfnBlock16, x, gamma, beta, w1, b1, w2, b2, alpha, p_latent, seed_latent, p_out, seed_out)
        y = self.ffn(self.layernorm(x, gamma, beta), w1, b1, w2, b2, p_latent, seed_latent)
        return self.axpy_dropout(alpha, y, x, p_out, seed_out)

    def attention_block(self, x, gamma, beta, wq, wkv, wo, bo, batch, p: float, seed: int):
        """x + dropout(to_out(attention(to_q | to_kv of LayerNorm(x)))) - the attention sub-block of conform_blocke.forward (Gconform.py:58)."""
        if self.can_block16(x, (gamma, beta, wq, wkv, wo, bo)) and self.joined(wq, wkv)[0] is not None:
            return self._apply(_AttnBlock16, x, gamma, beta, wq, wkv, wo, bo, batch, p, seed)
        n = self.layernorm(x, gamma, beta)
        out = self.linear(self.attention(self.linear(n, self.cat_rows(wq, wkv)), batch), wo, bo)
        return self.axpy_dropout(1.0, out, x, p, seed)

    def conv_block(self, x, gamma, beta, pw1_w, pw1_b, dw_w, dw_b, bn_g, bn_b, bn_rm, bn_rv, pw2_w, pw2_b, batch, p: float, seed: int):
        """x + dropout(conform_conv(LayerNorm(x))) - the conv sub-block (Gconform.py:59, modules/conv/base_conv.py:63-70)."""
        if self.can_block16(x, (gamma, beta, pw1_w, pw1_b, pw2_w, pw2_b)):
            return self._apply(_ConvBlock16, x, gamma, beta, pw1_w, pw1_b, dw_w, dw_b, bn_g, bn_b, bn_rm, bn_rv, pw2_w, pw2_b, batch, p, seed)
        h = self.glu(self.linear(self.layernorm(x, gamma, beta), pw1_w, pw1_b))
        h = self.batchnorm(self.dwconv(h, dw_w, dw_b, batch), bn_g, bn_b, bn_rm, bn_rv)
        return self.axpy_dropout(1.0, self.linear(self.silu(h), pw2_w, pw2_b), x, p, seed)

    def layernorm(self, x, gamma, beta):
        return self._apply(_LayerNorm, x, gamma, beta)

    def silu(self, x):
        return self._apply(_Silu, x)

    def sigmoid(self, x):
        return self._apply(_Sigmoid, x)

    def silu_dropout(self, x, p: float, seed: int):
        """dropout(silu(x)) in one pass (conform_ffn.forward: act + drop1, Gconform.py:31-32)."""
        return self._apply(_Silu, x) if p <= 0.0 else self._apply(_SiluDropout, x, p, seed)

    def axpy_dropout(self, alpha: float, y, x, p: float, seed: int):
        """alpha * dropout(y) + x in one pass (the residual sites of conform_blocke.forward, Gconform.py:57-61)."""
        return self._apply(_Axpy, alpha, y, x) if p <= 0.0 else self._apply(_AxpyDropout, alpha, y, x, p, seed)

    def glu(self, x):
        return self._apply(_Glu, x)

    def axpy(self, alpha: float, y, x):
        """alpha * y + x (``x = f(x) * 0.5 + x``, Gconform.py:57,60)."""
        return self._apply(_Axpy, alpha, y, x)

    def mask_rows(self, x, mask_u8):
        return self._apply(_MaskRows, x, mask_u8)

    def dropout(self, x, p: float, seed: int):
        if p <= 0.0:
            return x
        return self._apply(_Dropout, x, p, seed)

    def dwconv(self, x, weight, bias, batch):
        """Depthwise Conv1d(C, C, 31, padding 15, groups C): weight [C, 1, 31] as in the state dict."""
        return self._apply(_DwConv, x, weight, bias, batch)

    def batchnorm(self, x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
        return self._apply(_BatchNorm, x, gamma, beta, running_mean, running_var, momentum, eps)

    def attention(self, qkv, batch):
        """F.scaled_dot_product_attention on the fused projection qkv [M, 1536] -> merged heads [M, 512]."""
        return self._apply(_Attention, qkv, batch)

    def bce_with_logits(self, logits, target):
        return self._apply(_Bce, logits, target)

    def cross_entropy(self, logits, target, ignore_index: int = -1):
        """nn.CrossEntropyLoss(ignore_index) of logits [M, N] against int64 classes [M] (training/me_quant_task.py:42,77)."""
        return self._apply(_Ce, logits, target, ignore_index)

    def binary_emd(self, pred, gt, B: int, T: int):
        return self._apply(_Emd, pred, gt, B, T)


class _Ctx:
    """What the operator bodies below use of torch's FunctionCtx."""
    saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class Tape:
    """A minimal reverse-mode tape over the SAME ``forward`` / ``backward`` bodies torch.autograd would run.

    Why: measured (tools/host_overhead_probe.py) an operator call costs the host 9 us, torch.autograd.Function.apply around it another
    14 us and its backward node ~25 us more - 350 operators per step make the step HOST-bound at the reference's batch shape (8 phrases,
    ~4 100 frames: 15 ms of host time per 16.5 ms step).  The model is a fixed sequence of these operators, so the trainer records them
    itself: ``apply`` runs ``fn.forward`` (under torch.no_grad) and keeps (fn, ctx, inputs, output); ``backward`` walks the records in
    reverse, hands each output gradient to ``fn.backward`` and routes the results - summed where a tensor feeds two operators, added
    into ``.grad`` (the flat gradient buffer's views) for parameters, with ``ops.deposited`` standing in for the post-accumulate hook.
    Same kernels, same order of additions: bit-identical gradients to the torch.autograd path (tests/test_gpu_train_step.py)."""

    def __init__(self, ops: 'TrainOps'):
        self.ops = ops
        self.records = []
        self.produced = set()
        self.lane_of = {}             # id(tensor) -> the lane that produced it (absent: lane 0, or ready before the pass began)

    def apply(self, fn, *args):
        ctx = _Ctx()
        produced = self.produced
        ops = self.ops
        lane = ops._lane
        if ops.lanes == 2:
            lane_of = self.lane_of
            for a in args:                                      # operands produced on the other lane: order the streams
                if isinstance(a, torch.Tensor):
                    src = lane_of.get(id(a))
                    if src is not None and src[0] != lane:
                        ops.adopt(a, src[0], src[1])
        ctx.needs_input_grad = needs = tuple(isinstance(a, torch.Tensor) and (a.requires_grad or id(a) in produced) for a in args)
        out = fn.forward(ctx, *args)
        ops._lane_version[lane] += 1
        if ops.lanes == 2 and isinstance(out, torch.Tensor):
            self.lane_of[id(out)] = (lane, ops._lane_version[lane])
        if True in needs:
            self.records.append((fn, ctx, args, out, lane))
            produced.add(id(out))
        return out

    def backward(self, seeds):
        """seeds: (tensor, gradient) pairs - the losses with their weights."""
        grads = {}
        for t, g in seeds:
            grads[id(t)] = (g, 0) if id(t) not in grads else (grads[id(t)][0] + g, 0)
        records, ops = self.records, self.ops
        two = ops.two_lanes()
        ops.begin_wgrad_lanes()
        try:
            self._walk(records, ops, two, grads)
        finally:
            if two:
                ops._enter_lane(0)
            ops.end_wgrad_lanes()
        if two:
            ops.join_lanes()
        self.produced.clear()
        self.lane_of.clear()

    @staticmethod
    def _walk(records, ops, two, grads):
        while records:
            fn, ctx, args, out, lane = records.pop()            # releases the saved activations as the pass moves on
            entry = grads.pop(id(out), None)
            if entry is None:
                continue                                        # an output nothing differentiable consumed
            g, g_lane = entry
            if two and lane != ops._lane:
                ops._enter_lane(lane)
            if g_lane != lane and isinstance(g, torch.Tensor):
                ops.adopt(g, g_lane)
            results = fn.backward(ctx, g)
            ops._lane_version[lane] += 1
            for a, ga, need in zip(args, results, ctx.needs_input_grad):
                if ga is None or not need:
                    continue
                if a.requires_grad:                             # a parameter: accumulate where autograd's AccumulateGrad would
                    if a.grad is None:
                        a.grad = ga.clone()
                    else:
                        a.grad.add_(ga)
                    ops._lane_version[lane] += 1
                    ops.deposited(a)
                else:
                    prev = grads.get(id(a))
                    if prev is None:
                        grads[id(a)] = (ga, lane)
                    else:
                        if prev[1] != lane:
                            ops.adopt(prev[0], prev[1])
                        grads[id(a)] = (prev[0] + ga, lane)
                        ops._lane_version[lane] += 1


class _Lane:
    """``with ops.lane(i, after=event)``: see TrainOps.lane."""

    def __init__(self, ops: 'TrainOps', index: int, after):
        self.ops, self.index, self.after = ops, index, after
        self.prev = None

    def __enter__(self):
        ops = self.ops
        self.prev = ops._lane
        if self.index != self.prev:
            if self.after is not None:
                event, src, version = self.after
                ops._lane_streams[self.index].wait_event(event)
                ops._fork_cover = (src, version)
            else:
                ops._lane = self.index
                ops.wait_lane(self.prev)
            ops._enter_lane(self.index)
        return self

    def __exit__(self, *exc):
        if self.index != self.prev:
            self.ops._fork_cover = None
            self.ops._enter_lane(self.prev)
        return False


class _CatRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, a, b):
        ctx.rows = a.shape[0]
        return torch.cat([a, b], dim=0)

    @staticmethod
    def backward(ctx, d):
        return None, d[:ctx.rows], d[ctx.rows:]


class _Reshape(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x, shape):
        ctx.shape = x.shape
        return x.reshape(shape)

    @staticmethod
    def backward(ctx, d):
        return None, d.reshape(ctx.shape), None


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops: TrainOps, x, weight, bias):
        ctx.ops = ops
        w2 = weight.reshape(weight.shape[0], -1)                  # Conv1d k = 1 weights are [N, K, 1]
        ctx.save_for_backward(x, w2)
        ctx.wshape, ctx.has_bias = weight.shape, bias is not None
        ctx.wparam, ctx.bparam = weight, bias                     # identities only: looked up among the gradient sinks in backward
        return ops.gemm(x.contiguous(), w2.contiguous(), bias)

    @staticmethod
    def backward(ctx, dy):
        ops: TrainOps = ctx.ops
        x, w2 = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dx = ops.gemm_dx(dy, w2.contiguous())
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        sw = ops.sink(ctx.wparam) if ctx.needs_input_grad[2] else None
        sb = ops.sink(ctx.bparam) if want_db else None
        if sw is not None and (sb is not None or not want_db) and ops.can_wgrad_into(dy.shape[1], x.shape[1]):
            ops.gemm_dw_into(dy, x.contiguous(), sw, sb)          # written into the parameters' gradient arrays: nothing to return
            ops.deposited(ctx.wparam)
            if sb is not None:
                ops.deposited(ctx.bparam)
        elif ctx.needs_input_grad[2]:
            dw, db = ops.gemm_dw(dy, x.contiguous(), want_db)
            dw = dw.reshape(ctx.wshape)
        elif ctx.has_bias and ctx.needs_input_grad[3]:
            db = ops.colsum(dy)
        return None, dx, dw, db


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops: TrainOps, x, gamma, beta):
        M = x.shape[0]
        x = x.contiguous()
        y, mean, rstd = torch.empty_like(x), ops.new(M), ops.new(M)
        ops.check(ops.lib.some_train_layernorm_fwd(ops.h, _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, ops.stream()))
        ctx.ops = ops
        ctx.gparam, ctx.bparam = gamma, beta
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops: TrainOps = ctx.ops
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        M = x.shape[0]
        dx = torch.empty_like(x)
        sc = ops.scratch(M, 512)
        sg, sb = ops.sink(ctx.gparam), ops.sink(ctx.bparam)
        if sg is not None and sb is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]:
            # the column reductions add straight into the parameters' gradient arrays
            ops.check(ops.lib.some_train_layernorm_bwd(ops.h, _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(sg), _p(sb), 1, M,
                                                       _p(sc), sc.numel(), ops.stream()))
            ops.deposited(ctx.gparam)
            ops.deposited(ctx.bparam)
            return None, dx, None, None
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        ops.check(ops.lib.some_train_layernorm_bwd(ops.h, _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), 0, M,
                                                   _p(sc), sc.numel(), ops.stream()))
        return None, dx, dg, db


class _Silu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x):
        ctx.ops = ops
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.eltwise(_lib.ELT_SILU_FWD, x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return None, ctx.ops.eltwise(_lib.ELT_SILU_BWD, dy.contiguous(), x)


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x):
        ctx.ops = ops
        y = ops.eltwise(_lib.ELT_SIGMOID_FWD, x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return None, ctx.ops.eltwise(_lib.ELT_SIGMOID_BWD, dy.contiguous(), y)


class _Glu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x):
        ctx.ops = ops
        x = x.contiguous()
        M, C2 = x.shape
        y = ops.new(M, C2 // 2)
        ops.check(ops.lib.some_train_glu(ops.h, None, _p(x), _p(y), M, C2 // 2, 0, ops.stream()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = ctx.ops
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        ops.check(ops.lib.some_train_glu(ops.h, _p(dy.contiguous()), _p(x), _p(dx), x.shape[0], x.shape[1] // 2, 1, ops.stream()))
        return None, dx


class _Axpy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, alpha, y, x):
        ctx.ops, ctx.alpha = ops, alpha
        return ops.eltwise(_lib.ELT_AXPY, y.contiguous(), x.contiguous(), alpha=alpha)

    @staticmethod
    def backward(ctx, d):
        dy = ctx.ops.eltwise(_lib.ELT_AXPY, d.contiguous(), None, alpha=ctx.alpha) if ctx.alpha != 1.0 else d
        return None, None, dy, d


class _SiluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x, p, seed):
        ctx.ops, ctx.p, ctx.seed = ops, p, seed
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.eltwise(_lib.ELT_SILU_DROP_FWD, x, p=p, seed=seed)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return None, ctx.ops.eltwise(_lib.ELT_SILU_DROP_BWD, dy.contiguous(), x, p=ctx.p, seed=ctx.seed), None, None


class _AxpyDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, alpha, y, x, p, seed):
        ctx.ops, ctx.alpha, ctx.p, ctx.seed = ops, alpha, p, seed
        return ops.eltwise(_lib.ELT_AXPY_DROP, y.contiguous(), x.contiguous(), alpha=alpha, p=p, seed=seed)

    @staticmethod
    def backward(ctx, d):
        dy = ctx.ops.eltwise(_lib.ELT_AXPY_DROP, d.contiguous(), None, alpha=ctx.alpha, p=ctx.p, seed=ctx.seed)
        return None, None, dy, d, None, None


class _MaskRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x, mask_u8):
        ctx.ops = ops
        ctx.save_for_backward(mask_u8)
        y = torch.empty_like(x)
        ops.check(ops.lib.some_train_mask_rows(ops.h, _p(x.contiguous()), _p(mask_u8), _p(y), x.shape[0], x.shape[1], ops.stream()))
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = ctx.ops
        (mask_u8,) = ctx.saved_tensors
        dx = torch.empty_like(dy)
        ops.check(ops.lib.some_train_mask_rows(ops.h, _p(dy.contiguous()), _p(mask_u8), _p(dx), dy.shape[0], dy.shape[1], ops.stream()))
        return None, dx, None


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x, p, seed):
        ctx.ops, ctx.p, ctx.seed = ops, p, seed
        return ops.eltwise(_lib.ELT_DROPOUT, x.contiguous(), p=p, seed=seed)

    @staticmethod
    def backward(ctx, dy):
        return None, ctx.ops.eltwise(_lib.ELT_DROPOUT, dy.contiguous(), p=ctx.p, seed=ctx.seed), None, None


class _DwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x, weight, bias, batch):
        ctx.ops, ctx.batch = ops, batch
        x = x.contiguous()
        Cn = x.shape[1]
        taps = weight.reshape(Cn, -1).t().contiguous()             # [31, C] tap-major
        y = torch.empty_like(x)
        ops.check(ops.lib.some_train_dwconv(ops.h, _p(x), _p(taps), _p(bias), _p(batch.frame_offsets_dev), batch.B, batch.max_frames,
                                            _p(y), Cn, 0, ops.stream()))
        ctx.save_for_backward(x, taps)
        ctx.wshape, ctx.has_bias = weight.shape, bias is not None
        ctx.wparam, ctx.bparam = weight, bias                      # identities only: looked up among the gradient sinks in backward
        return y

    @staticmethod
    def backward(ctx, dy):
        ops, batch = ctx.ops, ctx.batch
        x, taps = ctx.saved_tensors
        dy = dy.contiguous()
        M, Cn = x.shape
        dx = torch.empty_like(x)
        ops.check(ops.lib.some_train_dwconv(ops.h, _p(dy), _p(taps), None, _p(batch.frame_offsets_dev), batch.B, batch.max_frames,
                                            _p(dx), Cn, 1, ops.stream()))
        sw = ops.sink(ctx.wparam) if ops.dwconv_sinks and ctx.needs_input_grad[2] else None
        sb = ops.sink(ctx.bparam) if sw is not None and ctx.has_bias and ctx.needs_input_grad[3] else None
        if sw is not None and (sb is not None or not ctx.has_bias) and sw.is_contiguous():
            # the parameter gradients straight into the gradient arrays, in the weight's own [C][31] layout (some_train_dwconv_bwd_params:
            # the same sums, no tap-major temporary, no transposed add, no column-sum tensor - four launches fewer) and, inside
            # Tape.backward with weight-gradient lanes, on the lane's side stream: 76 us per block that nothing downstream waits for
            sc = ops.wgrad_scratch(ops._bytes('some_train_scratch_bytes', M, Cn))
            ops.check(ops.lib.some_train_dwconv_bwd_params(ops.h, _p(dy), _p(x), _p(clip_of_row(batch)), _p(batch.frame_offsets_dev), M, Cn, _p(sw), _p(sb),
                                                           _p(sc), sc.numel(), ops.stream()))
            ops.wgrad_issued(dy, x, batch)                          # (the batch descriptor: clip_of_row / frame_offsets are read there too)
            ops.deposited(ctx.wparam)
            if sb is not None:
                ops.deposited(ctx.bparam)
            return None, dx, None, None, None
        dt = torch.empty_like(taps)
        sc = ops.scratch(M, Cn)
        ops.check(ops.lib.some_train_dwconv_bwd_taps(ops.h, _p(dy), _p(x), _p(clip_of_row(batch)), _p(batch.frame_offsets_dev), M, Cn, _p(dt), 0,
                                                     _p(sc), sc.numel(), ops.stream()))
        dw = dt.t().reshape(ctx.wshape)
        db = ops.colsum(dy) if ctx.has_bias else None
        return None, dx, dw, db, None


def clip_of_row(batch) -> torch.Tensor:
    """int32 [M]: clip index of every packed row (cached on the batch descriptor)."""
    cached = getattr(batch, '_clip_of_row_dev', None)
    if cached is None:
        idx = np.repeat(np.arange(batch.B, dtype=np.int32), batch.frame_counts)
        cached = batch._clip_of_row_dev = torch.from_numpy(idx).to(batch.frame_offsets_dev.device)
    return cached


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x, gamma, beta, running_mean, running_var, momentum, eps):
        x = x.contiguous()
        M, Cn = x.shape
        y, mean, rstd = torch.empty_like(x), ops.new(Cn), ops.new(Cn)
        sc = ops.scratch(M, Cn)
        ops.check(ops.lib.some_train_batchnorm_fwd(ops.h, _p(x), _p(gamma), _p(beta), M, Cn, float(eps), float(momentum), _p(running_mean),
                                                   _p(running_var), _p(y), _p(mean), _p(rstd), _p(sc), sc.numel(), ops.stream()))
        ctx.ops = ops
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = ctx.ops
        x, gamma, mean, rstd = ctx.saved_tensors
        M, Cn = x.shape
        dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        sc = ops.scratch(M, Cn)
        ops.check(ops.lib.some_train_batchnorm_bwd(ops.h, _p(dy.contiguous()), _p(x), _p(gamma), _p(mean), _p(rstd), M, Cn, _p(dx), _p(dg), _p(db),
                                                   _p(sc), sc.numel(), ops.stream()))
        return None, dx, dg, db, None, None, None, None


class _Attention(torch.autograd.Function):
    """Exact-f32 MFMA kernels (attention.hip / train_attention.hip) or, in f16x3 mode, the split-f16 kernels for forward
    AND backward: the two operand tensors R = split_rows(qkv) and Rt = transpose(qkv, split) are made once in the forward
    and kept for the backward instead of qkv itself."""

    @staticmethod
    def forward(ctx, ops, qkv, batch):
        qkv = qkv.contiguous()
        M = qkv.shape[0]
        assert qkv.shape[1] == 1536 and M == batch.total_frames
        out, lse = ops.new(M, 512), ops.new(8, M)
        ctx.ops, ctx.batch, ctx.prec = ops, batch, ops.attention_precision
        if ctx.prec == 'f16x3':
            R, Rt = ops.split_transpose(qkv, pad_to=64) if ops.device_prep else (ops.split_rows(qkv), ops.transpose(qkv, pad_to=64, split=True))
            ops.check(ops.lib.some_train_attention_fwd_f16x3(ops.h, _p(R), _p(Rt), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M,
                                                             Rt.shape[1], ops._hi_mode, _p(out), _p(lse), ops.stream()))
            ctx.save_for_backward(R, Rt, out, lse)
        else:
            ops.check(ops.lib.some_train_attention_fwd(ops.h, _p(qkv), _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M, _p(out),
                                                       _p(lse), ops.stream()))
            ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        ops, batch = ctx.ops, ctx.batch
        dout = dout.contiguous()
        M = dout.shape[0]
        dqkv, dsum = ops.new(M, 1536), ops.new(8, M)
        if ctx.prec == 'f16x3':
            R, Rt, out, lse = ctx.saved_tensors
            # dO is carried as f16 hi + lo halves whose absolute floor is 2^-25: bring its largest element to ~2^10 with a
            # power-of-two factor computed on the device (exact; no host sync) and take the factor out of dqkv again
            scale = torch.exp2(torch.floor(10.0 - torch.log2(dout.abs().amax().clamp_min(1e-30))))
            dout = dout * scale
            D, Dt = ops.split_rows(dout), ops.transpose(dout, pad_to=64, split=True)
            ops.check(ops.lib.some_train_attention_bwd_f16x3(ops.h, _p(R), _p(Rt), _p(D), _p(Dt), _p(out), _p(dout), _p(lse),
                                                             _p(batch.frame_offsets_dev), batch.B, batch.max_frames, M, Rt.shape[1], ops._hi_mode,
                                                             _p(dqkv), _p(dsum), ops.stream()))
            dqkv.mul_(1.0 / scale)
        else:
            qkv, out, lse = ctx.saved_tensors
            ops.check(ops.lib.some_train_attention_bwd(ops.h, _p(qkv), _p(out), _p(dout), _p(lse), _p(batch.frame_offsets_dev), batch.B,
                                                       batch.max_frames, M, _p(dqkv), _p(dsum), ops.stream()))
        return None, dqkv, None


def _attention_bwd16(ctx, dout):
    """_Attention.backward in mixed precision with dq | dk | dv written as the 16-bit GEMM operand by the kernels themselves (no fp32 array,
    no rescaling pass, no cast pass): [M, 1536] 16-bit."""
    ops, batch = ctx.ops, ctx.batch
    dout = dout.contiguous()
    M = dout.shape[0]
    R, Rt, out, lse = ctx.saved_tensors
    dqkv16 = torch.empty((M, 1536), dtype=ops.dtype16, device=ops.device)
    # dO is carried as 16-bit halves: its power-of-two factor (as in _Attention.backward), both split layouts, rowsum(dO O) and the 1 / factor
    # on the way out are made on the device inside the one call
    Mp = Rt.shape[1]
    if not ops.device_prep:
        scale = torch.exp2(torch.floor(10.0 - torch.log2(dout.abs().amax().clamp_min(1e-30))))
        dout = dout * scale
        inv, dsum = torch.reciprocal(scale).reshape(1), ops.new(8, M)
        D, Dt = ops.split_rows(dout), ops.transpose(dout, pad_to=64, split=True)
        ops.check(ops.lib.some_train_attention_bwd_f16x3_out16(ops.h, _p(R), _p(Rt), _p(D), _p(Dt), _p(out), _p(dout), _p(lse), _p(batch.frame_offsets_dev),
                                                               batch.B, batch.max_frames, M, Mp, ops._hi_mode, _p(dqkv16), _p(inv), _p(dsum), ops.stream()))
        return dqkv16
    work = torch.empty(ops._bytes('some_train_attention_bwd16_work_bytes', M, Mp), dtype=torch.uint8, device=ops.device)
    ops.check(ops.lib.some_train_attention_bwd_f16x3_auto16(ops.h, _p(R), _p(Rt), _p(out), _p(dout), _p(lse), _p(batch.frame_offsets_dev), batch.B,
                                                            batch.max_frames, M, Mp, ops._hi_mode, _p(dqkv16), _p(work), work.numel(), ops.stream()))
    return dqkv16


class _Bce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, logits, target):
        logits, target = logits.contiguous(), target.contiguous()
        n = logits.numel()
        dl, loss = torch.empty_like(logits), ops.new(1)
        sc = ops.scratch(1, 1)
        ops.check(ops.lib.some_train_bce_with_logits(ops.h, _p(logits), _p(target), n, _p(dl), _p(loss), _p(sc), sc.numel(), ops.stream()))
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return None, dl * g, None


class _Ce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, logits, target, ignore_index):
        logits, target = logits.contiguous(), target.contiguous().long()
        M, N = logits.shape
        assert target.numel() == M
        dl, loss = torch.empty_like(logits), ops.new(1)
        sc = ops.scratch(1, 1)
        ops.check(ops.lib.some_train_cross_entropy(ops.h, _p(logits), _p(target), M, N, int(ignore_index), _p(dl), _p(loss), _p(sc), sc.numel(), ops.stream()))
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return None, dl * g, None, None


class _Emd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, pred, gt, B, T):
        pred, gt = pred.contiguous(), gt.contiguous()
        assert pred.numel() == B * T == gt.numel()
        dp, loss = torch.empty_like(pred), ops.new(1)
        sc = ops.scratch(B * 8, 1)
        ops.check(ops.lib.some_train_binary_emd(ops.h, _p(pred), _p(gt), B, T, _p(dp), _p(loss), _p(sc), sc.numel(), ops.stream()))
        ctx.save_for_backward(dp)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return None, dp * g, None, None, None


# the fused 16-bit sub-block operators (mixed precision) live in blocks16.py; imported last: they are built from the operators above
from .blocks16 import _AttnBlock16, _ConvBlock16, _Ffn16, _FfnBlock16  # noqa: E402,F401
