"""Data-parallel gradient synchronisation overlapped with the backward pass ("DDP gradient
all-reduce, 51.77 M fp32 grads = 207 MB/step, bucketed and overlapped with backward"; the reference gets this from
Lightning's DDP strategy, configs/base.yaml:76-79).

All gradients live in ONE flat buffer (FlatParams).  It is cut into contiguous buckets in parameter order; the backward
pass produces gradients roughly last layer first, so a bucket is complete when the LAST of its parameters has received its
gradient (a post-accumulate hook per parameter - or ``mark()`` for parameters whose gradient a backward kernel wrote itself -
counts them down), and its slice is all-reduced asynchronously - RCCL runs it on its own stream behind an event of the compute
stream, while the backward kernels of the earlier layers keep the compute stream busy.  ``finish()`` launches whatever never
completed (parameters without a gradient this step) and makes the compute stream wait for every bucket.  Summation is over
ranks only, so the result is the same flat buffer a single all-reduce would give - bit for bit with two ranks, up to the
ring's summation order beyond.

Launch ORDER is fixed: bucket k goes out only after every bucket above it (torch DDP's rule).  Collectives pair up across ranks by
issue order, so an order that depended on which bucket happened to complete first on a rank (a parameter without a gradient on
one rank only, a different autograd schedule) would pair mismatched slices - a hang or silent corruption.  A bucket that
completes early waits in ``ready`` until its turn; ``finish()`` flushes the rest in the same descending order.  A parameter that
receives no gradient in a pass (an unused head, a frozen layer, a data-dependent path) therefore holds its bucket and every bucket
below it back until ``finish()``: that pass is not overlapped, and nothing else happens - a gradient path may switch on and off from
step to step and differ between ranks.  ``static_graph=True`` (config ``some_amd_ddp_static_graph``) is DDP's static-graph rule
instead: parameters without a gradient in the previous armed pass (``absent``) are counted as reported when the next pass is armed,
so one unused parameter does not cost the overlap; it is only sound when the set of used parameters never changes and is the same
on every rank - if an ``absent`` parameter does report after its bucket has gone the slice has been reduced without it, and the
pass raises.

Every parameter reports exactly once per armed backward pass: a second report (a weight used by two layers) would launch the
bucket before the later gradient is in the buffer, so it raises instead of counting (``fired``), and ``pending`` can never go
below zero.  The one legitimate double call - autograd's post-accumulate hook firing behind a ``mark()`` even though the backward
function returned None for the parameter - is recognised and ignored (see ``mark``).

Bucket size: xGMI is point-to-point and a ring all-reduce is bound by one link (about 50 GB/s effective per direction), so a
bucket must be large enough to amortise the ~20 us launch + ring latency of 8 hops but small enough that the last bucket -
the only one that cannot overlap - is short: 32 MiB (0.7 ms on the ring, 7 buckets for 207 MB) by default
(``some_amd_ddp_bucket_mb``)."""
from typing import List, Optional, Sequence, Tuple

import torch


class BucketedGradSync:
    def __init__(self, flat_grad: torch.Tensor, params: Sequence[Tuple[torch.Tensor, int, int]], process_group=None,
                 bucket_bytes: int = 32 << 20, names: Optional[Sequence[str]] = None, static_graph: bool = False):
        """params: (leaf tensor whose .grad is a view of ``flat_grad``, offset, padded numel) in buffer order."""
        self.flat_grad, self.pg = flat_grad, process_group
        self.static_graph = bool(static_graph)
        self.bounds: List[Tuple[int, int]] = []           # [start, end) of every bucket in the flat buffer
        self.bucket_of: List[int] = []                    # parameter index -> bucket
        self.size: List[int] = []                         # parameters per bucket
        self.names = list(names) if names is not None else [f'param{i}' for i in range(len(params))]
        limit = max(1, bucket_bytes // flat_grad.element_size())
        for _, off, n in params:                          # padded views are contiguous: off + n is the next offset
            if self.bounds and off + n - self.bounds[-1][0] <= limit:
                self.bounds[-1] = (self.bounds[-1][0], off + n)
                self.size[-1] += 1
            else:
                self.bounds.append((off, off + n))
                self.size.append(1)
            self.bucket_of.append(len(self.bounds) - 1)
        if self.bounds:
            assert self.bounds[0][0] == 0 and all(a[1] == b[0] for a, b in zip(self.bounds, self.bounds[1:]))
            self.bounds[-1] = (self.bounds[-1][0], flat_grad.numel())
        self.pending: List[int] = []
        self.ready: List[bool] = []                       # complete, waiting for its turn in the fixed launch order
        self.launched: List[bool] = []
        self.fired: List[bool] = []                       # per parameter: reported in this armed pass
        self.marked: List[bool] = []                      # ... through mark()
        self.echoed: List[bool] = []                      # ... and autograd's hook behind that mark() has been seen
        self.work: List[Optional[object]] = []
        self.next_bucket = -1                             # the only bucket allowed to go out next (descending)
        self.armed = False
        self.launch_order: List[int] = []                 # diagnostics: order in which buckets went out this step
        self.fire_order: List[int] = []                   # diagnostics: parameter indices in reporting order
        self.absent: set = set()                          # parameters without a gradient in the previous armed pass
        self.before_launch = None                         # optional callable run right before a bucket's all-reduce is issued
        self._hooks = []
        self._index = {}
        for i, (p, _, _) in enumerate(params):
            hook = self._make_hook(i)
            self._hooks.append(hook)
            self._index[id(p)] = i
            p.register_post_accumulate_grad_hook(hook)

    def mark(self, param):
        """The gradient of ``param`` was written into the flat buffer by a backward kernel itself (TrainOps gradient sinks) and the
        backward function returns None for it.  Call it AFTER the kernel has been enqueued, from the thread and under the current
        stream it was enqueued on: the all-reduce this may launch orders itself behind that stream's work at the moment of the call.

        autograd STILL runs the parameter's AccumulateGrad node afterwards (with an undefined gradient: nothing is accumulated) and,
        in torch >= 2.x, fires its post-accumulate hook - the echo of a mark.  Counting that echo as a second report was the cause of
        round 2's diverging data-parallel replicas (every sunk parameter counted twice -> buckets left when half of their gradients
        were in the buffer); the hook therefore ignores ONE echo per marked parameter and pass."""
        self._report(self._index[id(param)], True)

    def _make_hook(self, index: int):
        def hook(_param):
            self._report(index, False)
        return hook

    def _report(self, index: int, from_mark: bool):
        if not self.armed:
            return
        bucket = self.bucket_of[index]
        if self.fired[index]:
            if not from_mark and self.marked[index] and not self.echoed[index]:
                self.echoed[index] = True                 # AccumulateGrad's hook behind a mark(): already counted
                return
            raise RuntimeError(f'BucketedGradSync: {self.names[index]} reported its gradient twice in one backward pass (bucket {bucket}'
                               f'{", already launched" if self.launched[bucket] else ""}): a later write would land on a reduced slice')
        self.fired[index] = True
        self.marked[index] = from_mark
        self.fire_order.append(index)
        if index in self.absent:                          # counted at arm(): the gradient is in the buffer before the bucket leaves
            if self.launched[bucket]:
                raise RuntimeError(f'BucketedGradSync: {self.names[index]} had no gradient in the previous step and was counted as absent, '
                                   f'but reports one now, after bucket {bucket} was sent')
            self.absent.discard(index)
            return
        self.pending[bucket] -= 1
        assert self.pending[bucket] >= 0
        if self.pending[bucket] == 0:
            self.ready[bucket] = True
            self._drain()

    def _drain(self):
        while self.next_bucket >= 0 and self.ready[self.next_bucket]:
            self._launch(self.next_bucket)
            self.next_bucket -= 1

    def _launch(self, bucket: int):
        a, b = self.bounds[bucket]
        if self.before_launch is not None:
            self.before_launch()                   # the trainer's two lanes: the current stream first waits for the other lane's kernels
        self.work[bucket] = torch.distributed.all_reduce(self.flat_grad[a:b], op=torch.distributed.ReduceOp.SUM, group=self.pg,
                                                         async_op=True)
        self.launched[bucket] = True
        self.launch_order.append(bucket)

    def arm(self):
        """Call before the backward pass whose gradients are final (the LAST micro-batch of an accumulation group)."""
        n = len(self.bounds)
        self.pending, self.ready, self.launched, self.work = list(self.size), [False] * n, [False] * n, [None] * n
        self.fired = [False] * len(self.bucket_of)
        self.marked, self.echoed = [False] * len(self.bucket_of), [False] * len(self.bucket_of)
        self.next_bucket = n - 1
        self.launch_order, self.fire_order = [], []
        self.armed = True
        for index in self.absent:
            self.pending[self.bucket_of[index]] -= 1
        for bucket in range(n):
            self.ready[bucket] = self.pending[bucket] == 0

    def finish(self):
        """After backward: reduce the buckets that never completed (same descending order), then wait for all of them (stream-ordered
        for RCCL)."""
        if not self.armed:
            raise RuntimeError('BucketedGradSync.finish() without arm()')
        self.armed = False
        while self.next_bucket >= 0:
            self._launch(self.next_bucket)
            self.next_bucket -= 1
        for w in self.work:
            w.wait()
        self.absent = {i for i, f in enumerate(self.fired) if not f} if self.static_graph else set()

    def unreported(self) -> List[str]:
        """Names of the parameters that did not report in the last armed pass (no gradient this step)."""
        return [self.names[i] for i, f in enumerate(self.fired) if not f]
