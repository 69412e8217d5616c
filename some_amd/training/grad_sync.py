"""Data-parallel gradient synchronisation overlapped with the backward pass ("DDP gradient
all-reduce, 51.77 M fp32 grads = 207 MB/step, bucketed and overlapped with backward"; the reference gets this from
Lightning's DDP strategy, configs/base.yaml:76-79).

All gradients live in ONE flat buffer (FlatParams).  It is cut into contiguous buckets in parameter order; the backward
pass produces gradients roughly last layer first, so a bucket is complete when the LAST of its parameters has received its
gradient (a post-accumulate hook per parameter counts them down), and at that moment its slice is all-reduced
asynchronously - RCCL runs it on its own stream behind an event of the compute stream, while the backward kernels of the
earlier layers keep the compute stream busy.  ``finish()`` launches whatever never completed (parameters without a
gradient this step) and makes the compute stream wait for every bucket.  Summation is over ranks only, so the result is
the same flat buffer a single all-reduce would give - bit for bit with two ranks, up to the ring's summation order beyond.

Bucket size: xGMI is point-to-point and a ring all-reduce is bound by one link (about 50 GB/s effective per direction), so a
bucket must be large enough to amortise the ~20 us launch + ring latency of 8 hops but small enough that the last bucket -
the only one that cannot overlap - is short: 32 MiB (0.7 ms on the ring, 7 buckets for 207 MB) by default
(``some_amd_ddp_bucket_mb``)."""
from typing import List, Optional, Sequence, Tuple

import torch


class BucketedGradSync:
    def __init__(self, flat_grad: torch.Tensor, params: Sequence[Tuple[torch.Tensor, int, int]], process_group=None,
                 bucket_bytes: int = 32 << 20):
        """params: (leaf tensor whose .grad is a view of ``flat_grad``, offset, padded numel) in buffer order."""
        self.flat_grad, self.pg = flat_grad, process_group
        self.bounds: List[Tuple[int, int]] = []           # [start, end) of every bucket in the flat buffer
        self.bucket_of: List[int] = []                    # parameter index -> bucket
        self.size: List[int] = []                         # parameters per bucket
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
        self.launched: List[bool] = []
        self.work: List[Optional[object]] = []
        self.armed = False
        self.launch_order: List[int] = []                 # diagnostics: order in which buckets went out this step
        self._hooks = []
        self._index = {}
        for i, (p, _, _) in enumerate(params):
            hook = self._make_hook(i)
            self._hooks.append(hook)
            self._index[id(p)] = i
            p.register_post_accumulate_grad_hook(hook)

    def mark(self, param):
        """The gradient of ``param`` was written into the flat buffer by a backward kernel itself (TrainOps gradient sinks): autograd
        accumulates nothing for it, so its post-accumulate hook never fires - this call stands in for it."""
        self._hooks[self._index[id(param)]](param)

    def _make_hook(self, index: int):
        bucket = self.bucket_of[index]

        def hook(_param):
            if not self.armed:
                return
            self.pending[bucket] -= 1
            if self.pending[bucket] == 0:
                self._launch(bucket)
        return hook

    def _launch(self, bucket: int):
        a, b = self.bounds[bucket]
        self.work[bucket] = torch.distributed.all_reduce(self.flat_grad[a:b], op=torch.distributed.ReduceOp.SUM, group=self.pg,
                                                         async_op=True)
        self.launched[bucket] = True
        self.launch_order.append(bucket)

    def arm(self):
        """Call before the backward pass whose gradients are final (the LAST micro-batch of an accumulation group)."""
        n = len(self.bounds)
        self.pending, self.launched, self.work = list(self.size), [False] * n, [None] * n
        self.launch_order = []
        self.armed = True

    def finish(self):
        """After backward: reduce the buckets that never completed, then wait for all of them (stream-ordered for RCCL)."""
        if not self.armed:
            raise RuntimeError('BucketedGradSync.finish() without arm()')
        self.armed = False
        for bucket in reversed(range(len(self.bounds))):
            if not self.launched[bucket]:
                self._launch(bucket)
        for w in self.work:
            w.wait()
