"""Batch planning of the reference's trainer: ``batch_by_size`` (utils/__init__.py:60-111) and the two samplers built on it,
``DsBatchSampler`` (utils/training_utils.py:45-138, training) and ``DsEvalBatchSampler`` (:141-176, validation).

A plan is a pure function of (lengths, seed + epoch, replica layout), and every rank computes the whole plan and keeps its
own column - there is no communication.  To train on the same batches as the reference the random draws must be the same
numpy Generator calls in the same order (permutation of the items, permutation of the leftover batches, the per-row
``permuted`` of the batch grid, the optional final shuffle); tests/test_train_host.py pins plans to the reference's samplers on
length distributions of real singing datasets' shape.  Constructor arguments keep the reference's names."""
import math
from typing import Callable, List, Optional, Sequence

import numpy as np


def batch_by_size(indices: Sequence[int], num_frames_fn: Callable[[int], int], max_batch_frames: int = 80000,
                  max_batch_size: int = 48, required_batch_size_multiple: int = 1) -> List[List[int]]:
    """Greedy packing in the given order: an item joins the open batch unless the batch already holds
    ``max_batch_size`` items or (items + 1) x longest-so-far would exceed ``max_batch_frames`` (padded frames, which is
    what the model pays for).  A closed batch is trimmed to a multiple of ``required_batch_size_multiple``; the trimmed
    tail opens the next batch."""
    mult = required_batch_size_multiple
    plans: List[List[int]] = []
    open_items: List[int] = []
    open_lens: List[int] = []
    longest = 0
    for idx in indices:
        frames = num_frames_fn(idx)
        longest = max(longest, frames)
        assert longest <= max_batch_frames, (
            f'sentence at index {idx} of size {longest} exceeds max_batch_samples limit of {max_batch_frames}!')
        open_lens.append(frames)
        full = len(open_items) > 0 and (len(open_items) == max_batch_size or (len(open_items) + 1) * longest > max_batch_frames)
        if full:
            keep = max(mult * (len(open_items) // mult), len(open_items) % mult)
            plans.append(open_items[:keep])
            open_items, open_lens = open_items[keep:], open_lens[keep:]
            longest = max(open_lens) if open_lens else 0
        open_items.append(idx)
    if open_items:
        plans.append(open_items)
    return plans


_pack = batch_by_size      # the samplers take a ``batch_by_size`` flag of the same name


class DsBatchSampler:
    """Training batches of one rank for one epoch (``set_epoch`` re-plans).  ``dataset`` needs ``_sizes`` (clipped
    lengths, used for the similar-size ordering), ``num_frames(i)`` and ``__len__``."""

    def __init__(self, dataset, max_batch_frames, max_batch_size, sub_indices=None, num_replicas=None, rank=None,
                 frame_count_grid=200, required_batch_count_multiple=1, batch_by_size=True, sort_by_similar_size=True,
                 shuffle_sample=False, shuffle_batch=False, seed=0, drop_last=False) -> None:
        self.dataset = dataset
        self.max_batch_frames = max_batch_frames
        self.max_batch_size = max_batch_size
        self.sub_indices = sub_indices
        self.num_replicas = 1 if num_replicas is None else num_replicas
        self.rank = 0 if rank is None else rank
        self.frame_count_grid = frame_count_grid
        self.required_batch_count_multiple = required_batch_count_multiple
        self.batch_by_size = batch_by_size
        self.sort_by_similar_size = sort_by_similar_size
        self.shuffle_sample = shuffle_sample
        self.shuffle_batch = shuffle_batch
        self.seed = seed
        self.drop_last = drop_last
        self.epoch = 0
        self.batches: Optional[List[List[int]]] = None
        self._planned_for = None

    def _order(self, rng: np.random.Generator) -> List[int]:
        if not self.shuffle_sample:
            return list(self.sub_indices) if self.sub_indices is not None else list(range(len(self.dataset)))
        if self.sub_indices is not None:
            rng.shuffle(self.sub_indices)                      # in place, as the reference does: epochs compound
            order = np.array(self.sub_indices)
        else:
            order = rng.permutation(len(self.dataset))
        if self.sort_by_similar_size:
            grid = self.frame_count_grid
            assert grid > 0
            # lengths rounded to the grid: items of one bucket stay in shuffled order (stable sort), buckets ascend
            bucket = (np.round(np.array(self.dataset._sizes)[order] / grid) * grid).clip(grid, None).astype(np.int64)
            order = order[np.argsort(bucket, kind='mergesort')]
        return order.tolist()

    def _plan(self):
        key = self.epoch + self.seed
        # a plan is deterministic in (seed, epoch) - except with sub_indices, which the reference shuffles in place on
        # every call (its `formed` marker is never set): that case is re-planned each time, like there
        if self._planned_for == key and self.sub_indices is None:
            return
        rng = np.random.default_rng(self.seed + self.epoch)
        order = self._order(rng)
        if self.batch_by_size:
            pool = _pack(order, self.dataset.num_frames, max_batch_frames=self.max_batch_frames, max_batch_size=self.max_batch_size)
        else:
            pool = [order[i:i + self.max_batch_size] for i in range(0, len(order), self.max_batch_size)]
        world = self.num_replicas
        even = len(pool) // world * world
        if self.drop_last and len(pool) > even:
            pool, spare = pool[:even], []
        else:
            spare = (rng.permutation(len(pool) - even) + even).tolist()
        # batch grid [steps, world], every step's row shuffled across ranks; this rank keeps its column
        mine = rng.permuted(np.arange(even).reshape(-1, world).transpose(), axis=0)[self.rank].tolist()
        steps = len(mine)
        total = steps + (1 if spare else 0)
        if self.rank < len(spare):
            mine.append(spare[self.rank])
        elif spare:                                            # fewer spare batches than ranks: repeat one of this rank's own
            if steps == 0:       # the reference divides by zero here (utils/training_utils.py:115): same exception, with the reason
                raise ZeroDivisionError(f'DsBatchSampler: {len(pool)} batch(es) for {world} replicas - rank {self.rank} has none '
                                        f'(dataset too small for this world size)')
            mine.append(mine[self.epoch % steps])
        mult = self.required_batch_count_multiple
        if mult > 1 and total % mult != 0:                     # whole gradient-accumulation groups
            total = math.ceil(total / mult) * mult
            for i in range(total - len(mine)):
                mine.append(mine[(i + self.epoch * mult) % steps])
        self.batches = [list(pool[b]) for b in mine]
        if self.shuffle_batch:
            rng.shuffle(self.batches)
        self._planned_for = key

    def __iter__(self):
        self._plan()
        return iter(self.batches)

    def __len__(self):
        self._plan()
        return len(self.batches)

    def set_epoch(self, epoch):
        self.epoch = epoch


class DsEvalBatchSampler:
    """Validation runs on rank 0 only: every other rank gets the single batch ``[0]`` so that collectives stay matched."""

    def __init__(self, dataset, max_batch_frames, max_batch_size, rank=None, batch_by_size=True) -> None:
        self.dataset = dataset
        self.max_batch_frames = max_batch_frames
        self.max_batch_size = max_batch_size
        self.rank = 0 if rank is None else rank
        self.batch_by_size = batch_by_size
        self.batch_size = max_batch_size
        self.drop_last = False
        if self.rank != 0:
            self.batches = [[0]]
            return
        order = list(range(len(dataset)))
        if batch_by_size:
            self.batches = _pack(order, dataset.num_frames, max_batch_frames=max_batch_frames, max_batch_size=max_batch_size)
        else:
            self.batches = [order[i:i + max_batch_size] for i in range(0, len(order), max_batch_size)]

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)
