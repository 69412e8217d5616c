"""Prefetching batch loader of the training command - the role ``torch.utils.data.DataLoader(dataset, collate_fn=dataset.collater,
batch_sampler=..., num_workers=ds_workers, prefetch_factor=dataloader_prefetch_factor, pin_memory=True, persistent_workers=True)``
plays in the reference (training/base_task.py:374-380, 390-394), shaped for this trainer:

* the HOST half of the collater (reading the items of a batch out of the memory-mapped HDF5 container, padding them to the batch's
  longest item) runs in ``ds_workers`` worker THREADS - numpy / torch copies release the GIL, the container is an mmap, so threads
  scale without pickling batches between processes - into PINNED staging buffers, ``prefetch_factor`` batches per worker ahead of
  the consumer, in plan order;
* the DEVICE half (the Gaussian-blurred per-frame targets ``probs`` [B, T, bins] and the boundary flags, training/me_task.py:33-51)
  is computed on the GPU from the uploaded note arrays by the same torch expressions as before - the largest tensor of a batch
  (41 MB at 8 x 10 000 frames) never crosses PCIe;
* uploads run on a copy stream, one batch ahead of the training step; the compute stream waits on the copy's event only.

``stats`` records how long the consumer waited for a batch that was not ready (the "GPU waits for data" figure of bench.py --train).
Datasets whose items already live in device memory (SyntheticNoteDataset) are collated directly."""
import concurrent.futures
import time
from typing import Dict, Iterator, List, Sequence

import torch

from . import data

_FIELDS = (('units', torch.float32, 0.0), ('pitch', torch.float32, 0.0), ('note_midi', torch.float32, 0.0), ('note_rest', torch.bool, False),
           ('note_dur', torch.int64, 0), ('unit2note', torch.int64, 0))


def collate_host(samples: List[Dict[str, torch.Tensor]], pin: bool) -> Dict[str, torch.Tensor]:
    """Pad the raw item fields of a batch (utils.collate_nd, pad value 0) into pinned CPU tensors."""
    out = {}
    for key, dtype, pad in _FIELDS:
        values = [s[key] for s in samples]
        size = max(int(v.shape[0]) for v in values)
        buf = torch.full((len(values), size) + tuple(values[0].shape[1:]), pad, dtype=dtype, pin_memory=pin)
        for i, v in enumerate(values):
            buf[i, :v.shape[0]] = v
        out[key] = buf
    out['lengths'] = torch.tensor([int(s['note_midi'].shape[0]) for s in samples], dtype=torch.int64, pin_memory=pin)
    return out


def finish_on_device(host: Dict[str, torch.Tensor], config: dict) -> Dict[str, torch.Tensor]:
    """The rest of MIDIExtractionDataset.collater (training/me_task.py:26-52) on the device tensors - term for term what
    ``data.collater`` computes, so both paths give identical batches (tests/test_train_host.py)."""
    num_bins = config['midi_num_bins']
    interval = (config['midi_max'] - config['midi_min']) / (num_bins - 1)
    sigma = config['midi_prob_deviation'] / interval
    batch = {'size': int(host['units'].shape[0])}
    for key in ('units', 'pitch', 'note_midi', 'note_rest', 'note_dur'):
        batch[key] = host[key]
    miu = ((batch['note_midi'] - config['midi_min']) / interval)[:, :, None]
    x = torch.arange(num_bins, device=miu.device).float().reshape(1, 1, -1)
    probs = ((x - miu) / sigma).pow(2).div(-2).exp()
    n_max = batch['note_midi'].shape[1]
    note_mask = torch.arange(n_max, device=miu.device)[None, :] < host['lengths'][:, None]
    probs = probs * (note_mask[..., None] & ~batch['note_rest'][..., None])
    probs = torch.nn.functional.pad(probs, [0, 0, 1, 0])
    unit2note = host['unit2note']
    batch['probs'] = torch.gather(probs, 1, unit2note[..., None].repeat([1, 1, num_bins]))
    batch['unit2note'] = unit2note
    batch['bounds'] = (torch.diff(unit2note, dim=1, prepend=unit2note.new_zeros((unit2note.shape[0], 1))) > 0).float()
    return batch


class PrefetchLoader:
    def __init__(self, dataset, config: dict, device, workers: int = 4, prefetch_factor: int = 2):
        self.dataset, self.config, self.device = dataset, config, torch.device(device) if device is not None else None
        self.workers = max(0, int(workers))
        self.depth = max(1, self.workers) * max(1, int(prefetch_factor))
        # (quantised datasets - other item fields, no gaussian targets to build - are collated by the dataset itself, which moves the items)
        self.on_device = isinstance(dataset, data.SyntheticNoteDataset) or self.device is None or getattr(dataset, 'quantized', False)
        self.cuda = not self.on_device and self.device.type == 'cuda'        # (a CPU device runs the same pipeline without pinning / streams: tests)
        # worker threads pin memory (torch.full(..., pin_memory=True)): a new host thread defaults to device 0, so under one process per GPU
        # every rank's workers would create a context on GPU 0 and pin there - bind them to this rank's device first (what torch's
        # DataLoader pin thread does)
        # (an index-less torch.device('cuda') - the trainer's default - names the CURRENT device: resolve it here, set_device() rejects it)
        dev_index = (self.device.index if self.device.index is not None else torch.cuda.current_device()) if self.cuda else None
        init = (lambda: torch.cuda.set_device(dev_index)) if self.cuda else None
        self.pool = None if self.on_device or self.workers == 0 else concurrent.futures.ThreadPoolExecutor(self.workers, thread_name_prefix='some-loader',
                                                                                                             initializer=init)
        self.copy_stream = torch.cuda.Stream(dev_index) if self.cuda else None
        self.stats = {'batches': 0, 'wait_s': 0.0, 'host_collate_s': 0.0}

    # ---- host side ----------------------------------------------------------------------------------------------
    def _host_batch(self, indices: Sequence[int]):
        t0 = time.perf_counter()
        out = collate_host([self.dataset[int(i)] for i in indices], pin=self.cuda)
        return out, time.perf_counter() - t0

    def _upload(self, host: Dict[str, torch.Tensor]):
        if not self.cuda:
            return host, None, host
        with torch.cuda.stream(self.copy_stream):
            dev = {k: v.to(self.device, non_blocking=True) for k, v in host.items()}
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return dev, ev, host            # `host` is kept alive until the copy has been consumed

    # ---- iteration ----------------------------------------------------------------------------------------------
    def batches(self, plan: Sequence[Sequence[int]]) -> Iterator[Dict[str, torch.Tensor]]:
        """Device batches of ``plan`` (a list of index lists: DsBatchSampler's epoch) in order."""
        if self.on_device:
            for idx in plan:
                self.stats['batches'] += 1
                yield self.dataset.collater([self.dataset[int(i)] for i in idx])
            return
        futures, nxt = [], 0

        def fill():
            nonlocal nxt
            while nxt < len(plan) and len(futures) < self.depth:
                futures.append(self.pool.submit(self._host_batch, plan[nxt]) if self.pool is not None else None)
                nxt += 1

        def take():
            f = futures.pop(0)
            t0 = time.perf_counter()
            host, dt = f.result() if f is not None else self._host_batch(plan[taken[0]])
            self.stats['wait_s'] += time.perf_counter() - t0
            self.stats['host_collate_s'] += dt
            taken[0] += 1
            fill()
            return self._upload(host)

        taken = [0]
        fill()
        ahead = take() if futures or nxt < len(plan) else None
        while ahead is not None:
            dev, ev, host = ahead
            ahead = take() if futures else None                     # the next batch's upload runs under this batch's step
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for t in dev.values():
                    t.record_stream(torch.cuda.current_stream(self.device))
            self.stats['batches'] += 1
            yield finish_on_device(dev, self.config)
            del host

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=False, cancel_futures=True)
            self.pool = None
