"""Utterance sharding for batch_infer: one process per GPU, rows dealt round-robin after a size sort, weights
packed once on rank 0 and broadcast as one flat fp32 arena (RCCL over xGMI on GPUs, gloo on CPU tests), results
gathered to rank 0.  No collective inside the compute loop."""
import os
from typing import Any, List, Sequence, Tuple


def dist_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init_distributed(backend: str = None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1; returns the module or None."""
    rank, local_rank, world = dist_env()
    launched = 'RANK' in os.environ and 'MASTER_PORT' in os.environ     # torch.distributed.run, also with one process: the RCCL
    if world <= 1 and not launched:                                      # communicator / broadcast / gather path runs on a 1-GPU box too
        return None
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:      # 'nccl' is RCCL on ROCm; SOME_AMD_DIST_BACKEND=gloo: dry runs with several ranks on one GPU
            backend = os.environ.get('SOME_AMD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend)
    return dist


def host_workers(world: int) -> Tuple[int, int]:
    """(WAV reader threads, alignment worker processes) for ONE rank: the host is shared by all ranks of the node, so
    both pools are sized from the cores this process may run on divided by the world size (8 readers / 8 workers is
    what one rank can use - measured: 1024 x 30 s files, 0.5 s of reader wait and 0.07 s of alignment backlog)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
    share = max(2, cores // max(1, world))
    return max(2, min(8, share // 2)), max(1, min(8, share // 2))


def partition(sizes: Sequence[float], rank: int, world: int) -> List[int]:
    """Indices owned by ``rank``: largest-first order dealt round-robin (balances work when lengths vary;
    ties keep file order so the split is deterministic)."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    return order[rank::world]


def broadcast_arena(arena, src: int = 0):
    """In-place broadcast of the packed weight arena from ``src`` to every rank (one collective)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(arena, src=src)
    return arena


def gather_to_rank0(items: List[Tuple[int, Any]]) -> List[Tuple[int, Any]]:
    """Collect (row_index, payload) pairs from every rank on rank 0 (others get [])."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(items)
    world, rank = dist.get_world_size(), dist.get_rank()
    buckets = [None] * world if rank == 0 else None
    dist.gather_object(list(items), buckets, dst=0)
    if rank != 0:
        return []
    merged = [x for b in buckets for x in b]
    merged.sort(key=lambda x: x[0])
    return merged
