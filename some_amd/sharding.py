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
        bind_rank_to_cores(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend)
    return dist


_BOUND = False        # this process has been pinned to its own share of the host's cores (bind_rank_to_cores)
_SYSFS = '/sys'       # root of the topology files (tests point it at a fixture tree of a two-socket node)


def _core_groups(allowed) -> List[List[int]]:
    """The logical CPUs of ``allowed`` grouped by physical core (SMT siblings together), ordered by (package, core): from
    /sys/devices/system/cpu/cpu*/topology; without it every logical CPU is its own group, in numeric order."""
    groups = {}
    for cpu in sorted(allowed):
        base = f'{_SYSFS}/devices/system/cpu/cpu{cpu}/topology/'
        try:
            with open(base + 'physical_package_id') as f:
                pkg = int(f.read())
            with open(base + 'core_id') as f:
                core = int(f.read())
        except (OSError, ValueError):
            pkg, core = 0, cpu
        groups.setdefault((pkg, core), []).append(cpu)
    return [groups[k] for k in sorted(groups)]


def rank_core_slice(local_rank: int, local_world: int, allowed=None) -> List[int]:
    """The logical CPUs rank ``local_rank`` of ``local_world`` ranks on this node should run on: whole physical cores, a contiguous run in
    (package, core) order - ranks 0 .. n/2 - 1 on the first socket, the rest on the second, the layout of the GPUs on two-socket MI300 /
    MI355 nodes (GPUs 0 - 3 hang off socket 0)."""
    if allowed is None:
        allowed = os.sched_getaffinity(0)
    groups = _core_groups(allowed)
    per = len(groups) // max(1, local_world)
    if per == 0:
        return sorted(allowed)
    return sorted(c for g in groups[local_rank * per:(local_rank + 1) * per] for c in g)


def _parse_cpulist(text: str) -> List[int]:
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes(local_world: int) -> List[int]:
    """NUMA node of every local GPU (PCI sysfs, via the bus ids torch reports), or [] when any of it is unavailable."""
    try:
        import torch
        if not torch.cuda.is_available() or torch.cuda.device_count() < local_world:
            return []
        nodes = []
        for i in range(local_world):
            pr = torch.cuda.get_device_properties(i)
            addr = f'{getattr(pr, "pci_domain_id", 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
            with open(f'{_SYSFS}/bus/pci/devices/{addr}/numa_node') as f:
                node = int(f.read())
            if node < 0:
                return []
            nodes.append(node)
        return nodes
    except (AttributeError, OSError, ValueError, RuntimeError):
        return []


def rank_core_slice_numa(local_rank: int, local_world: int, nodes: Sequence[int], allowed=None) -> List[int]:
    """As ``rank_core_slice`` but inside the NUMA node the rank's GPU hangs off, shared evenly with the other ranks of that node."""
    if allowed is None:
        allowed = os.sched_getaffinity(0)
    node = nodes[local_rank]
    with open(f'{_SYSFS}/devices/system/node/node{node}/cpulist') as f:
        local = set(_parse_cpulist(f.read())) & set(allowed)
    peers = [r for r in range(local_world) if nodes[r] == node]
    groups = _core_groups(local)
    per = len(groups) // len(peers)
    if per == 0:
        raise ValueError('fewer cores than ranks on the node')
    k = peers.index(local_rank)
    return sorted(c for g in groups[k * per:(k + 1) * per] for c in g)


def bind_rank_to_cores(local_rank: int, local_world: int) -> List[int]:
    """Pin this process (and the threads / worker processes it starts afterwards) to ``rank_core_slice``.  Measured on the 2 x 64-core
    host of the MI355X box with 8 rank processes reading 10 000 distinct 30 s WAV files (tools/host_scaling_bench.py,
    profiles/r04_host_scaling.txt): unpinned, a file load takes 16.8 ms instead of 1.7 ms (threads and their freshly faulted buffers migrate
    across the two sockets) and the node sustains 1 630 rows/s; pinned, 2 400+.  ``SOME_AMD_BIND_CORES=0`` leaves the affinity alone."""
    global _BOUND
    if local_world <= 1 or os.environ.get('SOME_AMD_BIND_CORES', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    cores = None
    nodes = gpu_numa_nodes(local_world)
    if nodes:
        try:
            cores = rank_core_slice_numa(local_rank, local_world, nodes)      # next to the GPU's PCIe root
        except (OSError, ValueError):
            cores = None
    if not cores:
        cores = rank_core_slice(local_rank, local_world)
    os.sched_setaffinity(0, cores)
    _BOUND = True
    return cores


def host_workers(world: int) -> Tuple[int, int]:
    """(WAV reader threads, alignment worker processes) for ONE rank: the host is shared by all ranks of the node, so both pools are sized
    from this rank's share of the cores - the affinity set itself once ``bind_rank_to_cores`` has cut it, otherwise the cores this process
    may run on divided by the world size."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
    share = max(2, cores if _BOUND else cores // max(1, world))
    # 4 + 4 is the measured optimum with 8 ranks on 2 x 64 cores (profiles/r04_host_scaling.txt): a 30 s file loads in 0.9 ms, so 4 readers
    # feed 10 x the 420 files/s a GPU consumes, the alignment of a row costs ~2.5 ms of one core - and every further busy thread of the
    # rank makes all of its file loads slower (8 readers: 7 - 12 ms per file instead of 3.3; 8 alignment workers: -15 % rows/s)
    return max(2, min(4, share // 4)), max(1, min(4, share // 4))


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
