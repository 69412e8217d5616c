"""N > 1 path on CPU: two gloo processes shard the rows of a miniature dataset, rank 0 broadcasts the packed
weight arena, each rank processes only its rows, rank 0 gathers and writes the CSV (the model is the
deterministic FakeInference; the collectives and the sharding logic are the real ones)."""
import os
import pathlib
import subprocess
import sys

import numpy as np
import pytest

from some_amd import sharding

ROOT = pathlib.Path(__file__).resolve().parents[1]

WORKER = r'''
import os, sys, pathlib
sys.path.insert(0, os.environ['REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['REPO_ROOT'], 'tests'))
import numpy as np, torch
import batch_infer as bi, dataset_util
from some_amd import sharding
from some_amd.configs import get_config
from some_amd.engine import Engine
from some_amd import synth
root = pathlib.Path(os.environ['DATA_ROOT'])
dist = sharding.init_distributed('gloo')
rank, _, world = sharding.dist_env()
assert world == int(os.environ['EXPECT_WORLD']) and dist.get_world_size() == world
# weight broadcast: rank 0 packs, everyone ends up with the identical arena (gloo stands in for RCCL)
cfg = get_config('midi_conformer', lay=0)
eng = Engine(cfg, host_only=True)
arena = torch.zeros(eng.arena_numel)
if rank == 0:
    arena.copy_(eng.pack_state_dict(synth.synth_state_dict(cfg, 5)))
sharding.broadcast_arena(arena, src=0)
ref = eng.pack_state_dict(synth.synth_state_dict(cfg, 5))
assert torch.equal(arena, ref), 'arena broadcast mismatch'
# the command itself, with the model stubbed
seen = []
class Spy(dataset_util.FakeIngestInference):
    def infer(self, waves):
        seen.append(len(waves))
        return super().infer(waves)
bi.model_init = lambda p: (Spy(), get_config('midi_conformer'))
out = root / 'out_sharded.csv'
bi.batch_infer.callback(dataset=str(root), model=str(root / 'm.ckpt'), round_midi=False, csv=str(out), overwrite=True)
(root / f'rank{rank}.seen').write_text(str(sum(seen)))
'''


def test_partition_is_balanced_and_complete():
    sizes = [5, 1, 9, 3, 7, 2, 8]
    parts = [sharding.partition(sizes, r, 3) for r in range(3)]
    assert sorted(i for p in parts for i in p) == list(range(len(sizes)))
    assert parts[0][0] == 2 and parts[1][0] == 6 and parts[2][0] == 4      # largest first, dealt round-robin
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(sizes)
    assert sharding.partition(sizes, 0, 1) == sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world', [2, 8])
def test_gloo_ranks_batch_infer(tmp_path, golden_dir, world):
    """world 2 and world 8 (the node size BASELINE configs[3] names; /root/reference/batch_infer.py:164-226 is the loop being sharded):
    with 8 ranks the fixture's rows do not fill every rank - a rank with NO rows must still take part in the broadcast and the
    gather, and the CSV must come out in file order whatever the deal was."""
    sys.path.insert(0, str(ROOT / 'tests'))
    import dataset_util
    dataset_util.build_dataset(tmp_path)
    worker = tmp_path / 'worker.py'
    worker.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=str(ROOT), DATA_ROOT=str(tmp_path), MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', EXPECT_WORLD=str(world))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(worker)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # sharded output == the reference's single-process CSV
    assert (tmp_path / 'out_sharded.csv').read_bytes() == (golden_dir / 'batch_csv_full.csv').read_bytes()
    # the work was spread (chunks processed per rank), no rank did all of it
    seen = [int((tmp_path / f'rank{r}.seen').read_text()) for r in range(world)]
    assert sum(1 for s in seen if s > 0) >= 2 and max(seen) < sum(seen)
    if world == 2:
        assert all(s > 0 for s in seen)


def test_partition_over_eight_ranks():
    """partition(..., world=8): complete, disjoint, size-balanced; fewer rows than ranks leaves the last ranks empty (and they must cope)."""
    rng = np.random.default_rng(0)
    sizes = list(rng.uniform(5.0, 30.0, 10_000))
    parts = [sharding.partition(sizes, r, 8) for r in range(8)]
    assert sorted(i for p in parts for i in p) == list(range(10_000)) and {len(p) for p in parts} == {1250}
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert (max(loads) - min(loads)) / max(loads) < 2e-3                 # round-robin after the size sort: within one row of each other
    few = [sharding.partition([3.0, 1.0, 2.0], r, 8) for r in range(8)]
    assert few[:3] == [[0], [2], [1]] and few[3:] == [[]] * 5


def _fake_two_socket_node(root: pathlib.Path, cores_per_socket=64, smt=2):
    """A sysfs fixture of the MI355X box's host: 2 packages x 64 cores x 2 hardware threads, Linux numbering (cpu i and i + 128 are
    siblings), one NUMA node per package."""
    n_core = 2 * cores_per_socket
    for cpu in range(n_core * smt):
        phys = cpu % n_core
        d = root / 'devices' / 'system' / 'cpu' / f'cpu{cpu}' / 'topology'
        d.mkdir(parents=True)
        (d / 'physical_package_id').write_text(f'{phys // cores_per_socket}\n')
        (d / 'core_id').write_text(f'{phys % cores_per_socket}\n')
    for node in range(2):
        d = root / 'devices' / 'system' / 'node' / f'node{node}'
        d.mkdir(parents=True)
        lo = node * cores_per_socket
        (d / 'cpulist').write_text(f'{lo}-{lo + cores_per_socket - 1},{n_core + lo}-{n_core + lo + cores_per_socket - 1}\n')
    return set(range(n_core * smt))


def test_eight_ranks_on_a_two_socket_node_get_whole_cores_next_to_their_gpu(tmp_path, monkeypatch):
    """rank_core_slice / rank_core_slice_numa / host_workers at the node size the scaling run uses (8 ranks, 2 x 64 cores x SMT 2)."""
    allowed = _fake_two_socket_node(tmp_path)
    monkeypatch.setattr(sharding, '_SYSFS', str(tmp_path))
    groups = sharding._core_groups(allowed)
    assert len(groups) == 128 and groups[0] == [0, 128] and groups[64] == [64, 192]
    slices = [sharding.rank_core_slice(r, 8, allowed) for r in range(8)]
    assert all(len(s) == 32 for s in slices) and sorted(c for s in slices for c in s) == sorted(allowed)
    for r, s in enumerate(slices):
        assert s == sorted(list(range(16 * r, 16 * r + 16)) + list(range(128 + 16 * r, 128 + 16 * r + 16)))      # 16 physical cores + their siblings
        assert {(c % 128) // 64 for c in s} == {r // 4}                                                          # ranks 0-3 socket 0, 4-7 socket 1
    # NUMA-aware variant: GPUs 0-3 on node 0, 4-7 on node 1 (and an interleaved enumeration, which the plain slice would get wrong)
    for nodes in ([0, 0, 0, 0, 1, 1, 1, 1], [0, 1, 0, 1, 0, 1, 0, 1]):
        ns = [sharding.rank_core_slice_numa(r, 8, nodes, allowed) for r in range(8)]
        assert all(len(s) == 32 for s in ns) and sorted(c for s in ns for c in s) == sorted(allowed)
        for r, s in enumerate(ns):
            assert {(c % 128) // 64 for c in s} == {nodes[r]}
            assert all(set(g) <= set(s) or not (set(g) & set(s)) for g in groups)
    # a cgroup that leaves a rank's node fewer cores than ranks: an error the caller turns into the plain slice
    with pytest.raises(ValueError):
        sharding.rank_core_slice_numa(0, 8, [0] * 8, set(range(4)))
    # bound rank: pools sized from the 32 logical CPUs of its slice; 8 unbound ranks on the whole host: the same 4 + 4
    monkeypatch.setattr(os, 'sched_getaffinity', lambda _pid: set(slices[3]))
    monkeypatch.setattr(sharding, '_BOUND', True)
    assert sharding.host_workers(8) == (4, 4)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda _pid: allowed)
    monkeypatch.setattr(sharding, '_BOUND', False)
    assert sharding.host_workers(8) == (4, 4)
    # bind_rank_to_cores applies exactly that slice (no GPU here: falls back from the NUMA lookup to the plain slice)
    bound = {}
    monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cores: bound.update(pid=pid, cores=sorted(cores)))
    monkeypatch.setattr(sharding, 'gpu_numa_nodes', lambda n: [])
    assert sharding.bind_rank_to_cores(5, 8) == slices[5] and bound == {'pid': 0, 'cores': slices[5]} and sharding._BOUND


def test_rank_core_slices_are_disjoint_whole_cores_and_size_the_worker_pools(monkeypatch):
    """Every rank of a node gets its own run of physical cores (SMT siblings together); pools are sized from that share."""
    import os
    from some_amd import sharding
    allowed = sorted(os.sched_getaffinity(0))
    for world in (1, 2, 4):
        slices = [sharding.rank_core_slice(r, world, allowed) for r in range(world)]
        flat = [c for s in slices for c in s]
        assert len(flat) == len(set(flat)) and set(flat) <= set(allowed)
        if len(sharding._core_groups(allowed)) >= world:
            assert all(slices) and len({len(s) for s in slices}) == 1
            groups = sharding._core_groups(allowed)
            for s in slices:                                   # no physical core is split between two ranks
                assert all(set(g) <= set(s) or not (set(g) & set(s)) for g in groups)
    assert sharding._parse_cpulist('0-2,7,9-10\n') == [0, 1, 2, 7, 9, 10]
    # pool sizes: an unbound rank divides the visible cores by the world size, a bound one uses its own affinity set
    monkeypatch.setattr(os, 'sched_getaffinity', lambda _pid: set(range(256)))
    monkeypatch.setattr(sharding, '_BOUND', False)
    assert sharding.host_workers(8) == (4, 4) and sharding.host_workers(64) == (2, 1)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda _pid: set(range(32)))
    monkeypatch.setattr(sharding, '_BOUND', True)
    assert sharding.host_workers(8) == (4, 4)
