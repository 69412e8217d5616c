"""N > 1 path on CPU: two gloo processes shard the rows of a miniature dataset, rank 0 broadcasts the packed
weight arena, each rank processes only its rows, rank 0 gathers and writes the CSV (the model is the
deterministic FakeInference; the collectives and the sharding logic are the real ones)."""
import os
import pathlib
import subprocess
import sys

import numpy as np

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
assert world == 2 and dist.get_world_size() == 2
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


def test_two_rank_gloo_batch_infer(tmp_path, golden_dir):
    sys.path.insert(0, str(ROOT / 'tests'))
    import dataset_util
    dataset_util.build_dataset(tmp_path)
    worker = tmp_path / 'worker.py'
    worker.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=str(ROOT), DATA_ROOT=str(tmp_path), MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29517', str(worker)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # sharded output == the reference's single-process CSV
    assert (tmp_path / 'out_sharded.csv').read_bytes() == (golden_dir / 'batch_csv_full.csv').read_bytes()
    # both ranks did part of the work (chunks processed), neither did all of it
    seen = [int((tmp_path / f'rank{r}.seen').read_text()) for r in range(2)]
    assert all(s > 0 for s in seen)


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
