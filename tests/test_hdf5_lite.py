"""The reference's binarised-dataset container (utils/indexed_datasets.py: one HDF5 group per item, one dataset per
attribute) read WITHOUT h5py: some_amd/utils/hdf5_lite.py against files written by libhdf5 1.10.6 itself in that layout
(tests/golden/binary/, made by oracle/make_binary_fixture.py), then IndexedDataset / MIDIExtractionDataset / the samplers on top."""
import numpy as np
import pytest
import torch

from some_amd.configs import get_config
from some_amd.utils import hdf5_lite
from some_amd.utils.indexed_datasets import IndexedDataset

ATTRS = {'units': np.float32, 'pitch': np.float32, 'note_midi': np.float32, 'note_rest': np.bool_, 'note_dur': np.int64,
         'unit2note': np.int64}          # preprocessing/me_binarizer.py:22-29


@pytest.fixture(scope='module')
def binary_dir(golden_dir):
    return golden_dir / 'binary'


@pytest.mark.parametrize('prefix', ['train', 'valid', 'many'])
def test_every_item_of_every_attribute(binary_dir, prefix):
    exp = np.load(binary_dir / 'expected.npz')
    lengths = np.load(binary_dir / f'{prefix}.lengths')
    with hdf5_lite.File(binary_dir / f'{prefix}.data') as f:
        assert len(f) == len(lengths) and set(f.keys()) == {str(i) for i in range(len(lengths))}
        sums = []
        for i in range(len(f)):
            item = {k: v[()] for k, v in f[str(i)].items()}           # the reference's read (indexed_datasets.py:36)
            assert {k: v.dtype.type for k, v in item.items()} == ATTRS
            assert item['units'].shape == (lengths[i], 80) and item['unit2note'].shape == (lengths[i],)
            sums.append([float(np.asarray(item[k], dtype=np.float64).sum()) for k in sorted(item)])
        np.testing.assert_array_equal(np.asarray(sums), exp[f'{prefix}.sums'])
        for key in exp.files:
            if key.startswith(prefix + '.') and not key.endswith('.sums'):
                _, i, k = key.split('.')
                got = f[f'{i}/{k}'][()]
                assert got.dtype == exp[key].dtype and got.shape == exp[key].shape
                np.testing.assert_array_equal(got, exp[key])


def test_two_level_group_btree_is_walked(binary_dir):
    with hdf5_lite.File(binary_dir / 'many.data') as f:
        assert f._mm[f._btree + 5] >= 1                                # root B-tree node level: the 330 groups need two levels
        assert list(f.keys()) == sorted(str(i) for i in range(330))    # B-tree order = name order


def test_other_shapes_types_and_layouts(binary_dir):
    with hdf5_lite.File(binary_dir / 'misc.data') as f:
        g = f['0']
        assert g['scalar_f64'].shape == () and g['scalar_f64'][()].item() == 2.5
        assert g['scalar_i64'][()].item() == -7
        assert g['empty'][()].shape == (0, 80)
        np.testing.assert_array_equal(g['never_written'][()], np.zeros(4, np.float32))     # no storage allocated: fill value
        np.testing.assert_array_equal(g['i32'][()], np.arange(-3, 4, dtype=np.int32))
        np.testing.assert_array_equal(g['u8'][()], np.arange(250, 256, dtype=np.uint8))
        np.testing.assert_array_equal(g['f64'][()], np.linspace(0, 1, 5))
        np.testing.assert_array_equal(g['compact_f32'][()], np.arange(6, dtype=np.float32).reshape(2, 3))
        np.testing.assert_array_equal(f['1/nested/leaf'][()], [1, 2, 3])
        assert 'nested' in f['1'] and isinstance(f['1']['nested'], hdf5_lite.Group)
        with pytest.raises(KeyError):
            g['absent']


def test_rejects_what_it_does_not_parse(tmp_path, binary_dir):
    p = tmp_path / 'x.data'
    p.write_bytes(b'not an hdf5 file at all')
    with pytest.raises(hdf5_lite.Hdf5FormatError, match='not an HDF5 file'):
        hdf5_lite.File(p)
    raw = bytearray((binary_dir / 'misc.data').read_bytes())
    raw[8] = 2                                                         # superblock version of libver='latest'
    p.write_bytes(bytes(raw))
    with pytest.raises(hdf5_lite.Hdf5FormatError, match='superblock version 2'):
        hdf5_lite.File(p)
    with pytest.raises(ValueError):
        hdf5_lite.File(binary_dir / 'misc.data', 'w')


def test_indexed_dataset_interface(binary_dir):
    ds = IndexedDataset(binary_dir, 'train', num_cache=2)
    assert len(ds) == 20
    item = ds[7]
    exp = np.load(binary_dir / 'expected.npz')
    for k in ATTRS:
        assert torch.is_tensor(item[k])
        np.testing.assert_array_equal(item[k].numpy(), exp[f'train.7.{k}'])
    assert item['note_rest'].dtype == torch.bool and item['note_dur'].dtype == torch.int64
    assert ds[7] is item and ds[0] is not item                          # most-recently-used cache
    with pytest.raises(IndexError):
        ds[20]
    with pytest.raises(FileNotFoundError):
        IndexedDataset(binary_dir, 'absent')
    misc = IndexedDataset(binary_dir, 'misc')[0]                        # 0-d datasets come back as python scalars
    assert misc['scalar_f64'] == 2.5 and isinstance(misc['scalar_i64'], int)


def test_dataset_sampler_collater_pipeline(binary_dir, tmp_path):
    """training/base_task.py:135-142, 360-395 on the fixture: every epoch covers every item in budget-respecting batches
    and the collated batch has the shapes the model reads."""
    from some_amd.training import data
    from some_amd.training.samplers import DsBatchSampler, DsEvalBatchSampler
    cfg = get_config('two_head_model')
    train = data.MIDIExtractionDataset(cfg, binary_dir, 'train', allow_aug=True)
    valid = data.MIDIExtractionDataset(cfg, binary_dir, 'valid')
    assert len(train) == 20 and len(valid) == 3 and train.num_frames(0) == train[0]['units'].shape[0]
    sm = DsBatchSampler(train, max_batch_frames=600, max_batch_size=4, num_replicas=1, rank=0, sort_by_similar_size=True,
                        frame_count_grid=6, shuffle_sample=True, seed=cfg['seed'])
    plans = []
    for epoch in range(2):
        sm.set_epoch(epoch)
        plan = [list(map(int, b)) for b in sm]
        assert sorted(i for b in plan for i in b) == list(range(20))
        plans.append(plan)
        for idx in plan:
            batch = train.collater([train[i] for i in idx])
            t = max(int(train.sizes[i]) for i in idx)
            assert len(idx) * t <= 600 and batch['size'] == len(idx)
            assert batch['units'].shape == (len(idx), t, 80) and batch['probs'].shape == (len(idx), t, 128)
            assert batch['bounds'].shape == (len(idx), t) and batch['unit2note'].dtype == torch.int64
            assert float(batch['probs'].max()) <= 1.0 and float(batch['bounds'].sum()) > 0
    assert plans[0] != plans[1]
    assert list(DsEvalBatchSampler(valid, 10000, 1, rank=0, batch_by_size=False)) == [[0], [1], [2]]
    import shutil
    shutil.copy(binary_dir / 'valid.data', tmp_path / 'valid.data')
    shutil.copy(binary_dir / 'train.lengths', tmp_path / 'valid.lengths')
    with pytest.raises(ValueError, match='lengths'):
        data.MIDIExtractionDataset(cfg, tmp_path, 'valid')
