"""Binarised-dataset fixture generator - TEST INFRASTRUCTURE ONLY, runs only in the build container.

The reference stores its training data through h5py (utils/indexed_datasets.py:47-77: one HDF5 group per item, one
dataset per attribute, written by ``h5py.File(path, 'w').create_dataset(f'{item_no}/{k}', data=v)``) plus a numpy
``{prefix}.lengths`` file (preprocessing/base_binarizer.py:196-199).  h5py is not importable here, but the HDF5 library
itself is on this image (/opt/conda/lib/libhdf5.so, 1.10.6): this script drives its C API through ctypes with the same
calls h5py's ``create_dataset`` makes (default file / group / dataset creation property lists, intermediate groups,
contiguous layout, little-endian IEEE / two's-complement types, numpy bool as h5py's int8 enum {FALSE, TRUE}), so the
files under tests/golden/binary/ are genuine libhdf5 output in the reference's layout.  The product's reader
(some_amd/utils/hdf5_lite.py) is pinned against them.

Items follow preprocessing/me_binarizer.py:22-29 (MIDI_EXTRACTION_ITEM_ATTRIBUTES): units float32 [T, 80], pitch float32
[T], note_midi float32 [n], note_rest bool [n], note_dur int64 [n], unit2note int64 [T]; units come from the CPU oracle
(oracle/restate.logmel) of the synthetic sung clips in some_amd/training/data.py.

Usage:  python oracle/make_binary_fixture.py        (SOME_GOLDEN_OUT=DIR writes elsewhere; the containers carry libhdf5's object
modification times, so a regenerated file differs from the committed one in those 4-byte fields and nowhere else)
"""
import os
import pathlib
import sys

import numpy as np

REPO = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

from oracle import restate  # noqa: E402
from some_amd.configs import get_config  # noqa: E402
from some_amd.training import data  # noqa: E402

OUT = pathlib.Path(os.environ['SOME_GOLDEN_OUT']) if os.environ.get('SOME_GOLDEN_OUT') else REPO / 'tests' / 'golden' / 'binary'
from tools.h5_write import C, H5, _g, write_items  # noqa: E402,F401  (the ctypes binding of libhdf5 + the item writer)


def write_misc(path: pathlib.Path):
    """Shapes and types beside the binarizer's: a scalar (IndexedDataset reads it with .item()), an empty array, other
    widths, a compact-layout dataset and a nested group."""
    f = H5.H5Fcreate(str(path).encode(), 2, 0, 0)
    g = H5.H5Gcreate2(f, b'0', 0, 0, 0)
    other = H5.H5Gcreate2(f, b'1', 0, 0, 0)
    sub = H5.H5Gcreate2(other, b'nested', 0, 0, 0)
    compact = H5.H5Pcreate(_g('H5P_CLS_DATASET_CREATE_ID_g'))
    assert H5.H5Pset_layout(compact, 0) >= 0                  # H5D_COMPACT
    cases = [
        (g, b'scalar_f64', np.float64(2.5), 'H5T_IEEE_F64LE_g', 'H5T_NATIVE_DOUBLE_g', 0),
        (g, b'scalar_i64', np.int64(-7), 'H5T_STD_I64LE_g', 'H5T_NATIVE_INT64_g', 0),
        (g, b'empty', np.zeros((0, 80), np.float32), 'H5T_IEEE_F32LE_g', 'H5T_NATIVE_FLOAT_g', 0),
        (g, b'never_written', None, 'H5T_IEEE_F32LE_g', 'H5T_NATIVE_FLOAT_g', 0),
        (g, b'i32', np.arange(-3, 4, dtype=np.int32), 'H5T_STD_I32LE_g', 'H5T_NATIVE_INT32_g', 0),
        (g, b'u8', np.arange(250, 256, dtype=np.uint8), 'H5T_STD_U8LE_g', 'H5T_NATIVE_UINT8_g', 0),
        (g, b'f64', np.linspace(0, 1, 5), 'H5T_IEEE_F64LE_g', 'H5T_NATIVE_DOUBLE_g', 0),
        (g, b'compact_f32', np.arange(6, dtype=np.float32).reshape(2, 3), 'H5T_IEEE_F32LE_g', 'H5T_NATIVE_FLOAT_g', compact),
        (sub, b'leaf', np.asarray([1, 2, 3], dtype=np.int64), 'H5T_STD_I64LE_g', 'H5T_NATIVE_INT64_g', 0),
    ]
    for loc, name, v, ft, mt, dcpl in cases:
        if v is None:
            dims = (C.c_uint64 * 1)(4)
            s = H5.H5Screate_simple(1, dims, None)
        elif np.ndim(v) == 0:
            s = H5.H5Screate(0)                                # H5S_SCALAR
        else:
            dims = (C.c_uint64 * v.ndim)(*v.shape)
            s = H5.H5Screate_simple(v.ndim, dims, None)
        d = H5.H5Dcreate2(loc, name, _g(ft), s, 0, dcpl, 0)
        assert s >= 0 and d >= 0, name
        if v is not None and np.size(v):
            v = np.ascontiguousarray(v)
            assert H5.H5Dwrite(d, _g(mt), 0, 0, 0, v.ctypes.data_as(C.c_void_p)) >= 0
        H5.H5Dclose(d)
        H5.H5Sclose(s)
    H5.H5Pclose(compact)
    H5.H5Gclose(sub)
    H5.H5Gclose(other)
    H5.H5Gclose(g)
    assert H5.H5Fclose(f) >= 0


def sung_item(index: int, seconds: float, cfg: dict):
    """One binarised item (me_binarizer.py:144-223 with units_encoder 'mel')."""
    wave, note_midi, note_dur_sec, note_rest = data.synth_note_clip(index, seconds)
    units = restate.logmel(wave, cfg)
    length = units.shape[0]
    note_dur, unit2note = data.note_alignment(note_dur_sec, length, cfg['hop_size'] / cfg['audio_sample_rate'])
    rng = np.random.default_rng(index)
    return {'units': units.astype(np.float32), 'pitch': (60 + rng.standard_normal(length)).astype(np.float32),
            'note_midi': note_midi.astype(np.float32), 'note_rest': note_rest.astype(bool), 'note_dur': note_dur.astype(np.int64),
            'unit2note': unit2note.astype(np.int64)}


def tiny_item(index: int):
    rng = np.random.default_rng(900 + index)
    t, n = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    return {'units': rng.standard_normal((t, 80)).astype(np.float32), 'pitch': rng.standard_normal(t).astype(np.float32),
            'note_midi': rng.uniform(40, 80, n).astype(np.float32), 'note_rest': rng.uniform(size=n) < 0.3,
            'note_dur': rng.integers(1, 9, n).astype(np.int64), 'unit2note': rng.integers(1, n + 1, t).astype(np.int64)}


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    cfg = get_config('two_head_model')
    rng = np.random.default_rng(3)
    sets = {
        'train': [sung_item(i, float(rng.uniform(0.6, 2.4)), cfg) for i in range(20)],
        'valid': [sung_item(100 + i, 1.5, cfg) for i in range(3)],
        # enough groups under the root for a two-level group B-tree (32 symbol-table nodes per B-tree node, 8 symbols each)
        'many': [tiny_item(i) for i in range(330)],
    }
    expected = {}
    for prefix, items in sets.items():
        write_items(OUT / f'{prefix}.data', items)
        with open(OUT / f'{prefix}.lengths', 'wb') as fh:      # base_binarizer.py:197-199
            np.save(fh, [it['units'].shape[0] for it in items])
        picks = {'train': (0, 7, 19), 'valid': (0, 2), 'many': (0, 1, 9, 10, 99, 100, 199, 255, 256, 329)}[prefix]
        for i in picks:
            for k, v in items[i].items():
                expected[f'{prefix}.{i}.{k}'] = v
        # every item of every attribute, as one checksum each
        expected[f'{prefix}.sums'] = np.asarray([[float(np.asarray(it[k], dtype=np.float64).sum()) for k in sorted(it)] for it in items])
    write_misc(OUT / 'misc.data')
    np.savez_compressed(OUT / 'expected.npz', **expected)
    for p in sorted(OUT.iterdir()):
        print(p.name, p.stat().st_size)


if __name__ == '__main__':
    main()
