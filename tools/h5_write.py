#!/usr/bin/env python
"""HDF5 container writer in the layout of the reference's binarizer - measurement / test tooling, not product code.

The reference stores its training data through h5py (utils/indexed_datasets.py:47-77: one HDF5 group per item, one dataset per
attribute, ``h5py.File(path, 'w').create_dataset(f'{item_no}/{k}', data=v)``) plus a numpy ``{prefix}.lengths`` file
(preprocessing/base_binarizer.py:196-199).  h5py is not importable on this image, but the HDF5 library itself is
(/opt/conda/lib/libhdf5.so, 1.10.6): ``write_items`` drives its C API through ctypes with the same calls h5py's ``create_dataset``
makes (default file / group / dataset creation property lists, intermediate groups, contiguous layout, little-endian IEEE /
two's-complement types, numpy bool as h5py's int8 enum {FALSE, TRUE}), so the files are genuine libhdf5 output in the reference's
layout.  Used by oracle/make_binary_fixture.py (the fixtures the product's reader some_amd/utils/hdf5_lite.py is pinned to) and by
tools/make_train_dataset.py (the 3-hour synthetic dataset of bench.py --train)."""
import ctypes as C
import pathlib

import numpy as np

H5 = C.CDLL('/opt/conda/lib/libhdf5.so')
hid = C.c_int64
H5.H5open()
for fn, res, args in [
    ('H5Fcreate', hid, [C.c_char_p, C.c_uint, hid, hid]), ('H5Fclose', C.c_int, [hid]),
    ('H5Gcreate2', hid, [hid, C.c_char_p, hid, hid, hid]), ('H5Gclose', C.c_int, [hid]),
    ('H5Screate_simple', hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]), ('H5Sclose', C.c_int, [hid]),
    ('H5Dcreate2', hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ('H5Dclose', C.c_int, [hid]),
    ('H5Dwrite', C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
    ('H5Pcreate', hid, [hid]), ('H5Pset_layout', C.c_int, [hid, C.c_int]), ('H5Pclose', C.c_int, [hid]),
    ('H5Screate', hid, [C.c_int]),
    ('H5Tenum_create', hid, [hid]), ('H5Tenum_insert', C.c_int, [hid, C.c_char_p, C.c_void_p]), ('H5Tclose', C.c_int, [hid]),
]:
    getattr(H5, fn).restype, getattr(H5, fn).argtypes = res, args


def _g(name):
    return C.c_int64.in_dll(H5, name).value


def _bool_type():
    t = H5.H5Tenum_create(_g('H5T_STD_I8LE_g'))               # h5py maps numpy bool to this enum
    for name, v in ((b'FALSE', 0), (b'TRUE', 1)):
        val = C.c_int8(v)
        assert H5.H5Tenum_insert(t, name, C.byref(val)) >= 0
    return t


def write_items(path: pathlib.Path, items):
    f = H5.H5Fcreate(str(path).encode(), 2, 0, 0)             # H5F_ACC_TRUNC, default fcpl / fapl (libver earliest, as h5py)
    assert f >= 0
    bool_t = _bool_type()
    file_types = {np.dtype('float32'): _g('H5T_IEEE_F32LE_g'), np.dtype('int64'): _g('H5T_STD_I64LE_g'), np.dtype('bool'): bool_t}
    mem_types = {np.dtype('float32'): _g('H5T_NATIVE_FLOAT_g'), np.dtype('int64'): _g('H5T_NATIVE_INT64_g'), np.dtype('bool'): bool_t}
    for no, item in enumerate(items):
        g = H5.H5Gcreate2(f, str(no).encode(), 0, 0, 0)       # the intermediate group of f'{item_no}/{k}'
        assert g >= 0
        for k, v in item.items():
            v = np.ascontiguousarray(v)
            dims = (C.c_uint64 * max(v.ndim, 1))(*v.shape)
            s = H5.H5Screate_simple(v.ndim, dims, None)
            d = H5.H5Dcreate2(g, k.encode(), file_types[v.dtype], s, 0, 0, 0)
            assert s >= 0 and d >= 0
            if v.size:
                assert H5.H5Dwrite(d, mem_types[v.dtype], 0, 0, 0, v.ctypes.data_as(C.c_void_p)) >= 0
            H5.H5Dclose(d)
            H5.H5Sclose(s)
        H5.H5Gclose(g)
    H5.H5Tclose(bool_t)
    assert H5.H5Fclose(f) >= 0
