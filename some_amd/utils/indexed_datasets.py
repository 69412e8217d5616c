"""``IndexedDataset`` with the reference's interface (utils/indexed_datasets.py:10-45): item ``i`` of
``<path>/<prefix>.data`` as a dict of attribute -> torch tensor (python scalar for 0-d datasets), with the same small
most-recently-used cache.  The file is the HDF5 container the reference's binarizer writes; it is read through
some_amd/utils/hdf5_lite.py (no h5py on this image)."""
import pathlib
from collections import deque

import torch

from . import hdf5_lite


class IndexedDataset:
    def __init__(self, path, prefix, num_cache=0):
        self.path = pathlib.Path(path) / f'{prefix}.data'
        if not self.path.exists():
            raise FileNotFoundError(f'IndexedDataset not found: {self.path}')
        self.dset = None
        self.cache = deque(maxlen=num_cache)
        self.num_cache = num_cache

    def _file(self) -> hdf5_lite.File:
        if self.dset is None:                      # opened on first use, so a forked loader worker maps it itself
            self.dset = hdf5_lite.File(self.path)
        return self.dset

    def check_index(self, i):
        if i < 0 or i >= len(self._file()):
            raise IndexError('index out of range')

    def __del__(self):
        if getattr(self, 'dset', None):
            self.dset.close()

    def __getitem__(self, i):
        self.check_index(i)
        for j, item in self.cache:
            if j == i:
                return item
        item = {}
        for k, v in self._file()[str(i)].items():
            a = v[()]
            item[k] = a.item() if v.shape == () else torch.from_numpy(a)
        if self.num_cache > 0:
            self.cache.appendleft((i, item))
        return item

    def __len__(self):
        return len(self._file())
