"""Read-only HDF5 access for the reference's binarised datasets, without h5py (which this image does not have).

The reference stores training items as ``h5py.File(path, 'w').create_dataset(f'{item_no}/{k}', data=v)``
(utils/indexed_datasets.py:47-77) and reads them back as ``{k: v[()] for k, v in file[str(i)].items()}`` (:27-40).
h5py's defaults pin the on-disk structures to the original HDF5 format (``libver='earliest'``): superblock version 0,
version-1 object headers, groups as symbol tables (version-1 B-tree + local heap + symbol-table nodes), contiguous (or
compact) dataset layout, little-endian fixed-point / IEEE types and numpy ``bool`` as an int8 enum.  Exactly that
subset of the published format specification ("HDF5 File Format Specification Version 2.0") is parsed here, straight
from a read-only memory map; anything else (chunked / filtered datasets, new-style groups of ``libver='latest'``,
compound or variable-length types) raises ``Hdf5FormatError`` with the structure that was met.

Pinned by tests/test_hdf5_lite.py against files written by libhdf5 1.10.6 itself in the reference's layout
(oracle/make_binary_fixture.py)."""
import mmap
import struct
from typing import Dict, Iterator, Tuple

import numpy as np

_SIGNATURE = b'\x89HDF\r\n\x1a\n'
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(RuntimeError):
    pass


class Dataset:
    def __init__(self, file: 'File', name: str, shape: Tuple[int, ...], dtype: np.dtype, is_bool: bool, layout):
        self._file, self.name, self.shape, self._store_dtype, self._is_bool, self._layout = file, name, shape, dtype, is_bool, layout
        self.dtype = np.dtype(bool) if is_bool else dtype

    def __getitem__(self, key):
        """``ds[()]`` / ``ds[...]`` / ``ds[:]`` - the whole array (a copy), as the reference reads it."""
        if key not in ((), Ellipsis) and key != slice(None):
            raise Hdf5FormatError('hdf5_lite datasets are read whole: use ds[()]')
        count = int(np.prod(self.shape, dtype=np.int64))
        kind, a, b = self._layout
        if kind == 'compact':
            raw = np.frombuffer(a, dtype=self._store_dtype, count=count)
        elif a == _UNDEF or count == 0:                       # storage never allocated: the fill value (zero)
            raw = np.zeros(count, dtype=self._store_dtype)
        else:
            if b < count * self._store_dtype.itemsize or a + b > len(self._file._mm):
                raise Hdf5FormatError(f'{self.name}: data extent outside the file')
            raw = np.frombuffer(self._file._mm, dtype=self._store_dtype, count=count, offset=a)
        out = raw.reshape(self.shape)
        return (out != 0) if self._is_bool else out.copy()


class Group:
    def __init__(self, file: 'File', name: str, btree: int, heap: int):
        self._file, self.name, self._btree, self._heap = file, name, btree, heap
        self._links = None

    def _table(self) -> Dict[str, int]:
        if self._links is None:
            self._links = dict(self._file._walk_group(self._btree, self._heap))
        return self._links

    def keys(self):
        return self._table().keys()

    def __len__(self):
        return len(self._table())

    def __contains__(self, name):
        return name in self._table()

    def __iter__(self) -> Iterator[str]:
        return iter(self._table())

    def __getitem__(self, name: str):
        node = self
        for part in name.strip('/').split('/'):
            if not isinstance(node, Group):
                raise KeyError(name)
            table = node._table()
            if part not in table:
                raise KeyError(f"Unable to open object (object '{part}' doesn't exist)")
            node = self._file._open_object(table[part], f'{node.name.rstrip("/")}/{part}')
        return node

    def items(self):
        return [(k, self[k]) for k in self._table()]

    def values(self):
        return [self[k] for k in self._table()]


class File(Group):
    def __init__(self, path, mode: str = 'r'):
        if mode != 'r':
            raise ValueError('hdf5_lite is read-only')
        self._fh = open(path, 'rb')
        try:
            self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise Hdf5FormatError(f'{path}: empty file')
        mm = self._mm
        if mm[:8] != _SIGNATURE:
            self.close()
            raise Hdf5FormatError(f'{path}: not an HDF5 file (no signature at offset 0)')
        version = mm[8]
        if version > 1:
            self.close()
            raise Hdf5FormatError(f'{path}: superblock version {version} (written with libver="latest"?) is not supported; '
                                  f'the reference writes version 0')
        self._so, self._sl = mm[13], mm[14]
        if (self._so, self._sl) != (8, 8):
            self.close()
            raise Hdf5FormatError(f'{path}: {self._so}-byte offsets / {self._sl}-byte lengths are not supported')
        p = 24 + (4 if version == 1 else 0)
        self._base, _, self._eof, _ = struct.unpack_from('<4Q', mm, p)
        p += 32
        if self._base != 0:
            raise Hdf5FormatError(f'{path}: non-zero base address')
        # root group symbol-table entry: name offset, object header address, cache type, reserved, scratch pad
        _, header, cache_type, _ = struct.unpack_from('<QQII', mm, p)
        if cache_type == 1:
            btree, heap = struct.unpack_from('<QQ', mm, p + 24)
        else:
            btree, heap = self._symbol_table_of(header, '/')
        super().__init__(self, '/', btree, heap)

    # ---- life cycle ------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_mm', None) is not None:
            try:
                self._mm.close()
            except BufferError:          # arrays still view the map; it goes with them
                pass
            self._mm = None
        if getattr(self, '_fh', None) is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __bool__(self):
        return self._mm is not None

    # ---- groups ----------------------------------------------------------------------------------------------------
    def _heap_data(self, heap: int) -> int:
        mm = self._mm
        if mm[heap:heap + 4] != b'HEAP':
            raise Hdf5FormatError(f'no local heap at {heap}')
        return struct.unpack_from('<Q', mm, heap + 24)[0]

    def _walk_group(self, btree: int, heap: int):
        """(name, object header address) of every link of an old-style group, in B-tree (= name) order."""
        mm = self._mm
        names = self._heap_data(heap)
        stack = [btree]
        while stack:
            node = stack.pop()
            if mm[node:node + 4] != b'TREE':
                raise Hdf5FormatError(f'no B-tree node at {node}')
            node_type, level, used = struct.unpack_from('<BBH', mm, node + 4)
            if node_type != 0:
                raise Hdf5FormatError(f'B-tree node at {node} is not a group node')
            # keys and children interleave after the two sibling addresses: key0 child0 key1 child1 ... key_used
            children = [struct.unpack_from('<Q', mm, node + 24 + 16 * i + 8)[0] for i in range(used)]
            if level > 0:
                stack.extend(reversed(children))
                continue
            for snod in children:
                if mm[snod:snod + 4] != b'SNOD':
                    raise Hdf5FormatError(f'no symbol-table node at {snod}')
                count = struct.unpack_from('<H', mm, snod + 6)[0]
                for e in range(count):
                    name_off, header = struct.unpack_from('<QQ', mm, snod + 8 + 40 * e)
                    start = names + name_off
                    end = mm.find(b'\x00', start)
                    yield mm[start:end].decode('utf8'), header

    # ---- object headers --------------------------------------------------------------------------------------------
    def _messages(self, header: int):
        """(type, flags, offset, size) of every message of a version-1 object header, continuation blocks included."""
        mm = self._mm
        if mm[header:header + 4] == b'OHDR':
            raise Hdf5FormatError('version-2 object header (file written with libver="latest") is not supported')
        version, _, total, _, size = struct.unpack_from('<BBHII', mm, header)
        if version != 1:
            raise Hdf5FormatError(f'object header version {version} at {header} is not supported')
        blocks = [(header + 16, size)]
        seen = 0
        while blocks and seen < total:
            p, left = blocks.pop(0)
            while left >= 8 and seen < total:
                mtype, msize, flags = struct.unpack_from('<HHB', mm, p)
                seen += 1
                if mtype == 0x0010:
                    blocks.append(struct.unpack_from('<QQ', mm, p + 8))
                else:
                    yield mtype, flags, p + 8, msize
                p += 8 + msize
                left -= 8 + msize

    def _symbol_table_of(self, header: int, name: str):
        for mtype, _, p, _ in self._messages(header):
            if mtype == 0x0011:
                return struct.unpack_from('<QQ', self._mm, p)
            if mtype in (0x0002, 0x0006):
                raise Hdf5FormatError(f'{name}: new-style group (link messages) is not supported')
        raise Hdf5FormatError(f'{name}: object header has neither a symbol table nor a dataset')

    def _datatype(self, p: int, name: str):
        """-> (numpy dtype as stored, is_bool, bytes consumed)."""
        mm = self._mm
        head, b0, b1, _, size = struct.unpack_from('<BBBBI', mm, p)
        cls, version = head & 15, head >> 4
        if cls == 0:                                            # fixed point: bit 0 byte order, bit 3 signed
            if b0 & 1:
                raise Hdf5FormatError(f'{name}: big-endian integers are not supported')
            return np.dtype(f'<{"i" if b0 & 8 else "u"}{size}'), False, 8 + 4
        if cls == 1:                                            # floating point
            if b0 & 1 or size not in (2, 4, 8):
                raise Hdf5FormatError(f'{name}: unsupported floating-point type (size {size})')
            return np.dtype(f'<f{size}'), False, 8 + 12
        if cls == 8:                                            # enumeration: h5py's numpy bool
            members = b0 | (b1 << 8)
            base, _, used = self._datatype(p + 8, name)
            q = p + 8 + used
            labels = []
            for _ in range(members):
                end = mm.find(b'\x00', q)
                labels.append(mm[q:end])
                q = end + 1 if version >= 3 else q + ((end - q) // 8 + 1) * 8
            values = np.frombuffer(mm, dtype=base, count=members, offset=q).tolist()
            if dict(zip(labels, values)) != {b'FALSE': 0, b'TRUE': 1}:
                raise Hdf5FormatError(f'{name}: enumeration {labels} is not h5py\'s bool')
            return base, True, q + members * base.itemsize - p
        raise Hdf5FormatError(f'{name}: datatype class {cls} is not supported (only integers, floats and bool)')

    def _open_object(self, header: int, name: str):
        mm = self._mm
        shape = dtype = layout = None
        is_bool = False
        for mtype, _, p, msize in self._messages(header):
            if mtype == 0x0011:
                return Group(self, name, *struct.unpack_from('<QQ', mm, p))
            if mtype == 0x0001:
                version, rank, flags = struct.unpack_from('<BBB', mm, p)
                if version not in (1, 2):
                    raise Hdf5FormatError(f'{name}: dataspace version {version}')
                if version == 2 and mm[p + 3] == 2:
                    raise Hdf5FormatError(f'{name}: null dataspace')
                q = p + (8 if version == 1 else 4)
                shape = struct.unpack_from(f'<{rank}Q', mm, q)
            elif mtype == 0x0003:
                dtype, is_bool, _ = self._datatype(p, name)
            elif mtype == 0x0008:
                version, cls = struct.unpack_from('<BB', mm, p)
                if version != 3:
                    raise Hdf5FormatError(f'{name}: data layout message version {version} is not supported')
                if cls == 0:
                    n = struct.unpack_from('<H', mm, p + 2)[0]
                    layout = ('compact', bytes(mm[p + 4:p + 4 + n]), n)
                elif cls == 1:
                    layout = ('contiguous',) + struct.unpack_from('<QQ', mm, p + 2)
                else:
                    raise Hdf5FormatError(f'{name}: chunked / virtual layout (class {cls}) is not supported; the reference '
                                          f'writes contiguous datasets')
            elif mtype == 0x000B:
                raise Hdf5FormatError(f'{name}: filtered (compressed) datasets are not supported')
            elif mtype in (0x0002, 0x0006):
                raise Hdf5FormatError(f'{name}: new-style group (link messages) is not supported')
        if shape is None or dtype is None or layout is None:
            raise Hdf5FormatError(f'{name}: object header describes neither a group nor a dataset')
        return Dataset(self, name, tuple(int(d) for d in shape), dtype, is_bool, layout)
