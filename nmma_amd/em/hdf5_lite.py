"""A minimal pure-Python reader for the HDF5 files Keras 2.x writes for its legacy ``.h5`` models -- enough to get the Dense layers'
``kernel`` / ``bias`` arrays out of ``model_weights/<layer>/<layer>/{kernel:0, bias:0}`` (the layout ``nmma/em/model.py:635-648``
loads through ``keras.models.load_model``; written by ``nmma/em/training.py:KerasTrainingModel.save_routine``) without h5py, which
the build image does not have.  Host-side file ingestion, not the compute path.

Supported subset of the HDF5 file format (what h5py's default ``libver='earliest'`` produces): superblock version 0, old-style groups
(symbol-table message -> version-1 B-tree of type 0 -> ``SNOD`` symbol-table nodes -> names in a local heap), version-1 object
headers with continuation blocks, simple dataspaces (version 1 / 2), fixed-point and IEEE floating-point datatypes (little endian),
contiguous (layout class 1) and compact (class 0) storage.  Chunked / filtered datasets, new-style groups and variable-length
attributes are not read -- attributes are skipped altogether (the layer order is recovered from the kernel shapes, see ``em/io.py``).
A STRUCTURE outside the subset (superblock, groups, object headers), a truncated file or a cyclic group link raises
``Hdf5LiteError``; a single DATASET outside it (chunked, compressed, a string datatype ...) is skipped and listed in
``File.skipped`` -- a Keras file carries such datasets next to the Dense weights, and they must not abort the read.
"""
from __future__ import annotations

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5LiteError(ValueError):
    pass


class _Dataset:
    def __init__(self, shape, dtype, address, size, inline=None):
        self.shape, self.dtype, self.address, self.size, self.inline = shape, dtype, address, size, inline


class File:
    """``File(path)`` -> ``.datasets``: ``{"group/sub/name": numpy array}`` for every dataset reachable from the root group."""

    def __init__(self, path=None, data=None):
        if data is None:
            with open(path, "rb") as fh:
                data = fh.read()
        self.buf = bytes(data)
        b = self.buf
        base = b.find(_SIG)
        if base != 0:
            raise Hdf5LiteError("not an HDF5 file (or a user block in front of the superblock)")
        version = b[8]
        if version != 0:
            raise Hdf5LiteError(f"superblock version {version} (only 0 -- libver 'earliest' -- is read)")
        self.so, self.sl = b[13], b[14]              # size of offsets / of lengths
        if self.so != 8 or self.sl != 8:
            raise Hdf5LiteError("only 8-byte offsets and lengths are read")
        # 8 sig | ver, fs-ver, root-ver, 0, shm-ver, so, sl, 0 | leaf k (2), internal k (2) | flags (4) | base, fs-info, eof, driver (8 each)
        # then the root group's symbol-table entry
        root = 24 + 4 * 8
        self.datasets = {}
        self.skipped = {}                 # name -> reason, for datasets outside the supported subset
        self._seen = set()                # object-header addresses already visited (hard links may form cycles)
        try:
            link_off, header, cache, _ = struct.unpack_from("<QQII", b, root)
            scratch = b[root + 24: root + 40]
            self._walk_object(header, "", cache, scratch)
        except (struct.error, IndexError) as exc:             # an offset beyond the end of the buffer: a truncated or damaged file
            raise Hdf5LiteError(f"truncated or damaged HDF5 file: {exc}") from exc
        except RecursionError as exc:
            raise Hdf5LiteError("group nesting too deep") from exc

    # ---- object headers -------------------------------------------------------------------------------------------------------
    def _messages(self, addr):
        """(type, flags, payload bytes) of every message of the version-1 object header at ``addr``."""
        b = self.buf
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise Hdf5LiteError(f"object header version {ver} at {addr} (only version 1 is read)")
        out = []
        blocks = [(addr + 16, hsize)]            # (the 12-byte prefix is padded to 16)
        while blocks and len(out) < nmsg + 64:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg + 64:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8: pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                      # continuation: (address, length) of another block of messages
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr, clen))
                else:
                    out.append((mtype, mflags, body))
        return out

    def _walk_object(self, header, name, cache=0, scratch=b""):
        if header in self._seen:              # a second link to an object already read (or a cycle): nothing new below it
            return
        self._seen.add(header)
        msgs = self._messages(header)
        types = {m[0] for m in msgs}
        if 0x0011 in types or cache == 1:                 # a group: symbol-table message (B-tree address, local-heap address)
            if cache == 1 and len(scratch) >= 16:
                btree, heap = struct.unpack_from("<QQ", scratch, 0)
            else:
                body = next(m[2] for m in msgs if m[0] == 0x0011)
                btree, heap = struct.unpack_from("<QQ", body, 0)
            heap_data = self._local_heap(heap)
            for lname, lheader, lcache, lscratch in self._group_entries(btree, heap_data):
                self._walk_object(lheader, f"{name}/{lname}" if name else lname, lcache, lscratch)
            return
        if 0x0008 in types and 0x0001 in types and 0x0003 in types:          # a dataset
            try:
                self.datasets[name] = self._read(self._dataset(msgs))
            except Hdf5LiteError as exc:      # this dataset only: the others stay readable
                self.skipped[name] = str(exc)
            return
        if 0x0002 in types or 0x0006 in types:
            raise Hdf5LiteError(f"{name}: new-style group (link messages) -- not read")
        # an object with neither children nor data (e.g. an empty group header): nothing to collect

    # ---- groups -----------------------------------------------------------------------------------------------------------------
    def _local_heap(self, addr):
        b = self.buf
        if b[addr:addr + 4] != b"HEAP":
            raise Hdf5LiteError(f"no local heap at {addr}")
        size, _, data = struct.unpack_from("<QQQ", b, addr + 8)
        return b[data: data + size]

    def _group_entries(self, addr, heap):
        b = self.buf
        sig = b[addr:addr + 4]
        if sig == b"TREE":
            ntype, level, used = struct.unpack_from("<BBH", b, addr + 4)
            if ntype != 0:
                raise Hdf5LiteError("B-tree node of a chunked dataset where a group node was expected")
            pos = addr + 8 + 16                           # signature, type, level, entries used, left / right siblings
            for i in range(used):
                child = struct.unpack_from("<Q", b, pos + 8 + i * 16)[0]          # key (8) | child (8) | key ...
                yield from self._group_entries(child, heap)
        elif sig == b"SNOD":
            n = struct.unpack_from("<H", b, addr + 6)[0]
            for i in range(n):
                e = addr + 8 + i * 40
                link_off, header, cache, _ = struct.unpack_from("<QQII", b, e)
                end = heap.find(b"\x00", link_off)
                if link_off >= len(heap) or end < 0:
                    raise Hdf5LiteError(f"link name at heap offset {link_off} is not terminated")
                yield heap[link_off:end].decode("utf-8", "replace"), header, cache, b[e + 24: e + 40]
        else:
            raise Hdf5LiteError(f"neither a B-tree nor a symbol-table node at {addr}")

    # ---- datasets ---------------------------------------------------------------------------------------------------------------
    def _dataset(self, msgs):
        shape = dtype = None
        address = size = None
        inline = None
        for mtype, _, body in msgs:
            if mtype == 0x0001:                            # dataspace
                ver, rank, flags = body[0], body[1], body[2]
                off = 8 if ver == 1 else 4
                if ver not in (1, 2):
                    raise Hdf5LiteError(f"dataspace version {ver}")
                shape = tuple(struct.unpack_from("<Q", body, off + 8 * i)[0] for i in range(rank))
            elif mtype == 0x0003:                          # datatype
                cls, bits0 = body[0] & 0x0F, body[1]
                nbytes = struct.unpack_from("<I", body, 4)[0]
                if bits0 & 1:
                    raise Hdf5LiteError("big-endian data")
                if cls == 1 and nbytes in (2, 4, 8):
                    dtype = np.dtype(f"<f{nbytes}")
                elif cls == 0 and nbytes in (1, 2, 4, 8):
                    dtype = np.dtype(f"<{'i' if bits0 & 8 else 'u'}{nbytes}")
                else:
                    raise Hdf5LiteError(f"datatype class {cls} of {nbytes} bytes (only integers and IEEE floats are read)")
            elif mtype == 0x0008:                          # data layout
                ver = body[0]
                if ver == 3:
                    cls = body[1]
                    if cls == 1:
                        address, size = struct.unpack_from("<QQ", body, 2)
                    elif cls == 0:
                        n = struct.unpack_from("<H", body, 2)[0]
                        inline = body[4:4 + n]
                    else:
                        raise Hdf5LiteError("chunked dataset -- not read")
                elif ver in (1, 2):
                    rank, cls = body[1], body[2]
                    if cls != 1:
                        raise Hdf5LiteError("only contiguous datasets are read (layout versions 1 / 2)")
                    address = struct.unpack_from("<Q", body, 8)[0]
                    size = None
                else:
                    raise Hdf5LiteError(f"data layout version {ver}")
            elif mtype == 0x000B:
                raise Hdf5LiteError("filtered (compressed) dataset -- not read")
        if shape is None or dtype is None or (address is None and inline is None):
            raise Hdf5LiteError("dataset without dataspace, datatype or layout")
        return _Dataset(shape, dtype, address, size, inline)

    def _read(self, ds):
        n = int(np.prod(ds.shape)) if ds.shape else 1
        nbytes = n * ds.dtype.itemsize
        if ds.inline is not None:
            raw = ds.inline[:nbytes]
        else:
            if ds.address == UNDEF:
                return np.zeros(ds.shape, dtype=ds.dtype)            # never written: the fill value
            raw = self.buf[ds.address: ds.address + nbytes]
        if len(raw) != nbytes:
            raise Hdf5LiteError("dataset extends beyond the end of the file")
        return np.frombuffer(raw, dtype=ds.dtype).reshape(ds.shape).copy()


def read_datasets(path=None, data=None):
    """Every dataset of the file at ``path`` (or of the file image ``data``) as ``{"group/.../name": array}``."""
    return File(path, data).datasets
