"""Minimal read-only HDF5 reader for BubbleML trajectory files ("own reader; no h5py on either box").

The reference opens one HDF5 file per simulation with h5py and reads one ``(T_total, H, W)`` dataset per field
(bubbleformer/data/dataset.py:43-50, 134-136).  This module reads exactly that subset of the format, from the published HDF5 file
format specification (v1.x "old style" files, which is what ``samples/*.hdf5`` and h5py's default writer produce):

  superblock v0/v1 -> root symbol-table entry -> group B-tree (v1, node type 0) -> symbol nodes (SNOD) + local heap (names)
  -> object header v1 (incl. continuation blocks) -> dataspace / datatype / data-layout messages
  -> contiguous (or compact) data, returned as a zero-copy ``numpy.memmap``

Little-endian IEEE floats and fixed-point integers.  Contiguous layout (what h5py writes unless chunking / compression is asked
for, and what ``samples/*.hdf5`` contain) comes back as a zero-copy ``numpy.memmap``; chunked layout (data-layout message v3,
B-tree v1 chunk index) with the deflate / shuffle / fletcher32 filters -- what ``compression="gzip"`` produces -- is assembled into
an array on first access.  Other filters (szip, lzf, ...), new-style (v2) groups / object headers and layout message v4 raise
``Hdf5Error`` -- nothing is guessed.
"""
import struct
import zlib
from typing import Dict, List, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(RuntimeError):
    pass


class Dataset:
    """Lazy handle: ``shape``, ``dtype``, ``[...]`` / ``[a:b]`` indexing along the first axis like the h5py objects the reference uses."""

    def __init__(self, f: "File", name: str, shape: Tuple[int, ...], dtype: np.dtype, layout: dict, filters: List[Tuple[int, tuple]]):
        self._f, self.name, self.shape, self.dtype, self._layout, self._filters = f, name, tuple(shape), dtype, layout, filters
        self._array = None

    def array(self) -> np.ndarray:
        if self._array is None:
            lay = self._layout
            if lay["class"] == "contiguous":
                if lay["addr"] == UNDEF:
                    self._array = np.zeros(self.shape, self.dtype)
                else:
                    self._array = np.memmap(self._f.path, dtype=self.dtype, mode="r", offset=self._f.base + lay["addr"], shape=self.shape)
            elif lay["class"] == "compact":
                self._array = np.frombuffer(lay["data"], dtype=self.dtype).reshape(self.shape)
            elif lay["class"] == "chunked":
                self._array = self._read_chunked()
            else:
                raise Hdf5Error(f"dataset {self.name}: {lay['class']} layout is not supported")
        return self._array

    # ------------------------------------------------------------------ chunked storage
    def _unfilter(self, raw: bytes, mask: int) -> bytes:
        """Undo the filter pipeline of one chunk (filters were applied in list order when it was written)."""
        for i in range(len(self._filters) - 1, -1, -1):
            if (mask >> i) & 1:
                continue                               # this filter was skipped for the chunk
            fid, cd = self._filters[i]
            if fid == 1:                               # deflate
                raw = zlib.decompress(raw)
            elif fid == 2:                             # shuffle: byte j of every element is stored together
                es = cd[0] if cd else self.dtype.itemsize
                n = len(raw) // es
                body = np.frombuffer(raw, dtype=np.uint8, count=n * es).reshape(es, n).T.tobytes()
                raw = body + raw[n * es:]
            elif fid == 3:                             # fletcher32: 4 checksum bytes follow the data
                raw = raw[:-4]
            else:
                raise Hdf5Error(f"dataset {self.name}: filter id {fid} is not supported (deflate, shuffle, fletcher32 only)")
        return raw

    def _read_chunked(self) -> np.ndarray:
        lay, f = self._layout, self._f
        cdims = lay["chunk"]
        nd = len(self.shape)
        if len(cdims) != nd:
            raise Hdf5Error(f"dataset {self.name}: chunk rank {len(cdims)} != data rank {nd}")
        out = np.zeros(self.shape, self.dtype)
        if lay["btree"] == UNDEF:
            return out                                 # nothing was ever written
        nbytes = int(np.prod(cdims)) * self.dtype.itemsize

        def walk(addr: int):
            a = f.base + addr
            if f._bytes(a, 4) != b"TREE" or f._u(a + 4, 1) != 1:
                raise Hdf5Error(f"dataset {self.name}: chunk B-tree node expected")
            level, used = f._u(a + 5, 1), f._u(a + 6, 2)
            ksize = 8 + 8 * (nd + 1)
            p = a + 8 + 16
            for i in range(used):
                k = p + i * (ksize + 8)
                size, mask = f._u(k, 4), f._u(k + 4, 4)
                offs = struct.unpack_from("<" + "Q" * nd, f._bytes(k + 8, 8 * nd), 0)
                child = f._u(k + ksize, 8)
                if level > 0:
                    walk(child)
                    continue
                raw = self._unfilter(f._bytes(f.base + child, size), mask)
                if len(raw) != nbytes:
                    raise Hdf5Error(f"dataset {self.name}: chunk at {offs} has {len(raw)} bytes, expected {nbytes}")
                chunk = np.frombuffer(raw, dtype=self.dtype).reshape(cdims)
                take = tuple(slice(0, min(c, s - o)) for c, s, o in zip(cdims, self.shape, offs))     # edge chunks stick out
                out[tuple(slice(o, o + t.stop) for o, t in zip(offs, take))] = chunk[take]
        walk(lay["btree"])
        return out

    def __getitem__(self, key):
        return self.array()[key]

    def __len__(self):
        return self.shape[0]


class File:
    def __init__(self, path: str, mode: str = "r"):
        if mode != "r":
            raise Hdf5Error("hdf5_lite is read-only")
        self.path = path
        with open(path, "rb") as fh:
            self.buf = fh.read(1 << 20)          # metadata lives at the front of these files; grown on demand by _bytes()
        self._fh = open(path, "rb")
        self.base = 0
        self._parse_superblock()
        self._datasets: Dict[str, Dataset] = {}
        self._walk_group(self.root_btree, self.root_heap)

    # ------------------------------------------------------------------ raw access
    def _bytes(self, off: int, n: int) -> bytes:
        if off + n <= len(self.buf):
            return self.buf[off:off + n]
        self._fh.seek(off)
        b = self._fh.read(n)
        if len(b) != n:
            raise Hdf5Error(f"truncated file: wanted {n} bytes at {off}")
        return b

    def _u(self, off: int, n: int) -> int:
        return int.from_bytes(self._bytes(off, n), "little")

    # ------------------------------------------------------------------ superblock
    def _parse_superblock(self):
        if self._bytes(0, 8) != SIGNATURE:
            raise Hdf5Error("not an HDF5 file (signature must be at offset 0)")
        ver = self._u(8, 1)
        if ver not in (0, 1):
            raise Hdf5Error(f"superblock version {ver} not supported (only the v0/v1 'old style' layout)")
        self.so, self.sl = self._u(13, 1), self._u(14, 1)
        if self.so != 8 or self.sl != 8:
            raise Hdf5Error("only 8-byte offsets / lengths are supported")
        p = 24 + (4 if ver == 1 else 0)
        self.base = self._u(p, 8)
        p += 4 * 8                                   # base, free-space, end-of-file, driver-info addresses
        # root group symbol table entry
        _name_off, ohdr, cache_type = self._u(p, 8), self._u(p + 8, 8), self._u(p + 16, 4)
        if cache_type == 1:
            self.root_btree, self.root_heap = self._u(p + 24, 8), self._u(p + 32, 8)
        else:
            msgs = self._object_header(ohdr)
            st = [m for m in msgs if m[0] == 0x0011]
            if not st:
                raise Hdf5Error("root group has no symbol table message")
            self.root_btree, self.root_heap = struct.unpack_from("<QQ", st[0][1], 0)

    # ------------------------------------------------------------------ groups
    def _heap_name(self, heap_addr: int, off: int) -> str:
        a = self.base + heap_addr
        if self._bytes(a, 4) != b"HEAP":
            raise Hdf5Error("local heap signature missing")
        data_addr = self.base + self._u(a + 24, 8)
        out = bytearray()
        while True:
            c = self._bytes(data_addr + off + len(out), 1)
            if c == b"\x00":
                return out.decode()
            out += c

    def _walk_group(self, btree: int, heap: int, prefix: str = ""):
        a = self.base + btree
        if self._bytes(a, 4) != b"TREE":
            raise Hdf5Error("group B-tree signature missing")
        ntype, level, used = self._u(a + 4, 1), self._u(a + 5, 1), self._u(a + 6, 2)
        if ntype != 0:
            raise Hdf5Error("expected a group B-tree node")
        p = a + 8 + 16                                  # header + left / right sibling addresses
        for i in range(used):
            child = self._u(p + 8 + i * 16, 8)          # key_i (8) child_i (8) ...
            if level > 0:
                self._walk_group(child, heap, prefix)
            else:
                self._symbol_node(child, heap, prefix)

    def _symbol_node(self, addr: int, heap: int, prefix: str):
        a = self.base + addr
        if self._bytes(a, 4) != b"SNOD":
            raise Hdf5Error("symbol node signature missing")
        n = self._u(a + 6, 2)
        for i in range(n):
            e = a + 8 + i * 40
            name = self._heap_name(heap, self._u(e, 8))
            ohdr, cache_type = self._u(e + 8, 8), self._u(e + 16, 4)
            if cache_type == 1:                          # sub-group with cached B-tree / heap addresses
                self._walk_group(self._u(e + 24, 8), self._u(e + 32, 8), prefix + name + "/")
                continue
            msgs = self._object_header(ohdr)
            sub = [m for m in msgs if m[0] == 0x0011]
            if sub:
                bt, hp = struct.unpack_from("<QQ", sub[0][1], 0)
                self._walk_group(bt, hp, prefix + name + "/")
            elif any(m[0] == 0x0008 for m in msgs):
                self._datasets[prefix + name] = self._make_dataset(prefix + name, msgs)

    # ------------------------------------------------------------------ object headers (version 1)
    def _object_header(self, addr: int) -> List[Tuple[int, bytes]]:
        a = self.base + addr
        if self._u(a, 1) != 1:
            raise Hdf5Error("only version-1 object headers are supported")
        nmsg, size = self._u(a + 2, 2), self._u(a + 8, 4)
        blocks = [(a + 16, size)]
        msgs: List[Tuple[int, bytes]] = []
        while blocks and len(msgs) < nmsg:
            p, left = blocks.pop(0)
            while left >= 8 and len(msgs) < nmsg:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                data = self._bytes(p + 8, msize)
                if mtype == 0x0010:                       # continuation
                    off, ln = struct.unpack_from("<QQ", data, 0)
                    blocks.append((self.base + off, ln))
                msgs.append((mtype, data))
                p += 8 + msize
                left -= 8 + msize
        return msgs

    def _make_dataset(self, name: str, msgs) -> Dataset:
        shape = dtype = layout = None
        filters: List[Tuple[int, tuple]] = []
        for mtype, d in msgs:
            if mtype == 0x0001:                           # dataspace
                ver, rank, flags = d[0], d[1], d[2]
                p = 8 if ver == 1 else 4
                if ver not in (1, 2):
                    raise Hdf5Error(f"dataspace message version {ver}")
                shape = struct.unpack_from("<" + "Q" * rank, d, p)
            elif mtype == 0x0003:                         # datatype
                cls, bits0, size = d[0] & 0x0F, d[1], struct.unpack_from("<I", d, 4)[0]
                if bits0 & 1:
                    raise Hdf5Error("big-endian data is not supported")
                if cls == 1 and size in (2, 4, 8):
                    dtype = np.dtype("<f%d" % size)
                elif cls == 0 and size in (1, 2, 4, 8):
                    dtype = np.dtype("<%s%d" % ("i" if (bits0 >> 3) & 1 else "u", size))
                else:
                    raise Hdf5Error(f"datatype class {cls} size {size} is not supported")
            elif mtype == 0x0008:                         # data layout
                ver = d[0]
                if ver == 3:
                    cls = d[1]
                    if cls == 1:
                        layout = {"class": "contiguous", "addr": struct.unpack_from("<Q", d, 2)[0], "size": struct.unpack_from("<Q", d, 10)[0]}
                    elif cls == 2:
                        ndim1 = d[2]                          # rank + 1: the last "dimension" is the element size
                        btree = struct.unpack_from("<Q", d, 3)[0]
                        cd = struct.unpack_from("<" + "I" * ndim1, d, 11)
                        layout = {"class": "chunked", "btree": btree, "chunk": tuple(cd[:-1]), "esize": cd[-1]}
                    elif cls == 0:
                        n = struct.unpack_from("<H", d, 2)[0]
                        layout = {"class": "compact", "data": bytes(d[4:4 + n])}
                    else:
                        raise Hdf5Error(f"data layout class {cls}")
                elif ver in (1, 2):
                    nd, cls = d[1], d[2]
                    p = 8
                    addr = UNDEF
                    if cls != 0:
                        addr = struct.unpack_from("<Q", d, p)[0]
                        p += 8
                    if cls == 1:
                        layout = {"class": "contiguous", "addr": addr, "size": 0}
                    elif cls == 2:
                        cd = struct.unpack_from("<" + "I" * nd, d, p)
                        layout = {"class": "chunked", "btree": addr, "chunk": tuple(cd[:-1]), "esize": cd[-1]}
                    else:
                        raise Hdf5Error("compact layout (message version 1/2) is not supported")
                else:
                    raise Hdf5Error(f"data layout message version {ver}")
            elif mtype == 0x000B:                         # filter pipeline
                filters = self._parse_filters(d)
        if shape is None or dtype is None or layout is None:
            raise Hdf5Error(f"dataset {name}: missing dataspace / datatype / layout message")
        if filters and layout["class"] != "chunked":
            raise Hdf5Error(f"dataset {name}: a filter pipeline on {layout['class']} data")
        for fid, _ in filters:
            if fid not in (1, 2, 3):
                raise Hdf5Error(f"dataset {name}: filter id {fid} is not supported (deflate, shuffle, fletcher32 only)")
        return Dataset(self, name, shape, dtype, layout, filters)

    @staticmethod
    def _parse_filters(d: bytes) -> List[Tuple[int, tuple]]:
        ver, n = d[0], d[1]
        out: List[Tuple[int, tuple]] = []
        if ver == 1:
            p = 8
            for _ in range(n):
                fid, nlen, _flags, ncd = struct.unpack_from("<HHHH", d, p)
                p += 8 + ((nlen + 7) // 8) * 8
                cd = struct.unpack_from("<" + "I" * ncd, d, p)
                p += 4 * ncd + (4 if ncd % 2 else 0)
                out.append((fid, tuple(cd)))
        elif ver == 2:
            p = 2
            for _ in range(n):
                fid = struct.unpack_from("<H", d, p)[0]
                p += 2
                nlen = 0
                if fid >= 256:
                    nlen = struct.unpack_from("<H", d, p)[0]
                    p += 2
                _flags, ncd = struct.unpack_from("<HH", d, p)
                p += 4 + nlen
                cd = struct.unpack_from("<" + "I" * ncd, d, p)
                p += 4 * ncd
                out.append((fid, tuple(cd)))
        else:
            raise Hdf5Error(f"filter pipeline message version {ver}")
        return out

    # ------------------------------------------------------------------ mapping interface
    def keys(self):
        return self._datasets.keys()

    def __contains__(self, name):
        return name in self._datasets

    def __getitem__(self, name: str) -> Dataset:
        try:
            return self._datasets[name]
        except KeyError:
            raise KeyError(f"{name!r} not in {self.path} (datasets: {sorted(self._datasets)})") from None

    def close(self):
        self._fh.close()
