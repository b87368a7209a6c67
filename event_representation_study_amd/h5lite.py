"""h5lite -- the HDF5 subset on either side of the path, written from the published HDF5 File
Format Specification (version 1.x structures) because h5py / libhdf5 are not part of the MI355X image.

WRITE (``write_dataset_file``): one contiguous little-endian dataset per file -- the container the reference stores
every precomputed representation in: ``fh.create_dataset("repr", data=rep.astype("float32"), ...)``
(ev-YOLOv6/yolov6/data/gen4/precompute_reps.py:432-435), read back by ``gen4_2yolo.py:383-386`` with plain h5py.
Layout: superblock version 0, version-1 object headers, an old-style root group (symbol-table message -> v1 B-tree
-> symbol-table node -> local heap), a version-3 contiguous data layout.  The header is a pure function of
(name, shape, dtype), so a writer thread emits ``header + array bytes`` with two writes.

READ (``File``): what h5py's defaults (libver "earliest") produce, which is how the reference's inputs are made --
Gen1 / gen4 event datasets (precompute_reps.py:307-308,408-409) and ev-licious ``events/{x,y,p,t}``
(ev-licious/src/evlicious/io/utils/h5_writer.py:29-67): superblock 0/1, v1 object headers with continuation
blocks, symbol-table groups, fixed-point / IEEE datatypes, compact / contiguous / chunked (v1 B-tree) layouts,
deflate and shuffle filters, and Blosc (filter 32001: what ev-licious compresses with -- zstd, bit shuffle), decoded by
blosc_lite.py.

Validated in tests/test_h5lite_cpu.py against real HDF5 (libhdf5 1.10.6 / h5py 3.3 of the image's conda
environment) when that is present, and against committed files that h5py wrote (tests/golden/h5/).
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _datatype_message(dtype):
    """Datatype message (version 1) of a little-endian fixed-point or IEEE floating-point numpy dtype."""
    dt = np.dtype(dtype)
    if dt.byteorder == ">":
        raise ValueError("big-endian arrays are not written")
    size = dt.itemsize
    if dt.kind == "f":
        layout = {2: (15, 10, 5, 0, 10, 15), 4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023)}[size]
        sign, eloc, esize, mloc, msize, bias = layout
        head = struct.pack("<BBBBI", 0x11, 0x20, sign, 0, size)      # class 1 v1; LE, mantissa normalisation "implied"
        props = struct.pack("<HHBBBBI", 0, 8 * size, eloc, esize, mloc, msize, bias)
        return head + props
    if dt.kind in "iu":
        bits0 = 0x08 if dt.kind == "i" else 0x00                      # bit 3: two's complement signed
        head = struct.pack("<BBBBI", 0x10, bits0, 0, 0, size)        # class 0 v1; LE
        props = struct.pack("<HH", 0, 8 * size)
        return head + props
    raise ValueError("unsupported dtype %s" % dt)


def dataset_file_header(name, shape, dtype, data_align=4096):
    """(header bytes, data offset) of a file holding ONE contiguous dataset ``/name`` of ``shape`` / ``dtype``.
    The array's C-order little-endian bytes follow at ``data offset`` (the header is padded up to it)."""
    dt = np.dtype(dtype)
    shape = tuple(int(s) for s in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize if shape else dt.itemsize
    bname = name.encode("ascii")
    if not bname or b"/" in bname:
        raise ValueError("dataset name must be a single non-empty path component")
    # fixed addresses: superblock 0..96 | root header | B-tree | heap header | heap data | symbol node | dataset header
    off_root = 96
    root_msgs_len = 8 + 16
    off_btree = off_root + 16 + root_msgs_len
    btree_len = 24 + (2 * 16 + 1) * 8 + 2 * 16 * 8
    off_heap = off_btree + btree_len
    names = _pad8(b"\x00") + _pad8(bname + b"\x00")
    name_off = 8
    # the data segment ends in one genuine free block {next = 1 (libhdf5's end-of-list mark), size}: valid whichever
    # way a reader interprets "no free block"
    heap_data = names + struct.pack("<QQ", 1, 16)
    free_head = len(names)
    off_heap_data = off_heap + 32
    off_snod = off_heap_data + len(heap_data)
    snod_len = 8 + 2 * 4 * 40
    off_dset = off_snod + snod_len
    # dataset object header messages
    dataspace = struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", s) for s in shape)
    fill = struct.pack("<BBBB", 2, 1, 0, 0)             # v2: allocate early, write at allocation, no fill value defined
    msgs = _msg(0x0001, dataspace) + _msg(0x0003, _datatype_message(dt), flags=1) + _msg(0x0005, fill, flags=1)
    dset_len = 16 + len(msgs) + 8 + 24                  # prefix + the three messages + the layout message
    data_off = -(-(off_dset + dset_len) // data_align) * data_align
    msgs += _msg(0x0008, struct.pack("<BBQQ", 3, 1, data_off, nbytes))
    eof = data_off + nbytes
    out = bytearray()
    # superblock, version 0
    out += SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", 4, 16, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQII", 0, off_root, 1, 0) + struct.pack("<QQ", off_btree, off_heap)   # root symbol-table entry
    assert len(out) == off_root
    # root group object header (v1): one symbol-table message
    out += struct.pack("<BBHII4x", 1, 0, 1, 1, root_msgs_len) + _msg(0x0011, struct.pack("<QQ", off_btree, off_heap))
    assert len(out) == off_btree
    # v1 B-tree, group node, leaf level, one child
    out += b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, off_snod, name_off)
    out += b"\x00" * (off_heap - len(out))
    # local heap
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_head, off_heap_data) + heap_data
    assert len(out) == off_snod
    # symbol-table node with one entry
    out += b"SNOD" + struct.pack("<BBH", 1, 0, 1) + struct.pack("<QQII16x", name_off, off_dset, 0, 0)
    out += b"\x00" * (off_dset - len(out))
    # dataset object header (v1)
    out += struct.pack("<BBHII4x", 1, 0, 4, 1, len(msgs)) + msgs
    out += b"\x00" * (data_off - len(out))
    return bytes(out), data_off


def write_dataset_file(path, name, array, data_align=4096):
    """Write ``array`` as the single dataset ``/name`` of a new HDF5 file."""
    a = np.ascontiguousarray(array)
    if a.dtype.byteorder == ">":
        a = a.astype(a.dtype.newbyteorder("<"))
    head, _ = dataset_file_header(name, a.shape, a.dtype, data_align)
    with open(path, "wb") as f:
        f.write(head)
        f.write(memoryview(a).cast("B") if a.size else b"")


# ------------------------------------------------------------------------------------------------ reader
class _Reader:
    def __init__(self, buf):
        self.b = buf

    def u(self, off, size):
        return int.from_bytes(self.b[off:off + size], "little")


class Dataset:
    def __init__(self, f, name, shape, dtype, layout, filters):
        self.file, self.name, self.shape, self.dtype = f, name, shape, dtype
        self._layout, self._filters = layout, filters

    def __repr__(self):
        return "<h5lite dataset %r shape %r dtype %s>" % (self.name, self.shape, self.dtype)

    def __len__(self):
        return self.shape[0]

    def _apply_filters(self, raw, mask, nbytes):
        for k in range(len(self._filters) - 1, -1, -1):       # undo in reverse pipeline order
            fid, cd = self._filters[k]
            if mask & (1 << k):
                continue
            if fid == 1:
                # bounded (r05): a chunk never inflates past its declared size (+ a trailing fletcher32 word, + slack for
                # filters further up the pipeline whose output is a few bytes longer): more is a corrupt file, not a bomb
                d = zlib.decompressobj()
                cap = int(nbytes) + 64
                raw = d.decompress(raw, cap + 1)
                if len(raw) > cap or d.unconsumed_tail:
                    raise ValueError("corrupt HDF5 chunk: deflate stream inflates past the chunk's %d bytes" % nbytes)
            elif fid == 2:                                     # shuffle: bytes of equal significance were grouped
                es = cd[0] if cd else self.dtype.itemsize
                n = len(raw) // es
                raw = np.frombuffer(raw, np.uint8, n * es).reshape(es, n).T.tobytes() + raw[n * es:]
            elif fid == 3:                                     # fletcher32: checksum appended
                raw = raw[:-4]
            elif fid == 32001:
                # ev-licious' default (h5_writer.py:8-26).  The chunk is one Blosc 1 frame: decoded by blosc_lite (pinned against
                # frames of the real libblosc; zstd / lz4 through the system's shared libraries)
                from . import blosc_lite
                raw = blosc_lite.decompress(raw, expected_nbytes=nbytes if k == 0 else None)   # (the first filter's output is the chunk)
            else:
                raise NotImplementedError("HDF5 filter %d is not supported" % fid)
        return raw

    def read(self):
        r = self.file._r
        shape, dt = self.shape, self.dtype
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        kind = self._layout[0]
        if kind == "compact":
            return np.frombuffer(self._layout[1], dt, count).reshape(shape).copy()
        if kind == "contiguous":
            addr = self._layout[1]
            if addr == UNDEF:
                return np.zeros(shape, dt)
            return np.frombuffer(r.b, dt, count, addr).reshape(shape).copy()
        if kind != "chunked":
            raise NotImplementedError("data layout %r" % (kind,))
        _, btree, cdims = kind, self._layout[1], self._layout[2]
        out = np.zeros(shape, dt)
        if btree == UNDEF:
            return out
        rank = len(shape)
        cshape = tuple(cdims[:rank])
        cbytes = int(np.prod(cshape)) * dt.itemsize

        def walk(addr):
            if bytes(r.b[addr:addr + 4]) != b"TREE":
                raise ValueError("bad chunk B-tree node at %d" % addr)
            ntype, level, used = r.u(addr + 4, 1), r.u(addr + 5, 1), r.u(addr + 6, 2)
            assert ntype == 1
            p = addr + 24
            ksz = 8 + 8 * (rank + 1)
            for _ in range(used):
                size, mask = r.u(p, 4), r.u(p + 4, 4)
                offs = tuple(r.u(p + 8 + 8 * d, 8) for d in range(rank))
                child = r.u(p + ksz, 8)
                p += ksz + 8
                if level > 0:
                    walk(child)
                    continue
                raw = self._apply_filters(bytes(r.b[child:child + size]), mask, cbytes)
                chunk = np.frombuffer(raw, dt, int(np.prod(cshape))).reshape(cshape)
                sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, shape))
                sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
                out[sel_out] = chunk[sel_in]
        walk(btree)
        return out

    def _chunk_index(self):
        """[(offsets, address, size, filter mask)] of a chunked dataset, read once: a window of a Gen1 recording is a
        slice [event_idx - N, event_idx) of datasets of 10^7..10^8 events (gen1_2yolo.py:188-198), not the whole of them."""
        if getattr(self, "_index", None) is None:
            r, rank = self.file._r, len(self.shape)
            index = []

            def walk(addr):
                if bytes(r.b[addr:addr + 4]) != b"TREE":
                    raise ValueError("bad chunk B-tree node at %d" % addr)
                level, used = r.u(addr + 5, 1), r.u(addr + 6, 2)
                p = addr + 24
                ksz = 8 + 8 * (rank + 1)
                for _ in range(used):
                    size, mask = r.u(p, 4), r.u(p + 4, 4)
                    offs = tuple(r.u(p + 8 + 8 * d, 8) for d in range(rank))
                    child = r.u(p + ksz, 8)
                    p += ksz + 8
                    if level > 0:
                        walk(child)
                    else:
                        index.append((offs, child, size, mask))
            if self._layout[1] != UNDEF:
                walk(self._layout[1])
            index.sort()
            self._index = index
        return self._index

    def read_rows(self, start, stop):
        """rows [start, stop) along axis 0 -- only the chunks that hold them are read and decoded"""
        n0 = self.shape[0] if self.shape else 0
        start, stop = max(0, min(int(start), n0)), max(0, min(int(stop), n0))
        if stop <= start:
            return np.zeros((0,) + tuple(self.shape[1:]), self.dtype)
        if self._layout[0] != "chunked":
            return self.read()[start:stop]
        r, dt, shape = self.file._r, self.dtype, self.shape
        rank = len(shape)
        cshape = tuple(self._layout[2][:rank])
        cbytes = int(np.prod(cshape)) * dt.itemsize
        out = np.zeros((stop - start,) + tuple(shape[1:]), dt)
        for offs, addr, size, mask in self._chunk_index():
            if offs[0] >= stop or offs[0] + cshape[0] <= start:
                continue
            raw = self._apply_filters(bytes(r.b[addr:addr + size]), mask, cbytes)
            chunk = np.frombuffer(raw, dt, int(np.prod(cshape))).reshape(cshape)
            lo, hi = max(start, offs[0]), min(stop, offs[0] + cshape[0], shape[0])
            rest_out = tuple(slice(o, min(o + c, s_)) for o, c, s_ in zip(offs[1:], cshape[1:], shape[1:]))
            rest_in = tuple(slice(0, s_.stop - s_.start) for s_ in rest_out)
            out[(slice(lo - start, hi - start),) + rest_out] = chunk[(slice(lo - offs[0], hi - offs[0]),) + rest_in]
        return out

    def __getitem__(self, key):
        # a plain slice along axis 0 (what the reference's loaders take: handle["x"][idx0:idx1]) reads its chunks only
        if isinstance(key, slice) and key.step in (None, 1) and self.shape and self._layout[0] == "chunked":
            a, b, _ = key.indices(self.shape[0])
            return self.read_rows(a, b)
        return self.read()[key]

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a.astype(dtype) if dtype is not None else a


class Group:
    def __init__(self, f, name, links):
        self.file, self.name, self._links = f, name, links

    def keys(self):
        return list(self._links)

    def __contains__(self, k):
        try:
            self[k]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            node = node.file._object(node._links[part], (node.name.rstrip("/") + "/" + part))
        return node

    def get(self, path, default=None):
        try:
            return self[path]
        except KeyError:
            return default


class File(Group):
    """Read-only view of an HDF5 file written with h5py's defaults; ``f["events/x"][:]`` / ``np.array(f.get(name))``
    work as the reference uses them."""

    def __init__(self, path):
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")
        self._r = _Reader(self._mm)
        r = self._r
        if bytes(r.b[:8]) != SIGNATURE:
            raise ValueError("%s is not an HDF5 file (no signature at offset 0)" % path)
        ver = r.u(8, 1)
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d (written with libver='latest'?) is not supported" % ver)
        if r.u(13, 1) != 8 or r.u(14, 1) != 8:
            raise NotImplementedError("only 8-byte offsets / lengths are supported")
        base = 24 + (4 if ver == 1 else 0)
        if r.u(base, 8) != 0:
            raise NotImplementedError("non-zero base address")
        root_entry = base + 32
        self._cache = {}
        root = self._object(r.u(root_entry + 8, 8), "/")
        Group.__init__(self, self, "/", root._links)

    def close(self):
        self._cache = {}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- object headers ------------------------------------------------------------------------------
    def _messages(self, addr):
        r = self._r
        if r.u(addr, 1) != 1:
            raise NotImplementedError("version-2 object headers (libver='latest') are not supported")
        nmsg, size = r.u(addr + 2, 2), r.u(addr + 8, 4)
        blocks, msgs = [(addr + 16, size)], []
        while blocks and len(msgs) < nmsg:
            p, left = blocks.pop(0)
            while left >= 8 and len(msgs) < nmsg:
                mtype, msize, flags = r.u(p, 2), r.u(p + 2, 2), r.u(p + 4, 1)
                body = p + 8
                if mtype == 0x0010:                              # continuation
                    blocks.append((r.u(body, 8), r.u(body + 8, 8)))
                msgs.append((mtype, body, msize, flags))
                p += 8 + msize
                left -= 8 + msize
        return msgs

    def _object(self, addr, name):
        if addr in self._cache:
            return self._cache[addr]
        r = self._r
        msgs = self._messages(addr)
        by = {}
        for mtype, body, msize, flags in msgs:
            by.setdefault(mtype, (body, msize))
        if 0x0011 in by:                                         # group: symbol table
            body, _ = by[0x0011]
            obj = Group(self, name, self._symbols(r.u(body, 8), r.u(body + 8, 8)))
        elif 0x0008 in by:
            obj = self._dataset(by, name)
        else:
            raise NotImplementedError("object %r is neither an old-style group nor a dataset" % name)
        self._cache[addr] = obj
        return obj

    def _symbols(self, btree, heap):
        r = self._r
        if bytes(r.b[heap:heap + 4]) != b"HEAP":
            raise ValueError("bad local heap")
        hdata = r.u(heap + 24, 8)

        def cstr(off):
            a = hdata + off
            e = a
            while r.b[e] != 0:
                e += 1
            return bytes(r.b[a:e]).decode("utf-8")
        links = {}

        def walk(addr):
            sig = bytes(r.b[addr:addr + 4])
            if sig == b"TREE":
                used = r.u(addr + 6, 2)
                p = addr + 24 + 8
                for _ in range(used):
                    walk(r.u(p, 8))
                    p += 16
            elif sig == b"SNOD":
                n = r.u(addr + 6, 2)
                p = addr + 8
                for _ in range(n):
                    links[cstr(r.u(p, 8))] = r.u(p + 8, 8)
                    p += 40
            else:
                raise ValueError("bad group node at %d" % addr)
        walk(btree)
        return links

    def _dataset(self, by, name):
        r = self._r
        body, _ = by[0x0001]                                      # dataspace
        ver, rank, flags = r.u(body, 1), r.u(body + 1, 1), r.u(body + 2, 1)
        p = body + (8 if ver == 1 else 4)
        shape = tuple(r.u(p + 8 * d, 8) for d in range(rank))
        body, _ = by[0x0003]                                      # datatype
        cls, b0, size = r.u(body, 1) & 0x0F, r.u(body + 1, 1), r.u(body + 4, 4)
        order = ">" if b0 & 1 else "<"
        if cls == 0:
            dt = np.dtype("%s%s%d" % (order, "i" if b0 & 0x08 else "u", size))
        elif cls == 1:
            dt = np.dtype("%sf%d" % (order, size))
        else:
            raise NotImplementedError("datatype class %d of %r is not supported" % (cls, name))
        filters = []
        if 0x000B in by:                                          # filter pipeline
            body, _ = by[0x000B]
            fver, nf = r.u(body, 1), r.u(body + 1, 1)
            p = body + (8 if fver == 1 else 2)
            for _ in range(nf):
                fid = r.u(p, 2)
                if fver == 1 or fid >= 256:
                    nlen = r.u(p + 2, 2)
                    ncd = r.u(p + 6, 2)
                    p += 8 + (-(-nlen // 8) * 8 if fver == 1 else nlen)
                else:
                    ncd = r.u(p + 4, 2)
                    p += 6
                cd = [r.u(p + 4 * k, 4) for k in range(ncd)]
                p += 4 * ncd + (4 if fver == 1 and ncd % 2 else 0)
                filters.append((fid, cd))
        body, msize = by[0x0008]                                  # layout
        lver, lcls = r.u(body, 1), r.u(body + 1, 1)
        if lver != 3:
            raise NotImplementedError("data layout message version %d" % lver)
        if lcls == 0:
            n = r.u(body + 2, 2)
            layout = ("compact", bytes(r.b[body + 4:body + 4 + n]))
        elif lcls == 1:
            layout = ("contiguous", r.u(body + 2, 8))
        elif lcls == 2:
            nd = r.u(body + 2, 1)
            cd = [r.u(body + 11 + 4 * k, 4) for k in range(nd)]
            layout = ("chunked", r.u(body + 3, 8), cd)
        else:
            raise NotImplementedError("layout class %d" % lcls)
        return Dataset(self, name, shape, dt, layout, filters)
