#!/opt/conda/bin/python3.9
"""Blosc fixtures made with the REAL libblosc (c-blosc 1.21.0, /opt/conda/lib/libblosc.so.1, driven through ctypes) and the
REAL libhdf5 (h5py 3.3.0 / HDF5 1.10.6 of /opt/conda/bin/python3.9).  Build container only:
``/opt/conda/bin/python3.9 tests/golden/h5/make_blosc_fixtures.py``.

* blosc_frames.npz -- blosc_compress() outputs ("frames") for the element types, shuffle modes and codecs that matter, with
  the arrays that went in: what event_representation_study_amd/blosc_lite.py (a decoder written from the published c-blosc 1
  frame format) is pinned against.  ev-licious writes zstd, level 1, BIT shuffle (h5_writer.py:8-20).
* events_evlicious_blosc.h5 -- the ev-licious container with its own filter: events/{x:u2, y:u2, p:i1, t:i8}, chunked and
  resizable, HDF5 filter 32001 with ev-licious' compression_opts, every chunk a libblosc frame.  The image has no HDF5
  Blosc plugin (hdf5plugin / libH5Zblosc), so the filter is declared on the dataset-creation property list (flagged
  optional, or libhdf5 refuses to create the dataset without the plugin) and the chunks are compressed here with libblosc
  exactly as the plugin does (blosc_compress(clevel, shuffle, typesize, ...) over the whole chunk, cd_values[0:4] = filter
  revision 2, Blosc format 2, typesize, chunk bytes) and stored with H5Dwrite_chunk -- the bytes in the file are what the
  plugin writes.
* blosc_expected.npz -- the arrays that went in.
"""
import ctypes
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
B = ctypes.CDLL("/opt/conda/lib/libblosc.so.1")
B.blosc_init()
B.blosc_compress.restype = ctypes.c_int
B.blosc_compress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_size_t]
B.blosc_set_compressor.argtypes = [ctypes.c_char_p]
B.blosc_decompress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]


def compress(arr, cname, clevel, shuffle):
    arr = np.ascontiguousarray(arr)
    assert B.blosc_set_compressor(cname.encode()) >= 0
    dst = np.empty(arr.nbytes + 16 + 4096, dtype=np.uint8)
    n = B.blosc_compress(clevel, shuffle, arr.dtype.itemsize, arr.nbytes, arr.ctypes.data, dst.ctypes.data, dst.nbytes)
    assert n > 0, n
    back = np.empty_like(arr)
    assert B.blosc_decompress(dst.ctypes.data, back.ctypes.data, back.nbytes) == arr.nbytes and np.array_equal(back, arr)
    return dst[:n].copy()


rng = np.random.default_rng(2026)
frames, exp = {}, {}


def sample(dt, n):
    if dt == "i8":      # ascending microsecond timestamps
        return (np.sort(rng.integers(0, 10 ** 9, n)) + 1_700_000_000_000_000).astype("i8")
    if dt == "u2":
        return rng.integers(0, 1280, n).astype("u2")
    if dt == "i1":
        return (rng.integers(0, 2, n) * 2 - 1).astype("i1")
    if dt == "f4":
        return rng.random(n).astype("f4")
    return rng.integers(-1000, 1000, n).astype(dt)


k = 0
for dt in ("u2", "i1", "i8", "f4", "i4"):
    for n in (1, 7, 100, 4099, 20000, 140001):
        for cname, clevel, shuffle in (("zstd", 1, 2), ("zstd", 1, 1), ("zstd", 5, 0), ("lz4", 5, 1), ("zlib", 3, 2), ("lz4hc", 4, 2)):
            if n > 4099 and not (cname == "zstd" and shuffle == 2 and dt in ("i8", "u2")):   # the fixtures stay small:
                continue                                                                    # several blocks only for ev-licious' own settings
            a = sample(dt, n)
            frames["f%03d" % k] = compress(a, cname, clevel, shuffle)
            exp["f%03d" % k] = a
            k += 1
# incompressible input: libblosc stores it (the "memcpyed" flag) or stores single streams raw
a = rng.integers(0, 256, 30000).astype("u1")
frames["f%03d" % k] = compress(a, "zstd", 1, 0); exp["f%03d" % k] = a; k += 1
a = rng.integers(0, 2 ** 63, 5000).astype("i8")
frames["f%03d" % k] = compress(a, "zstd", 1, 2); exp["f%03d" % k] = a; k += 1
np.savez_compressed(os.path.join(HERE, "blosc_frames.npz"), **frames)
np.savez_compressed(os.path.join(HERE, "blosc_frames_expected.npz"), **exp)
print("frames:", k, "bytes:", sum(v.nbytes for v in frames.values()))

# ---- the ev-licious container with its Blosc filter
n = 40000
cols = {"x": sample("u2", n), "y": rng.integers(0, 720, n).astype("u2"), "p": sample("i1", n), "t": sample("i8", n)}
path = os.path.join(HERE, "events_evlicious_blosc.h5")
if os.path.exists(path):
    os.remove(path)
with h5py.File(path, "w") as f:
    g = f.create_group("events")
    for name, arr in cols.items():
        # h5py's chunks=True guess for shape (65536,) of this dtype, as H5Writer.__init__ gets it
        guess = h5py.File(os.path.join(HERE, "_tmp.h5"), "w")
        chunk = guess.create_dataset("d", shape=(2 ** 16,), dtype=arr.dtype, maxshape=(None,), chunks=True).chunks[0]
        guess.close()
        os.remove(os.path.join(HERE, "_tmp.h5"))
        space = h5py.h5s.create_simple((n,), (h5py.h5s.UNLIMITED,))
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_chunk((chunk,))
        cd = (2, 2, arr.dtype.itemsize, chunk * arr.dtype.itemsize, 1, 2, 5)     # ev-licious: level 1, bit shuffle, zstd
        dcpl.set_filter(32001, h5py.h5z.FLAG_OPTIONAL, cd)
        did = h5py.h5d.create(g.id, name.encode(), h5py.h5t.py_create(arr.dtype), space, dcpl)
        d = h5py.Dataset(did)
        for c0 in range(0, n, chunk):
            piece = np.zeros(chunk, dtype=arr.dtype)                             # a partial last chunk is padded to full size
            part = arr[c0:c0 + chunk]
            piece[:len(part)] = part
            d.id.write_direct_chunk((c0,), compress(piece, "zstd", 1, 2).tobytes(), filter_mask=0)
        print(name, "chunk", chunk, "chunks", -(-n // chunk))
    g.create_dataset("width", data=1280, dtype="i4")
    g.create_dataset("height", data=720, dtype="i4")
    g.create_dataset("divider", data=1, dtype="i4")
np.savez_compressed(os.path.join(HERE, "blosc_expected.npz"), **{"evl_" + k_: v for k_, v in cols.items()})
print("wrote", path, os.path.getsize(path))
