#!/opt/conda/bin/python3.9
"""Write the HDF5 fixtures of tests/golden/h5/ with REAL h5py / libhdf5 (the image's conda environment:
/opt/conda/bin/python3.9, h5py 3.3.0, HDF5 1.10.6) -- what the reference's own tools produce their inputs with.
Run in the build container: ``/opt/conda/bin/python3.9 tests/golden/h5/make_h5_fixtures.py``.

* events_evlicious_layout.h5 -- the ev-licious container (ev-licious/src/evlicious/io/utils/h5_writer.py:29-67):
  events/{x:u2, y:u2, p:i1, t:i8} as resizable chunked datasets (written in two add_data-style appends) plus the
  scalar events/{width,height,divider}; gzip + shuffle stand in for the reference's Blosc filter (hdf5plugin is
  absent from every interpreter of the image).
* events_gen4_layout.h5 -- flat (N, 4) event arrays under string keys, as precompute_reps.py:408-409 reads them
  (np.array(hf.get(event_file))): one contiguous float64, one chunked int32 without filters, one inside a group.
* expected.npz -- the arrays that went in.
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(31)
exp = {}

n = 5000
x = rng.integers(0, 1280, n).astype("u2")
y = rng.integers(0, 720, n).astype("u2")
p = rng.integers(0, 2, n).astype("i1") * 2 - 1
t = np.sort(rng.integers(0, 10**9, n)).astype("i8") + 1_700_000_000_000_000
path = os.path.join(HERE, "events_evlicious_layout.h5")
if os.path.exists(path):
    os.remove(path)
with h5py.File(path, "w") as f:
    kw = dict(compression="gzip", compression_opts=4, shuffle=True, chunks=True)
    for name, dt in (("x", "u2"), ("y", "u2"), ("p", "i1"), ("t", "i8")):
        f.create_dataset("events/" + name, shape=(2 ** 16,), dtype=dt, maxshape=(None,), **kw)
    f.create_dataset("events/width", data=1280, dtype="i4")
    f.create_dataset("events/height", data=720, dtype="i4")
    f.create_dataset("events/divider", data=1, dtype="i4")
    row = 0
    for lo, hi in ((0, 3000), (3000, n)):                      # H5Writer.add_data: resize, then assign the slice
        for name, arr in (("x", x), ("y", y), ("p", p), ("t", t)):
            f["events/" + name].resize(hi, axis=0)
            f["events/" + name][lo:hi] = arr[lo:hi]
exp.update(evl_x=x, evl_y=y, evl_p=p, evl_t=t)

a = rng.random((777, 4)) * 1000
b = rng.integers(-5, 1000, size=(4097, 4)).astype("i4")
c = rng.integers(0, 100, size=(33, 4)).astype("i8")
path = os.path.join(HERE, "events_gen4_layout.h5")
if os.path.exists(path):
    os.remove(path)
with h5py.File(path, "w") as f:
    f.create_dataset("moorea_2019-02-19_004_td_2257500000_2317500000_td_000012", data=a)
    f.create_dataset("chunked_nofilter", data=b, chunks=(512, 4))
    f.create_dataset("train/seq/0001", data=c)
    f.create_dataset("tiny_compact", data=np.arange(6, dtype="f4").reshape(2, 3))
exp.update(g4_a=a, g4_b=b, g4_c=c, g4_tiny=np.arange(6, dtype="f4").reshape(2, 3))
np.savez_compressed(os.path.join(HERE, "expected.npz"), **exp)
print("wrote", sorted(os.listdir(HERE)))
