#!/opt/conda/bin/python3.9
"""A container in the reference's Gen1 layout (gen1_2yolo.py:72-82,168-198), written by REAL h5py / libhdf5:
two recordings, each ``<name>/events/{x,y,t,p,height,width}`` (chunked, gzip + shuffle) and ``<name>/bbox/{t_unique, event_idx,
offsets, class_id, x, y, w, h}``.  ``/opt/conda/bin/python3.9 tests/golden/h5/make_gen1_fixture.py``; expected windows are
cut here with h5py itself (what Gen1H5._load_events reads)."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(404)
path = os.path.join(HERE, "gen1_layout.h5")
if os.path.exists(path):
    os.remove(path)
N_EV = 3000       # window size of the test
exp = {}
with h5py.File(path, "w") as f:
    for name, n in (("17-03-30_12-53-58_1037500000_1097500000", 20000), ("17-04-04_11-00-13_cut_15_500000_60500000", 9000)):
        g = f.create_group(name)
        e = g.create_group("events")
        cols = {"x": rng.integers(0, 304, n).astype("u2"), "y": rng.integers(0, 240, n).astype("u2"),
                "t": (np.sort(rng.integers(0, 6 * 10 ** 7, n)) + 1_000_000).astype("i8"), "p": rng.integers(0, 2, n).astype("i1")}
        for k, a in cols.items():
            e.create_dataset(k, data=a, chunks=(2048,), compression="gzip", compression_opts=4, shuffle=True)
        e.create_dataset("height", data=240, dtype="i4")
        e.create_dataset("width", data=304, dtype="i4")
        b = g.create_group("bbox")
        nb = 7
        ev_idx = np.sort(rng.integers(100, n, nb)).astype("i8")
        ev_idx[0] = 1500                                   # fewer than N_EV events in front of it: idx0 clamps to 0
        ev_idx.sort()
        b.create_dataset("event_idx", data=ev_idx)
        b.create_dataset("t_unique", data=cols["t"][ev_idx - 1])
        b.create_dataset("offsets", data=np.arange(1, nb + 1, dtype="i8"))
        for k, dt in (("class_id", "i4"), ("x", "f4"), ("y", "f4"), ("w", "f4"), ("h", "f4")):
            b.create_dataset(k, data=rng.integers(0, 2, nb).astype(dt))
with h5py.File(path, "r") as f:
    k = 0
    for name in sorted(f.keys()):
        for i in range(len(f[name + "/bbox/t_unique"])):
            idx1 = int(f[name + "/bbox/event_idx"][i])
            idx0 = max(0, idx1 - N_EV)
            h = f[name + "/events"]
            xyt = np.stack([h["x"][idx0:idx1].astype("i8"), h["y"][idx0:idx1].astype("i8"), h["t"][idx0:idx1]], axis=-1)
            xyt[:, -1] -= xyt[0, -1]
            exp["w%02d" % k] = np.concatenate([xyt, h["p"][idx0:idx1].astype("i8")[:, None]], axis=1).astype("i4")
            k += 1
np.savez_compressed(os.path.join(HERE, "gen1_layout_expected.npz"), **exp)
print("wrote", path, os.path.getsize(path), "windows", k)
