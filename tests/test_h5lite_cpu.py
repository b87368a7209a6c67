"""h5lite: the HDF5 writer for "repr" files and the reader for the reference's event containers.

* reader vs files REAL h5py wrote (tests/golden/h5/, made by make_h5_fixtures.py with the image's conda h5py);
* writer -> own reader round trips for every dtype / rank the path uses;
* writer -> REAL libhdf5 (h5py of /opt/conda/bin/python3.9) when that interpreter exists on the box, i.e. the file
  opens with the library gen4_2yolo.py:383-386 would use.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN

from event_representation_study_amd import h5lite

H5 = os.path.join(GOLDEN, "h5")
CONDA_PY = "/opt/conda/bin/python3.9"


def test_reader_on_evlicious_layout_written_by_h5py():
    e = dict(np.load(os.path.join(H5, "expected.npz")))
    with h5lite.File(os.path.join(H5, "events_evlicious_layout.h5")) as f:
        assert f.keys() == ["events"] and sorted(f["events"].keys()) == ["divider", "height", "p", "t", "width", "x", "y"]
        for k, dt in (("x", "<u2"), ("y", "<u2"), ("p", "|i1"), ("t", "<i8")):
            d = f["events/" + k]
            assert d.shape == (5000,) and d.dtype == np.dtype(dt)
            np.testing.assert_array_equal(d[:], e["evl_" + k])           # chunked, shuffle + deflate, resized twice
        assert int(np.array(f["events/width"])) == 1280 and int(f["events/height"][()]) == 720
        assert int(f["events"]["divider"][()]) == 1 and f.get("events/nope") is None and "events/x" in f


def test_blosc_frames_of_the_real_libblosc():
    """blosc_lite against 126 frames the real libblosc 1.21.0 produced (make_blosc_fixtures.py): u2 / i1 / i8 / f4 / i4 x no /
    byte / bit shuffle x zstd / lz4 / lz4hc / zlib, one element to several blocks, split and unsplit blocks, stored
    ("memcpyed") frames, bit-shuffled blocks whose element count is no multiple of eight (c-blosc copies those)."""
    from event_representation_study_amd import blosc_lite
    fr = np.load(os.path.join(H5, "blosc_frames.npz"))
    ex = np.load(os.path.join(H5, "blosc_frames_expected.npz"))
    kinds = set()
    for k in fr.files:
        frame = fr[k].tobytes()
        got = np.frombuffer(blosc_lite.decompress(frame), dtype=ex[k].dtype)
        np.testing.assert_array_equal(got, ex[k], err_msg=k)
        kinds.add((frame[2] >> 5, frame[2] & 0x7))
    assert {(4, 4), (4, 1), (4, 0), (1, 1), (3, 4)} <= kinds          # (codec format, shuffle bits) really present
    with pytest.raises(ValueError):
        blosc_lite.decompress(b"\x02\x01\x80\x08" + b"\x00" * 8)
    bad = bytearray(fr["f025"].tobytes())
    bad[2] &= 0x1f                                                    # claim blosclz: not decoded here, loudly
    with pytest.raises(NotImplementedError):
        blosc_lite.decompress(bytes(bad))


def test_reader_on_the_evlicious_container_with_its_blosc_filter():
    """ev-licious' own file: events/{x,y,p,t} chunked, resizable, HDF5 filter 32001 with its compression_opts (zstd, level 1,
    bit shuffle; h5_writer.py:8-44), written by the real libhdf5 with chunks the real libblosc compressed."""
    e = dict(np.load(os.path.join(H5, "blosc_expected.npz")))
    with h5lite.File(os.path.join(H5, "events_evlicious_blosc.h5")) as f:
        for k, dt in (("x", "<u2"), ("y", "<u2"), ("p", "|i1"), ("t", "<i8")):
            d = f["events/" + k]
            assert d.shape == (40000,) and d.dtype == np.dtype(dt)
            np.testing.assert_array_equal(d[:], e["evl_" + k])
        assert int(f["events/width"][()]) == 1280 and int(f["events/height"][()]) == 720 and int(f["events/divider"][()]) == 1


def test_gen1_container_windows_and_partial_reads():
    """The reference's Gen1 layout (gen1_2yolo.py:72-82,168-198), written by real h5py: sample idx = the num_events events in
    front of the idx-th labelled timestamp, recordings in name order, t rebased; only the chunks a slice touches are decoded."""
    from event_representation_study_amd.gen1_h5 import Gen1H5Events
    ex = np.load(os.path.join(H5, "gen1_layout_expected.npz"))
    d = Gen1H5Events(os.path.join(H5, "gen1_layout.h5"), num_events=3000)
    assert len(d) == 14 and (d.height, d.width) == (240, 304)
    for i in range(len(d)):
        w = d.window(i)
        assert w.dtype == np.int32 and w.shape[1] == 4
        np.testing.assert_array_equal(w, ex["w%02d" % i])
    assert len(d.window(0)) == 1500                                  # fewer events than the window in front of the first label
    assert d.locate(7)[0] == 0 and d.locate(6)[0] == 6 and d.locate(7)[1] != d.locate(6)[1]
    with pytest.raises(IndexError):
        d.window(14)
    # partial reads: slices along axis 0 agree with the whole dataset, whatever the chunk boundaries
    with h5lite.File(os.path.join(H5, "gen1_layout.h5")) as f:
        t = f[d._file_names[0] + "/events/t"]
        whole = t.read()
        for a, b in ((0, 1), (2047, 2049), (4096, 4096), (5000, 19999), (19990, 30000), (0, 20000)):
            np.testing.assert_array_equal(t[a:b], whole[a:b])
        np.testing.assert_array_equal(t[::3], whole[::3])           # anything else falls back to the full read
    with h5lite.File(os.path.join(H5, "events_evlicious_blosc.h5")) as f:
        np.testing.assert_array_equal(f["events/t"][3000:9001], np.load(os.path.join(H5, "blosc_expected.npz"))["evl_t"][3000:9001])


def test_reader_on_gen4_layout_written_by_h5py():
    e = dict(np.load(os.path.join(H5, "expected.npz")))
    f = h5lite.File(os.path.join(H5, "events_gen4_layout.h5"))
    key = "moorea_2019-02-19_004_td_2257500000_2317500000_td_000012"
    np.testing.assert_array_equal(np.array(f.get(key)), e["g4_a"])       # as precompute_reps.py:408-409 reads it
    np.testing.assert_array_equal(f["chunked_nofilter"][:], e["g4_b"])    # chunked, edge chunk, no filter
    np.testing.assert_array_equal(f["train/seq/0001"][:], e["g4_c"])      # nested old-style groups
    np.testing.assert_array_equal(f["tiny_compact"][:], e["g4_tiny"])
    with pytest.raises(KeyError):
        f["train/seq/0002"]


@pytest.mark.parametrize("shape,dtype", [((640, 640, 12), "f4"), ((7, 5), "f8"), ((1000, 4), "i4"), ((3,), "u2"), ((0, 4), "i4"),
                                         ((2, 3, 4, 5), "i8"), ((5,), "i1")])
def test_writer_round_trip(tmp_path, shape, dtype):
    rng = np.random.default_rng(1)
    a = (rng.random(shape) * 100).astype(dtype)
    p = str(tmp_path / "x.h5")
    h5lite.write_dataset_file(p, "repr", a)
    head, off = h5lite.dataset_file_header("repr", shape, dtype)
    assert off % 4096 == 0 and os.path.getsize(p) == off + a.nbytes and open(p, "rb").read(len(head)) == head
    d = h5lite.File(p)["repr"]
    assert d.shape == tuple(shape) and d.dtype == np.dtype(dtype)
    np.testing.assert_array_equal(d[()], a)


def test_not_hdf5_and_unsupported_are_loud(tmp_path):
    p = tmp_path / "junk.h5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(ValueError):
        h5lite.File(str(p))
    with pytest.raises(ValueError):
        h5lite.dataset_file_header("a/b", (3,), "f4")


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with real h5py on this box")
def test_written_files_open_with_real_libhdf5(tmp_path):
    probe = subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True)
    if probe.returncode != 0:
        pytest.skip("the conda interpreter has no working h5py")
    rng = np.random.default_rng(2)
    cases = {"a": rng.random((64, 64, 12)).astype("f4"), "b": rng.integers(-9, 9999, (321, 4)).astype("i4"),
             "c": rng.random((5, 7)), "d": rng.integers(0, 60000, (17,)).astype("u2")}
    for k, a in cases.items():
        h5lite.write_dataset_file(str(tmp_path / (k + ".h5")), "repr", a)
        np.save(str(tmp_path / (k + ".npy")), a)
    code = ("import h5py, numpy as np, sys\n"
            "for k in 'abcd':\n"
            "    with h5py.File(sys.argv[1] + '/' + k + '.h5', 'r') as fh:\n"
            "        rep = fh['repr'][()]\n"
            "    ref = np.load(sys.argv[1] + '/' + k + '.npy')\n"
            "    assert rep.dtype == ref.dtype and rep.shape == ref.shape and np.array_equal(rep, ref), k\n"
            "print('ok')\n")
    r = subprocess.run([CONDA_PY, "-W", "ignore", "-c", code, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]
