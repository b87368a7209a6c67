"""-m gpu: the STREAM builders of round 6 (k_voxel_stream, k_tore_stream, k_event_stack_stream, k_polstats_stream,
k_mdes_stream) against the oracle AND against the ordered kernels they replace after the key-sorted pass (the
EVREP_X_*_ORDERED plan flags switch those back on): same tensors bit for bit, on uniform, clustered and degenerate
windows, both polarity encodings, escaped polarity values, unsorted timestamps, empty and one-event windows."""
import numpy as np
import pytest
import torch

from conftest import assert_bit_equal

from event_representation_study_amd.synthetic import GENERATORS, make_events

pytestmark = pytest.mark.gpu

ORDERED = ("EVREP_X_VOXEL_ORDERED", "EVREP_X_TORE_ORDERED", "EVREP_X_POLSTATS_ORDERED", "EVREP_X_ESTACK_ORDERED", "EVREP_X_MDES_ORDERED",
           "EVREP_X_TS_ORDERED")
CLOSE = ("ts64", "ts32", "ts_scaled")     # float exponentials: 1e-12 relative between the two forms (budget 1e-5)


def _windows(kind, W, H):
    rng = np.random.default_rng(17)
    if kind == "uniform":
        return [make_events(n, W, H, seed=50 + i, polarity=("pm1" if i % 2 else "01")) for i, n in enumerate((3001, 2, 9000))]
    if kind == "clustered":
        wins = [GENERATORS["circle"](12000, W, H, seed=3), GENERATORS["edges"](12000, W, H, seed=4)]
        hot = make_events(6000, W, H, seed=5)                   # one pixel holds 40 % of the window, in bursts
        idx = np.sort(rng.choice(6000, 2400, replace=False))
        hot[idx, 0], hot[idx, 1] = W // 3, H // 2
        return wins + [hot]
    if kind == "escaped":                                       # arbitrary integer polarity values (operations.py takes them as they come)
        ev = make_events(5000, W, H, seed=8)
        ev[:, 3] = rng.integers(-3, 6, size=5000)
        return [ev, make_events(4000, W, H, seed=9)]
    if kind == "dense":
        return [make_events(40000, W, H, seed=11), make_events(40001, W, H, seed=12, polarity="01")]
    if kind == "sweeps":
        # units of more than 256 records, both ways a stream sweeps them.  15 block runs of 4 096 events with ~40 records of a unit
        # each: 64 consecutive records of the UNIT per batch for the voxel grid / ERGO-12 (every lane finds its record's run by the
        # readlane chain), run by run for the order-free streams
        return [make_events(60000, W, H, seed=21), make_events(0, W, H, seed=1)]
    if kind == "sweeps19":
        # 19 runs of 8 192 events (the run of a record by an LDS search) with ~41 records of a unit each, and a unit that holds
        # 3 000 more in bursts (run by run for everyone)
        b = make_events(150000, W, H, seed=22, polarity="01")
        idx = np.sort(rng.choice(150000, 3000, replace=False))
        b[idx, 0], b[idx, 1] = 130 + (idx % 7), H // 2
        return [b]
    raise ValueError(kind)


def _build_all(eng, wins, H, W, monkeypatch, ordered):
    for name in ORDERED:
        if ordered:
            monkeypatch.setenv(name, "1")
        else:
            monkeypatch.delenv(name, raising=False)
    for name in ("EVREP_X_MDES_STREAM", "EVREP_X_TS_STREAM"):    # the streams at every density (their default gates are by density)
        if ordered:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, "1")
    eb = eng.EventBatch.from_numpy(wins, H, W)
    tn = torch.rand(eb.total, dtype=torch.float64, device=eb.device, generator=torch.Generator(device=eb.device).manual_seed(3))
    out = {"ergo64": eb.optimized(), "ergo32": eb.optimized(dtype=torch.float32), "ergo_x255": eb.optimized(scale=255.0),
           "es": eb.event_stack(), "tore": eb.tore(6, frame_mode=2), "tore_scaled": eb.tore(6, frame_mode=2, scale=255.0),
           "ts64": eb.time_surface(), "ts32": eb.time_surface(dtype=torch.float32), "ts_scaled": eb.time_surface(premap=1, scale=255.0),
           "voxel": eb.voxel(5), "voxel12": eb.voxel(12, mode=1, scale=255.0), "evl": eb.voxel(9, mode=2),
           "acc_all": eb.polstats(tn, [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2]),
           "acc_exp": eb.polstats(tn, [0, 1, 2, 0], [4, 4, 0, 5], tau=0.3)}
    bbox = eb.tore(6, frame_mode=0)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res["tore_bbox"] = [t.cpu().numpy() for t in bbox]
    eb.check_built("stream builders")
    return res


@pytest.mark.parametrize("kind", ["uniform", "clustered", "escaped", "dense", "sweeps", "sweeps19"])
def test_stream_builders_equal_ordered_builders_and_oracle(kind, monkeypatch, oracle):
    from event_representation_study_amd import engine as eng
    H, W = {"dense": (48, 160), "sweeps": (48, 160), "sweeps19": (48, 480)}.get(kind, (60, 200))
    wins = _windows(kind, W, H)
    got = _build_all(eng, wins, H, W, monkeypatch, ordered=False)
    ref = _build_all(eng, wins, H, W, monkeypatch, ordered=True)
    for k in got:
        if k == "tore_bbox":
            for a, b in zip(got[k], ref[k]):
                assert_bit_equal(a, b, "tore bbox stream vs ordered (%s)" % kind)
        elif k in CLOSE:
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-6 if k == "ts32" else 1e-12, atol=0, err_msg="%s stream vs ordered (%s)" % (k, kind))
            assert np.array_equal(got[k] == 0, ref[k] == 0)          # dead slices stay exactly 0
        else:
            assert_bit_equal(got[k], ref[k], "%s stream vs ordered (%s)" % (k, kind))
    for b, ev in enumerate(wins):
        if ev.shape[0] == 0:
            assert not got["ergo64"][b].any() and not got["es"][b].any()
            continue
        assert_bit_equal(got["ergo64"][b], oracle.ergo12(ev, H, W), "ergo12 stream vs oracle (%s, window %d)" % (kind, b))
        assert_bit_equal(got["es"][b], oracle.event_stack(ev, H, W), "event_stack stream vs oracle (%s, window %d)" % (kind, b))
        want = oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W))
        np.testing.assert_allclose(got["tore"][b], want, rtol=1e-6, atol=1e-6)
        if kind != "escaped":      # (the time surface reads p & 1 of whatever the dispatcher mapped: the oracle takes {0, 1} / {-1, +1} streams)
            np.testing.assert_allclose(got["ts64"][b], oracle.time_surface(ev, H, W), rtol=1e-12)


def test_stream_builders_on_unsorted_windows(monkeypatch, oracle):
    """Array order whatever the timestamps are: ERGO-12 (MDES works by index), EventStack (put is last-write-wins), TORE (the
    reference's np.partition on its k-vector): the streams against the ordered kernels, bit for bit."""
    from event_representation_study_amd import engine as eng
    H, W = 60, 200
    rng = np.random.default_rng(23)
    wins = []
    for s in (1, 2):
        ev = make_events(7000, W, H, seed=70 + s)
        ev = ev[rng.permutation(ev.shape[0])]
        wins.append(np.ascontiguousarray(ev))
    got = _build_all(eng, wins, H, W, monkeypatch, ordered=False)
    ref = _build_all(eng, wins, H, W, monkeypatch, ordered=True)
    for k in ("ergo64", "ergo32", "es", "tore", "tore_scaled", "acc_all", "acc_exp"):
        assert_bit_equal(got[k], ref[k], "%s stream vs ordered on unsorted windows" % k)
    for a, b in zip(got["tore_bbox"], ref["tore_bbox"]):
        assert_bit_equal(a, b, "tore bbox stream vs ordered on unsorted windows")
    for b, ev in enumerate(wins):
        assert_bit_equal(got["ergo64"][b], oracle.ergo12(ev, H, W), "ergo12 stream vs oracle, unsorted window %d" % b)


def test_ordered_float32_ergo12_monsters_by_sixteen_waves(monkeypatch, oracle):
    """k_mdes_coop (r06): on a SPARSE window (the ordered float32 builder) a unit of >= 4096 records is one hot item for a workgroup of
    sixteen waves; EVREP_X_MDES_NO_COOP brings r05's time slices back.  Both against the oracle, bit for bit; the voxel grid's burst
    units (k_voxel_hot) ride along."""
    from event_representation_study_amd import engine as eng
    H, W = 480, 640
    rng = np.random.default_rng(31)
    wins = []
    for s, (n_hot, span) in enumerate(((9000, 200), (5000, 40))):
        ev = make_events(60000, W, H, seed=90 + s)
        idx = np.sort(rng.choice(60000, n_hot, replace=False))
        ev[idx, 1] = 100 + s
        ev[idx, 0] = rng.integers(130, 130 + span, size=n_hot)
        wins.append(ev)
    res = {}
    for flag in (False, True):
        if flag:
            monkeypatch.setenv("EVREP_X_MDES_NO_COOP", "1")
        else:
            monkeypatch.delenv("EVREP_X_MDES_NO_COOP", raising=False)
        eb = eng.EventBatch.from_numpy(wins, H, W)
        assert eb.plan.reserved == 2
        res[flag] = (eb.optimized(dtype=torch.float32).cpu().numpy(), eb.voxel(5).cpu().numpy())
        eb.check_built("ergo12 float32")
    assert_bit_equal(res[False][0], res[True][0], "ergo12 float32: sixteen waves vs time slices")
    for b, ev in enumerate(wins):
        assert_bit_equal(res[False][0][b], oracle.ergo12(ev, H, W).astype(np.float32), "ergo12 float32 monster window %d" % b)
        assert_bit_equal(res[False][1][b], oracle.voxel(ev, H, W, 5), "voxel burst window %d" % b)


@pytest.mark.parametrize("k", [1, 3, 6])
def test_tore_big_units_queue_against_oracle(k, oracle):
    """k_tore_stream, units of more than 256 records on ascending timestamps (r06b): swept last records first, a per-FIFO counter
    admits what still has a FIFO to enter into the LDS queue.  Windows that stress it: a pixel with hundreds of records in bursts
    (far more than K per batch), many duplicate timestamps (ties at the FIFO's edge), arbitrary integer polarity values (their sign
    picks the FIFO), a sample time in the MIDDLE of the window (later records are dropped before they are counted), K = 1."""
    from event_representation_study_amd import engine as eng
    H, W = 40, 200
    rng = np.random.default_rng(77)
    wins = []
    for s in range(3):
        n = 30000
        ev = make_events(n, W, H, seed=900 + s, polarity="pm1")
        ev[:, 2] = np.sort(rng.integers(0, 4000, size=n)).astype(np.int32)          # ~7 events per microsecond: ties everywhere
        idx = np.sort(rng.choice(n, 9000, replace=False))
        ev[idx, 0] = 70 + (idx % 5)                                               # 9 000 records on five pixels of one unit, in bursts
        ev[idx, 1] = H // 2
        if s == 1:
            ev[:, 3] = rng.integers(-4, 5, size=n)                                # escaped polarity values: pol > 0 / pol <= 0
        wins.append(ev)
    eb = eng.EventBatch.from_numpy(wins, H, W)
    t_mid = np.array([int(w[len(w) // 2, 2]) for w in wins], dtype=np.int32)
    for times in (None, t_mid):
        got = eb.tore(k, frame_mode=2, sample_times=times).cpu().numpy()
        for b, ev in enumerate(wins):
            T = ev[-1, 2] if times is None else times[b]
            want = oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], T, k, (H, W))
            np.testing.assert_allclose(got[b], want, rtol=1e-6, atol=1e-6, err_msg="tore k=%d window %d sample time %s" % (k, b, "end" if times is None else "mid"))
    eb.check_built("tore")
