"""-m gpu: the HIP path (through the C ABI) against the golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import STANDARD_CASES, assert_bit_equal, load_golden

from event_representation_study_amd.synthetic import to_structured  # noqa: E402

from event_representation_study_amd.synthetic import make_events

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from event_representation_study_amd import engine
    return engine


def _batch(eng, ev, H, W):
    return eng.EventBatch.from_numpy(ev, H, W)


@pytest.mark.parametrize("name", STANDARD_CASES)
def test_golden_all_builders(eng, name):
    g = load_golden(name)
    H, W = int(g["H"]), int(g["W"])
    eb = _batch(eng, g["events"], H, W)
    assert_bit_equal(eb.optimized()[0].cpu().numpy(), g["ergo12"], name + " ergo12")
    assert_bit_equal(eb.event_stack(12, premap=True)[0].cpu().numpy(), g["event_stack"], name + " event_stack")
    if "voxel5" in g:
        assert_bit_equal(eb.voxel(5)[0].cpu().numpy(), g["voxel5"], name + " voxel5")
    if "time_surface" in g:
        ts = eb.time_surface(6, 50000.0, premap=True)[0].cpu().numpy()
        np.testing.assert_allclose(ts, g["time_surface"], rtol=1e-12, atol=0)   # budget: 1e-5 rel
        assert np.array_equal(ts == 0, g["time_surface"] == 0)
    tore = eb.tore(6, frame_mode=0)[0].cpu().numpy()
    assert tore.shape == g["tore"].shape
    np.testing.assert_allclose(tore, g["tore"], rtol=1e-6, atol=1e-6)             # budget: 1e-5 rel
    assert np.array_equal(tore == 0, g["tore"] == 0)


@pytest.mark.parametrize("enc", ["pm1", "01"])
def test_golden_mdes_all_triples(eng, enc):
    g = load_golden("mdes_all_triples_40x30_n3001_" + enc)
    eb = _batch(eng, g["events"], int(g["H"]), int(g["W"]))
    got = eb.mdes(list(g["windows"]), [str(s) for s in g["funcs"]], [str(s) for s in g["aggs"]])[0].cpu().numpy()
    assert_bit_equal(got, g["rep"], enc)


@pytest.mark.parametrize("name", ["sbt_all_triples_40x30_n3001_pm1", "sbt_all_triples_40x30_n3001_01",
                                  "sbt_shortspan_40x30_n2000_pm1", "sbt_latestart_40x30_n1500_pm1"])
def test_golden_mdes_sbt(eng, name):
    """stacking_type="SBT": evrep_mdes_sbt_windows + evrep_mdes_ex against the reference's own output, bit for bit."""
    g = load_golden(name)
    eb = _batch(eng, g["events"], int(g["H"]), int(g["W"]))
    got = eb.mdes(list(g["windows"]), [str(s) for s in g["funcs"]], [str(s) for s in g["aggs"]], stacking="SBT")[0].cpu().numpy()
    assert_bit_equal(got, g["rep"], name)


def test_mdes_sbt_batch_against_oracle(eng, oracle):
    """Several windows per launch, float32 output, an out-of-frame event inside some windows only, a window of one timestamp
    (t_s = NaN: every window but the first is empty), both stackings from one binning pass."""
    H, W = 37, 150
    rng = np.random.default_rng(8)
    wins = [make_events(n, W, H, seed=90 + i, polarity="pm1" if i % 2 else "01") for i, n in enumerate((5000, 333, 2, 1200))]
    wins[0][4000, 0] = W + 7                  # out of frame, polarity as drawn, late in the window
    wins[0][4000, 1] = H + 3
    wins[3][:, 2] = 77                        # one timestamp
    triples = [(w, f, a) for w in range(8) for f in ("timestamp", "polarity", "count_neg", "timestamp_pos")
               for a in ("sum", "mean", "max", "variance")][:48]
    wi, fu, ag = [t[0] for t in triples], [t[1] for t in triples], [t[2] for t in triples]
    eb = eng.EventBatch.from_numpy(wins, H, W)
    got = eb.mdes(wi, fu, ag, stacking="SBT").cpu().numpy()
    got32 = eb.mdes(wi[:16], fu[:16], ag[:16], stacking="SBT", dtype=torch.float32).cpu().numpy()
    sbn = eb.mdes([w % 7 for w in wi[:16]], fu[:16], ag[:16]).cpu().numpy()
    for b, ev in enumerate(wins):
        ref = oracle.mdes_sbt(ev, H, W, wi, fu, ag)
        assert_bit_equal(got[b], ref, "window %d" % b)
        assert_bit_equal(got32[b], ref[..., :16].astype(np.float32), "window %d float32" % b)
        assert_bit_equal(sbn[b], oracle.mdes(ev, H, W, [w % 7 for w in wi[:16]], fu[:16], ag[:16]), "window %d SBN" % b)


def test_golden_mdes_none_channels(eng):
    g = load_golden("mdes_none_channels_40x30_n999")
    eb = _batch(eng, g["events"], int(g["H"]), int(g["W"]))
    got = eb.mdes([0, None, 6, None], ["count", None, "polarity", None], ["sum", None, "sum", None])[0].cpu().numpy()
    assert_bit_equal(got, g["rep"])


def test_batch_of_ragged_windows_vs_oracle(eng, oracle):
    H, W = 60, 80
    sizes = [5000, 1, 0, 777, 4097, 64, 65, 3]
    wins = [make_events(n, W, H, seed=900 + i, polarity="pm1" if i % 2 else "01") for i, n in enumerate(sizes)]
    eb = eng.EventBatch.from_numpy(wins, H, W)
    st = eb.status()
    opt = eb.optimized().cpu().numpy()
    es = eb.event_stack().cpu().numpy()
    for b, (n, ev) in enumerate(zip(sizes, wins)):
        if n == 0:
            assert st[b] & 1 and not opt[b].any() and not es[b].any()
            continue
        assert not (st[b] & 1)
        assert_bit_equal(opt[b], oracle.ergo12(ev, H, W), "ergo12 window %d" % b)
        assert_bit_equal(es[b], oracle.event_stack(ev, H, W), "event_stack window %d" % b)


def test_scale_and_f32_output(eng, oracle):
    H, W = 48, 64
    ev = make_events(9000, W, H, seed=31)
    eb = _batch(eng, ev, H, W)
    ref = oracle.ergo12(ev, H, W)
    assert_bit_equal(eb.optimized(scale=255.0)[0].cpu().numpy(), ref * 255, "x255")
    assert_bit_equal(eb.optimized(dtype=torch.float32)[0].cpu().numpy(), ref.astype(np.float32), "f32")


def test_big_640x480_digest(eng):
    import hashlib
    g = load_golden("big_640x480_n50000_pm1")
    W, H, N, seed = int(g["W"]), int(g["H"]), int(g["N"]), int(g["seed"])
    ev = make_events(N, W, H, seed=seed, polarity="pm1")
    eb = _batch(eng, ev, H, W)
    for key, t in (("ergo12", eb.optimized()), ("event_stack", eb.event_stack()), ("voxel5", eb.voxel(5))):
        a = np.ascontiguousarray(t[0].cpu().numpy())
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(g[key + "_sha256"]), key
    ts = eb.time_surface()[0].cpu().numpy().reshape(-1)
    np.testing.assert_allclose(ts[g["time_surface_pos"]], g["time_surface_val"], rtol=1e-12)
    tore = eb.tore(6, 0)[0].cpu().numpy()
    assert tore.shape == tuple(g["tore_shape"])
    np.testing.assert_allclose(tore.reshape(-1)[g["tore_pos"]], g["tore_val"], rtol=1e-6, atol=1e-6)


def test_out_of_frame_events(eng, oracle):
    H, W = 12, 16
    ev = make_events(100, W, H, seed=5)
    ev[50, 0] = 16 * 12 * 4
    eb = _batch(eng, ev, H, W)
    assert eb.status()[0] & 2
    got = eb.mdes([0, 1, 6], ["count"] * 3, ["sum"] * 3)[0].cpu().numpy()
    assert_bit_equal(got, oracle.mdes(ev, H, W, [0, 1, 6], ["count"] * 3, ["sum"] * 3))


def test_status_bits_and_exceptions(eng):
    from event_representation_study_amd import _lib
    from event_representation_study_amd.representations.optimized_representation import get_optimized_representation
    from event_representation_study_amd.synthetic import to_structured
    H, W = 24, 32
    ok = make_events(500, W, H, seed=1)
    unsorted = ok.copy()
    unsorted[[100, 300], 2] = unsorted[[300, 100], 2]
    flat = ok.copy()
    flat[:, 2] = 7
    eb = eng.EventBatch.from_numpy([ok, unsorted, flat, np.zeros((0, 4), np.int32)], H, W)
    st = eb.status()
    assert st[0] == 0
    assert st[1] & _lib.ST_UNSORTED and not (st[1] & _lib.ST_EMPTY)
    assert st[2] & _lib.ST_FLAT_TIME
    assert st[3] & _lib.ST_EMPTY
    bb = eb.bbox()
    assert tuple(bb[0]) == (ok[:, 0].min(), ok[:, 1].min(), ok[:, 0].max(), ok[:, 1].max())
    # MixedDensityEventStack takes timestamps in any order, as the reference does (array order rules) -- the other builders do not
    from event_representation_study_amd.representations import gen1_transforms
    from conftest import assert_bit_equal as abe
    import oracle as orc
    abe(get_optimized_representation(to_structured(unsorted), 500, H, W), orc.ergo12(unsorted, H, W))
    # TORE runs in array order too (r05; goldens: test_tore_on_unsorted_timestamps): here against the oracle
    rep = gen1_transforms.get_item_transform(to_structured(unsorted), "<function events2ToreFeature at 0x0>", None, H, W, 500, 50000)
    np.testing.assert_allclose(rep, orc.tore_bbox(unsorted, 6) * 255, rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("enc", ["pm1", "01"])
def test_tore_on_unsorted_timestamps(enc):
    """events2ToreFeature on timestamps that are NOT ascending (r05): the reference keeps np.partition([t] + v[:k-1], k-1)[:k]
    per event in array order (tore.py:22-25); goldens from the reference's own function and dispatcher
    (tests/golden/make_golden_r05.py; numpy's sorting partition), host result and device result, and every binning pass."""
    import torch
    from event_representation_study_amd import _lib, engine as eng
    from event_representation_study_amd.representations import gen1_transforms
    from event_representation_study_amd.representations.tore import events2ToreFeature
    g = load_golden("tore_unsorted_40x30_n3000")
    H, W = int(g["H"]), int(g["W"])
    ev = g["events_" + enc]
    N = ev.shape[0]
    x, y, ts, pol = ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3]
    np.testing.assert_allclose(events2ToreFeature(x, y, ts, pol, ts[-1], 6, (H, W)), g["tore6_" + enc], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(events2ToreFeature(x, y, ts, pol, 30000, 3, (H, W)), g["tore3_mid_" + enc], rtol=1e-6, atol=1e-6)
    # float64 timestamps (seconds), the same order: the float path of the kernel
    np.testing.assert_allclose(events2ToreFeature(x, y, ts + 0.25, pol, float(ts[-1]) + 0.25, 6, (H, W)),
                               g["tore6_" + enc], rtol=0, atol=2e-3)     # (non-integral microseconds: the same intervals)
    rec = to_structured(ev)
    rep = gen1_transforms.get_item_transform(rec, "<function events2ToreFeature at 0x0>", None, H, W, N, 50000)
    want = g["dispatch_" + enc]
    assert rep.shape == want.shape and rep.dtype == want.dtype
    np.testing.assert_allclose(rep, want, rtol=1e-6, atol=1e-4)
    assert np.array_equal(rec["p"], g["p_after_" + enc])
    dev = gen1_transforms.get_item_transform_cuda(to_structured(ev), "<function events2ToreFeature at 0x0>", None, H, W, N, 50000)
    assert np.array_equal(dev.cpu().numpy(), rep)
    for flags in (None, _lib.PLAN_NO_KEY_PASS, _lib.PLAN_FORCE_KEY_SORTED, _lib.PLAN_THREE_KERNEL):
        offs = torch.tensor([0, N, 2 * N], dtype=torch.int64)
        eb = eng.EventBatch(torch.from_numpy(np.concatenate([ev, ev])).cuda(), offs, H, W, plan_flags=flags)
        got = eb.tore(6, frame_mode=2).cpu().numpy()
        for b in range(2):
            np.testing.assert_allclose(got[b], g["tore6_" + enc], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("enc", ["pm1", "01"])
def test_unsorted_windows_through_the_dispatcher(enc):
    """Timestamps that are NOT ascending (r04): EventStack (the past half t <= t[-1] in array order) and ToTimesurface
    (array-order scan, cuts from numpy's searchsorted on the unsorted t_norm) against goldens from the reference's own
    dispatcher (tests/golden/make_golden_r04.py), host result and device result."""
    from event_representation_study_amd.representations import gen1_transforms
    g = load_golden("unsorted_80x60_n4000_" + enc)
    H, W, N = int(g["H"]), int(g["W"]), g["events"].shape[0]
    for label, name in (("event_stack", "EventStack"), ("time_surface", "ToTimesurface")):
        rec = to_structured(g["events"])
        rep = gen1_transforms.get_item_transform(rec, name, None, H, W, N, 50000)
        want = g["rep_" + label]
        assert rep.shape == want.shape and rep.dtype == want.dtype, label
        if label == "event_stack":
            assert_bit_equal(rep, want, label)
        else:
            np.testing.assert_allclose(rep, want, rtol=1e-12, atol=0)      # budget 1e-5
        assert np.array_equal(rec["p"], g["p_after_" + label])
        dev = gen1_transforms.get_item_transform_cuda(to_structured(g["events"]), name, None, H, W, N, 50000)
        assert np.array_equal(dev.cpu().numpy(), rep), label


def test_compute_repr_with_the_callers_own_float_time():
    """compute_repr(x, y, t, p, width, height, bins) -- the reference's name and signature -- with caller-normalised float64 t
    in no particular order: bit-exact against the reference's own function (np.add.at order: lower bins, then upper bins)."""
    from event_representation_study_amd.representations.representation_search.gromov_wasserstein import compute_repr
    g = load_golden("compute_repr_float_t_64x48")
    for bins in (5, 9):
        got = compute_repr(g["x"], g["y"], g["t"], g["p"], int(g["W"]), int(g["H"]), bins=bins)
        assert_bit_equal(got, g["voxel%d" % bins], "compute_repr bins=%d" % bins)
    # and the demo's own normalisation of integer timestamps (gromov_wasserstein.py:96) equals the integer-time entry point
    c1 = load_golden("c1_304x240_n10000_01")
    ev = c1["events"]
    t = ev[:, 2].astype(np.float64)
    t = (t - t[0]) / (t[-1] - t[0])
    assert_bit_equal(compute_repr(ev[:, 0], ev[:, 1], t, ev[:, 3], int(c1["W"]), int(c1["H"])), c1["voxel5"], "compute_repr c1")


def test_mdes_unsorted_timestamps(eng, monkeypatch):
    """Timestamps in any order: golden from the reference's own MixedDensityEventStack / get_optimized_representation, under
    every binning pass (the window's time range is then a reduction over all blocks, not first / last), plus a batch in
    which only one window is unsorted, against the oracle."""
    import oracle as orc
    g = load_golden("mdes_unsorted_40x30_n2500_pm1")
    ev, H, W = g["events"], int(g["H"]), int(g["W"])
    trip = (list(g["windows"]), [str(s) for s in g["funcs"]], [str(s) for s in g["aggs"]])
    for var in (None, "EVREP_BIN_KEY_SORTED", "EVREP_BIN_CLASSIC", "EVREP_BIN_THREE_KERNEL"):
        for v in ("EVREP_BIN_KEY_SORTED", "EVREP_BIN_CLASSIC", "EVREP_BIN_THREE_KERNEL"):
            monkeypatch.delenv(v, raising=False)
        if var:
            monkeypatch.setenv(var, "1")
        eb = _batch(eng, ev, H, W)
        assert_bit_equal(eb.mdes(*trip)[0].cpu().numpy(), g["rep"], str(var))
        assert_bit_equal(eb.optimized()[0].cpu().numpy(), g["ergo12"], str(var))
    for v in ("EVREP_BIN_KEY_SORTED", "EVREP_BIN_CLASSIC", "EVREP_BIN_THREE_KERNEL"):
        monkeypatch.delenv(v, raising=False)
    # a larger frame, several blocks per window: the extremes sit in the middle blocks
    H2, W2 = 120, 200
    wins = [make_events(n, W2, H2, seed=60 + i) for i, n in enumerate((30000, 9000, 12000))]
    rng = np.random.default_rng(3)
    wins[1][:, 2] = wins[1][rng.permutation(len(wins[1])), 2]
    wins[2][5000, 2] = wins[2][-1, 2] + 1000          # one late event early in the array: t.max() is not the last timestamp
    eb = eng.EventBatch.from_numpy(wins, H2, W2)
    got = eb.optimized().cpu().numpy()
    for b, w in enumerate(wins):
        assert_bit_equal(got[b], orc.ergo12(w, H2, W2), "window %d" % b)


def test_mdes_arbitrary_polarity_values(eng, oracle):
    """`polarity` is used as a VALUE by Operations: anything outside {-1,0,1} still sums / averages."""
    H, W = 24, 32
    ev = make_events(2000, W, H, seed=4)
    ev[::7, 3] = 3
    ev[::11, 3] = -2
    trip = ([0, 5, 2, 0, 1], ["polarity", "polarity", "count_pos", "timestamp_neg", "polarity"],
            ["sum", "variance", "sum", "mean", "mean"])
    got = eng.EventBatch.from_numpy(ev, H, W).mdes(*trip)[0].cpu().numpy()
    assert_bit_equal(got, oracle.mdes(ev, H, W, *trip))


def test_to_timesurface_with_float_timestamps():
    """ToTimesurface.__call__ with NON-INTEGRAL float64 timestamps (seconds; time_surface.py:66-74 is dtype-agnostic), ascending
    and not, against the reference's own class (tests/golden/make_golden_r04.py)."""
    from event_representation_study_amd.representations.time_surface import ToTimesurface
    g = load_golden("time_surface_float_t_40x30")
    W, H = int(g["W"]), int(g["H"])
    for tag in ("asc", "unsorted"):
        rec = np.zeros(len(g["x"]), dtype=[("x", "<i8"), ("y", "<i8"), ("t", "<f8"), ("p", "<i8")])
        rec["x"], rec["y"], rec["t"], rec["p"] = g["x"], g["y"], g["t_" + tag], g["p"]
        got = ToTimesurface(sensor_size=(W, H, 2), surface_dimensions=None, tau=0.01, decay="exp")(rec, g["idx"])
        want = g["surf_" + tag]
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=0)      # budget 1e-5 (x * (1/tau) instead of x / tau: ~1e-14 here)
        assert np.array_equal(got == 0, want == 0)
