"""The key-sorted binning pass (plan.reserved == 2: k_block_keysort alone, the builder waves gather their unit's records
from the block runs and finish the order by pixel, evrep_bin.hip / evrep_builders.hip unit_records) against the oracle
and against the classic passes (EVREP_BIN_CLASSIC), bit for bit.

EVREP_BIN_KEY_SORTED forces the pass for windows denser than the 30-records-per-unit average it is chosen for, which
exercises the spill layout (units of more than 64 records) on every builder."""
import numpy as np
import pytest
import torch

from event_representation_study_amd.synthetic import make_events

pytestmark = pytest.mark.gpu


def assert_bit_equal(a, b, msg):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, "%s: %s %s vs %s %s" % (msg, a.shape, a.dtype, b.shape, b.dtype)
    assert np.array_equal(a, b, equal_nan=True), "%s: %d elements differ" % (msg, int((a != b).sum()))


def _batches(eng, wins, H, W, monkeypatch, force=True):
    """the same windows under the key-sorted pass and under the classic pass"""
    monkeypatch.delenv("EVREP_BIN_CLASSIC", raising=False)
    monkeypatch.delenv("EVREP_BIN_THREE_KERNEL", raising=False)   # the suite may run under a globally forced pass
    if force:
        monkeypatch.setenv("EVREP_BIN_KEY_SORTED", "1")
    ks = eng.EventBatch.from_numpy(wins, H, W)
    monkeypatch.delenv("EVREP_BIN_KEY_SORTED", raising=False)
    monkeypatch.setenv("EVREP_BIN_CLASSIC", "1")
    cl = eng.EventBatch.from_numpy(wins, H, W)
    monkeypatch.delenv("EVREP_BIN_CLASSIC", raising=False)
    assert ks.plan.reserved == 2 and cl.plan.reserved in (0, 1)
    return ks, cl


def _builders(eb, rng_seed=0):
    g = torch.Generator(device="cpu").manual_seed(rng_seed)
    tn = torch.rand(eb.total, dtype=torch.float64, generator=g).to("cuda:0")
    out = {
        "ergo12": eb.optimized(),
        "ergo12_f32": eb.optimized(dtype=torch.float32),
        "mdes": eb.mdes([0, 3, 5, 1, 6], [0, 6, 1, 3, 4], [1, 0, 3, 2, 3]),
        "event_stack": eb.event_stack(),
        "time_surface": eb.time_surface(),
        "time_surface_f32": eb.time_surface(dtype=torch.float32),
        "tore_full": eb.tore(6, frame_mode=2),
        "tore_shift": eb.tore(4, frame_mode=1),
        "voxel5": eb.voxel(5),
        "voxel_evl": eb.voxel(7, mode=2),
        "polstats": eb.polstats(tn, [1, 2, 1, 2, 0, 0], [0, 0, 1, 2, 4, 5]),
    }
    # the float-time TORE of n_imagenet, an explicit ev-licious time range, the raw-polarity EventStack, the EST layer
    tf = (torch.arange(eb.total, dtype=torch.float64) * 1e-6).to("cuda:0")  # any per-event float64 times: gathered by rank
    out["tore_ftime"] = eb.tore(5, frame_mode=2, times_f64=tf)
    # ToTimesurface with the caller's indices: integer timestamps, the caller's float64 times, the array-order scan (premap + 2)
    # and 8 slices -- units beyond the record stage are VISITED after the key-sorted pass (r04), ordered after the classic ones
    oh = np.asarray(eb.offsets_host, dtype=np.int64)
    n_of = oh[1:] - oh[:-1]
    idx = torch.tensor(np.stack([(n_of * (s + 1)) // 5 for s in range(4)], axis=1).astype(np.int32))
    out["ts_idx"] = eb.time_surface(4, tau=20000.0, indices=idx)
    out["ts_ftime"] = eb.time_surface(4, tau=0.02, indices=idx, times_f64=tf)
    out["ts_array_order"] = eb.time_surface(4, tau=20000.0, premap=3, indices=idx)
    out["ts_8"] = eb.time_surface(8, tau=30000.0)
    tr = torch.tensor([[1000, 30000]] * eb.B, dtype=torch.int64)
    out["voxel_range"] = eb.voxel(4, mode=2, t_range=tr)
    out["event_stack_raw"] = eb.event_stack(7, premap=False)
    seg = torch.tensor([[-0.25, 0.5, 0.1], [0.3, -1.5, 0.6], [1e9, 0.25, -0.2]], dtype=torch.float64, device="cuda:0")
    bucket = torch.zeros(16, dtype=torch.int32, device="cuda:0")
    out["est"] = eb.est_voxel(tn.to(torch.float32), 3, seg, bucket, -1.0, 1.0)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    out["tore_bbox"] = [t.cpu().numpy() for t in eb.tore(6, frame_mode=0)]
    return out


# r06: after the key-sorted pass dense windows take the time surface's STREAM (k_time_surface_stream), whose exponentials are
# factorised per event for units of every size; the classic passes' builder factorises fully staged units only.  The two forms
# agree to an ulp or two (budget 1e-5): these keys are compared to 1e-13 relative, with the exact zeros (dead slices) in place.
TS_KEYS = ("time_surface", "time_surface_f32", "ts_idx", "ts_8")


def _same(a, b, tag):
    for k in a:
        if isinstance(a[k], list):
            for i, (x, y) in enumerate(zip(a[k], b[k])):
                assert_bit_equal(x, y, "%s %s[%d]" % (tag, k, i))
        elif k in TS_KEYS:
            np.testing.assert_allclose(a[k], b[k], rtol=1e-6 if a[k].dtype == np.float32 else 1e-13, atol=0, err_msg="%s %s" % (tag, k))
            assert np.array_equal(a[k] == 0, b[k] == 0), "%s %s: zeros" % (tag, k)
        else:
            assert_bit_equal(a[k], b[k], "%s %s" % (tag, k))


@pytest.mark.parametrize("geom", [(48, 64, (3000, 900, 1)), (30, 200, (20000, 64, 65)), (480, 640, (50000, 49999, 12345)),
                                  (240, 304, (10000, 70000)), (5, 129, (4000, 130)),
                                  # units of 65-128 records behind a 128-record stage (fuzz seed 3239, r03): the classic
                                  # front end has to stage them whole, or TimeSurface picks its other exponential form
                                  (49, 129, (2799, 2525)), (40, 256, (9000, 7000))])
def test_every_builder_matches_the_classic_pass(geom, monkeypatch):
    from event_representation_study_amd import engine as eng
    H, W, sizes = geom
    wins = [make_events(n, W, H, seed=11 * i + n % 97, polarity="pm1" if i % 2 == 0 else "01", dup_last=(3 if n > 10 else 0))
            for i, n in enumerate(sizes)]
    ks, cl = _batches(eng, wins, H, W, monkeypatch)
    _same(_builders(ks), _builders(cl), "%dx%d" % (W, H))
    np.testing.assert_array_equal(ks.status(), cl.status())
    np.testing.assert_array_equal(ks.bbox(), cl.bbox())


def test_the_pass_is_chosen_for_the_headline_windows_and_matches_the_oracle(oracle, monkeypatch):
    """640x480, 50 000 events per window (BASELINE config 2): chosen without being forced; ERGO-12, EventStack, voxel
    bit-exact vs the oracle."""
    from event_representation_study_amd import engine as eng
    monkeypatch.delenv("EVREP_BIN_CLASSIC", raising=False)
    monkeypatch.delenv("EVREP_BIN_KEY_SORTED", raising=False)
    monkeypatch.delenv("EVREP_BIN_THREE_KERNEL", raising=False)
    H, W = 480, 640
    wins = [make_events(50000, W, H, seed=900 + i, polarity="pm1" if i else "01") for i in range(3)]
    eb = eng.EventBatch.from_numpy(wins, H, W)
    assert eb.plan.reserved == 2
    got, es, vx = eb.optimized().cpu().numpy(), eb.event_stack().cpu().numpy(), eb.voxel(5).cpu().numpy()
    for b, ev in enumerate(wins):
        assert_bit_equal(got[b], oracle.ergo12(ev, H, W), "ergo12 w%d" % b)
        assert_bit_equal(es[b], oracle.event_stack(ev, H, W), "event stack w%d" % b)
        assert_bit_equal(vx[b], oracle.voxel(ev, H, W, 5), "voxel w%d" % b)
    assert not eb.status().any()
    # r03: up to ~110 records per unit the builder waves still order their own records (two LDS batches) ...
    mid = make_events(200000, W, H, seed=1)
    ebm = eng.EventBatch.from_numpy([mid], H, W)
    assert ebm.plan.reserved == 2
    assert_bit_equal(ebm.optimized().cpu().numpy()[0], oracle.ergo12(mid, H, W), "ergo12, 83 records per unit")
    # ... r04: up to ~220 per unit (500 000 events here) the builders' warm path -- a unit sorted in LDS over its tile --
    # still beats the per-key column sort for one builder per binning pass
    d5 = make_events(500000, W, H, seed=1)
    ebd = eng.EventBatch.from_numpy([d5], H, W)
    assert ebd.plan.reserved == 2
    assert_bit_equal(ebd.optimized().cpu().numpy()[0], oracle.ergo12(d5, H, W), "ergo12, 208 records per unit")
    assert_bit_equal(ebd.event_stack().cpu().numpy()[0], oracle.event_stack(d5, H, W), "event stack, 208 records per unit")
    # ... a denser window of the same sensor: k_block_keysort + the per-key column sort, then the classic builders
    assert eng.EventBatch.from_numpy([make_events(600000, W, H, seed=1)], H, W).plan.reserved == 3
    # the reference's own Gen1 shape (304x240, 50 000 events: ~69 records per unit) is on the key-sorted pass
    g1 = make_events(50000, 304, 240, seed=2)
    eg = eng.EventBatch.from_numpy([g1], 240, 304)
    assert eg.plan.reserved == 2
    assert_bit_equal(eg.optimized().cpu().numpy()[0], oracle.ergo12(g1, 240, 304), "ergo12, Gen1 shape")
    assert_bit_equal(eg.event_stack().cpu().numpy()[0], oracle.event_stack(g1, 240, 304), "event stack, Gen1 shape")


def test_gen4_sensor_two_round_stage(oracle, monkeypatch):
    """1280x720 (BASELINE config 3): 7200 keys leave room for a 4096-record stage, the block is written in two rounds."""
    from event_representation_study_amd import engine as eng
    monkeypatch.delenv("EVREP_BIN_CLASSIC", raising=False)
    monkeypatch.delenv("EVREP_BIN_THREE_KERNEL", raising=False)
    H, W = 720, 1280
    wins = [make_events(200000, W, H, seed=31), make_events(8000, W, H, seed=32, polarity="01")]
    eb = eng.EventBatch.from_numpy(wins, H, W)
    assert eb.plan.reserved == 2
    got, es = eb.optimized().cpu().numpy(), eb.event_stack().cpu().numpy()
    tf = eb.tore(6, frame_mode=2).cpu().numpy()
    for b, ev in enumerate(wins):
        assert_bit_equal(got[b], oracle.ergo12(ev, H, W), "ergo12 w%d" % b)
        assert_bit_equal(es[b], oracle.event_stack(ev, H, W), "event stack w%d" % b)
        want = oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W))
        np.testing.assert_allclose(tf[b], want, rtol=1e-6, atol=1e-6)


def test_hot_pixels(oracle, monkeypatch):
    """A pixel that fires thousands of times: one key group of the block sort holds most of a block (its members are
    put in time order by counting smaller ranks), the builder unit spills; a second hot pixel sits in the last chunk."""
    from event_representation_study_amd import engine as eng
    H, W = 60, 300
    rng = np.random.default_rng(5)
    wins = []
    for n, hot in ((30000, 9000), (9000, 8500)):
        ev = make_events(n, W, H, seed=n)
        idx = np.sort(rng.choice(n, hot, replace=False))
        ev[idx[: hot // 2], 0], ev[idx[: hot // 2], 1] = 17, 33
        ev[idx[hot // 2:], 0], ev[idx[hot // 2:], 1] = W - 1, H - 1
        wins.append(ev)
    ks, cl = _batches(eng, wins, H, W, monkeypatch)
    a, b = _builders(ks), _builders(cl)
    _same(a, b, "hot")
    for i, ev in enumerate(wins):
        assert_bit_equal(a["ergo12"][i], oracle.ergo12(ev, H, W), "hot ergo12 w%d" % i)
        assert_bit_equal(a["event_stack"][i], oracle.event_stack(ev, H, W), "hot event stack w%d" % i)


@pytest.mark.parametrize("n", [17 * 8192 - 3, 40 * 8192 + 11, 64 * 8192])
def test_many_block_runs(oracle, n, monkeypatch):
    """More than 16 block runs per window: the builder waves find a record's run by the LDS search."""
    from event_representation_study_amd import engine as eng
    H, W = 100, 260
    wins = [make_events(n, W, H, seed=n % 1000), make_events(500, W, H, seed=3)]
    ks, cl = _batches(eng, wins, H, W, monkeypatch)
    for b, ev in enumerate(wins):
        assert_bit_equal(ks.optimized()[b].cpu().numpy(), oracle.ergo12(ev, H, W), "ergo12 n=%d" % len(ev))
    assert_bit_equal(ks.event_stack().cpu().numpy(), cl.event_stack().cpu().numpy(), "event stack")
    assert_bit_equal(ks.tore(6, frame_mode=1).cpu().numpy(), cl.tore(6, frame_mode=1).cpu().numpy(), "tore")


@pytest.mark.parametrize("n", [64 * 8192 + 1, 100 * 8192 + 77, 128 * 8192])
def test_more_than_64_block_runs(oracle, n, monkeypatch):
    """r06: windows of 65 .. 128 block runs stay on the key-sorted pass -- the stream builders gather two runs per lane; an
    ordered builder (one lane per run) gets the pixel-sorted stream from the per-key column sort, run in front of its launch."""
    import torch
    from event_representation_study_amd import engine as eng
    H, W = 100, 260
    wins = [make_events(n, W, H, seed=n % 1000), make_events(700, W, H, seed=4, polarity="01")]
    monkeypatch.delenv("EVREP_BIN_CLASSIC", raising=False)
    monkeypatch.delenv("EVREP_BIN_THREE_KERNEL", raising=False)
    monkeypatch.setenv("EVREP_BIN_KEY_SORTED", "1")
    ks = eng.EventBatch.from_numpy(wins, H, W)
    assert ks.plan.reserved == 2 and ks.plan.nblk == (n + 8191) // 8192 > 64
    monkeypatch.delenv("EVREP_BIN_KEY_SORTED", raising=False)
    monkeypatch.setenv("EVREP_BIN_CLASSIC", "1")
    cl = eng.EventBatch.from_numpy(wins, H, W)
    assert cl.plan.reserved in (0, 1)
    for b, ev in enumerate(wins):
        assert_bit_equal(ks.optimized()[b].cpu().numpy(), oracle.ergo12(ev, H, W), "ergo12 (stream) n=%d" % len(ev))
        assert_bit_equal(ks.voxel(5)[b].cpu().numpy(), oracle.voxel(ev, H, W, 5), "voxel (stream) n=%d" % len(ev))
    tn = torch.rand(ks.total, dtype=torch.float64, device=ks.device, generator=torch.Generator(device=ks.device).manual_seed(5))
    for tag, fn in {
        "event stack (stream)": lambda eb: eb.event_stack(),
        "tore, shifted frame (stream)": lambda eb: eb.tore(6, frame_mode=1),
        "tore, negative scale (ordered, behind the column sort)": lambda eb: eb.tore(6, frame_mode=2, scale=-1.0),
        "accumulators (stream)": lambda eb: eb.polstats(tn, [1, 2, 0, 0], [0, 1, 2, 4], tau=0.4),
        "ergo12 float32 (stream)": lambda eb: eb.optimized(dtype=torch.float32),
        "mdes, other triples (ordered, behind the column sort)": lambda eb: eb.mdes([0, 4, 6, 2], [0, 1, 5, 6], [2, 3, 0, 1]),
    }.items():
        assert_bit_equal(fn(ks).cpu().numpy(), fn(cl).cpu().numpy(), "%s n=%d" % (tag, n))
    np.testing.assert_allclose(ks.time_surface().cpu().numpy(), cl.time_surface().cpu().numpy(), rtol=1e-13, atol=0)
    np.testing.assert_array_equal(ks.status(), cl.status())
    np.testing.assert_array_equal(ks.bbox(), cl.bbox())


def test_status_bbox_and_failed_channels(monkeypatch):
    """Out-of-frame events (zeroed MDES channels, OOB status), an unsorted window, an empty one, flat time: the block
    statistics merged by the builder waves / k_window_meta say what WindowMeta of the classic passes says."""
    from event_representation_study_amd import engine as eng
    H, W = 40, 150
    a = make_events(5000, W, H, seed=1)
    a[100, 0] = W + 7            # in-frame flat index, column outside [0, W)
    a[4000, 1] = H + 3           # out of frame
    a[4500, 0] = -5
    b = make_events(3000, W, H, seed=2)
    b[10, 2] = b[11, 2] + 5      # unsorted pair
    c = make_events(700, W, H, seed=3)
    c[:, 2] = 77                 # flat time
    d = np.zeros((0, 4), dtype=np.int32)
    e = make_events(2500, W, H, seed=4, polarity="01")
    wins = [a, b, c, d, e]
    ks, cl = _batches(eng, wins, H, W, monkeypatch)
    np.testing.assert_array_equal(ks.status(), cl.status())
    np.testing.assert_array_equal(ks.bbox(), cl.bbox())
    assert_bit_equal(ks.optimized().cpu().numpy(), cl.optimized().cpu().numpy(), "ergo12 with failed channels")
    assert_bit_equal(ks.event_stack().cpu().numpy(), cl.event_stack().cpu().numpy(), "event stack")
    assert_bit_equal(ks.tore(6, frame_mode=2).cpu().numpy(), cl.tore(6, frame_mode=2).cpu().numpy(), "tore")


def test_subpixel_voxel_gets_its_column_sorted_stream(monkeypatch):
    """k_voxel_subpixel walks the pixel-sorted stream: after the key-sorted pass the column sort runs on demand."""
    from event_representation_study_amd import engine as eng
    H, W = 60, 200
    wins = [make_events(6000, W, H, seed=8), make_events(300, W, H, seed=9)]
    ks, cl = _batches(eng, wins, H, W, monkeypatch)
    rng = np.random.default_rng(0)
    xy = torch.from_numpy(np.concatenate([w[:, :2] for w in wins]).astype(np.float64) + rng.random((6300, 2)) * 0.99).to("cuda:0")
    assert_bit_equal(ks.voxel_subpixel(xy, 5).cpu().numpy(), cl.voxel_subpixel(xy, 5).cpu().numpy(), "sub-pixel voxel")
    # and the builders still read the key-sorted runs afterwards
    assert_bit_equal(ks.optimized().cpu().numpy(), cl.optimized().cpu().numpy(), "ergo12 after the on-demand column sort")


def test_rebinning_is_idempotent_and_deterministic(monkeypatch):
    from event_representation_study_amd import engine as eng
    H, W = 480, 640
    wins = [make_events(50000, W, H, seed=40 + i) for i in range(4)]
    monkeypatch.delenv("EVREP_BIN_CLASSIC", raising=False)
    monkeypatch.delenv("EVREP_BIN_THREE_KERNEL", raising=False)
    eb = eng.EventBatch.from_numpy(wins, H, W)
    assert eb.plan.reserved == 2
    first = eb.optimized().clone()
    for _ in range(20):
        eb.rebin()
        assert torch.equal(eb.optimized(), first)


def test_placement_probe_writes_zeros_and_returns_a_candidate():
    """evrep_probe_store (the builder's write footprint, NOTES.md 8) zero-fills whole 12 KiB tiles of the tensor;
    probe_output_placement returns one of its candidates with one timing per candidate."""
    from event_representation_study_amd import engine as eng
    best, us, all_us = eng.probe_output_placement((4, 48, 64, 12), torch.float64, candidates=3, launches=2)
    assert best.shape == (4, 48, 64, 12) and len(all_us) == 3 and us == min(all_us)
    nbytes = best.numel() * 8
    covered = (nbytes // 12288) * 12288 // 8
    assert not best.reshape(-1)[:covered].any()
    # a caller-supplied writer is timed instead when given
    calls = []
    best2, _, t2 = eng.probe_output_placement((2, 8, 8, 2), torch.float32, launch=lambda o: calls.append(o.data_ptr()) or o.fill_(1.0),
                                              candidates=2, launches=1)
    assert len(t2) == 2 and len(set(calls)) == 2 and bool((best2 == 1).all())
