"""-m gpu: randomized geometry / density sweep of every builder against the oracle, aimed at the
internal boundaries of the HIP path (64-record LDS stage, 64-segment fast path, 256-record column-sort
batches, 2048-event partition blocks, part tiles, two-chunk units, ragged batches)."""
import numpy as np
import pytest

from conftest import assert_bit_equal

from event_representation_study_amd.synthetic import make_events

pytestmark = pytest.mark.gpu


def _check_all(eng, oracle, wins, H, W, tag):
    eb = eng.EventBatch.from_numpy(wins, H, W)
    opt = eb.optimized().cpu().numpy()
    opt32 = eb.optimized(dtype=__import__("torch").float32).cpu().numpy()
    es = eb.event_stack().cpu().numpy()
    ts = eb.time_surface().cpu().numpy()
    tore_full = eb.tore(6, frame_mode=2).cpu().numpy()
    vox = eb.voxel(5).cpu().numpy()
    for b, ev in enumerate(wins):
        if ev.shape[0] == 0:
            continue
        ref = oracle.ergo12(ev, H, W)
        assert_bit_equal(opt[b], ref, "%s ergo12 w%d" % (tag, b))
        assert_bit_equal(opt32[b], ref.astype(np.float32), "%s ergo12 f32 w%d" % (tag, b))
        assert_bit_equal(es[b], oracle.event_stack(ev, H, W), "%s event_stack w%d" % (tag, b))
        if ev.shape[0] >= 2 and ev[-1, 2] != ev[0, 2]:
            np.testing.assert_allclose(ts[b], oracle.time_surface(ev, H, W), rtol=1e-12, err_msg="%s ts w%d" % (tag, b))
            assert_bit_equal(vox[b], oracle.voxel(ev, H, W, 5), "%s voxel w%d" % (tag, b))
        # full-frame TORE (frame_mode 2) against the oracle's 1-based entry point
        want = oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W))
        np.testing.assert_allclose(tore_full[b], want, rtol=1e-6, atol=1e-6, err_msg="%s tore w%d" % (tag, b))


@pytest.mark.parametrize("seed", range(12))
def test_random_geometry_and_density(seed, oracle):
    from event_representation_study_amd import engine as eng
    rng = np.random.default_rng(1000 + seed)
    W = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 300, 640]))
    H = int(rng.integers(1, 40))
    B = int(rng.integers(1, 6))
    dens = float(rng.choice([0.02, 0.16, 0.5, 1.0, 4.0]))          # events per pixel
    wins = []
    for b in range(B):
        n = max(1, int(dens * W * H * rng.uniform(0.5, 1.5)))
        if rng.random() < 0.15:
            n = int(rng.choice([1, 2, 3, 63, 64, 65, 2047, 2048, 2049]))
        wins.append(make_events(n, W, H, seed=7 * seed + b, polarity="pm1" if rng.random() < 0.5 else "01",
                                span_us=int(rng.choice([3, 1000, 50000]))))
    _check_all(eng, oracle, wins, H, W, "seed%d %dx%d" % (seed, W, H))


@pytest.mark.parametrize("per_chunk", [63, 64, 65, 127, 128, 129])
def test_records_per_chunk_boundaries(per_chunk, oracle):
    """Exactly `per_chunk` records in one 128-pixel chunk: the LDS stage (64) and one-lane-per-pixel (64 segments) edges."""
    from event_representation_study_amd import engine as eng
    H, W = 3, 256
    rng = np.random.default_rng(per_chunk)
    n = per_chunk
    ev = np.zeros((n, 4), np.int32)
    ev[:, 0] = rng.permutation(128)[:n] if n <= 128 else np.concatenate([np.arange(128), rng.integers(0, 128, n - 128)])
    ev[:, 1] = 1
    ev[:, 2] = np.arange(n) * 3
    ev[:, 3] = rng.choice([-1, 1], n)
    _check_all(eng, oracle, [ev], H, W, "chunk%d" % per_chunk)


@pytest.mark.parametrize("per_row", [255, 256, 257, 511, 513, 1500])
def test_records_per_row_boundaries(per_row, oracle):
    """Row lengths around the column sort's 256-record register batches."""
    from event_representation_study_amd import engine as eng
    H, W = 4, 96
    rng = np.random.default_rng(per_row)
    n = per_row + 40
    ev = np.zeros((n, 4), np.int32)
    ev[:, 0] = rng.integers(0, W, n)
    ev[:, 1] = 2
    ev[rng.choice(n, 40, replace=False), 1] = rng.integers(0, H, 40)
    ev[:, 2] = np.sort(rng.integers(0, 5000, n))
    ev[:, 3] = rng.choice([-1, 1], n)
    _check_all(eng, oracle, [ev], H, W, "row%d" % per_row)


def test_window_sizes_around_partition_blocks(oracle):
    """Windows of 2047..4097 events: the 2048-event partition blocks and their wave quarters."""
    from event_representation_study_amd import engine as eng
    H, W = 30, 40
    wins = [make_events(n, W, H, seed=n) for n in (2047, 2048, 2049, 4095, 4096, 4097, 511, 513)]
    _check_all(eng, oracle, wins, H, W, "blocks")


@pytest.mark.parametrize("sizes", [(8191, 8192, 8193), (16 * 8192, 16 * 8192 + 1), (17 * 8192 - 5, 40000),
                                   (64 * 8192, 3), (64 * 8192 + 1, 9000), (128 * 8192, 70 * 8192 + 7), (128 * 8192 + 1, 100)])
def test_window_sizes_around_two_kernel_binning_limits(oracle, sizes, monkeypatch):
    """The two-kernel binning pass: 8192-event workgroup blocks, <= 16 block runs per row found by a readlane chain,
    17..128 by the LDS search (two runs per lane), more than 128 x 8192 events per window -> the three-kernel pass.  Every size next to a
    short window in the same batch; ERGO-12 / EventStack / voxel bit-exact vs the oracle.
    (EVREP_BIN_CLASSIC keeps the key-sorted pass, tests/test_gpu_key_sorted.py, out of the choice.)"""
    from event_representation_study_amd import engine as eng
    monkeypatch.setenv("EVREP_BIN_CLASSIC", "1")
    monkeypatch.delenv("EVREP_BIN_THREE_KERNEL", raising=False)
    H, W = 36, 200
    wins = [make_events(n, W, H, seed=n % 1000 + 3) for n in sizes]
    eb = eng.EventBatch.from_numpy(wins, H, W)
    assert eb.plan.reserved == (1 if max(sizes) <= 128 * 8192 else 0)
    got, es, vx = eb.optimized().cpu().numpy(), eb.event_stack().cpu().numpy(), eb.voxel(5).cpu().numpy()
    for b, ev in enumerate(wins):
        assert_bit_equal(got[b], oracle.ergo12(ev, H, W), "ergo12 n=%d" % len(ev))
        assert_bit_equal(es[b], oracle.event_stack(ev, H, W), "event stack n=%d" % len(ev))
        if len(ev) > 3:
            assert_bit_equal(vx[b], oracle.voxel(ev, H, W, 5), "voxel n=%d" % len(ev))
    assert not eb.status().any()


@pytest.mark.parametrize("H", [880, 882, 1200])
def test_tall_sensors_around_the_lds_limit_of_the_row_sort(oracle, H, monkeypatch):
    """k_block_rowsort keeps 16 x H packed row counters next to its 128 KB record stage: sensors of up to ~881 rows
    take the two-kernel pass, taller ones the three-kernel pass; same tensors either way."""
    from event_representation_study_amd import engine as eng
    monkeypatch.setenv("EVREP_BIN_CLASSIC", "1")
    monkeypatch.delenv("EVREP_BIN_THREE_KERNEL", raising=False)
    W = 70
    wins = [make_events(30000, W, H, seed=H), make_events(100, W, H, seed=H + 1)]
    eb = eng.EventBatch.from_numpy(wins, H, W)
    assert eb.plan.reserved == (1 if H <= 881 else 0)
    got = eb.optimized().cpu().numpy()
    for b, ev in enumerate(wins):
        assert_bit_equal(got[b], oracle.ergo12(ev, H, W), "ergo12 H=%d" % H)
    assert_bit_equal(eb.event_stack()[0].cpu().numpy(), oracle.event_stack(wins[0], H, W), "event stack H=%d" % H)


def test_three_kernel_binning_pass_still_agrees(oracle, monkeypatch):
    """EVREP_BIN_THREE_KERNEL forces the round-1 pass for windows the two-kernel pass would take: same tensors."""
    from event_representation_study_amd import engine as eng
    H, W = 120, 160
    wins = [make_events(20000, W, H, seed=5), make_events(9000, W, H, seed=6, polarity="01")]
    monkeypatch.setenv("EVREP_BIN_CLASSIC", "1")
    monkeypatch.delenv("EVREP_BIN_THREE_KERNEL", raising=False)
    a = eng.EventBatch.from_numpy(wins, H, W)
    monkeypatch.setenv("EVREP_BIN_THREE_KERNEL", "1")
    b = eng.EventBatch.from_numpy(wins, H, W)
    assert a.plan.reserved == 1 and b.plan.reserved == 0
    assert_bit_equal(a.optimized().cpu().numpy(), b.optimized().cpu().numpy(), "two- vs three-kernel ergo12")
    assert_bit_equal(a.time_surface().cpu().numpy(), b.time_surface().cpu().numpy(), "two- vs three-kernel time surface")
    np.testing.assert_array_equal(a.bbox(), b.bbox())
    np.testing.assert_array_equal(a.status(), b.status())


@pytest.mark.parametrize("seed", range(4))
def test_tore_frame_modes(seed, oracle):
    """frame_mode 0 (bounding box, the dispatcher) and 1 (full frame, origin-shifted: n_imagenet's call) with
    events confined to a sub-rectangle, so the shift straddles chunk boundaries."""
    from event_representation_study_amd import engine as eng
    rng = np.random.default_rng(seed)
    H, W = 50, 700
    x0, y0 = int(rng.integers(1, 300)), int(rng.integers(1, 20))
    wbb, hbb = int(rng.integers(130, 390)), int(rng.integers(5, 25))
    n = 6000
    ev = make_events(n, wbb, hbb, seed=seed + 50)
    ev[:, 0] += x0
    ev[:, 1] += y0
    eb = eng.EventBatch.from_numpy(ev, H, W)
    got0 = eb.tore(6, frame_mode=0)[0].cpu().numpy()
    np.testing.assert_allclose(got0, oracle.tore_bbox(ev, 6), rtol=1e-6, atol=1e-6)
    got1 = eb.tore(6, frame_mode=1)[0].cpu().numpy()
    x = ev[:, 0] - ev[:, 0].min() + 1
    y = ev[:, 1] - ev[:, 1].min() + 1
    want1 = oracle.tore(x, y, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W))
    np.testing.assert_allclose(got1, want1, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n", [30000, 65536, 65537])
def test_clustered_windows(oracle, n):
    """Clustered windows: every event inside seven sensor rows and a third of the stream on ONE pixel (long
    rows for the column sort, dense chunks and a 10 000+-event pixel segment for the builders), next to a
    sparse window in the same batch -- bit-exact vs the oracle."""
    from event_representation_study_amd import engine as eng
    H, W = 480, 640
    ev = make_events(n, W, H, seed=n)
    ev[:, 1] = ev[:, 1] % 7 + 180            # rows 180..186: one band of 60 rows gets everything
    ev[: n // 3, 0] = 321                    # and a third of the stream on one column
    ev[: n // 3, 1] = 183
    eb = eng.EventBatch.from_numpy([ev, make_events(777, W, H, seed=1)], H, W)
    got = eb.optimized().cpu().numpy()
    assert_bit_equal(got[0], oracle.ergo12(ev, H, W), "clustered ergo12 n=%d" % n)
    assert_bit_equal(got[1], oracle.ergo12(make_events(777, W, H, seed=1), H, W), "sparse neighbour window")
    assert_bit_equal(eb.event_stack()[0].cpu().numpy(), oracle.event_stack(ev, H, W), "clustered event stack")
    assert_bit_equal(eb.voxel(5)[0].cpu().numpy(), oracle.voxel(ev, H, W, 5), "clustered voxel")


def test_batch_of_only_empty_windows():
    """No event at all (a NULL events pointer is legal when total_events == 0): every builder returns its
    background and every window reports EVREP_ST_EMPTY."""
    import torch
    from event_representation_study_amd import _lib, engine as eng
    H, W = 33, 130
    eb = eng.EventBatch.from_numpy([np.zeros((0, 4), np.int32)] * 3, H, W)
    assert all(int(s) & _lib.ST_EMPTY for s in eb.status())
    assert float(eb.optimized().abs().sum()) == 0.0
    assert float(eb.event_stack().abs().sum()) == 0.0
    assert float(eb.voxel(5).abs().sum()) == 0.0
    assert float(eb.polstats(torch.zeros(0, dtype=torch.float64, device="cuda:0"), [0, 1], [0, 1]).abs().sum()) == 0.0
    # TORE's empty-FIFO value (tore.py:69-79: inf -> 5e8 -> log(5e8 + 1) - log(151)) in EVERY element, in both
    # full-frame modes, and for an empty window sitting between non-empty ones of a batch
    bgv = np.float32(np.float64(np.log(np.float32(5e8) + np.float32(1.0))) - np.log(151.0))
    for mode in (1, 2):
        tr = eb.tore(6, frame_mode=mode)
        assert tuple(tr.shape) == (3, H, W, 12)
        vals = np.unique(tr.cpu().numpy())
        assert vals.size == 1 and abs(float(vals[0]) - float(bgv)) <= 1e-6 * float(bgv)   # device logf vs libm: 1 ulp
    some = make_events(500, W, H, seed=5)
    mixed = eng.EventBatch.from_numpy([some, np.zeros((0, 4), np.int32), some], H, W)
    for mode in (1, 2):
        out = torch.full((3, H, W, 12), float("nan"), device="cuda:0")     # poisoned: every element must be written
        tr = mixed.tore(6, frame_mode=mode, out=out).cpu().numpy()
        vals = np.unique(tr[1])
        assert vals.size == 1 and abs(float(vals[0]) - float(bgv)) <= 1e-6 * float(bgv)
        assert np.array_equal(tr[0], tr[2]) and not np.isnan(tr).any()
    ts = eb.time_surface().cpu().numpy()
    assert not np.isnan(ts).any() and float(np.abs(ts).sum()) == 0.0       # no slice is ever reached: all zero
