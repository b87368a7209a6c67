"""-m gpu: full-size (BASELINE.json configs[1]/[2]) checks through size-independent properties,
plus dense / hot-pixel / ragged edge cases against the oracle."""
import numpy as np
import pytest
import torch

from conftest import assert_bit_equal

from event_representation_study_amd.synthetic import make_events

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from event_representation_study_amd import engine
    return engine


def test_c2_batch32_properties(eng):
    """640x480x12, 50k events/window, 32 windows: per-window channel identities that hold for any input."""
    H, W, N, B = 480, 640, 50000, 32
    wins = [make_events(N, W, H, seed=i) for i in range(B)]
    eb = eng.EventBatch.from_numpy(wins, H, W)
    rep = eb.optimized()
    assert tuple(rep.shape) == (B, H, W, 12)
    # ch5 = count over window 6 (sum): integer-valued and sums to the window length N - (N//2+N//4+N//8)
    w6 = N - (N // 2 + N // 4 + N // 8)
    ch5 = rep[..., 5]
    assert torch.equal(ch5, ch5.round()) and torch.all(ch5.sum(dim=(1, 2)) == w6)
    # ch11 = occupancy mask of the first third; ch2/ch4/ch7 are masks too
    for c in (2, 4, 7, 11):
        assert torch.all((rep[..., c] == 0) | (rep[..., c] == 1))
    # ch3 = polarity sum over window 6: |ch3| <= ch5 and same parity
    assert torch.all(rep[..., 3].abs() <= ch5) and torch.all((rep[..., 3] - ch5) % 2 == 0)
    # maxima of normalised timestamps lie in [0, 1]; variance channels are >= -1e-12
    for c in (8, 9, 10):
        assert float(rep[..., c].min()) >= 0.0 and float(rep[..., c].max()) <= 1.0
    assert float(rep[..., 0].min()) >= 0.0 and float(rep[..., 1].min()) > -1e-12
    # windows are independent: window 7 alone reproduces slice 7 bit for bit (idempotence / no cross-talk)
    single = eng.EventBatch.from_numpy(wins[7], H, W).optimized()
    assert torch.equal(single[0], rep[7])
    # re-running the whole pipeline is deterministic
    assert torch.equal(eb.rebin().optimized(), rep)
    # f32 output = rounded f64 output; x255 = exact float64 product
    assert torch.equal(eb.optimized(dtype=torch.float32), rep.to(torch.float32))
    assert torch.equal(eb.optimized(scale=255.0), rep * 255.0)


def test_c2_batch32_every_slice_vs_oracle(eng, oracle):
    """The headline batch as a batch: all 32 slices of one 640x480 launch against the oracle, bit
    for bit, float64 and float32 outputs, both polarity encodings in the same batch."""
    H, W, N, B = 480, 640, 50000, 32
    wins = [make_events(N, W, H, seed=900 + i, polarity=("pm1" if i % 2 == 0 else "01")) for i in range(B)]
    eb = eng.EventBatch.from_numpy(wins, H, W)
    rep = eb.optimized().cpu().numpy()
    rep32 = eb.optimized(dtype=torch.float32).cpu().numpy()
    es = eb.event_stack().cpu().numpy()
    for b, ev in enumerate(wins):
        ref = oracle.ergo12(ev, H, W)
        assert_bit_equal(rep[b], ref, "ergo12 slice %d of the c2 batch" % b)
        assert_bit_equal(rep32[b], ref.astype(np.float32), "ergo12 f32 slice %d" % b)
        assert_bit_equal(es[b], oracle.event_stack(ev, H, W), "event_stack slice %d" % b)


def test_c2_window_vs_oracle_fullsize(eng, oracle):
    H, W, N = 480, 640, 50000
    ev = make_events(N, W, H, seed=424242)
    eb = eng.EventBatch.from_numpy(ev, H, W)
    assert_bit_equal(eb.optimized()[0].cpu().numpy(), oracle.ergo12(ev, H, W), "ergo12 640x480")
    assert_bit_equal(eb.event_stack()[0].cpu().numpy(), oracle.event_stack(ev, H, W), "event_stack 640x480")


@pytest.mark.parametrize("n", [200000, 1000000])
def test_c3_1mpx_sweep_vs_oracle(eng, oracle, n):
    """BASELINE.json configs[2]: TimeSurface + EventStack + ToRE on 1280x720 windows."""
    H, W = 720, 1280
    ev = make_events(n, W, H, seed=n)
    eb = eng.EventBatch.from_numpy(ev, H, W)
    assert_bit_equal(eb.event_stack()[0].cpu().numpy(), oracle.event_stack(ev, H, W), "event_stack 1Mpx")
    ts = eb.time_surface()[0].cpu().numpy()
    np.testing.assert_allclose(ts, oracle.time_surface(ev, H, W), rtol=1e-12)     # budget 1e-5 rel
    tore = eb.tore(6, frame_mode=0)[0].cpu().numpy()
    ref = oracle.tore_bbox(ev, 6)
    assert tore.shape == ref.shape
    np.testing.assert_allclose(tore, ref, rtol=1e-6, atol=1e-6)                  # budget 1e-5 rel
    if n == 200000:
        assert_bit_equal(eb.optimized()[0].cpu().numpy(), oracle.ergo12(ev, H, W), "ergo12 1Mpx")


def test_dense_stress_500k(eng, oracle):
    """SURVEY 8(d) dense stress point: 500 000 events in one 640x480 window (chunks overflow the LDS stage)."""
    H, W, N = 480, 640, 500000
    ev = make_events(N, W, H, seed=5)
    eb = eng.EventBatch.from_numpy(ev, H, W)
    assert_bit_equal(eb.optimized()[0].cpu().numpy(), oracle.ergo12(ev, H, W), "ergo12 dense")
    assert_bit_equal(eb.voxel(5)[0].cpu().numpy(), oracle.voxel(ev, H, W, 5), "voxel dense")
    # r05b: on windows this dense the MAIN launches of TORE and of the n_imagenet accumulators run their order-free sweeps
    # themselves (k_tore / k_polstats, SM): full frame, the shifted bounding-box frame (ordered ways inside the same launch), a
    # second window in array order (unsorted timestamps: ordered ways), the accumulators incl. one with an empty-pixel background
    np.testing.assert_allclose(eb.tore(6, frame_mode=2)[0].cpu().numpy(),
                               oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W)), rtol=1e-6, atol=1e-6)
    ev2 = ev.copy()
    ev2[:, 0] = 5 + ev2[:, 0] % 600          # a bounding box that does not start at a chunk boundary
    un = ev.copy()
    k = np.random.default_rng(3).random(N) < 0.2
    un[k, 2] = np.random.default_rng(4).integers(0, 60000, size=int(k.sum()))
    eb2 = eng.EventBatch.from_numpy([ev2, un], H, W)
    tb = eb2.tore(6, frame_mode=0)
    np.testing.assert_allclose(tb[0].cpu().numpy(), oracle.tore_bbox(ev2, 6), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(eb2.tore(6, frame_mode=2)[1].cpu().numpy(),
                               oracle.tore(un[:, 0] + 1, un[:, 1] + 1, un[:, 2], un[:, 3], un[-1, 2], 6, (H, W)), rtol=1e-6, atol=1e-6)
    ni = ev.copy()
    ni[:, 3] = np.where(ev[:, 3] > 0, 1, -1)
    rows = ni.astype(np.float64)
    t = ev[:, 2].astype(np.float64)
    tn = torch.from_numpy((t - t[0]) / (t[-1] - t[0])).cuda()
    ebn = eng.EventBatch.from_numpy(ni, H, W)
    for name, pol, stat in (("acc_all", [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2]), ("acc_exp", [1, 2], [4, 4])):
        got = np.moveaxis(ebn.polstats(tn, pol, stat, tau=0.3)[0].cpu().numpy(), -1, 0)
        np.testing.assert_allclose(got, oracle.nimagenet_acc(name, rows, H, W), rtol=1e-6, atol=1e-7, err_msg=name)


def test_hot_pixels_and_hot_rows(eng, oracle):
    """A flickering pixel (20k events on one pixel) and a saturated row (long-row column sort path)."""
    H, W = 60, 80
    ev = make_events(40000, W, H, seed=9)
    ev[::2, 0] = 17
    ev[::2, 1] = 23                      # every other event on pixel (23, 17)
    ev[1::4, 1] = 40                     # a quarter of the events on row 40
    eb = eng.EventBatch.from_numpy(ev, H, W)
    assert_bit_equal(eb.optimized()[0].cpu().numpy(), oracle.ergo12(ev, H, W), "ergo12 hot")
    assert_bit_equal(eb.event_stack()[0].cpu().numpy(), oracle.event_stack(ev, H, W), "event_stack hot")
    np.testing.assert_allclose(eb.time_surface()[0].cpu().numpy(), oracle.time_surface(ev, H, W), rtol=1e-12)
    np.testing.assert_allclose(eb.tore(6, 0)[0].cpu().numpy(), oracle.tore_bbox(ev, 6), rtol=1e-6, atol=1e-6)
    assert_bit_equal(eb.voxel(5)[0].cpu().numpy(), oracle.voxel(ev, H, W, 5), "voxel hot")


@pytest.mark.parametrize("W,H", [(1, 1), (3, 2), (127, 5), (128, 3), (129, 4), (4096, 2), (7, 4096)])
def test_odd_geometries(eng, oracle, W, H):
    n = 3000
    ev = make_events(n, W, H, seed=W * 10007 + H)
    eb = eng.EventBatch.from_numpy(ev, H, W)
    assert_bit_equal(eb.optimized()[0].cpu().numpy(), oracle.ergo12(ev, H, W), "ergo12 %dx%d" % (W, H))
    assert_bit_equal(eb.event_stack()[0].cpu().numpy(), oracle.event_stack(ev, H, W), "es %dx%d" % (W, H))
    tore = eb.tore(6, 0)[0].cpu().numpy()
    np.testing.assert_allclose(tore, oracle.tore_bbox(ev, 6), rtol=1e-6, atol=1e-6)


def test_maximum_frame(eng, oracle):
    """EVREP_MAX_DIM x EVREP_MAX_DIM (4096 x 4096): the widest rows (32 chunks, 16 KB of column counters),
    the tallest row table (the unfused scan/scatter pair), 1.6 GB of float64 output -- bit-exact vs the oracle;
    one window more than the 16-bit grid limit is refused by the plan."""
    import ctypes
    from event_representation_study_amd import _lib
    H = W = 4096
    ev = make_events(300000, W, H, seed=4096)
    eb = eng.EventBatch.from_numpy(ev, H, W)
    got = eb.optimized()[0].cpu().numpy()
    assert_bit_equal(got, oracle.ergo12(ev, H, W), "ergo12 4096x4096")
    del got
    es = eb.event_stack()[0].cpu().numpy()
    assert_bit_equal(es, oracle.event_stack(ev, H, W), "event stack 4096x4096")
    assert _lib.load().evrep_plan_init(ctypes.byref(_lib.Plan()), 1, H + 1, W, 10, 10) == _lib.EVREP_EINVAL


def test_gwd_fullsize_properties(eng, oracle):
    """GWD at the reference's size (n ~ 12.5k, m = 14.4k): symmetry-free invariants + sampled check."""
    rng = np.random.default_rng(3)
    n, m = 12500, 14400
    Xs = rng.random((n, 4))
    Xt = rng.random((m, 14)) * np.array([255.0] * 12 + [1.0, 1.0])
    c = float(eng.gwd_padded_l1(Xs, Xt).item())
    assert 0.0 < c < 1.0
    # invariance: translating a cloud or permuting feature columns leaves the Gaussian kernels unchanged
    c2 = float(eng.gwd_padded_l1(Xs + 3.0, Xt[:, ::-1].copy()).item())
    assert abs(c - c2) <= 1e-6 * c
    # identical clouds -> exactly zero
    assert float(eng.gwd_padded_l1(Xs, Xs).item()) == 0.0
    # sub-sampled problem against the oracle (seconds on one core)
    sub = float(eng.gwd_padded_l1(Xs[:1500], Xt[:1700]).item())
    ref = oracle.gwd(Xs[:1500], Xt[:1700])
    assert abs(sub - ref) <= 1e-5 * ref


def test_gwd_fullsize_golden(eng):
    """The 12 500 x 4 vs 14 400 x 14 cost bench.py's GWD leg prints, against the float64 oracle value committed as a golden
    (tests/golden/make_golden_gwd_fullsize.py; budget 1e-5 relative), single solve and batched solve."""
    import json
    import os
    from conftest import GOLDEN
    sys_path = os.path.join(GOLDEN, "gwd_fullsize.json")
    g = json.load(open(sys_path))
    rng = np.random.default_rng(g["seed"])
    Xs = rng.random((g["n"], 4))
    Xt = rng.random((g["m"], 14)) * np.array([255.0] * 12 + [1.0, 1.0])
    c = float(eng.gwd_padded_l1(Xs, Xt).item())
    assert abs(c - g["cost_f64"]) <= 1e-5 * g["cost_f64"], (c, g["cost_f64"])
    xs, xt = torch.from_numpy(Xs).cuda(), torch.from_numpy(Xt).cuda()
    one = torch.ones(3, dtype=torch.int64, device="cuda")
    cb = eng.gwd_padded_l1_batch(xs, one * g["n"], xt, one * g["m"], g["n"], g["m"], xs_row=one * 0, xt_row=one * 0)
    assert all(abs(float(v) - g["cost_f64"]) <= 1e-5 * g["cost_f64"] for v in cb.cpu())


def test_bin_build_pipeline_matches_serial(eng):
    """bin(k+1) overlapped with build(k) on a second stream gives the same tensors as the serial path."""
    H, W, N, B = 120, 160, 6000, 4
    batches = [eng.EventBatch.from_numpy([make_events(N, W, H, seed=50 * j + i) for i in range(B)], H, W) for j in range(2)]
    serial = [b.rebin().optimized().clone() for b in batches]
    outs = [torch.empty((B, H, W, 12), dtype=torch.float64, device="cuda") for _ in range(2)]
    pipe = eng.BinBuildPipeline("cuda:0")
    for k in range(6):
        j = k % 2
        pipe.submit(batches[j], lambda b, j=j: b.optimized(out=outs[j]))
    pipe.drain()
    torch.cuda.synchronize()
    for j in range(2):
        assert torch.equal(outs[j], serial[j])


def test_bin_build_pipeline_refilled_and_resubmitted_batch(eng):
    """The stream-of-batches use: two resident batches alternate and the caller REFILLS a batch's events on its
    own stream once that batch's build has been issued.  The side-stream binning must see the refilled events
    (it is ordered behind the caller's stream) and must not start while a build of the same workspace is
    outstanding -- also when the same batch is submitted twice in a row."""
    H, W, N, B, G = 240, 320, 40000, 4, 6
    streams = [np.concatenate([make_events(N, W, H, seed=900 + 10 * g + i) for i in range(B)]) for g in range(G)]
    dev = [torch.from_numpy(s).cuda() for s in streams]
    mk = lambda g: eng.EventBatch.from_numpy([streams[g][i * N:(i + 1) * N] for i in range(B)], H, W)  # noqa: E731
    batches = [mk(0), mk(1)]
    expect = []
    for g in range(G):
        batches[g % 2].events.copy_(dev[g])
        expect.append(batches[g % 2].rebin().optimized().clone())
    batches[0].events.copy_(dev[0])
    batches[1].events.copy_(dev[1])
    torch.cuda.synchronize()
    pipe = eng.BinBuildPipeline("cuda:0")
    outs = [torch.empty((B, H, W, 12), dtype=torch.float64, device="cuda") for _ in range(G)]
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    for g in range(G):
        j = g % 2
        if g >= 2:                             # the build of generation g-2 (same buffer) was issued by submit(g-1)
            big.normal_()                      # keeps the caller's stream busy: a missing wait would race the copy
            batches[j].events.copy_(dev[g])    # refill on the caller's stream
        pipe.submit(batches[j], lambda b, g=g: b.optimized(out=outs[g]))
    pipe.drain()
    torch.cuda.synchronize()
    for g in range(G):
        assert torch.equal(outs[g], expect[g]), "generation %d" % g
    # the same batch twice in a row: the second submission builds the first before re-binning the workspace
    again = [torch.empty_like(outs[0]) for _ in range(2)]
    pipe.submit(batches[1], lambda b: b.optimized(out=again[0]))
    pipe.submit(batches[1], lambda b: b.optimized(out=again[1]))
    pipe.drain()
    torch.cuda.synchronize()
    assert torch.equal(again[0], expect[G - 1]) and torch.equal(again[1], expect[G - 1])


def test_step_is_graph_capturable(eng):
    """bin + build make no synchronising call, so the whole step records into a hipGraph and replays to the
    same tensor (what a launch-bound caller would do with many small batches)."""
    wins = [make_events(5000, 304, 240, seed=70 + i) for i in range(4)]
    eb = eng.EventBatch.from_numpy(wins, 240, 304)
    out = torch.empty((4, 240, 304, 12), dtype=torch.float64, device="cuda:0")

    def step():
        eb.rebin()
        eb.optimized(out=out)
    step()
    ref = out.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_two_plans_two_streams_two_host_threads(eng, oracle):
    """include/evrep.h: "no global state ... any number of host threads may drive any number of devices and streams".
    Two host threads, each with its own plan / workspace / HIP stream, run bin + build concurrently (ctypes drops the GIL
    inside the C calls); every result equals the oracle bit for bit.  Different sensors, so the two plans differ in every
    derived field (pass, block geometry, workspace layout)."""
    import threading
    import torch
    from event_representation_study_amd.synthetic import make_events
    jobs = [(480, 640, 50000, 11), (240, 304, 20000, 12)]
    want = [oracle.ergo12(make_events(n, w, h, seed=s), h, w) for h, w, n, s in jobs]
    errors, results = [], [None, None]

    def work(k):
        try:
            h, w, n, s = jobs[k]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                ev = make_events(n, w, h, seed=s)
                batch = eng.EventBatch.from_numpy([ev, ev], h, w)
                out = None
                for _ in range(50):
                    batch.rebin()
                    out = batch.optimized(out=out)
                stream.synchronize()
                results[k] = out.cpu().numpy()
        except Exception as e:          # surfaced below: a thread's exception must fail the test
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        for b in range(2):
            assert np.array_equal(results[k][b], want[k])


def _clustered_batch(eng, seeds, H=240, W=304, N=50000):
    from event_representation_study_amd.synthetic import GENERATORS
    wins = [GENERATORS["circle"](N, W, H, seed=s) for s in seeds]
    return wins, eng.EventBatch.from_numpy(wins, H, W)


def test_one_plan_two_workspaces_two_streams_two_host_threads(eng, oracle):
    """ABI 3: a plan is read-only after evrep_plan_init (the hot-unit list and its state live in the workspace), so ONE plan
    may drive two workspaces on two streams from two host threads.  Clustered windows: the voxel / TORE builders defer hot
    units to their hot launch on every call, which is what used to flip a bit in the caller's plan."""
    import threading
    import torch
    H, W = 240, 304
    sets = [[301, 302], [303, 304]]
    batches, wants = [], []
    for seeds in sets:
        wins, eb = _clustered_batch(eng, seeds, H, W)
        batches.append(eb)
        wants.append([(oracle.voxel(ev, H, W, 5), oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W)))
                      for ev in wins])
    plan_bytes = bytes(batches[0].plan)
    assert bytes(batches[1].plan) == plan_bytes      # same geometry and sizes: the same plan
    batches[1].plan = batches[0].plan                # ... literally the same struct
    errors, results = [], [None, None]

    def work(k):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                eb = batches[k]
                for _ in range(20):
                    eb.rebin()
                    vox = eb.voxel(5)
                    tore = eb.tore(6, frame_mode=2)
                stream.synchronize()
                results[k] = (vox.cpu().numpy(), tore.cpu().numpy())
        except Exception as e:
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert bytes(batches[0].plan) == plan_bytes      # nothing wrote through the const pointer
    for k in range(2):
        for b in range(2):
            assert np.array_equal(results[k][0][b].view(np.uint64), wants[k][b][0].view(np.uint64))
            np.testing.assert_allclose(results[k][1][b], wants[k][b][1], rtol=1e-6, atol=1e-6)


def test_graph_of_builder_calls_with_hot_units_replays(eng, oracle):
    """Two builder calls that both defer hot units, captured in ONE hipGraph: every replay leaves the workspace's hot list
    empty again (exit tickets), so replays neither accumulate items nor depend on host state."""
    H, W = 240, 304
    wins, eb = _clustered_batch(eng, [311, 312], H, W)
    vox = torch.empty((2, H, W, 5), dtype=torch.float64, device="cuda:0")
    acc = None

    def step():
        eb.voxel(5, out=vox)
        return eb.optimized(dtype=torch.float32)

    eb.rebin()
    ref32 = step().clone()
    refv = vox.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            acc = step()
    torch.cuda.synchronize()
    for _ in range(5):
        vox.zero_()
        acc.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(vox, refv) and torch.equal(acc, ref32)
    for b, ev in enumerate(wins):
        assert np.array_equal(refv[b].cpu().numpy().view(np.uint64), oracle.voxel(ev, H, W, 5).view(np.uint64))
