"""Long fuzz campaign (not part of the test suite): python tools/fuzz_campaign.py <first seed> <seconds>
Random geometry / density / batch / hot pixels; every builder under the key-sorted pass (forced), the classic passes
(forced) and the pass evrep_plan_init chooses, bit for bit against each other; ERGO-12 and EventStack against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from event_representation_study_amd import engine as eng
from event_representation_study_amd.synthetic import GENERATORS, make_events
import oracle
def builders(eb):
    out = {"ergo12": eb.optimized(), "ergo12_f32": eb.optimized(dtype=torch.float32), "es": eb.event_stack(),
           "ts": eb.time_surface(), "tore": eb.tore(6, frame_mode=2), "tore1": eb.tore(5, frame_mode=1), "vox": eb.voxel(5), "vox2": eb.voxel(9, mode=2),
           "mdes": eb.mdes([0, 1, 2, 3, 4, 5, 6, 2], [0, 1, 2, 3, 4, 5, 6, 0], [0, 1, 2, 3, 0, 1, 2, 3])}
    # r05: the n_imagenet accumulators (order-free hand-over / sweeping main launch): counts, time maxima / minima, the exp channel
    ev = eb.events.cpu().numpy(); off = eb.offsets_host.numpy()
    tn = np.zeros(len(ev), np.float64)
    for b in range(len(off) - 1):
        t = ev[off[b]:off[b + 1], 2].astype(np.float64)
        if len(t) and t[-1] != t[0]: tn[off[b]:off[b + 1]] = (t - t[0]) / (t[-1] - t[0])
    out["acc"] = eb.polstats(torch.from_numpy(tn).cuda(), [1, 2, 1, 2, 1, 2, 0, 1], [0, 0, 1, 1, 2, 2, 5, 4], tau=0.3)
    return {k: v.cpu().numpy() for k, v in out.items()}
t0 = time.time(); n = 0; seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 300:
    seed = seed0 + n; n += 1
    rng = np.random.default_rng(seed)
    if os.environ.get("FUZZ_BIG"):   # the reference's sensors: the 4096-event blocks, the two-round stage of 1280x720
        W, H = [(640, 480), (1280, 720), (304, 240), (346, 260)][int(rng.integers(0, 4))]
        B = int(rng.integers(1, 4))
        dens = float(rng.choice([0.01, 0.05, 0.16, 0.25, 0.4, 1.0, 1.7]))   # the last two: dense units, the 256-record stage of the classic passes
    else:
        W = int(rng.choice([1, 7, 64, 127, 128, 129, 200, 256, 300, 640, 1000]))
        H = int(rng.integers(1, 60))
        B = int(rng.integers(1, 5))
        dens = float(rng.choice([0.01, 0.05, 0.2, 0.25, 0.5, 2.0]))
    wins = []
    for b in range(B):
        nn = max(2, int(dens * W * H * rng.uniform(0.3, 1.7)))
        # r04: a third of the windows come from the clustered generators (hot / warm builder units: the moving circle of the
        # reference's fake_events, the edge-cluster model), with hot densities
        dist = str(rng.choice(["uniform", "uniform", "circle", "edges"]))
        ev = GENERATORS[dist](nn, W, H, seed=seed * 7 + b, polarity="pm1" if rng.random() < 0.5 else "01", span_us=int(rng.choice([5, 2000, 50000])))
        if rng.random() < 0.2:   # a hot UNIT: a fifth of the window in a 100-pixel stretch of one row
            k = rng.integers(0, nn, size=nn // 5); ev[k, 0] = rng.integers(0, min(W, 100), size=len(k)) + int(rng.integers(0, max(1, W - 100))); ev[k, 1] = int(rng.integers(0, H))
        if rng.random() < 0.3:   # a hot pixel
            k = rng.integers(0, nn, size=nn // 3); ev[k, 0] = int(rng.integers(0, W)); ev[k, 1] = int(rng.integers(0, H))
        wins.append(ev)
    os.environ.pop("EVREP_BIN_CLASSIC", None); os.environ["EVREP_BIN_KEY_SORTED"] = "1"
    ks = eng.EventBatch.from_numpy(wins, H, W)
    os.environ.pop("EVREP_BIN_KEY_SORTED", None); os.environ["EVREP_BIN_CLASSIC"] = "1"
    cl = eng.EventBatch.from_numpy(wins, H, W)
    os.environ.pop("EVREP_BIN_CLASSIC", None)
    au = eng.EventBatch.from_numpy(wins, H, W)
    a, c, d = builders(ks), builders(cl), builders(au)
    for k in a:
        if k == "ts":   # r06: the stream takes one exponential per event, the ordered builder one per slice beyond its stage: an ulp or two
            for other, what in ((c, "classic"), (d, "auto(%d)" % au.plan.reserved)):
                assert np.allclose(a[k], other[k], rtol=1e-13, atol=0, equal_nan=True) and np.array_equal(a[k] == 0, other[k] == 0), \
                    (seed, k, "key-sorted vs " + what, W, H, [len(w) for w in wins])
            continue
        assert np.array_equal(a[k], c[k], equal_nan=True), (seed, k, "key-sorted vs classic", W, H, [len(w) for w in wins])
        assert np.array_equal(a[k], d[k], equal_nan=True), (seed, k, "key-sorted vs auto(%d)" % au.plan.reserved, W, H)
    for b, ev in enumerate(wins):
        if ev[-1, 2] != ev[0, 2]:
            assert np.array_equal(a["ergo12"][b], oracle.ergo12(ev, H, W), equal_nan=True), (seed, "ergo12 vs oracle", W, H, len(ev))
        assert np.array_equal(a["es"][b], oracle.event_stack(ev, H, W)), (seed, "event stack vs oracle")
    if n % 4 == 0:   # r03: the "SBT" stacking (windows cut by time) and MDES on timestamps in any order, against the oracle
        trip = ([0, 1, 2, 3, 4, 5, 6, 7, 2, 5], ["count", "timestamp", "polarity", "count_neg", "timestamp_pos", "count_pos", "timestamp_neg", "polarity", "timestamp", "count"],
                ["sum", "mean", "variance", "sum", "max", "mean", "mean", "max", "variance", "mean"])
        sbt = au.mdes(*trip, stacking="SBT").cpu().numpy()
        for b, ev in enumerate(wins):
            if ev[-1, 2] != ev[0, 2]:
                assert np.array_equal(sbt[b], oracle.mdes_sbt(ev, H, W, *trip), equal_nan=True), (seed, "SBT vs oracle", W, H, len(ev))
        shuf = [ev.copy() for ev in wins]
        for ev in shuf:
            ev[:, 2] = ev[np.random.default_rng(seed).permutation(len(ev)), 2]
        us = eng.EventBatch.from_numpy(shuf, H, W).optimized().cpu().numpy()
        for b, ev in enumerate(shuf):
            if ev[:, 2].max() != ev[:, 2].min():
                assert np.array_equal(us[b], oracle.ergo12(ev, H, W), equal_nan=True), (seed, "unsorted ergo12 vs oracle", W, H, len(ev))
        # r05: TORE in array order on the shuffled windows (every pass, and the oracle), with a few escaped polarity values thrown in
        for ev in shuf:
            k = np.random.default_rng(seed + 1).random(len(ev)) < 0.01
            ev[k, 3] = 5
        os.environ["EVREP_BIN_KEY_SORTED"] = "1"; tk = eng.EventBatch.from_numpy(shuf, H, W); os.environ.pop("EVREP_BIN_KEY_SORTED")
        os.environ["EVREP_BIN_CLASSIC"] = "1"; tc = eng.EventBatch.from_numpy(shuf, H, W); os.environ.pop("EVREP_BIN_CLASSIC")
        ta, tb = tk.tore(6, frame_mode=2).cpu().numpy(), tc.tore(6, frame_mode=2).cpu().numpy()
        assert np.array_equal(ta, tb), (seed, "unsorted tore: key-sorted vs classic", W, H)
        e32a, e32b = tk.optimized(dtype=torch.float32).cpu().numpy(), tc.optimized(dtype=torch.float32).cpu().numpy()
        assert np.array_equal(e32a, e32b, equal_nan=True), (seed, "unsorted / escaped ergo12 f32: key-sorted vs classic", W, H)
        for b, ev in enumerate(shuf):
            want = oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W))
            assert np.allclose(ta[b], want, rtol=1e-6, atol=1e-6), (seed, "unsorted tore vs oracle", W, H, len(ev))
print("fuzz campaign: %d cases ok in %.0f s" % (n, time.time() - t0))
