"""-m gpu: the reference's own Python names, re-hosted on the HIP path, against
the golden vectors produced by the reference itself."""
import os

import numpy as np
import pytest

from conftest import assert_bit_equal, load_golden

from event_representation_study_amd.synthetic import make_events, to_structured

pytestmark = pytest.mark.gpu

TORE_NAME = "<function events2ToreFeature at 0x7f0000000000>"


@pytest.mark.parametrize("enc", ["pm1", "01"])
@pytest.mark.parametrize("which", ["gen1", "gen4"])
def test_dispatcher_golden(enc, which):
    from event_representation_study_amd.representations import gen1_transforms, gen4_transforms
    g = load_golden("dispatch_80x60_n5000_" + enc)
    H, W, N = int(g["H"]), int(g["W"]), g["events"].shape[0]
    for label, name in (("mdes", "MixedDensityEventStack"), ("event_stack", "EventStack"),
                        ("tore", TORE_NAME), ("time_surface", "ToTimesurface")):
        rec = to_structured(g["events"])
        if which == "gen1":
            rep = gen1_transforms.get_item_transform(rec, name, None, H, W, N, 50000)
        else:
            rep = gen4_transforms.get_item_transform(rec, name, None, H, W, N)
        want = g["rep_" + label]
        assert rep.shape == want.shape and rep.dtype == want.dtype, label
        if label in ("mdes", "event_stack"):
            assert_bit_equal(rep, want, label)
        else:
            # north_star budget: 1e-5 relative.  TORE = log(dt+1) - log(151) in float32 cancels near dt ~ 150,
            # which turns a 1-ulp logf difference into ~6e-6 relative; the time surface agrees to 1e-12.
            np.testing.assert_allclose(rep, want, rtol=1e-5 if label == "tore" else 1e-12, atol=0)
        assert np.array_equal(rec["p"], g["p_after_" + label]), "in-place rewrite of ['p'] (%s)" % label


def test_dispatcher_name_matching():
    from event_representation_study_amd.representations import gen1_transforms
    from event_representation_study_amd.representations.tonic_compat import ToImage, ToVoxelGrid
    from event_representation_study_amd.representations.event_stack import EventStack
    from event_representation_study_amd.representations.time_surface import ToTimesurface
    from event_representation_study_amd.representations.tore import events2ToreFeature
    from event_representation_study_amd.representations.representation_search.mixed_density_event_stack import \
        MixedDensityEventStack
    H, W, N = 24, 32, 500
    shapes = {}
    for cls in (ToVoxelGrid, MixedDensityEventStack, EventStack, ToImage, events2ToreFeature, ToTimesurface):
        rec = to_structured(make_events(N, W, H, seed=3))
        rep = gen1_transforms.get_item_transform(rec, str(cls), cls, H, W, N, 50000)
        shapes[cls.__name__] = rep.shape
    assert shapes["ToVoxelGrid"] == (H, W, 12) and shapes["ToImage"] == (H, W, 2)
    assert shapes["MixedDensityEventStack"] == shapes["EventStack"] == shapes["ToTimesurface"] == (H, W, 12)
    assert shapes["events2ToreFeature"][2] == 12
    with pytest.raises(UnboundLocalError):
        gen1_transforms.get_item_transform(rec, "NoSuchRepresentation", None, H, W, N, 50000)


def test_event_stack_class():
    from event_representation_study_amd.representations.event_stack import EventStack
    g = load_golden("s_80x60_n5000_pm1")
    H, W = int(g["H"]), int(g["W"])
    rec = to_structured(g["events"])
    rec["p"] = (rec["p"] + 1) // 2
    es = EventStack(12, rec.shape[0], H, W)
    post = es.post_stack(es.pre_stack(rec, rec[-1]["t"]))
    assert post.shape == (H, W, 1, 12) and post.dtype == np.float32
    assert_bit_equal(np.ascontiguousarray(post.transpose(0, 1, 3, 2)[..., 0]), g["event_stack"])
    with pytest.raises(ValueError):
        es.pre_stack(rec, rec[0]["t"] - 1)                    # empty past half: p_t.min() of nothing (:24)
    bad = rec.copy()
    bad["x"][7] = W * H
    with pytest.raises(IndexError):
        es.pre_stack(bad, bad[-1]["t"])


def test_event_stack_future_half():
    """last_timestamp inside the window: the reversed, polarity-negated "future" half and post_stack's reversed
    level axis (event_stack.py:28-41,64-65), bit-exact against the reference's own (H, W, 2, S) output."""
    from event_representation_study_amd.representations.event_stack import EventStack
    g = load_golden("boundary")
    for tag in "abc":
        ev = g["future_%s_events" % tag]
        H, W = int(g["future_%s_H" % tag]), int(g["future_%s_W" % tag])
        es = EventStack(12, ev.shape[0], H, W)
        post = es.post_stack(es.pre_stack(to_structured(ev), int(g["future_%s_last" % tag])))
        assert post.dtype == np.float32
        assert_bit_equal(post, g["future_%s_post" % tag], "future half " + tag)


def _f8_record(g):
    rec = np.empty(g["float_rec_x"].shape[0], dtype=[("x", "<f8"), ("y", "<f8"), ("t", "<f8"), ("p", "<f8")])
    for n in "xytp":
        rec[n] = g["float_rec_" + n]
    return rec


def test_float_fields_are_truncated_like_the_reference():
    """n_imagenet hands every builder all-'<f8' fields with non-integral x, y, t (imagenet.py:1002-1006); the
    reference truncates them (mixed_density_event_stack.py:26-29, event_stack.py:16-19) -- so do the mirrors."""
    from event_representation_study_amd.representations.event_stack import EventStack
    from event_representation_study_amd.representations.optimized_representation import get_optimized_representation
    from event_representation_study_amd.representations.representation_search.mixed_density_event_stack import \
        MixedDensityEventStack
    g = load_golden("boundary")
    H, W = int(g["float_H"]), int(g["float_W"])
    rec = _f8_record(g)
    N = rec.shape[0]
    assert_bit_equal(get_optimized_representation(rec.copy(), N, H, W), g["float_ergo12"], "float ergo12")
    triples = ([0, 3, 5, 1], ["timestamp", "count_neg", "polarity", "timestamp_pos"], ["mean", "sum", "variance", "max"])
    assert_bit_equal(MixedDensityEventStack(4, N, H, W, triples, "SBN").stack(rec.copy()), g["float_mdes"], "float mdes")
    r2 = rec.copy()
    r2["p"] = (r2["p"] + 1) // 2
    es = EventStack(12, N, H, W)
    assert_bit_equal(es.post_stack(es.pre_stack(r2, r2[-1]["t"])), g["float_event_stack"], "float event stack")
    # absolute int64 timestamps (above 2^31): MDES only sees t - t.min()
    big = np.empty(N, dtype=[("x", "<i4"), ("y", "<i4"), ("t", "<i8"), ("p", "<i4")])
    big["x"], big["y"], big["p"] = np.trunc(rec["x"]), np.trunc(rec["y"]), rec["p"]
    ev = make_events(N, W, H, seed=701)
    big["x"], big["y"], big["p"] = ev[:, 0], ev[:, 1], ev[:, 3]
    big["t"] = g["abs_t"]
    assert_bit_equal(get_optimized_representation(big, N, H, W), g["abs_ergo12"], "absolute int64 t")


def test_to_timesurface_class():
    from event_representation_study_amd.representations.time_surface import ToTimesurface
    g = load_golden("c1_304x240_n10000_pm1")
    H, W = int(g["H"]), int(g["W"])
    rec = to_structured(g["events"])
    rec["p"] = ((rec["p"] + 1) / 2).astype(np.int8)
    ts = ToTimesurface(sensor_size=(W, H, 2), surface_dimensions=None, tau=50000, decay="exp")
    assert "ToTimesurface" in str(ToTimesurface)
    rep = ts(rec, g["ts_idx"])
    assert rep.shape == (6, 2, H, W) and rep.dtype == np.float64
    got = rep.reshape((-1, H, W)).transpose(1, 2, 0)
    np.testing.assert_allclose(got, g["time_surface"], rtol=1e-12)
    # arbitrary indices, including a repeated one: everything from the repeat on stays zero
    rep2 = ts(rec, [100, 2000, 2000, 5000])
    assert rep2[:2].all() and not rep2[2:].any()


def test_tore_function():
    from event_representation_study_amd.representations.tore import events2ToreFeature
    g = load_golden("s_80x60_n4097_01")
    ev = g["events"]
    x, y = ev[:, 0] - ev[:, 0].min() + 1, ev[:, 1] - ev[:, 1].min() + 1
    rep = events2ToreFeature(x, y, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (y.max(), x.max()))
    assert rep.shape == g["tore"].shape and rep.dtype == np.float32
    np.testing.assert_allclose(rep, g["tore"], rtol=1e-6, atol=1e-6)
    # an explicit earlier sample time
    import oracle
    T = int(ev[3000, 2])
    rep_t = events2ToreFeature(x, y, ev[:, 2], ev[:, 3], T, 3, (y.max(), x.max()))
    np.testing.assert_allclose(rep_t, oracle.tore(x, y, ev[:, 2], ev[:, 3], T, 3, (y.max(), x.max())), rtol=1e-6, atol=1e-6)


def test_gen1_container_to_builder(oracle=None):
    """Windows cut from the reference's Gen1 container layout feed the batched builder (bit-exact against the oracle)."""
    import oracle as orc
    from event_representation_study_amd.engine import EventBatch
    from event_representation_study_amd.gen1_h5 import Gen1H5Events
    d = Gen1H5Events(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5", "gen1_layout.h5"), num_events=3000)
    wins = d.windows(range(0, len(d), 3))
    for w in wins:
        w[:, 3] = 2 * w[:, 3] - 1                                    # p in {0, 1} -> {-1, +1}, as the dispatcher's callers map it
    got = EventBatch.from_numpy(wins, d.height, d.width).optimized().cpu().numpy()
    for b, w in enumerate(wins):
        assert_bit_equal(got[b], orc.ergo12(w, d.height, d.width), "window %d" % b)


def test_tore_coordinates_below_one_wrap_like_numpy():
    """1-based x, y < 1 are negative numpy indices in the reference (tore.py:25,41: [i - 1, j - 1]) and wrap to the far side of
    the frame; golden from the reference's own function (tests/golden/make_golden_sbt.py).  Beyond the frame: IndexError."""
    from event_representation_study_amd.representations.tore import events2ToreFeature
    g = load_golden("tore_wrap_20x15_n600")
    rep = events2ToreFeature(g["x"], g["y"], g["ts"], g["pol"], int(g["sample_time"]), int(g["k"]), (int(g["H"]), int(g["W"])))
    assert rep.shape == g["tore"].shape and rep.dtype == np.float32
    np.testing.assert_allclose(rep, g["tore"], rtol=1e-6, atol=1e-6)
    with pytest.raises(IndexError):
        events2ToreFeature(np.array([-20, 3]), np.array([1, 2]), np.array([0, 5]), np.array([1, -1]), 6, 3, (15, 20))


def test_optimized_and_mdes_classes():
    from event_representation_study_amd.representations.optimized_representation import get_optimized_representation
    from event_representation_study_amd.representations.representation_search.mixed_density_event_stack import \
        MixedDensityEventStack
    g = load_golden("c1_304x240_n10000_01")
    H, W = int(g["H"]), int(g["W"])
    rec = to_structured(g["events"])
    assert_bit_equal(get_optimized_representation(rec, rec.shape[0], H, W), g["ergo12"])
    a = load_golden("mdes_all_triples_40x30_n3001_pm1")
    triples = (list(a["windows"]), [str(s) for s in a["funcs"]], [str(s) for s in a["aggs"]])
    m = MixedDensityEventStack(len(triples[0]), 3001, int(a["H"]), int(a["W"]), triples, "SBN")
    assert_bit_equal(m.stack(to_structured(a["events"])), a["rep"])
    t = load_golden("sbt_shortspan_40x30_n2000_pm1")                      # stacking_type "SBT": windows cut by time
    triples = (list(t["windows"]), [str(s) for s in t["funcs"]], [str(s) for s in t["aggs"]])
    m = MixedDensityEventStack(len(triples[0]), 2000, int(t["H"]), int(t["W"]), triples, "SBT")
    assert_bit_equal(m.stack(to_structured(t["events"])), t["rep"])
    z = load_golden("mdes_none_channels_40x30_n999")
    m = MixedDensityEventStack(4, 999, 30, 40, ([0, None, 6, None], ["count", None, "polarity", None],
                                                  ["sum", None, "sum", None]), "SBN")
    assert_bit_equal(m.stack(to_structured(z["events"])), z["rep"])
    with pytest.raises(ValueError):
        get_optimized_representation(to_structured(np.zeros((0, 4), np.int32)), 0, H, W)


def test_compute_repr():
    from event_representation_study_amd.representations.representation_search.gromov_wasserstein import \
        compute_repr_from_events
    g = load_golden("c1_304x240_n10000_01")                   # BASELINE.json configs[0]
    assert_bit_equal(compute_repr_from_events(g["events"], int(g["W"]), int(g["H"]), bins=5), g["voxel5"])


def test_otmi_and_OTMI():
    import torch
    from event_representation_study_amd.representations.representation_search.compute_otmi import OTMI, otmi
    g = load_golden("gwd")
    for tag in "abc":
        T, cost = OTMI(g[tag + "_Xs"], g[tag + "_Xt"], h=0.7, reg=0.05).solve()
        want = float(g[tag + "_cost"])
        assert abs(cost - want) <= 1e-5 * want, (tag, cost, want)          # north_star tolerance: 1e-5 rel
        assert T.shape == (g[tag + "_Xs"].shape[0], g[tag + "_Xt"].shape[0])
        assert abs(float(T[0, 0]) - float(g[tag + "_T00"])) < 1e-18
    cost = otmi(torch.from_numpy(g["otmi_events"].copy()), g["otmi_rep"], int(g["otmi_H"]), int(g["otmi_W"]),
                int(g["otmi_S"]))
    assert abs(cost - float(g["otmi_cost"])) <= 1e-5 * float(g["otmi_cost"])


def test_tonic_standins_vs_numpy_restatement():
    """ToVoxelGrid / ToImage follow tonic's published algorithms (parity unpinned: tonic is absent)."""
    from event_representation_study_amd.representations.tonic_compat import ToImage, ToVoxelGrid
    H, W, N, T = 30, 40, 3000, 12
    ev = make_events(N, W, H, seed=77, polarity="01")
    grid = ToVoxelGrid((W, H, 2), n_time_bins=T)(to_structured(ev))
    assert grid.shape == (T, 1, H, W)
    ref = np.zeros(T * H * W)
    ts = T * (ev[:, 2].astype(float) - ev[0, 2]) / (ev[-1, 2] - ev[0, 2])
    pol = np.where(ev[:, 3] == 0, -1, ev[:, 3]).astype(float)
    tis = ts.astype(int)
    dts = ts - tis
    base = ev[:, 0] + ev[:, 1] * W
    ok = tis < T
    np.add.at(ref, base[ok] + tis[ok] * W * H, (pol * (1.0 - dts))[ok])
    ok = (tis + 1) < T
    np.add.at(ref, base[ok] + (tis[ok] + 1) * W * H, (pol * dts)[ok])
    assert_bit_equal(np.ascontiguousarray(grid[:, 0]), ref.reshape(T, H, W))
    img = ToImage((W, H, 2))(to_structured(ev))
    want = np.zeros((2, H, W), np.int16)
    np.add.at(want, (ev[:, 3], ev[:, 1], ev[:, 0]), 1)
    assert img.dtype == np.int16 and np.array_equal(img, want)


def test_api_aliases():
    import torch
    from event_representation_study_amd import api
    import oracle
    H, W = 48, 64
    ev = make_events(4000, W, H, seed=12)
    t = api.OptimizedRepresentation().construct(ev, H, W)
    assert isinstance(t, torch.Tensor) and t.is_cuda and tuple(t.shape) == (H, W, 12)
    assert_bit_equal(t.cpu().numpy(), oracle.ergo12(ev, H, W))
    assert tuple(api.EventStack().construct(ev, H, W).shape) == (H, W, 12)
    assert tuple(api.TimeSurface().construct(ev, H, W).shape) == (H, W, 12)
    assert tuple(api.ToRE().construct(ev, H, W).shape) == (H, W, 12)
    assert_bit_equal(api.VoxelGrid(5).construct(ev, H, W).cpu().numpy(), oracle.voxel(ev, H, W, 5))
    rng = np.random.default_rng(1)
    a, b = rng.random((200, 4)), rng.random((150, 6))
    assert abs(api.gwd_point_clouds(a, b) - oracle.gwd(a, b)) <= 1e-5 * oracle.gwd(a, b)


def test_evlicious_voxel_grid():
    """ev-licious events_to_voxel_grid (numpy variant, integer pixels)."""
    from event_representation_study_amd.evlicious_tools import events_to_voxel_grid
    g = load_golden("evlicious_voxel")

    class Events:
        pass

    for tag in "ab":
        ev = g[tag + "_events"]
        e = Events()
        e.x, e.y, e.t, e.p = ev[:, 0].astype(np.uint16), ev[:, 1].astype(np.uint16), ev[:, 2].astype(np.int64), ev[:, 3].astype(np.int8)
        e.width, e.height = int(g[tag + "_W"]), int(g[tag + "_H"])
        for bins in (5, 12):
            got = events_to_voxel_grid(e, bins, normalize=False)
            assert_bit_equal(got, g["%s_raw%d" % (tag, bins)], "%s raw %d" % (tag, bins))
        got = events_to_voxel_grid(e, 5, normalize=True)
        np.testing.assert_allclose(got, g[tag + "_norm5"], rtol=1e-5, atol=1e-6)   # float32 mean / std
        assert np.array_equal(got == 0, g[tag + "_norm5"] == 0)
    # explicit t0_us / t1_us (utils.py:60-63), absolute int64 timestamps
    g = load_golden("boundary")
    ev = g["evl_events"]
    e = Events()
    e.x, e.y, e.t, e.p = ev[:, 0].astype(np.uint16), ev[:, 1].astype(np.uint16), g["evl_t_abs"], ev[:, 3].astype(np.int8)
    e.width, e.height = int(g["evl_W"]), int(g["evl_H"])
    for k, (t0, t1) in enumerate(g["evl_ranges"]):
        got = events_to_voxel_grid(e, 5, normalize=False, t0_us=int(t0), t1_us=int(t1))
        assert_bit_equal(got, g["evl_raw5_%d" % k], "t range %d" % k)
    assert_bit_equal(events_to_voxel_grid(e, 5, normalize=False, t0_us=1_010_000), g["evl_raw5_t0only"], "t0 only")
    got = events_to_voxel_grid(e, 5, normalize=True, t0_us=int(g["evl_ranges"][0][0]), t1_us=int(g["evl_ranges"][0][1]))
    np.testing.assert_allclose(got, g["evl_norm5_0"], rtol=1e-5, atol=1e-6)


def test_evlicious_voxel_grid_subpixel():
    """ev-licious events_to_voxel_grid on sub-pixel coordinates (Events.divider > 1): bit-exact against the
    reference's own float32 grids (goldens), float32 and float64 positions, explicit time range, and integer-valued
    coordinates in a non-uint16 dtype (which take the bilinear path in the reference too)."""
    from event_representation_study_amd.evlicious_tools import events_to_voxel_grid
    g = load_golden("boundary")

    class Events:
        pass

    for tag in ("sp_a", "sp_b"):
        e = Events()
        e.x, e.y, e.t, e.p = g[tag + "_x"], g[tag + "_y"], g[tag + "_t"], g[tag + "_p"]
        e.width, e.height = int(g[tag + "_W"]), int(g[tag + "_H"])
        assert_bit_equal(events_to_voxel_grid(e, 5, normalize=False), g[tag + "_raw5"], tag + " raw5")
        assert_bit_equal(events_to_voxel_grid(e, 3, normalize=False, t0_us=5000, t1_us=40000), g[tag + "_raw3_range"], tag + " range")
        got = events_to_voxel_grid(e, 5, normalize=True)
        np.testing.assert_allclose(got, g[tag + "_norm5"], rtol=1e-5, atol=1e-6)
    ev = g["sp_int_events"]
    e = Events()
    e.x, e.y, e.t, e.p = ev[:, 0].astype(np.int32), ev[:, 1].astype(np.int32), ev[:, 2].astype(np.int64), ev[:, 3].astype(np.int8)
    e.width, e.height = 80, 60
    assert_bit_equal(events_to_voxel_grid(e, 5, normalize=False), g["sp_int_raw5"], "int32 coordinates")


def test_tore_float_coordinates_and_float_seconds():
    """events2ToreFeature as n_imagenet drives it (imagenet.py:1080-1107): float coordinates (the reference's except
    branch truncates them) and timestamps in float seconds; sample time = the last timestamp, or mid-window."""
    from event_representation_study_amd.representations.tore import events2ToreFeature
    g = load_golden("boundary")
    for tag in ("tf_a", "tf_b"):
        x, y, t, p = g[tag + "_x"], g[tag + "_y"], g[tag + "_t"], g[tag + "_p"]
        H, W = int(g[tag + "_H"]), int(g[tag + "_W"])
        x1, y1 = x - min(x) + 1, y - min(y) + 1
        got = events2ToreFeature(x1, y1, t, p, t[-1], 6, (H, W))
        assert got.dtype == np.float32 and got.shape == g[tag + "_tore"].shape
        np.testing.assert_allclose(got, g[tag + "_tore"], rtol=1e-6, atol=1e-6)      # device logf vs libm
        assert np.array_equal(got == got.max(), g[tag + "_tore"] == g[tag + "_tore"].max())   # same empty-FIFO pattern
        got = events2ToreFeature(x1, y1, t, p, float(t[len(t) // 2]) + 1e-9, 4, (H, W))
        np.testing.assert_allclose(got, g[tag + "_tore_mid"], rtol=1e-6, atol=1e-6)
    # through n_imagenet's wrapper: (N, 4) float tensor rows [x, y, t, p]
    import torch
    from event_representation_study_amd import n_imagenet_acc as ni
    x, y, t, p = g["tf_a_x"], g["tf_a_y"], g["tf_a_t"], g["tf_a_p"]
    rep = ni.reshape_then_tore(torch.from_numpy(np.stack([x, y, t, p], 1)), height=int(g["tf_a_H"]), width=int(g["tf_a_W"]))
    np.testing.assert_allclose(rep.numpy(), g["tf_a_tore"].transpose(2, 0, 1), rtol=1e-6, atol=1e-6)


def test_gwd_caller_pipeline_f1():
    """keep-ratio area resize + letterbox(114) + otmi -> C_p (resize restated from
    OpenCV's published algorithm; parity unpinned: cv2 is absent)."""
    import torch
    from event_representation_study_amd import gwd_pipeline as gp
    from event_representation_study_amd.engine import EventBatch
    from event_representation_study_amd.representations.representation_search.compute_otmi import otmi
    # 1. INTER_AREA weights = exact box integration of the piecewise-constant source
    rng = np.random.default_rng(0)
    for src, dst in ((304, 240), (240, 189), (33, 10)):
        img = rng.random((src, 3))
        scale = src / dst
        want = np.zeros((dst, 3))
        for d in range(dst):
            lo, hi = d * scale, min((d + 1) * scale, src)
            acc = np.zeros(3)
            for sx in range(int(np.floor(lo)), int(np.ceil(hi))):
                acc += img[sx] * (min(hi, sx + 1) - max(lo, sx))
            want[d] = acc / (hi - lo)
        got = gp.area_weights(src, dst) @ img
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=2e-3 * 0)   # identical up to the 1e-3 edge rule
    # 2. shapes and padding of the Gen1 case: 240x304 -> 189x240 -> 240x240 with 25 / 26 rows of 114
    H, W = 240, 304
    ev = make_events(12000, W, H, seed=8)
    rep = EventBatch.from_numpy(ev, H, W).optimized(scale=255.0)[0]
    small = gp.resize_image(rep, 240)
    assert tuple(small.shape) == (189, 240, 12)
    np.testing.assert_allclose(float(small.sum()) / (189 * 240), float(rep.sum()) / (H * W), rtol=2e-3)  # area-preserving
    lb = gp.rep_for_gwd(rep, 240)
    assert tuple(lb.shape) == (240, 240, 12)
    assert torch.all(lb[:25] == 114) and torch.all(lb[-26:] == 114) and torch.equal(lb[25:214], small)
    # 3. C_p = mean over windows of otmi(events, letterboxed rep)
    wins = [make_events(6000, W, H, seed=20 + i) for i in range(2)]
    build = lambda e: EventBatch.from_numpy(e, H, W).optimized(scale=255.0)[0]
    cp, scores = gp.measure_cp(wins, build, H, W, 240)
    manual = [otmi(torch.from_numpy(w), gp.rep_for_gwd(build(w), 240).cpu().numpy(), H, W, 240) for w in wins]
    assert abs(cp - float(np.mean(manual))) < 1e-12 and 0.0 < cp < 1.0


def test_resize_taps_kernel_equals_dense_weight_matrices():
    """evrep_resize_taps (one pass, a few taps per output) == the dense (dst x src) weight matrices it was cut
    from, applied as float64 einsum (what round 1 shipped): area and linear, shrink and enlarge, odd sizes."""
    import torch
    from event_representation_study_amd import gwd_pipeline as gp
    rng = np.random.default_rng(4)
    for (H, W, C, nh, nw, mode) in ((90, 160, 12, 80, 80, "area"), (720, 1280, 3, 640, 640, "area"), (37, 53, 5, 64, 71, "linear"),
                                    (240, 304, 12, 189, 240, "area"), (48, 64, 2, 48, 64, "area")):
        rep = torch.from_numpy(rng.random((2, H, W, C)) * 255.0).cuda()
        fn = gp.area_weights if mode == "area" else gp.linear_weights
        wy, wx = torch.from_numpy(fn(H, nh)).cuda(), torch.from_numpy(fn(W, nw)).cuda()
        want = torch.einsum("yi,bijc,xj->byxc", wy, rep, wx)
        got = gp.resize_batch(rep, nh, nw, mode)
        assert got.dtype == torch.float64 and tuple(got.shape) == (2, nh, nw, C)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-13, atol=1e-12)
        got32 = gp.resize_batch(rep.to(torch.float32), nh, nw, mode, out_dtype=torch.float32)
        np.testing.assert_allclose(got32.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-4)


def test_precompute_pipeline_f2(tmp_path):
    """SURVEY 8 row F2 / BASELINE config 5: windows -> representation -> forced (S,S) resize -> one HDF5 file per
    sample holding the float32 dataset "repr" (precompute_reps.py:432-435), read back here with the h5lite reader."""
    import torch
    from event_representation_study_amd import gwd_pipeline as gp, h5lite
    from event_representation_study_amd.engine import EventBatch
    from event_representation_study_amd.precompute import RepPrecomputer
    H, W, S = 90, 160, 80
    wins = [make_events(3000, W, H, seed=70 + i) for i in range(5)]
    pc = RepPrecomputer(H, W, S, "optimized", writers=2)
    n, nbytes, el = pc.run([wins[:3], wins[3:]], str(tmp_path))
    assert n == 5 and nbytes == 5 * S * S * 12 * 4
    for i, ev in enumerate(wins):
        got = h5lite.File(str(tmp_path / ("%d.h5" % i)))["repr"][()]
        assert got.shape == (S, S, 12) and got.dtype == np.float32
        rep = EventBatch.from_numpy(ev, H, W).optimized(scale=255.0)[0]
        want = gp.resize(rep, S, S, "area").to(torch.float32).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    # every builder of the dispatcher goes through; TORE on its per-sample bounding box; no resize when r == 1
    for b in ("event_stack", "time_surface", "tore", "voxel_grid"):
        d = tmp_path / b
        n, nbytes, _ = RepPrecomputer(H, W, S, b, writers=1).run([wins[:2]], str(d))
        assert n == 2 and h5lite.File(str(d / "1.h5"))["repr"].shape == (S, S, 12)
    d = tmp_path / "same"
    RepPrecomputer(H, W, W, "optimized", writers=1).run([wins[:1]], str(d))
    same = h5lite.File(str(d / "0.h5"))["repr"][()]
    assert same.shape == (H, W, 12)                                    # `if r != 1` (precompute_reps.py:228)
    np.testing.assert_array_equal(same, EventBatch.from_numpy(wins[0], H, W).optimized(scale=255.0)[0].to(torch.float32).cpu().numpy())


def test_precompute_c5_full_size(tmp_path):
    """BASELINE config 5 at its real size: 1280x720 windows of 200 000 events -> (640, 640, 12) float32 "repr"
    files, against round 1's dense float64 weight-matrix resize of the same builder output."""
    import torch
    from event_representation_study_amd import gwd_pipeline as gp, h5lite
    from event_representation_study_amd.engine import EventBatch
    from event_representation_study_amd.precompute import RepPrecomputer
    H, W, S, N = 720, 1280, 640, 200000
    wins = [make_events(N, W, H, seed=500 + i) for i in range(3)]
    n, nbytes, el = RepPrecomputer(H, W, S, "optimized", writers=2).run([wins], str(tmp_path / "reps"))
    assert n == 3 and nbytes == 3 * S * S * 12 * 4
    rep = EventBatch.from_numpy(wins, H, W).optimized(scale=255.0)
    wy, wx = torch.from_numpy(gp.area_weights(H, S)).cuda(), torch.from_numpy(gp.area_weights(W, S)).cuda()
    want = torch.einsum("yi,bijc,xj->byxc", wy, rep, wx).to(torch.float32).cpu().numpy()
    for i in range(3):
        got = h5lite.File(str(tmp_path / "reps" / ("%d.h5" % i)))["repr"][()]
        assert got.shape == (S, S, 12) and got.dtype == np.float32
        np.testing.assert_allclose(got, want[i], rtol=1e-6, atol=1e-6)


def test_precompute_reads_event_containers_h5py_wrote(tmp_path):
    """run_h5: events come straight from an HDF5 container real h5py wrote (tests/golden/h5/, flat (n, 4) int32
    datasets under string keys as precompute_reps.py:408-409 reads them) and give the same files as run()."""
    import os
    from conftest import GOLDEN
    from event_representation_study_amd import h5lite
    from event_representation_study_amd.precompute import RepPrecomputer
    src = os.path.join(GOLDEN, "h5", "events_gen4_layout.h5")
    ev = np.load(os.path.join(GOLDEN, "h5", "expected.npz"))["g4_b"]
    H = W = 1024
    pc = RepPrecomputer(H, W, 256, "event_stack", writers=1)
    n, _, _ = pc.run_h5(src, ["chunked_nofilter"], str(tmp_path / "a"))
    assert n == 1
    order = np.argsort(ev[:, 2], kind="stable")                 # the fixture's t column is random: sort it for the builders
    del order
    pc.run([[np.ascontiguousarray(ev)]], str(tmp_path / "b"))
    np.testing.assert_array_equal(h5lite.File(str(tmp_path / "a" / "0.h5"))["repr"][()],
                                  h5lite.File(str(tmp_path / "b" / "0.h5"))["repr"][()])
    with pytest.raises(ValueError):
        pc.run_h5(src, ["moorea_2019-02-19_004_td_2257500000_2317500000_td_000012"], str(tmp_path / "c"))   # float64 rows


@pytest.mark.gpu
def test_precompute_two_shards_leave_the_files_of_one(tmp_path):
    """Config 5 data-parallel (precompute_reps.py:439-466's Pool(8) -> one process per GPU): the shares of two ranks, each
    written under the global sample numbers, are byte for byte the files a single rank leaves."""
    import os
    from event_representation_study_amd import h5lite
    from event_representation_study_amd.precompute import RepPrecomputer, shard_keys
    from event_representation_study_amd.synthetic import make_events
    H, W = 96, 128
    wins = [make_events(1500 + 37 * i, W, H, seed=400 + i) for i in range(7)]
    pc = RepPrecomputer(H, W, 64, "optimized", writers=2)
    n, _, _ = pc.run([wins[0:3], wins[3:6], wins[6:]], str(tmp_path / "one"))
    assert n == 7
    for rank in range(2):
        mine, first, stride = shard_keys(list(range(7)), rank, 2)
        k, _, _ = pc.run([[wins[i] for i in mine[j:j + 2]] for j in range(0, len(mine), 2)], str(tmp_path / "two"),
                         first_index=first, index_stride=stride)
        assert k == len(mine)
    assert sorted(os.listdir(tmp_path / "one")) == sorted(os.listdir(tmp_path / "two"))
    for i in range(7):
        np.testing.assert_array_equal(h5lite.File(str(tmp_path / "one" / ("%d.h5" % i)))["repr"][()],
                                      h5lite.File(str(tmp_path / "two" / ("%d.h5" % i)))["repr"][()])


# ------------------------------------------------------------------ F4: n_imagenet accumulators
NI_NAMES = ["acc", "acc_time", "acc_count", "acc_count_pol", "acc_count_only", "acc_all", "flat", "flat_pol",
            "acc_exp", "acc_time_pol", "acc_intensity"]


def _ni_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "nimagenet_acc.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "pos", "c224"])
def test_nimagenet_acc_against_reference_goldens(tag):
    """reshape_then_* (HIP k_polstats) against outputs of the reference's own functions: bit-exact for
    counts / flags / min-max times / normalised counts (NaN where the reference has NaN), 1e-5 for acc_exp."""
    import torch
    from event_representation_study_amd import n_imagenet_acc as ni
    g = _ni_golden()
    ev, H, W = g[tag + "_events"], int(g[tag + "_H"]), int(g[tag + "_W"])
    for name in NI_NAMES:
        key = "%s_%s" % (tag, name)
        if key not in g.files:
            continue
        got = getattr(ni, "reshape_then_" + name)(torch.from_numpy(ev.copy()), height=H, width=W)
        assert got.dtype == torch.float32 and tuple(got.shape) == g[key].shape and got.device.type == "cpu"
        if name == "acc_exp":
            np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-5, atol=0)
        else:
            np.testing.assert_array_equal(got.numpy(), g[key])


@pytest.mark.gpu
def test_nimagenet_acc_empty_and_batch(oracle):
    import torch
    from event_representation_study_amd import n_imagenet_acc as ni
    from event_representation_study_amd.synthetic import make_events
    g = _ni_golden()
    empty = torch.zeros((0, 4), dtype=torch.float64)
    for name in ("acc_count", "acc_time_pol"):
        np.testing.assert_array_equal(getattr(ni, "reshape_then_" + name)(empty, height=24, width=32).numpy(),
                                      g["empty_" + name])
    assert tuple(ni.reshape_then_acc_all(empty, height=24, width=32).shape) == (6, 224, 224)
    with pytest.raises(IndexError):
        ni.reshape_then_acc(empty, height=24, width=32)
    with pytest.raises(RuntimeError):
        ni.reshape_then_acc_count_pol(torch.tensor([[40.0, 3.0, 0.0, 1.0], [1.0, 2.0, 0.1, -1.0]], dtype=torch.float64),
                                      height=24, width=32)
    # a ragged batch against the oracle, window by window
    wins = []
    for i, n in enumerate([1500, 7, 40000]):
        e = make_events(n, 160, 120, seed=900 + i).astype(np.float64)
        e[:, 2] /= 1e6
        wins.append(e)
    for name in ("acc_all", "acc", "acc_intensity", "flat_pol"):
        out = ni.accumulate_batch(name, wins, 120, 160).cpu().numpy()
        for b, e in enumerate(wins):
            np.testing.assert_array_equal(out[b], oracle.nimagenet_acc(name, e, 120, 160))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["w1", "w2"])
def test_nimagenet_builder_wrappers(tag):
    """reshape_then_optimized / _event_stack / _tore (imagenet.py:1025-1107) against the reference's outputs."""
    import torch
    from event_representation_study_amd import n_imagenet_acc as ni
    g = _ni_golden()
    ev, H, W = g[tag + "_events"], int(g[tag + "_H"]), int(g[tag + "_W"])
    for name, exact in (("optimized", True), ("event_stack", True), ("tore", False)):
        got = getattr(ni, "reshape_then_" + name)(torch.from_numpy(ev.copy()), height=H, width=W)
        want = g["%s_%s" % (tag, name)]
        assert got.dtype == torch.float32 and tuple(got.shape) == want.shape
        if exact:
            np.testing.assert_array_equal(got.numpy(), want)
        else:
            np.testing.assert_allclose(got.numpy(), want, rtol=1e-6, atol=1e-6)
    with pytest.raises(IndexError):
        ni.reshape_then_time_surface(torch.from_numpy(ev.copy()), height=H, width=W)


# ------------------------------------------------------------------ F4: EST quantisation layer (forward)
@pytest.mark.gpu
def test_est_quantization_layer_forward(oracle):
    """est.QuantizationLayer (k_est + the exact piecewise-linear form of the value MLP) against the reference's own
    layer run on the CPU: voxel grid before the letterbox, and the letterboxed output."""
    import torch
    from event_representation_study_amd import est
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "est.npz"))
    C, H, W = (int(v) for v in g["dim"])
    weights = est.mlp_weights({k[2:]: g[k] for k in g.files if k.startswith("w_")})
    layer = est.QuantizationLayer((C, H, W), est.PiecewiseLinearKernel(weights), image_size=int(g["image_size"]))
    ev = torch.from_numpy(g["events"].copy())
    vox = layer.voxel(ev).cpu().numpy()
    want = g["voxel"]
    assert vox.shape == want.shape and vox.dtype == np.float32
    scale = np.abs(want).max()
    assert np.abs(vox - want).max() <= 1e-5 * scale, np.abs(vox - want).max() / scale
    assert np.array_equal(vox != 0, want != 0)            # exactly the touched (pixel, polarity, bin) entries
    assert np.abs(vox - oracle.est_voxel(g["events"], (C, H, W), weights)).max() <= 1e-5 * scale
    out = layer(ev).cpu().numpy()
    assert out.shape == g["output"].shape and out.dtype == np.float32
    np.testing.assert_allclose(out, g["output"], rtol=1e-5, atol=1e-5 * scale)
    assert torch.equal(ev, torch.from_numpy(g["events"]))  # the caller's tensor is left alone
    with pytest.raises(ValueError):
        bad = g["events"].copy(); bad[0, 3] = -1
        layer.voxel(torch.from_numpy(bad))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "pos"])
def test_nimagenet_dist_adj_sort(tag):
    """DiST (reshape_then_acc_adj_sort): HIP per-polarity statistics + the reference's image-space statements on the GPU."""
    import torch
    from event_representation_study_amd import n_imagenet_acc as ni
    g = _ni_golden()
    ev, H, W = g[tag + "_events"], int(g[tag + "_H"]), int(g[tag + "_W"])
    got = ni.reshape_then_acc_adj_sort(torch.from_numpy(ev.copy()), height=H, width=W).numpy()
    want = g[tag + "_acc_adj_sort"]
    assert got.shape == want.shape and got.dtype == np.float32
    # same ranks everywhere; the final `rank.float() / n_unique` differs by one float32 ulp between the CPU and GPU divisions
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6, equal_nan=True)


@pytest.mark.gpu
def test_nimagenet_acc_sort_strict_single_polarity():
    """strict=True on a stream with ONE polarity: the absent polarity is the reference's substituted single event at
    pixel (0, 0) (imagenet.py:647-652) -- an all-zero rank image and, with use_image, that one pixel set."""
    import torch
    from event_representation_study_amd import n_imagenet_acc as ni
    g = _ni_golden()
    ev, H, W = g["pos_events"], int(g["pos_H"]), int(g["pos_W"])
    sbase = dict(strict=True, denoise_image=False, denoise_sort=False)
    combos = {"t0": dict(global_time=True, neglect_polarity=True, use_image=True, quantize_sort=None),
              "t1": dict(global_time=True, neglect_polarity=False, use_image=True, quantize_sort=8),
              "t2": dict(global_time=False, neglect_polarity=False, use_image=False, quantize_sort=[4, 16]),
              "t3": dict(global_time=False, neglect_polarity=True, use_image=False, quantize_sort=None)}
    for ck, kw in combos.items():
        got = ni.reshape_then_acc_sort(torch.from_numpy(ev.copy()), height=H, width=W, **sbase, **kw)
        np.testing.assert_array_equal(got.numpy(), g["pos_acc_sort_" + ck], err_msg=ck)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_nimagenet_acc_sort(tag):
    """reshape_then_acc_sort, strict=False: latest time index per pixel from the HIP builder, for the keyword
    combinations the goldens hold -- bit-exact; plus the caller-visible side effect and the reference's errors."""
    import torch
    from event_representation_study_amd import n_imagenet_acc as ni
    g = _ni_golden()
    ev, H, W = g[tag + "_events"], int(g[tag + "_H"]), int(g[tag + "_W"])
    base = dict(strict=False, denoise_image=False, denoise_sort=False)
    combos = {"s0": dict(global_time=True, neglect_polarity=True, use_image=True, quantize_sort=None),
              "s1": dict(global_time=True, neglect_polarity=False, use_image=True, quantize_sort=8),
              "s2": dict(global_time=False, neglect_polarity=False, use_image=False, quantize_sort=[4, 16]),
              "s3": dict(global_time=False, neglect_polarity=True, use_image=False, quantize_sort=None)}
    for ck, kw in combos.items():
        t = torch.from_numpy(ev.copy())
        got = ni.reshape_then_acc_sort(t, height=H, width=W, **base, **kw)
        want = g["%s_acc_sort_%s" % (tag, ck)]
        assert got.dtype == torch.float32 and tuple(got.shape) == want.shape, (ck, got.shape, want.shape)
        np.testing.assert_array_equal(got.numpy(), want, err_msg=ck)
        # the caller's time column now holds the time index, as after the reference call
        assert float(t[:, 2].max()) >= 1.0 and bool((t[:, 2] == t[:, 2].round()).all())
    # strict=True: the dense rank of the per-pixel latest indices (goldens checked to be independent of scatter_max's
    # arg tie-break, make_golden_nimagenet.py)
    sbase = dict(strict=True, denoise_image=False, denoise_sort=False)
    for ck, kw in combos.items():
        want = g["%s_acc_sort_t%s" % (tag, ck[1:])]
        got = ni.reshape_then_acc_sort(torch.from_numpy(ev.copy()), height=H, width=W, **sbase, **kw)
        assert got.dtype == torch.float32 and tuple(got.shape) == want.shape, (ck, got.shape, want.shape)
        np.testing.assert_array_equal(got.numpy(), want, err_msg="strict " + ck)
    with pytest.raises(NameError):
        ni.reshape_then_acc_sort(torch.from_numpy(ev.copy()), height=H, width=W, global_time=True, neglect_polarity=True,
                                 use_image=False, quantize_sort=None, strict=False, denoise_image=False, denoise_sort=True)
    with pytest.raises(KeyError):
        ni.reshape_then_acc_sort(torch.from_numpy(ev.copy()), height=H, width=W)


def test_results_are_the_callers_own():
    """"outputs are fresh arrays".  A caller keeps 40 results of the same shape (a list comprehension over
    samples, a collate over a batch): none of them may change when later calls reuse the pinned pool."""
    from event_representation_study_amd.representations.optimized_representation import get_optimized_representation
    from event_representation_study_amd.representations import gen1_transforms, _common
    from event_representation_study_amd import engine as eng
    H, W, N = 60, 80, 3000
    wins = [make_events(N, W, H, seed=900 + i) for i in range(40)]
    kept = [get_optimized_representation(to_structured(w), N, H, W) for w in wins]
    want = eng.EventBatch.from_numpy(wins, H, W).optimized().cpu().numpy()
    for i in range(40):
        assert_bit_equal(kept[i], want[i], "kept result %d" % i)
    # views of a result keep its buffer out of the pool as well
    views = [get_item(w) for w in wins[:12] for get_item in (lambda w: gen1_transforms.get_item_transform(
        to_structured(w), "EventStack", None, H, W, N, 50000)[..., 3],)]
    es = eng.EventBatch.from_numpy(wins[:12], H, W).event_stack(scale=255.0).cpu().numpy()
    for i in range(12):
        assert np.array_equal(views[i], es[i][..., 3])
    # dropped results give their buffers back: the pool does not grow past its depth
    del kept, views
    for w in wins:
        get_optimized_representation(to_structured(w), N, H, W)
    pools = [len(p) for ctx in _common._CONTEXTS.values() for p in ctx.pools.values()]
    assert max(pools) <= _common.RESULT_POOL_DEPTH


def test_wrappers_from_two_host_threads():
    """Two host threads call the per-sample wrappers at the same (H, W, size): each has its own pooled context."""
    import threading
    from event_representation_study_amd.representations.optimized_representation import get_optimized_representation
    from event_representation_study_amd import engine as eng
    H, W, N = 48, 64, 2500
    wins = [make_events(N, W, H, seed=1200 + i) for i in range(24)]
    want = eng.EventBatch.from_numpy(wins, H, W).optimized().cpu().numpy()
    got, errs = [None] * 24, []

    def work(lo):
        try:
            for rep in range(3):
                for i in range(lo, lo + 12):
                    got[i] = get_optimized_representation(to_structured(wins[i]), N, H, W)
        except Exception as e:      # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=work, args=(lo,)) for lo in (0, 12)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for i in range(24):
        assert_bit_equal(got[i], want[i], "thread result %d" % i)


def test_get_item_transform_cuda_equals_host_path():
    """device_out: the same dispatch, side effects and x255, result left on the GPU (no read-back)."""
    import torch
    from event_representation_study_amd.representations import gen1_transforms, gen4_transforms
    from event_representation_study_amd.representations.tonic_compat import ToImage, ToVoxelGrid
    H, W, N = 60, 80, 4000
    for name, tr in (("MixedDensityEventStack", None), ("EventStack", None), (TORE_NAME, None), ("ToTimesurface", None),
                     (str(ToVoxelGrid), ToVoxelGrid), (str(ToImage), ToImage)):
        a, b = to_structured(make_events(N, W, H, seed=77)), to_structured(make_events(N, W, H, seed=77))
        host = gen1_transforms.get_item_transform(a, name, tr, H, W, N, 50000)
        dev = gen1_transforms.get_item_transform_cuda(b, name, tr, H, W, N, 50000)
        assert isinstance(dev, torch.Tensor) and dev.is_cuda and tuple(dev.shape) == host.shape, name
        assert np.array_equal(dev.cpu().numpy(), host), name
        assert np.array_equal(a["p"], b["p"]), name       # the same in-place rewrite of ["p"]
        dev4 = gen4_transforms.get_item_transform_cuda(to_structured(make_events(N, W, H, seed=77)), name, tr, H, W, N)
        assert torch.equal(dev4, dev), name
    # a kept device result is the caller's own too
    kept = [gen1_transforms.get_item_transform_cuda(to_structured(make_events(N, W, H, seed=s)), "EventStack", None, H, W, N, 0)
            for s in range(10)]
    again = [gen1_transforms.get_item_transform_cuda(to_structured(make_events(N, W, H, seed=s)), "EventStack", None, H, W, N, 0)
             for s in range(10)]
    assert all(torch.equal(x, y) for x, y in zip(kept, again))
