"""-m gpu: clustered event streams at BASELINE config 3's shape (1280x720) and for EVERY builder, against the oracle, under
the automatic choice of the binning pass and under every forced pass.

tests/test_gpu_clustered.py stops at 640x480 and at the five headline builders; the 1 Mpx circle is where the narrow and
float32 builders take the spill slot, the hot launch and the quarter pieces most.  Here: 200 000 and 1 000 000 events per
window on 1280x720, and k_polstats (acc_all, acc_time_pol), k_est, TORE's bounding-box frame and the ev-licious voxel mode
beside ERGO-12 / EventStack / ToTimesurface / TORE / voxel.  The oracle's answers are computed once per stream and shared by
the passes."""
import numpy as np
import pytest

from conftest import assert_bit_equal

from event_representation_study_amd import _lib
from event_representation_study_amd.synthetic import GENERATORS

pytestmark = pytest.mark.gpu

PASSES = {"auto": None, "classic": _lib.PLAN_NO_KEY_PASS, "key_sorted": _lib.PLAN_FORCE_KEY_SORTED,
          "three_kernel": _lib.PLAN_THREE_KERNEL}


def _batch(eng, wins, H, W, flags):
    import torch
    offs = np.zeros(len(wins) + 1, dtype=np.int64)
    np.cumsum([w.shape[0] for w in wins], out=offs[1:])
    ev = torch.from_numpy(np.concatenate(wins)).cuda()
    return eng.EventBatch(ev, torch.from_numpy(offs), H, W, plan_flags=flags)


def _tnorm(ev):
    """n_imagenet's normalised time (imagenet.py:198-199) of an int32 window, float64."""
    t = ev[:, 2].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (t - t[0]) / (t[-1] - t[0])


def _ni_rows(ev):
    """[x, y, t, p] with p in {-1, +1} as float64 rows, what oracle.nimagenet_acc reads (a 0 polarity counts as negative there
    only if it is < 0: windows of the "01" encoding are mapped to -1 / +1 first, as parse_event does)."""
    e = ev.astype(np.float64)
    e[:, 3] = np.where(ev[:, 3] > 0, 1.0, -1.0)
    return e


def _ni_events(ev):
    e = ev.copy()
    e[:, 3] = np.where(ev[:, 3] > 0, 1, -1)
    return e


@pytest.mark.parametrize("dist", ["circle", "edges"])
@pytest.mark.parametrize("shape", [(1280, 720, 200000, 2), (1280, 720, 1000000, 1)])
def test_clustered_1mpx_every_builder_every_pass(oracle, dist, shape):
    import torch
    from event_representation_study_amd import engine as eng
    W, H, N, B = shape
    wins = [GENERATORS[dist](N, W, H, seed=300 + i, polarity=("pm1", "01")[i % 2]) for i in range(B)]
    ni_wins = [_ni_events(ev) for ev in wins]           # the n_imagenet accumulators read p in {-1, +1}
    tn = torch.from_numpy(np.concatenate([_tnorm(ev) for ev in wins])).cuda()
    refs = []
    for ev, nev in zip(wins, ni_wins):
        ref = oracle.ergo12(ev, H, W)
        refs.append({
            "ergo12": ref, "ergo12_f32": ref.astype(np.float32),
            "event_stack": oracle.event_stack(ev, H, W),
            "time_surface": oracle.time_surface(ev, H, W),
            "tore": oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W)),
            "tore_bbox": oracle.tore_bbox(ev, 6),
            "voxel": oracle.voxel(ev, H, W, 5),
            "evl_voxel": oracle.evl_voxel(ev, H, W, 9),
            "acc_all": oracle.nimagenet_acc("acc_all", _ni_rows(ev), H, W),
            "acc_time_pol": oracle.nimagenet_acc("acc_time_pol", _ni_rows(ev), H, W),
        })
    # the EST layer has no oracle entry point on raw windows: the passes are checked against each other (the classic pass
    # walks an ordered stream; tests/test_gpu_reference_api.py pins the kernel to the reference's layer)
    seg = torch.tensor([[-0.25, 0.5, 0.1], [0.3, -1.5, 0.6], [1e9, 0.25, -0.2]], dtype=torch.float64, device="cuda:0")
    bucket = torch.zeros(16, dtype=torch.int32, device="cuda:0")
    est_ref = None
    for pass_name, flags in PASSES.items():
        tag = "%s %dx%d N=%d %s" % (dist, W, H, N, pass_name)
        eb = _batch(eng, wins, H, W, flags)
        got = {
            "ergo12": eb.optimized(), "ergo12_f32": eb.optimized(dtype=torch.float32),
            "event_stack": eb.event_stack(), "time_surface": eb.time_surface(),
            "tore": eb.tore(6, frame_mode=2), "voxel": eb.voxel(5), "evl_voxel": eb.voxel(9, mode=2),
        }
        got = {k: v.cpu().numpy() for k, v in got.items()}
        bbox = [t.cpu().numpy() for t in eb.tore(6, frame_mode=0)]
        est = eb.est_voxel(tn.to(torch.float32), 3, seg, bucket, -1.0, 1.0).cpu().numpy()
        if est_ref is None:
            est_ref = est
        else:
            assert_bit_equal(est, est_ref, "est " + tag)
        eb_ni = _batch(eng, ni_wins, H, W, flags)
        acc_all = eb_ni.polstats(tn, [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2]).cpu().numpy()
        acc_tp = eb_ni.polstats(tn, [1, 2], [1, 1]).cpu().numpy()
        for b, r in enumerate(refs):
            for k in ("ergo12", "ergo12_f32", "event_stack", "voxel"):
                assert_bit_equal(got[k][b], r[k], "%s %s w%d" % (k, tag, b))
            # ev-licious' grid is (bins, H, W) float32 (evlicious_tools.events_to_voxel_grid casts the builder's float64 counts)
            assert_bit_equal(np.ascontiguousarray(np.moveaxis(got["evl_voxel"][b], -1, 0)).astype(np.float32), r["evl_voxel"],
                             "evl_voxel %s w%d" % (tag, b))
            np.testing.assert_allclose(got["time_surface"][b], r["time_surface"], rtol=1e-12, err_msg="ts " + tag)
            np.testing.assert_allclose(got["tore"][b], r["tore"], rtol=1e-6, atol=1e-6, err_msg="tore " + tag)
            np.testing.assert_allclose(bbox[b], r["tore_bbox"], rtol=1e-6, atol=1e-6, err_msg="tore bbox " + tag)
            np.testing.assert_array_equal(np.moveaxis(acc_all[b], -1, 0), r["acc_all"], err_msg="acc_all " + tag)
            np.testing.assert_array_equal(np.moveaxis(acc_tp[b], -1, 0), r["acc_time_pol"], err_msg="acc_time_pol " + tag)


def test_hot_units_under_every_pass_at_gen1(oracle):
    """The reference's own shape with units far beyond every LDS stage (a quarter of each window on 90 pixels of one row, and a
    flickering pixel): ERGO-12's split path with its kept records in the spill slot, the narrow builders' deferred quarters."""
    import torch
    from event_representation_study_amd import engine as eng
    W, H, N = 304, 240, 120000
    wins = []
    for i in range(2):
        ev = GENERATORS["edges"](N, W, H, seed=60 + i, polarity=("pm1", "01")[i])
        rng = np.random.default_rng(80 + i)
        k = rng.integers(0, N, size=N // 4)
        ev[k, 0] = rng.integers(100, 190, size=len(k)); ev[k, 1] = 77 + i
        k = rng.integers(0, N, size=N // 10)
        ev[k, 0], ev[k, 1] = 17, 200 + i
        wins.append(ev)
    for pass_name, flags in PASSES.items():
        eb = _batch(eng, wins, H, W, flags)
        rep, rep32 = eb.optimized().cpu().numpy(), eb.optimized(dtype=torch.float32).cpu().numpy()
        vox, tore = eb.voxel(5).cpu().numpy(), eb.tore(6, frame_mode=2).cpu().numpy()
        for b, ev in enumerate(wins):
            ref = oracle.ergo12(ev, H, W)
            assert_bit_equal(rep[b], ref, "ergo12 hot %s" % pass_name)
            assert_bit_equal(rep32[b], ref.astype(np.float32), "ergo12 f32 hot %s" % pass_name)
            assert_bit_equal(vox[b], oracle.voxel(ev, H, W, 5), "voxel hot %s" % pass_name)
            want = oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W))
            np.testing.assert_allclose(tore[b], want, rtol=1e-6, atol=1e-6)


def test_ergo12_hot_units_with_escaped_polarities(oracle):
    """Polarity values outside {-1, 0, 1} are escaped in the 8-byte records; a unit of ERGO-12's split path that meets one takes
    the ordered paths inside the same launch (no hot launch behind the float64 ERGO-12 builder): hot and warm units with a few
    such records, against the oracle and the classic pass."""
    from event_representation_study_amd import engine as eng
    W, H, N = 304, 240, 60000
    wins = []
    for i in range(2):
        ev = GENERATORS["circle"](N, W, H, seed=140 + i, polarity="pm1")
        rng = np.random.default_rng(170 + i)
        odd = rng.random(N) < 0.002
        ev[odd, 3] = rng.choice(np.array([-2, 3, 7], dtype=np.int32), size=int(odd.sum()))
        k = rng.integers(0, N, size=N // 4)            # a hot unit: a quarter of the window in 90 pixels of one row
        ev[k, 0] = rng.integers(100, 190, size=len(k)); ev[k, 1] = 77 + i
        wins.append(ev)
    ks = _batch(eng, wins, H, W, PASSES["key_sorted"])
    cl = _batch(eng, wins, H, W, PASSES["classic"])
    assert ks.plan.reserved == 2 and cl.plan.reserved in (0, 1)
    a, c = ks.optimized().cpu().numpy(), cl.optimized().cpu().numpy()
    assert_bit_equal(a, c, "ergo12 escaped polarities: key-sorted vs classic")
    for b, ev in enumerate(wins):
        assert_bit_equal(a[b], oracle.ergo12(ev, H, W), "ergo12 escaped polarities vs oracle w%d" % b)


def test_hand_over_whole_units_and_time_slices_at_gen1(oracle):
    """r05b: at the reference's own shape, sparse enough that hot units are the exception (<= 90 records per unit on average),
    the float32 ERGO-12 and the n_imagenet accumulators hand every unit beyond their record stage to the hot launch WHOLE
    (order-free sweep there); a unit of >= 4096 records is swept in time slices by several hot waves that merge through its
    spill slot (here: ~15 000 records on 90 pixels of one row, and a flickering pixel).  Window 2 holds escaped polarity values:
    its units keep the ordered ways (float32 ERGO-12), in the same launches.  Against the oracle, under every pass."""
    import torch
    from event_representation_study_amd import engine as eng
    W, H, N = 304, 240, 60000
    wins = []
    for i in range(3):
        ev = GENERATORS["circle"](N, W, H, seed=240 + i, polarity=("pm1", "01", "pm1")[i])
        rng = np.random.default_rng(250 + i)
        k = rng.integers(0, N, size=N // 4)
        ev[k, 0] = rng.integers(100, 190, size=len(k)); ev[k, 1] = 99 + i
        k = rng.integers(0, N, size=N // 12)
        ev[k, 0], ev[k, 1] = 17, 200 + i
        if i == 2:
            odd = rng.random(N) < 0.002
            ev[odd, 3] = rng.choice(np.array([-2, 3, 7], dtype=np.int32), size=int(odd.sum()))
        wins.append(ev)
    ni_wins = [_ni_events(ev) for ev in wins]
    tn = torch.from_numpy(np.concatenate([_tnorm(ev) for ev in wins])).cuda()
    refs = [(oracle.ergo12(ev, H, W), oracle.nimagenet_acc("acc_all", _ni_rows(ev), H, W),
             oracle.nimagenet_acc("acc_time_pol", _ni_rows(ev), H, W)) for ev in wins]
    # TORE (full frame and the bounding-box frame) and the accumulators with an empty-pixel background (acc_exp)
    tore_refs = [(oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W)), oracle.tore_bbox(ev, 6),
                  oracle.nimagenet_acc("acc_exp", _ni_rows(ev), H, W)) for ev in wins]
    for pass_name, flags in PASSES.items():
        eb = _batch(eng, wins, H, W, flags)
        if pass_name in ("auto", "key_sorted"):
            assert eb.plan.reserved == 2
        rep, rep32 = eb.optimized().cpu().numpy(), eb.optimized(dtype=torch.float32).cpu().numpy()
        eb_ni = _batch(eng, ni_wins, H, W, flags)
        acc_all = eb_ni.polstats(tn, [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2]).cpu().numpy()
        acc_tp = eb_ni.polstats(tn, [1, 2], [1, 1]).cpu().numpy()
        assert not any(int(s) & _lib.ST_HOT_OVERFLOW for s in eb.status())
        tore = eb.tore(6, frame_mode=2).cpu().numpy()
        bbox = [t.cpu().numpy() for t in eb.tore(6, frame_mode=0)]
        acc_exp = eb_ni.polstats(tn, [1, 2], [4, 4], tau=0.3).cpu().numpy()
        for b, (t_full, t_bbox, a_exp) in enumerate(tore_refs):
            np.testing.assert_allclose(tore[b], t_full, rtol=1e-6, atol=1e-6, err_msg="tore %s w%d" % (pass_name, b))
            np.testing.assert_allclose(bbox[b], t_bbox, rtol=1e-6, atol=1e-6, err_msg="tore bbox %s w%d" % (pass_name, b))
            np.testing.assert_allclose(np.moveaxis(acc_exp[b], -1, 0), a_exp, rtol=1e-6, atol=1e-7, err_msg="acc_exp %s w%d" % (pass_name, b))
        for b, (ref, a_all, a_tp) in enumerate(refs):
            assert_bit_equal(rep[b], ref, "ergo12 %s w%d" % (pass_name, b))
            assert_bit_equal(rep32[b], ref.astype(np.float32), "ergo12 f32 %s w%d" % (pass_name, b))
            np.testing.assert_array_equal(np.moveaxis(acc_all[b], -1, 0), a_all, err_msg="acc_all %s w%d" % (pass_name, b))
            np.testing.assert_array_equal(np.moveaxis(acc_tp[b], -1, 0), a_tp, err_msg="acc_time_pol %s w%d" % (pass_name, b))
