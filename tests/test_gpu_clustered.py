"""-m gpu: non-uniform event streams (the reference's own moving-circle generator restated, and an edge-cluster model)
against the oracle, under the automatic choice of the binning pass and under every forced pass.  Uniform (x, y) is the
SURVEY 8(d) contract; real windows are edge-clustered: a handful of builder units hold a thousand records while most of the
frame is empty (hot units must be right, not just fast)."""
import os

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


@pytest.mark.parametrize("dist", ["circle", "edges"])
@pytest.mark.parametrize("shape", [(304, 240, 50000), (640, 480, 50000), (640, 480, 8000), (346, 260, 20000)])
@pytest.mark.parametrize("pass_name", list(PASSES))
def test_clustered_vs_oracle(oracle, dist, shape, pass_name):
    from event_representation_study_amd import engine as eng
    W, H, N = shape
    wins = [GENERATORS[dist](N, W, H, seed=100 + i, polarity=("pm1", "01")[i % 2]) for i in range(3)]
    eb = _batch(eng, wins, H, W, PASSES[pass_name])
    rep, es, ts = eb.optimized().cpu().numpy(), eb.event_stack().cpu().numpy(), eb.time_surface().cpu().numpy()
    rep32 = eb.optimized(dtype=__import__("torch").float32).cpu().numpy()
    vox = eb.voxel(5).cpu().numpy()
    tore = eb.tore(6, frame_mode=2).cpu().numpy()
    for b, ev in enumerate(wins):
        ref = oracle.ergo12(ev, H, W)
        assert_bit_equal(rep[b], ref, "ergo12 %s %s" % (dist, pass_name))
        assert_bit_equal(rep32[b], ref.astype(np.float32), "ergo12 f32 %s %s" % (dist, pass_name))
        assert_bit_equal(es[b], oracle.event_stack(ev, H, W), "event_stack %s %s" % (dist, pass_name))
        np.testing.assert_allclose(ts[b], oracle.time_surface(ev, H, W), rtol=1e-12)
        want = oracle.tore(ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3], ev[-1, 2], 6, (H, W))
        np.testing.assert_allclose(tore[b], want, rtol=1e-6, atol=1e-6)
        assert_bit_equal(vox[b], oracle.voxel(ev, H, W, 5), "voxel %s %s" % (dist, pass_name))


def test_one_hot_pixel_window(oracle):
    """40 % of a window on ONE pixel (a flickering pixel), the rest uniform: the longest per-pixel walk there is."""
    from event_representation_study_amd import engine as eng
    W, H, N = 640, 480, 50000
    ev = GENERATORS["uniform"](N, W, H, seed=5)
    rng = np.random.default_rng(9)
    hot = rng.random(N) < 0.4
    ev[hot, 0], ev[hot, 1] = 333, 222
    for name, flags in PASSES.items():
        eb = _batch(eng, [ev], H, W, flags)
        assert_bit_equal(eb.optimized()[0].cpu().numpy(), oracle.ergo12(ev, H, W), "ergo12 hot pixel %s" % name)
        assert_bit_equal(eb.event_stack()[0].cpu().numpy(), oracle.event_stack(ev, H, W), "event_stack hot pixel %s" % name)
        np.testing.assert_allclose(eb.time_surface()[0].cpu().numpy(), oracle.time_surface(ev, H, W), rtol=1e-12)


def test_order_free_units_with_odd_polarities_and_small_stacks():
    """EventStack / ToTimesurface units beyond the record stage are not ordered after the key-sorted pass (one atomicMax word
    per pixel / per pixel, polarity class and slice): polarity values outside {-1, 0, 1} (escaped in the 8-byte records),
    stacks too small for the survivors to fit (the ordered paths and the hot launch take over), 8 slices, float32 surfaces --
    bit for bit against the classic pass, whose builders walk an ordered stream."""
    import torch
    from event_representation_study_amd import engine as eng
    W, H, N = 304, 240, 60000
    wins = []
    for i in range(2):
        ev = GENERATORS["circle"](N, W, H, seed=40 + i, polarity="pm1")
        rng = np.random.default_rng(70 + i)
        odd = rng.random(N) < 0.1
        ev[odd, 3] = rng.choice(np.array([-2, 3, 0, 7], dtype=np.int32), size=int(odd.sum()))
        k = rng.integers(0, N, size=N // 4)            # a hot unit: a quarter of the window in 90 pixels of one row
        ev[k, 0] = rng.integers(100, 190, size=len(k)); ev[k, 1] = 77 + i
        wins.append(ev)
    ks = _batch(eng, wins, H, W, PASSES["key_sorted"])
    cl = _batch(eng, wins, H, W, PASSES["classic"])
    assert ks.plan.reserved == 2 and cl.plan.reserved in (0, 1)
    n_of = np.array([len(w) for w in wins], dtype=np.int64)
    idx = np.stack([(n_of * (s + 1)) // 9 for s in range(8)], axis=1).astype(np.int32)
    for tag, fn in {
        "stack 12": lambda eb: eb.event_stack(),
        "stack 3 (survivors do not fit the stage)": lambda eb: eb.event_stack(3),
        "stack 16 raw polarity": lambda eb: eb.event_stack(16, premap=False),
        "surface": lambda eb: eb.time_surface(),
        "surface, 8 slices, caller's cuts": lambda eb: eb.time_surface(8, tau=30000.0, indices=idx),
        "surface float32 (ordered paths)": lambda eb: eb.time_surface(dtype=torch.float32),
        "surface, raw polarity": lambda eb: eb.time_surface(premap=False),
    }.items():
        a, b = fn(ks).cpu().numpy(), fn(cl).cpu().numpy()
        if tag.startswith("surface"):   # r06: the stream's per-event exponentials vs the ordered builder's per-slice ones: an ulp or two
            np.testing.assert_allclose(a, b, rtol=1e-6 if a.dtype == np.float32 else 1e-13, atol=0, err_msg=tag)
            assert np.array_equal(a == 0, b == 0), tag
        else:
            assert_bit_equal(a, b, tag)
