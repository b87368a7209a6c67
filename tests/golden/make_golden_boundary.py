#!/usr/bin/env python3
"""Golden vectors for what the reference ACCEPTS at its boundary beyond the '<i4' happy path (round 2).

Run in the build container only (``/root/reference`` must exist): ``python tests/golden/make_golden_boundary.py``.
Like make_golden.py it imports the reference where it lies (same in-process stand-ins for the absent
third-party packages) and stores inputs and outputs only -> ``boundary.npz``.

* ``float_*``  -- all-'<f8' structured fields as n_imagenet hands them over
  (n_imagenet/real_cnn_model/data/imagenet.py:1002-1006): non-integral x, y (after the sensor -> image
  rescale, :104-108) and non-integral t.  The reference truncates (mixed_density_event_stack.py:26-29,
  event_stack.py:16-19): outputs of get_optimized_representation, MixedDensityEventStack.stack and
  EventStack.pre_stack/post_stack.
* ``abs_*``    -- absolute int64 timestamps above 2^31 (MDES rebases with t - t.min(), :33).
* ``future_*`` -- EventStack.pre_stack with last_timestamp inside the window: the "future" half
  (event_stack.py:28-41) and post_stack's reversed level axis (:64-65).
* ``evl_*``    -- ev-licious events_to_voxel_grid with explicit t0_us / t1_us (tools/utils.py:60-63).
* ``sp_*``     -- the same function on sub-pixel (float) coordinates: the bilinear-in-x/y draw (:86-102).
* ``tf_*``     -- events2ToreFeature on float coordinates and float-second timestamps, as n_imagenet calls it.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from event_representation_study_amd.synthetic import make_events  # noqa: E402

F8 = np.dtype([("x", "<f8"), ("y", "<f8"), ("t", "<f8"), ("p", "<f8")])


def float_record(ev, rng, jitter_xy=True, jitter_t=True):
    rec = np.empty(ev.shape[0], dtype=F8)
    rec["x"] = ev[:, 0] + (rng.random(ev.shape[0]) * 0.999 if jitter_xy else 0.0)
    rec["y"] = ev[:, 1] + (rng.random(ev.shape[0]) * 0.999 if jitter_xy else 0.0)
    t = ev[:, 2].astype(np.float64) + (rng.random(ev.shape[0]) * 0.999 if jitter_t else 0.0)
    rec["t"] = np.sort(t)
    rec["p"] = ev[:, 3]
    return rec


def main():
    ref = mg._import_reference()
    rng = np.random.default_rng(2024)
    out = {}
    with np.errstate(all="ignore"):
        # ---- float fields ------------------------------------------------------------------
        W, H, N = 80, 60, 5000
        ev = make_events(N, W, H, seed=701)
        rec = float_record(ev, rng)
        out["float_rec_x"], out["float_rec_y"], out["float_rec_t"], out["float_rec_p"] = \
            rec["x"].copy(), rec["y"].copy(), rec["t"].copy(), rec["p"].copy()
        out["float_W"], out["float_H"] = W, H
        out["float_ergo12"] = ref["opt"](rec.copy(), N, H, W)
        triples = ([0, 3, 5, 1], ["timestamp", "count_neg", "polarity", "timestamp_pos"], ["mean", "sum", "variance", "max"])
        out["float_mdes"] = ref["MDES"](4, N, H, W, triples, "SBN").stack(rec.copy())
        r2 = rec.copy()
        r2["p"] = (r2["p"] + 1) // 2                       # reshape_then_event_stack, imagenet.py:1052
        es = ref["EventStack"](12, N, H, W)
        out["float_event_stack"] = es.post_stack(es.pre_stack(r2, r2[-1]["t"]))
        # ---- absolute int64 timestamps -------------------------------------------------------
        big = np.empty(N, dtype=[("x", "<i4"), ("y", "<i4"), ("t", "<i8"), ("p", "<i4")])
        big["x"], big["y"], big["p"] = ev[:, 0], ev[:, 1], ev[:, 3]
        big["t"] = ev[:, 2].astype(np.int64) + 1_700_000_000_000_000
        out["abs_t"] = big["t"].copy()
        out["abs_ergo12"] = ref["opt"](big.copy(), N, H, W)
        assert np.array_equal(out["abs_ergo12"], ref["opt"](mg.to_structured(ev), N, H, W), equal_nan=True)
        # ---- EventStack future half ----------------------------------------------------------
        for tag, Wf, Hf, Nf, seed, frac in (("a", 80, 60, 5000, 711, 0.6), ("b", 40, 30, 4097, 712, 0.25),
                                            ("c", 16, 12, 40, 713, 0.5)):
            e = make_events(Nf, Wf, Hf, seed=seed, polarity="01")
            r = mg.to_structured(e)
            last = int(e[int(Nf * frac), 2])
            es = ref["EventStack"](12, Nf, Hf, Wf)
            post = es.post_stack(es.pre_stack(r, last))
            assert post.shape == (Hf, Wf, 2, 12), post.shape
            out["future_%s_events" % tag] = e
            out["future_%s_W" % tag], out["future_%s_H" % tag] = Wf, Hf
            out["future_%s_last" % tag] = np.int64(last)
            out["future_%s_post" % tag] = post
        # ---- ev-licious explicit t0_us / t1_us -----------------------------------------------
        evl = types.ModuleType("evlicious")
        evl.Events = object
        sys.modules["evlicious"] = evl
        spec = importlib.util.spec_from_file_location(
            "evl_utils", os.path.join(mg.REF, "ev-licious", "src", "evlicious", "tools", "utils.py"))
        evl_utils = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(evl_utils)

        class _Events:
            def __len__(self):
                return len(self.x)

        W, H, N = 80, 60, 6000
        ev = make_events(N, W, H, seed=721)
        e = _Events()
        e.x, e.y = ev[:, 0].astype(np.uint16), ev[:, 1].astype(np.uint16)
        e.t, e.p = ev[:, 2].astype(np.int64) + 1_000_000, ev[:, 3].astype(np.int8)
        e.width, e.height = W, H
        out["evl_events"], out["evl_W"], out["evl_H"] = ev, W, H
        out["evl_t_abs"] = e.t.copy()
        ranges = [(1_010_000, 1_040_000), (1_000_000, 1_050_000), (990_000, 1_020_000), (1_020_000, 1_020_000)]
        out["evl_ranges"] = np.array(ranges, dtype=np.int64)
        for k, (t0, t1) in enumerate(ranges):
            out["evl_raw5_%d" % k] = evl_utils.events_to_voxel_grid(e, 5, normalize=False, t0_us=t0, t1_us=t1)
        out["evl_norm5_0"] = evl_utils.events_to_voxel_grid(e, 5, normalize=True, t0_us=ranges[0][0], t1_us=ranges[0][1])
        out["evl_raw5_t0only"] = evl_utils.events_to_voxel_grid(e, 5, normalize=False, t0_us=1_010_000)
        # sub-pixel coordinates (divider > 1): the bilinear-in-x/y draw (:86-102), float32 and float64 positions
        for tag, Ws, Hs, Ns, seed, dt in (("sp_a", 80, 60, 6000, 731, np.float32), ("sp_b", 40, 30, 9000, 732, np.float64)):
            evs = make_events(Ns, Ws, Hs, seed=seed)
            r2 = np.random.default_rng(seed)
            s_ = _Events()
            s_.x = (evs[:, 0] + r2.random(Ns) * 0.999).astype(dt)
            s_.y = (evs[:, 1] + r2.random(Ns) * 0.999).astype(dt)
            s_.t, s_.p = evs[:, 2].astype(np.int64), evs[:, 3].astype(np.int8)
            s_.width, s_.height = Ws, Hs
            out[tag + "_x"], out[tag + "_y"], out[tag + "_t"], out[tag + "_p"] = s_.x, s_.y, s_.t, s_.p
            out[tag + "_W"], out[tag + "_H"] = Ws, Hs
            out[tag + "_raw5"] = evl_utils.events_to_voxel_grid(s_, 5, normalize=False)
            out[tag + "_norm5"] = evl_utils.events_to_voxel_grid(s_, 5, normalize=True)
            out[tag + "_raw3_range"] = evl_utils.events_to_voxel_grid(s_, 3, normalize=False, t0_us=5000, t1_us=40000)
        # integer-valued coordinates in a non-uint16 dtype go through the same bilinear path
        s_ = _Events()
        evs = make_events(3000, 80, 60, seed=733)
        s_.x, s_.y, s_.t, s_.p = evs[:, 0].astype(np.int32), evs[:, 1].astype(np.int32), evs[:, 2].astype(np.int64), evs[:, 3].astype(np.int8)
        s_.width, s_.height = 80, 60
        out["sp_int_events"] = evs
        out["sp_int_raw5"] = evl_utils.events_to_voxel_grid(s_, 5, normalize=False)
        # ---- TORE on float inputs, as n_imagenet's reshape_then_tore drives it (imagenet.py:1080-1107): float
        # coordinates (truncated by the except branch, tore.py:29-33), timestamps in float SECONDS ------------
        for tag, Wt, Ht, Nt, seed, tscale in (("tf_a", 64, 48, 4000, 741, 1e-6), ("tf_b", 40, 30, 2500, 742, 1.0)):
            evt = make_events(Nt, Wt, Ht, seed=seed, dup_last=3)
            r3 = np.random.default_rng(seed)
            xf = evt[:, 0] + r3.random(Nt) * 0.999
            yf = evt[:, 1] + r3.random(Nt) * 0.999
            tf = (evt[:, 2] + np.sort(r3.random(Nt)) * 0.5) * tscale      # non-integral, ascending
            tf[-3:] = tf[-1]
            pf = evt[:, 3].astype(np.float64)
            x1 = xf - min(xf) + 1
            y1 = yf - min(yf) + 1
            out[tag + "_x"], out[tag + "_y"], out[tag + "_t"], out[tag + "_p"] = xf, yf, tf, pf
            out[tag + "_W"], out[tag + "_H"] = Wt, Ht
            out[tag + "_tore"] = ref["tore"](x1, y1, tf, pf, tf[-1], 6, (Ht, Wt))
            out[tag + "_tore_mid"] = ref["tore"](x1, y1, tf, pf, float(tf[Nt // 2]) + 1e-9, 4, (Ht, Wt))
    np.savez_compressed(os.path.join(HERE, "boundary.npz"), **out)
    print("wrote boundary.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
