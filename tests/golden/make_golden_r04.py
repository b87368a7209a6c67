#!/usr/bin/env python3
"""Round-4 goldens, produced by IMPORTING the reference (same stand-ins as make_golden.py):
  * unsorted_80x60_n4000_{pm1,01}: a window whose timestamps are NOT ascending through the reference's own dispatcher
    (gen1_transforms.get_item_transform) -- EventStack (the past half t <= t[-1], array order: event_stack.py:23,125) and
    ToTimesurface (array-order scan, time_surface.py:66-74, cuts from numpy's searchsorted on the unsorted t_norm);
  * compute_repr_float_t_64x48: compute_repr(x, y, t, p, W, H, bins) with the CALLER's own float64 t in [0, 1]
    (gromov_wasserstein.py:72-82), unsorted, bins 5 and 9;
  * time_surface_float_t_40x30: ToTimesurface.__call__(events, indices) with non-integral float64 timestamps (seconds),
    ascending and not (time_surface.py:25-74).
Run from anywhere:  python tests/golden/make_golden_r04.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from event_representation_study_amd.synthetic import make_events, to_structured  # noqa: E402
import make_golden  # noqa: E402


def main():
    cwd = os.getcwd()
    ref = make_golden._import_reference()
    os.chdir(cwd)
    for enc, seed in (("pm1", 301), ("01", 302)):
        W, H, N = 80, 60, 4000
        ev = make_events(N, W, H, seed=seed, polarity=enc)
        rng = np.random.default_rng(seed)
        # not ascending: a third of the timestamps are redrawn, so some events lie AFTER the last one in time
        k = rng.random(N) < 0.33
        ev[k, 2] = rng.integers(0, 60000, size=int(k.sum()))
        ev[0, 2] = 0
        ev[-1, 2] = 41000
        out = {"events": ev, "W": W, "H": H}
        for label, name in (("event_stack", "EventStack"), ("time_surface", "ToTimesurface")):
            rec = to_structured(ev)
            out["rep_" + label] = ref["gen1"].get_item_transform(rec, name, None, H, W, N, 50000)
            out["p_after_" + label] = rec["p"].copy()
        np.savez_compressed(os.path.join(HERE, "unsorted_80x60_n4000_%s.npz" % enc), **out)
        print("unsorted", enc, {k2: v.shape for k2, v in out.items() if hasattr(v, "shape")})
    W, H, N = 64, 48, 6000
    rng = np.random.default_rng(77)
    x, y = rng.integers(0, W, size=N), rng.integers(0, H, size=N)
    t = rng.random(N)                      # the caller's own normalised time: any float64 in [0, 1), in no order
    t[:5] = [0.0, 1.0, 0.25, 0.5, 0.75]    # exact bin positions, and t == 1 (upper bin masked out)
    p = rng.integers(0, 2, size=N)
    out = {"x": x, "y": y, "t": t, "p": p, "W": W, "H": H}
    for bins in (5, 9):
        out["voxel%d" % bins] = ref["compute_repr"](x, y, t, p, W, H, bins=bins)
    np.savez_compressed(os.path.join(HERE, "compute_repr_float_t_64x48.npz"), **out)
    print("compute_repr", out["voxel5"].shape, float(out["voxel5"].sum()))


def float_time_surface():
    """ToTimesurface.__call__(events, indices) with NON-INTEGRAL float64 timestamps (seconds), ascending and not: the
    reference's own class (numba stubbed to the identity, as in make_golden.py)."""
    cwd = os.getcwd()
    ref = make_golden._import_reference()
    os.chdir(cwd)
    W, H, N = 40, 30, 2500
    ev = make_events(N, W, H, seed=411, polarity="01")
    rng = np.random.default_rng(411)
    out = {"W": W, "H": H, "x": ev[:, 0], "y": ev[:, 1], "p": ev[:, 3]}
    for tag in ("asc", "unsorted"):
        t = np.sort(rng.random(N) * 0.05)                 # seconds, non-integral
        if tag == "unsorted":
            k = rng.random(N) < 0.3
            t[k] = rng.random(int(k.sum())) * 0.05
        rec = np.zeros(N, dtype=[("x", "<i8"), ("y", "<i8"), ("t", "<f8"), ("p", "<i8")])
        rec["x"], rec["y"], rec["t"], rec["p"] = ev[:, 0], ev[:, 1], t, ev[:, 3]
        idx = np.array([300, 1100, 1101, 2499])
        tr = ref["ToTimesurface"](sensor_size=(W, H, 2), surface_dimensions=None, tau=0.01, decay="exp")
        out["t_" + tag] = t
        out["idx"] = idx
        out["surf_" + tag] = tr(rec, idx)
    np.savez_compressed(os.path.join(HERE, "time_surface_float_t_40x30.npz"), **out)
    print("float time surface", out["surf_asc"].shape)


if __name__ == "__main__":
    main()
    float_time_surface()
