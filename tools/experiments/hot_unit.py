"""Lifetime of ONE hot builder unit: a window whose events all lie in k units of n records each (everything else empty),
B = 1, so the launch time is the (hot) wave's lifetime + the empty units around it.
    python tools/experiments/hot_unit.py [W H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd import _lib

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)


def window(n_per_unit, k_units, npx=128, seed=0):
    rng = np.random.default_rng(seed)
    n = n_per_unit * k_units
    ev = np.zeros((n, 4), np.int32)
    unit = rng.integers(0, k_units, size=n)
    ev[:, 0] = rng.integers(0, npx, size=n) + 128 * (unit % 2)
    ev[:, 1] = 10 + unit // 2
    ev[:, 2] = np.sort(rng.integers(0, 50000, size=n))
    ev[:, 3] = 2 * rng.integers(0, 2, size=n) - 1
    return ev


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for k_units in (1, 64):
    for n in (32, 100, 200, 400, 800, 1600, 3200):
        ev = window(n, k_units)
        # a sparse plan (as a clustered 50k window gets): declare a large max so that N / NK stays small
        eb = EventBatch(torch.from_numpy(ev).cuda(), torch.tensor([0, len(ev)]), H, W, max_events_per_window=len(ev),
                        plan_flags=_lib.PLAN_FORCE_KEY_SORTED)
        eb.bin()
        row = "units %3d x %5d rec (pass %d)  bin %6.1f" % (k_units, n, eb.plan.reserved, timed(eb.rebin))
        for name, fn in (("ergo64", lambda: eb.optimized()), ("ergo32", lambda: eb.optimized(dtype=torch.float32)),
                         ("es", lambda: eb.event_stack()), ("ts", lambda: eb.time_surface()), ("vox", lambda: eb.voxel(5)),
                         ("tore", lambda: eb.tore(6, frame_mode=2))):
            out = fn()
            row += "  %s %7.1f" % (name, timed(fn))
        print(row, flush=True)
