#!/usr/bin/env python3
"""Round-5 goldens, produced by IMPORTING the reference (same stand-ins as make_golden.py):
  * tore_unsorted_40x30_n3000: events2ToreFeature (representations/tore.py:6-83) on windows whose timestamps are NOT
    ascending -- directly (k = 6 and k = 3, a sample time in the middle of the window) and through the reference's own
    dispatcher (gen1_transforms.get_item_transform, "TORE": bounding-box frame, sample time ts[-1]).  The kept values' ORDER
    is what this container's numpy (recorded below) leaves in np.partition's result: numpy >= 2.0 on AVX2 / AVX-512 hosts
    sorts so short a vector (x86-simd-sort), the scalar introselect of other builds only moves the maximum to the end.
Run from anywhere:  python tests/golden/make_golden_r05.py"""
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
    W, H, N = 40, 30, 3000
    out = {"W": W, "H": H, "numpy_version": np.__version__,
           "partition_probe": np.partition(np.array([3., 1., 2., 5., 4., 0.]), 5)}   # [0 1 2 3 4 5]: the sorting path
    for enc, seed in (("pm1", 501), ("01", 502)):
        ev = make_events(N, W, H, seed=seed, polarity=enc)
        rng = np.random.default_rng(seed)
        k = rng.random(N) < 0.4                       # not ascending: 40 % of the timestamps redrawn, some beyond ts[-1]
        ev[k, 2] = rng.integers(0, 60000, size=int(k.sum()))
        ev[rng.integers(0, N, size=200), 2] = 777     # ties
        ev[0, 2] = 0
        ev[-1, 2] = 45000
        hot = rng.random(N) < 0.2                     # a busy pixel: far more than k events per polarity
        ev[hot, 0], ev[hot, 1] = 7, 5
        out["events_" + enc] = ev
        x, y, ts, pol = ev[:, 0] + 1, ev[:, 1] + 1, ev[:, 2], ev[:, 3]
        out["tore6_" + enc] = ref["tore"](x, y, ts, pol, ts[-1], 6, (H, W))
        out["tore3_mid_" + enc] = ref["tore"](x, y, ts, pol, 30000, 3, (H, W))
        rec = to_structured(ev)
        out["dispatch_" + enc] = ref["gen1"].get_item_transform(rec, "<function events2ToreFeature at 0x0>", None, H, W, N, 50000)
        out["p_after_" + enc] = rec["p"].copy()
    np.savez_compressed(os.path.join(HERE, "tore_unsorted_40x30_n3000.npz"), **out)
    print({k2: getattr(v, "shape", v) for k2, v in out.items()})


if __name__ == "__main__":
    main()
