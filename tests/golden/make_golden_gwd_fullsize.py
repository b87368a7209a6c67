#!/usr/bin/env python3
"""Full-size GWD golden: the closed form of OTMI(Xs, Xt, h=0.7).solve()[1] (compute_otmi.py:61-93) at the
reference's problem size -- n = 12 500 event points x 4, m = 14 400 representation points x 14 -- on the clouds bench.py's
GWD leg draws (numpy default_rng(77)), evaluated in float64 by the C oracle (oracle/evrep_oracle.c, ~9 s on one core).
The oracle itself is pinned to the reference's kernels + loss on tests/golden/gwd.npz (make_golden.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def clouds(n=12500, m=14400, seed=77):
    rng = np.random.default_rng(seed)
    Xs = rng.random((n, 4))
    Xt = rng.random((m, 14)) * np.array([255.0] * 12 + [1.0, 1.0])
    return Xs, Xt


if __name__ == "__main__":
    oracle.build()
    Xs, Xt = clouds()
    out = {"n": 12500, "m": 14400, "ds": 4, "dt": 14, "seed": 77, "h": 0.7, "cost_f64": float(oracle.gwd(Xs, Xt)),
           "generator": "numpy.random.default_rng(77): Xs = random((n,4)); Xt = random((m,14)) * [255]*12 + [1,1]"}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gwd_fullsize.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)
