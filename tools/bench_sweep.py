#!/usr/bin/env python3
"""Per-builder timings at the reference's real Gen1 shape and at BASELINE.json configs 2 and 3 (bench.py runs a short form
of this as its `sweep` leg).

For every (builder, geometry, events/window, batch) it times the binning pass and the builder launch
separately with HIP events and prints one JSON object per line: algorithmic bytes per launch
(16 B per event + output element size x H x W x C), achieved GB/s and events/s.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from event_representation_study_amd.engine import EventBatch  # noqa: E402
from event_representation_study_amd.synthetic import GENERATORS  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


CONFIGS = {
    "gen1": (304, 240, 50000, 32),        # the reference's real Gen1 shape (gen1_2yolo.py:41-42,81-82)
    "c2": (640, 480, 50000, 32), "c2-150k": (640, 480, 150000, 16), "c2-250k": (640, 480, 250000, 8),
    "c2-dense": (640, 480, 500000, 8),
    "c3": (1280, 720, 200000, 8), "c3-1M": (1280, 720, 1000000, 4),
}
# experiments: SWEEP_SHAPES="w76:76,960,50000,32;w128:128,570,50000,32" adds geometries
for _item in filter(None, os.environ.get("SWEEP_SHAPES", "").split(";")):
    _name, _, _dims = _item.partition(":")
    CONFIGS[_name] = tuple(int(v) for v in _dims.split(","))
HBM_PEAK_GBPS = 8000.0


def split_tag(tag):
    """"c2@circle" -> ("c2", "circle"): a geometry of CONFIGS and an event distribution of synthetic.GENERATORS
    (uniform -- SURVEY 8(d)'s contract -- when none is named)."""
    base, _, dist = tag.partition("@")
    return base, (dist or "uniform")


def sweep(tags=("gen1", "c2", "c2-dense", "c3", "c3-1M"), iters=20, builders=None, device="cuda:0"):
    """Rows of the sweep as dicts (bench.py's `sweep` leg calls this with fewer iterations)."""
    rows = []
    for tag in tags:
        base, dist = split_tag(tag)
        W, H, N, B = CONFIGS[base]
        wins = [GENERATORS[dist](N, W, H, seed=7000 + i) for i in range(B)]
        eb = EventBatch.from_numpy(wins, H, W, device=device)
        t_bin = timed(lambda: eb.rebin(), iters)
        tnorm = torch.rand(eb.total, dtype=torch.float64, device=device)  # one normalised time per event
        table = {
            "optimized_f64": (lambda o: eb.optimized(out=o), 12, torch.float64),
            "optimized_f32": (lambda o: eb.optimized(dtype=torch.float32, out=o), 12, torch.float32),
            "event_stack_f32": (lambda o: eb.event_stack(out=o), 12, torch.float32),
            "time_surface_f64": (lambda o: eb.time_surface(out=o), 12, torch.float64),
            "tore_full_frame_f32": (lambda o: eb.tore(6, frame_mode=2, out=o), 12, torch.float32),
            "voxel5_f64": (lambda o: eb.voxel(5, out=o), 5, torch.float64),
            # F4: n_imagenet reshape_then_acc_all = 6 per-polarity statistics (count, latest, earliest)
            "nimagenet_acc_all_f32": (lambda o: eb.polstats(tnorm, [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2], out=o),
                                      6, torch.float32),
        }
        for name, (fn, C, dt) in table.items():
            if builders is not None and name not in builders:
                continue
            out = torch.empty((B, H, W, C), dtype=dt, device=device)   # the first allocation, as any caller gets it (r03)
            ms = timed(lambda: fn(out), iters)
            elem = out.element_size()
            alg = B * (16 * N + elem * H * W * C)
            rows.append({"config": base, "distribution": dist, "W": W, "H": H, "events_per_window": N, "batch": B, "builder": name,
                         "binning_pass": int(eb.plan.reserved), "bin_ms": round(t_bin, 4), "build_ms": round(ms, 4),
                         "algorithmic_bytes": alg, "build_GBps": round(alg / ms / 1e6, 1),
                         "build_frac_of_8TBps": round(alg / ms / 1e6 / HBM_PEAK_GBPS, 3),
                         "events_per_s_bin_plus_build": round(B * N / ((t_bin + ms) * 1e-3))})
            del out
        del eb
        torch.cuda.empty_cache()
    return rows


def main():
    tags = [a for a in sys.argv[1:] if split_tag(a)[0] in CONFIGS and split_tag(a)[1] in GENERATORS] or list(CONFIGS)
    only = [a[2:] for a in sys.argv[1:] if a.startswith("b=")] or None     # b=optimized_f64 b=event_stack_f32 ...
    for row in sweep(tags, builders=only):
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
