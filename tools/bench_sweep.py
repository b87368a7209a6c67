#!/usr/bin/env python3
"""Per-builder timings at BASELINE.json configs 2 and 3 (not the driver's bench line; see bench.py).

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

from event_representation_study_amd.engine import EventBatch, probe_output_placement  # noqa: E402
from event_representation_study_amd.synthetic import make_events  # noqa: E402


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


def main():
    configs = [
        ("c2", 640, 480, 50000, 32), ("c2-dense", 640, 480, 500000, 8),
        ("c3", 1280, 720, 200000, 8), ("c3-1M", 1280, 720, 1000000, 4),
    ]
    for tag, W, H, N, B in configs:
        wins = [make_events(N, W, H, seed=7000 + i) for i in range(B)]
        eb = EventBatch.from_numpy(wins, H, W)
        t_bin = timed(lambda: eb.rebin(), 20)
        tnorm = torch.rand(eb.total, dtype=torch.float64, device="cuda:0")  # one normalised time per event
        builders = {
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
        for name, (fn, C, dt) in builders.items():
            # float64 outputs of ~1 GB are placement-sensitive: like bench.py, take the fastest of a few
            # candidate allocations; the narrower outputs are indifferent
            if dt == torch.float64 and B * H * W * C * 8 >= (512 << 20):
                out, _, cand = probe_output_placement((B, H, W, C), dt, candidates=12)
            else:
                out, cand = torch.empty((B, H, W, C), dtype=dt, device="cuda:0"), None
            ms = timed(lambda: fn(out), 20)
            elem = out.element_size()
            alg = B * (16 * N + elem * H * W * C)
            print(json.dumps({"config": tag, "W": W, "H": H, "events_per_window": N, "batch": B, "builder": name,
                              "bin_ms": round(t_bin, 4), "build_ms": round(ms, 4),
                              "algorithmic_bytes": alg, "build_GBps": round(alg / ms / 1e6, 1),
                              "events_per_s_bin_plus_build": round(B * N / ((t_bin + ms) * 1e-3)),
                              "placement_probe_us": [round(x, 1) for x in cand] if cand else None}))
            del out
        del eb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
