#!/usr/bin/env python3
"""Workload for the per-builder PMC passes (profiles/rNN/pmc_builders_*): every builder of the sweep at the reference's Gen1
shape (32 windows x 50 000 events, 304x240), BASELINE config 2 (32 x 50 000, 640x480) and config 3 (8 x 200 000, 1280x720),
a few launches each, after the same 1 GiB fill / copy calibration kernels as tools/pmc_workload.py.

Arguments (r06): geometry@distribution tags of tools/bench_sweep.py (`gen1@circle c3@circle ...`) replace the three uniform
shapes, so that the clustered rows are steered by counters too; `b=<builder>` restricts the builders."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_representation_study_amd.engine import EventBatch  # noqa: E402
from event_representation_study_amd.synthetic import GENERATORS  # noqa: E402

dev = torch.device("cuda:0")
a = torch.empty((1 << 30) // 4, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    a.fill_(1.0)
    b.copy_(a)
del a, b
# the reference's real Gen1 shape first (gen1_2yolo.py:41-42,81-82), then BASELINE configs 2 and 3
SHAPES = {"gen1": (304, 240, 50000, 32), "c2": (640, 480, 50000, 32), "c3": (1280, 720, 200000, 8),
          "c2-dense": (640, 480, 500000, 8), "c3-1M": (1280, 720, 1000000, 4)}
tags = [a for a in sys.argv[1:] if not a.startswith("b=")] or ["gen1", "c2", "c3"]
only = [a[2:] for a in sys.argv[1:] if a.startswith("b=")]
for tag in tags:
    base, _, dist = tag.partition("@")
    W, H, N, B = SHAPES[base]
    gen = GENERATORS[dist or "uniform"]
    print("shape", tag, W, H, N, B, flush=True)
    eb = EventBatch.from_numpy([gen(N, W, H, seed=7000 + i) for i in range(B)], H, W, device=dev)
    tn = torch.rand(eb.total, dtype=torch.float64, device=dev)
    o64 = torch.empty((B, H, W, 12), dtype=torch.float64, device=dev)
    o32 = torch.empty((B, H, W, 12), dtype=torch.float32, device=dev)
    o5 = torch.empty((B, H, W, 5), dtype=torch.float64, device=dev)
    o6 = torch.empty((B, H, W, 6), dtype=torch.float32, device=dev)
    calls = {"optimized_f64": lambda: eb.optimized(out=o64),
             "optimized_f32": lambda: eb.optimized(dtype=torch.float32, out=o32),
             "event_stack_f32": lambda: eb.event_stack(out=o32),
             "time_surface_f64": lambda: eb.time_surface(out=o64),
             "tore_full_frame_f32": lambda: eb.tore(6, frame_mode=2, out=o32),
             "voxel5_f64": lambda: eb.voxel(5, out=o5),
             "nimagenet_acc_all_f32": lambda: eb.polstats(tn, [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2], out=o6)}
    for _ in range(4):
        eb.rebin()
        for name, fn in calls.items():
            if not only or name in only:
                fn()
    torch.cuda.synchronize()
    del eb, o64, o32, o5, o6, tn
print("pmc builders workload done")
