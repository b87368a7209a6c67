#!/usr/bin/env python3
"""Workload for the per-builder PMC passes (profiles/rNN/pmc_builders_*): every builder of the sweep at the reference's Gen1
shape (32 windows x 50 000 events, 304x240), BASELINE config 2 (32 x 50 000, 640x480) and config 3 (8 x 200 000, 1280x720),
a few launches each, after the same 1 GiB fill / copy calibration kernels as tools/pmc_workload.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_representation_study_amd.engine import EventBatch  # noqa: E402
from event_representation_study_amd.synthetic import make_events  # noqa: E402

dev = torch.device("cuda:0")
a = torch.empty((1 << 30) // 4, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    a.fill_(1.0)
    b.copy_(a)
del a, b
# the reference's real Gen1 shape first (gen1_2yolo.py:41-42,81-82), then BASELINE configs 2 and 3
for W, H, N, B in ((304, 240, 50000, 32), (640, 480, 50000, 32), (1280, 720, 200000, 8)):
    eb = EventBatch.from_numpy([make_events(N, W, H, seed=7000 + i) for i in range(B)], H, W, device=dev)
    tn = torch.rand(eb.total, dtype=torch.float64, device=dev)
    o64 = torch.empty((B, H, W, 12), dtype=torch.float64, device=dev)
    o32 = torch.empty((B, H, W, 12), dtype=torch.float32, device=dev)
    o5 = torch.empty((B, H, W, 5), dtype=torch.float64, device=dev)
    o6 = torch.empty((B, H, W, 6), dtype=torch.float32, device=dev)
    for _ in range(4):
        eb.rebin()
        eb.optimized(out=o64)
        eb.optimized(dtype=torch.float32, out=o32)
        eb.event_stack(out=o32)
        eb.time_surface(out=o64)
        eb.tore(6, frame_mode=2, out=o32)
        eb.voxel(5, out=o5)
        eb.polstats(tn, [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2], out=o6)
    torch.cuda.synchronize()
    del eb, o64, o32, o5, o6, tn
print("pmc builders workload done")
