#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: two calibration kernels with a known HBM byte count in
the builder's own access pattern class (wide coalesced streaming), then a few bench steps.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o p -- python tools/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -o p -- python tools/pmc_workload.py

tools/parse_pmc.py turns the two counter CSVs into profiles/traffic.json.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from event_representation_study_amd.engine import EventBatch  # noqa: E402
from event_representation_study_amd.synthetic import make_events  # noqa: E402

H, W, N, B = 480, 640, 50000, 32
CAL_BYTES = 1 << 30

dev = torch.device("cuda:0")
# calibration: a 1 GiB fill (write-only) and a 1 GiB -> 1 GiB copy (read + write), float32x4 streaming
a = torch.empty(CAL_BYTES // 4, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    a.fill_(1.0)
    b.copy_(a)
torch.cuda.synchronize()

wins = [make_events(N, W, H, seed=i) for i in range(B)]
batch = EventBatch.from_numpy(wins, H, W, device=dev)
out = torch.empty((B, H, W, 12), dtype=torch.float64, device=dev)
out32 = torch.empty((B, H, W, 12), dtype=torch.float32, device=dev)
for _ in range(5):
    batch.rebin()
    batch.optimized(out=out)
    batch.optimized(dtype=torch.float32, out=out32)
torch.cuda.synchronize()
print("pmc workload done")
