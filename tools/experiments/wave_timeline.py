#!/usr/bin/env python3
"""When the waves of ONE k_voxel_stream launch start and end (experiment build: -DEVREP_TIMING, EVREP_LIB_PATH=<that .so>):
   SHAPE=304,240,50000,32 DIST=circle python tools/experiments/wave_timeline.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from event_representation_study_amd.engine import EventBatch  # noqa: E402
from event_representation_study_amd.synthetic import GENERATORS  # noqa: E402

W, H, N, B = (int(v) for v in os.environ.get("SHAPE", "304,240,50000,32").split(","))
DIST = os.environ.get("DIST", "circle")
eb = EventBatch.from_numpy([GENERATORS[DIST](N, W, H, seed=7000 + i) for i in range(B)], H, W)
eb.bin()
out = torch.empty((B, H, W, 5), dtype=torch.float64, device="cuda:0")
nunit = B * H * ((W + 127) // 128)
assert nunit * 64 <= (eb.total + 1) * 8
idle = eb.plan.off_sorted1 + (((eb.total * 8 + 255) // 256) * 256 if eb.plan.reserved == 2 else 0)
dbg = eb.workspace[idle: idle + nunit * 64].view(torch.int64).view(nunit, 8)
for _ in range(3):
    eb.voxel(5, out=out)
torch.cuda.synchronize()
dbg.zero_()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
eb.voxel(5, out=out)
b.record()
torch.cuda.synchronize()
d = dbg.cpu().numpy()
ran = d[:, 7] > 0
start = (d[ran, 7] - d[ran, 7].min()) / 100.0
life = d[ran, 5] / 100.0
nrec = d[ran, 6]
end = start + life
print("%s %s  build %.1f us (main + hot launch); main waves %d of %d units (others deferred)  last start %.1f  last end %.1f us" % (
    os.environ.get("SHAPE", "gen1"), DIST, a.elapsed_time(b) * 1e3, ran.sum(), nunit, start.max(), end.max()))
for lo, hi in ((0, 64), (64, 192), (192, 384), (384, 768), (768, 1 << 30)):
    m = (nrec >= lo) & (nrec < hi)
    if m.any():
        print("  records %4d-%-6d waves %6d  life mean %6.1f max %6.1f  start mean %6.1f max %6.1f  end max %6.1f" % (
            lo, min(hi, int(nrec.max()) + 1), m.sum(), life[m].mean(), life[m].max(), start[m].mean(), start[m].max(), end[m].max()))
edges = np.arange(0, end.max() + 5, 5)
print("  waves alive per 5 us:", [int(((start < t + 5) & (end > t)).sum()) for t in edges[:-1]])
