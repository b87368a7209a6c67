#!/usr/bin/env python3
"""When the waves of ONE k_voxel_stream (BUILDER=tore: k_tore_stream) launch start and end (experiment build: -DEVREP_TIMING, EVREP_LIB_PATH=<that .so>):
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
TORE = os.environ.get("BUILDER") == "tore"
out = torch.empty((B, H, W, 12), dtype=torch.float32, device="cuda:0") if TORE else torch.empty((B, H, W, 5), dtype=torch.float64, device="cuda:0")


def build():
    if TORE:
        eb.tore(6, frame_mode=2, out=out)
    else:
        eb.voxel(5, out=out)


nunit = B * H * ((W + 127) // 128)
assert nunit * 64 <= (eb.total + 1) * 8
idle = eb.plan.off_sorted1 + (((eb.total * 8 + 255) // 256) * 256 if eb.plan.reserved == 2 else 0)
dbg = eb.workspace[idle: idle + nunit * 64].view(torch.int64).view(nunit, 8)
for _ in range(3):
    build()
torch.cuda.synchronize()
dbg.zero_()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
build()
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
for lo, hi in ((0, 64), (64, 192), (192, 384), (384, 768), (768, 2048), (2048, 1 << 30)):
    m = (nrec >= lo) & (nrec < hi)
    if m.any():
        print("  records %4d-%-6d waves %6d  life mean %6.1f max %6.1f  start mean %6.1f max %6.1f  end max %6.1f" % (
            lo, min(hi, int(nrec.max()) + 1), m.sum(), life[m].mean(), life[m].max(), start[m].mean(), start[m].max(), end[m].max()))
if TORE:   # the longest waves' sweeps (stream_unit_records, units of more than 256 records): issue / wait for the records / process
    dr = d[ran]
    for i in np.argsort(-life)[:4]:
        print("  wave of %5d records: life %.1f us = sweep issue %.1f + wait %.1f + process %.1f (%d batches) + the rest" % (
            nrec[i], life[i], dr[i, 0] / 100.0, dr[i, 1] / 100.0, dr[i, 2] / 100.0, dr[i, 3]))
edges = np.arange(0, end.max() + 5, 5)
print("  waves alive per 5 us:", [int(((start < t + 5) & (end > t)).sum()) for t in edges[:-1]])
