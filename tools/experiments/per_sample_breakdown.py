#!/usr/bin/env python3
"""Where a per-sample device-route call (get_item_transform_cuda, 50 000 events, 640x480) spends its ~105 us: host clock between the steps
of representations/_common.py (sample_batch -> builder -> finish), medians of 200 calls."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from event_representation_study_amd.representations import _common as C  # noqa: E402
from event_representation_study_amd.synthetic import make_events, to_structured  # noqa: E402

H, W, N = 480, 640, 50000
wins = [to_structured(make_events(N, W, H, seed=40 + i, polarity="pm1")) for i in range(8)]
T = {k: [] for k in ("stage+h2d_enqueue", "builder_enqueue", "finish(sync)", "total")}
for it in range(260):
    w = wins[it % 8]
    t0 = time.perf_counter()
    sb = C.sample_batch(w, H, W, device_out=True)
    t1 = time.perf_counter()
    dev = sb.optimized(scale=255.0)
    t2 = time.perf_counter()
    out = C.finish(sb, dev, what="x")
    t3 = time.perf_counter()
    if it >= 60:
        T["stage+h2d_enqueue"].append(t1 - t0); T["builder_enqueue"].append(t2 - t1); T["finish(sync)"].append(t3 - t2); T["total"].append(t3 - t0)
for k, v in T.items():
    print("%-20s %.1f us" % (k, np.median(v) * 1e6))
# the GPU side alone: events already resident
sb = C.sample_batch(wins[0], H, W, device_out=True)
b = sb.batch
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for it in range(50):
    a.record(); b.rebin(); o = b.optimized(scale=255.0); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 1e3)
print("bin + build on the GPU (B = 1, events resident): %.1f us" % np.median(ts))
ev = wins[0].view(np.int32).reshape(-1, 4)
ctx = sb.ctx
ts = []
for it in range(200):
    t0 = time.perf_counter(); ctx.ev_pinned[:N].numpy()[...] = ev; ts.append(time.perf_counter() - t0)
print("host copy into pinned staging: %.1f us" % (np.median(ts) * 1e6))
ts = []
for it in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.buf_dev[:N + 1].copy_(ctx.buf_pinned[:N + 1], non_blocking=True); t1 = time.perf_counter(); torch.cuda.synchronize(); ts.append((t1 - t0, time.perf_counter() - t0))
print("H2D 0.8 MB: enqueue %.1f us, done after %.1f us" % (np.median([x[0] for x in ts]) * 1e6, np.median([x[1] for x in ts]) * 1e6))
