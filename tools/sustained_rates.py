#!/usr/bin/env python3
"""Side-by-side sustained rates on one box: the headline builder (k_mdes, 32 x 50 000 events, 640x480x12 f64), a
1 GiB fill and a 512 MiB device copy, alternated for ~0.7 s.  Run together with
tools/microbench/store_patterns5/6 in ONE gpurun call: boxes and their power state differ between calls.

    python tools/sustained_rates.py
"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
B, N, H, W = 32, 50000, 480, 640
wins = [make_events(N, W, H, seed=i) for i in range(B)]
eb = EventBatch.from_numpy(wins, H, W)
out = torch.empty((B, H, W, 12), dtype=torch.float64, device="cuda:0")
fillbuf = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0")   # 1 GiB
src = torch.empty(1 << 27, dtype=torch.float32, device="cuda:0"); dst = torch.empty_like(src)  # 512 MiB copy
eb.rebin(); eb.optimized(out=out); fillbuf.fill_(1.0); torch.cuda.synchronize()
def tm(fn, K):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K
t0 = time.perf_counter()
for rnd in range(12):
    tb = tm(lambda: eb.optimized(out=out), 300)
    tf = tm(lambda: fillbuf.fill_(2.0), 30)
    tc = tm(lambda: dst.copy_(src), 30)
    print("t=%5.2fs  k_mdes %.4f ms (%.2f TB/s)   fill 1GiB %.4f ms (%.2f TB/s)   copy 512MiB %.4f ms (%.2f TB/s r+w)" % (
        time.perf_counter() - t0, tb, 969318400 / tb / 1e9, tf, (1 << 30) / tf / 1e9, tc, (1 << 30) / tc / 1e9))
