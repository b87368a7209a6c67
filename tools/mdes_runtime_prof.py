import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
eb.bin()
out = torch.empty((B, H, W, 12), dtype=torch.float64, device="cuda")
def timed(fn, it=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
ergo = ([0, 3, 2, 6, 5, 6, 2, 5, 1, 0, 4, 1],
        ["polarity", "timestamp_neg", "count_neg", "polarity", "count_pos", "count", "timestamp_pos", "count_neg", "timestamp_neg", "timestamp_pos", "timestamp", "count"],
        ["variance", "variance", "mean", "sum", "mean", "sum", "mean", "mean", "max", "max", "max", "mean"])
print("static ERGO-12     %.3f ms" % timed(lambda: eb.optimized(out=out)))
perm = [1, 0] + list(range(2, 12))
rt = tuple([t[i] for i in perm] for t in ergo)
print("runtime 12 triples %.3f ms" % timed(lambda: eb.mdes(*rt, out=out)))
out16 = torch.empty((B, H, W, 16), dtype=torch.float64, device="cuda")
rt16 = tuple(t + t[:4] for t in ergo)
print("runtime 16 triples %.3f ms" % timed(lambda: eb.mdes(*rt16, out=out16)))
