import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
def t(out, n=100):
    for _ in range(10): eb.optimized(out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): eb.optimized(out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
outs = [torch.empty((B, H, W, 12), dtype=torch.float64, device="cuda:0") for _ in range(24)]
for o in outs:
    p = o.data_ptr()
    print("%x  mod1G %4d MiB  mod4G %5d MiB  %.1f" % (p, (p % (1 << 30)) >> 20, (p % (1 << 32)) >> 20, t(o)))
