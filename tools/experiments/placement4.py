import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
outs = [torch.empty((B, H, W, 12), dtype=torch.float64, device="cuda:0") for _ in range(16)]
src = torch.empty((B, H, W, 12), dtype=torch.float64, device="cuda:0")
for o in outs:
    o32 = o.view(torch.float32).reshape(-1)[: B * H * W * 12].view(B, H, W, 12)
    print("%x  ergo64 %.1f  ergo32 %.1f  evstack %.1f  zero %.1f  copy %.1f" % (
        o.data_ptr(), t(lambda: eb.optimized(out=o)), t(lambda: eb.optimized(dtype=torch.float32, out=o32)),
        t(lambda: eb.event_stack(out=o32)), t(lambda: o.zero_()), t(lambda: o.copy_(src))))
