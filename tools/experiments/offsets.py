"""Does the builder's launch time depend on where the output tensor / workspace sits?  (same process, same box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
nel = B * H * W * 12
big = torch.empty(nel + (64 << 20) // 8, dtype=torch.float64, device="cuda:0")
print("base address mod 2MB:", big.data_ptr() % (2 << 20), "ws mod 2MB", eb.workspace.data_ptr() % (2 << 20) if hasattr(eb, "workspace") else None)
def t(out, n=300):
    for _ in range(30): eb.optimized(out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): eb.optimized(out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for off in (0, 256, 4096, 12288, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (8 << 20) + 256, 0):
    out = big[off // 8: off // 8 + nel].view(B, H, W, 12)
    print("offset %9d B: %.1f us" % (off, t(out)))
