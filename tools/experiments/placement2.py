import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
H, W, N, B = 480, 640, 50000, 32
mode = sys.argv[1] if len(sys.argv) > 1 else "after"
nel = B * H * W * 12
if mode == "before":
    big = torch.empty(5 * nel, dtype=torch.float64, device="cuda:0")
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
if mode != "before":
    big = torch.empty(5 * nel, dtype=torch.float64, device="cuda:0")
def t(out, n=200):
    for _ in range(20): eb.optimized(out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): eb.optimized(out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print(mode, "views of one allocation:", " ".join("%.1f" % t(big[k * nel:(k + 1) * nel].view(B, H, W, 12)) for k in range(5)),
      "| ptr %x ws %x ev %x" % (big.data_ptr(), eb.workspace.data_ptr() if hasattr(eb, "workspace") else 0, eb.events.data_ptr()))
