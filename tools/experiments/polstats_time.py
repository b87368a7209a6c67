import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
tn = torch.rand(eb.total, dtype=torch.float64, device="cuda:0")
out = torch.empty((B, H, W, 6), dtype=torch.float32, device="cuda:0")
f = lambda: eb.polstats(tn, [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2], out=out)
for _ in range(20): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200): f()
b.record(); torch.cuda.synchronize()
print(os.environ.get("EVREP_LIB_PATH", "default"), os.environ.get("EVREP_BIN_CLASSIC", ""), "polstats us/launch: %.2f" % (a.elapsed_time(b) / 200 * 1e3))
