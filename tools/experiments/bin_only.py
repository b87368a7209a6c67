"""Time evrep_bin_events alone (HIP events over N launches) for the library EVREP_LIB_PATH points at."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
for _ in range(50): eb.rebin()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(500): eb.rebin()
b.record(); torch.cuda.synchronize()
print(os.environ.get("EVREP_LIB_PATH", "default"), "bin us/launch: %.2f" % (a.elapsed_time(b) / 500 * 1e3))
