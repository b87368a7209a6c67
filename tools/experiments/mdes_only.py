"""The headline builder alone (one binning pass, five ERGO-12 float64 launches): the workload of the per-phase instruction
counts (libraries built with -DEVREP_STOP_AFTER=n, see WaveLds::mark) and of ad-hoc rocprofv3 passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events

H, W, N, B = 480, 640, 50000, 32
dt = torch.float32 if "f32" in sys.argv[1:] else torch.float64
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
eb.bin()
out = torch.empty((B, H, W, 12), dtype=dt, device="cuda:0")
for _ in range(5):
    eb.optimized(dtype=dt, out=out)
torch.cuda.synchronize()
print("done")
