import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
for (W, H, N, B) in ((1280, 720, 1000000, 4), (640, 480, 500000, 8)):
    eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
    for _ in range(30): eb.rebin()
    torch.cuda.synchronize()
    print(W, H, N, B, "reserved", eb.plan.reserved)
