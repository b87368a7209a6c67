import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events
H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
print("reserved", eb.plan.reserved)
ref = eb.optimized().clone()
ref_es = eb.event_stack().clone()
ref_tore = eb.tore(6, frame_mode=2).clone()
bad = 0
for it in range(1000):
    eb.rebin()
    o = eb.optimized()
    if not torch.equal(o, ref): bad += 1
    if it % 100 == 0:
        if not torch.equal(eb.event_stack(), ref_es): bad += 1
        if not torch.equal(eb.tore(6, frame_mode=2), ref_tore): bad += 1
print("mismatches", bad)
