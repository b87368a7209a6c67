"""k_block_keysort with parts switched off (libraries built with -DKS_DEBUG=x: 1 no group repair, 2 no stage / write-out,
4 no statistics, 8 no table copy, 15 all of them): where the binning pass's ~21 us at config 2 go.
    EVREP_LIB_PATH=tools/variants/libevrep_ks<x>.so python tools/experiments/ks_phases.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events

W, H, N, B = (int(v) for v in os.environ.get("SHAPE", "640,480,50000,32").split(","))
ebs = [EventBatch.from_numpy([make_events(N, W, H, seed=j * B + i) for i in range(B)], H, W) for j in range(12)]
for eb in ebs:
    eb.rebin()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    a.record()
    for k in range(240):
        ebs[k % 12].rebin()
    b.record()
    torch.cuda.synchronize()
print(os.environ.get("EVREP_LIB_PATH", "library"), "%.2f us per binning pass (12 rotating batches)" % (a.elapsed_time(b) / 240 * 1e3))
