"""Workload for rocprofv3: one builder on a clustered batch.  python tools/experiments/hot_prof.py c2 circle optimized_f64 [iters]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench_sweep
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import GENERATORS
tag, dist, name = sys.argv[1], sys.argv[2], sys.argv[3]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
W, H, N, B = bench_sweep.CONFIGS[tag]
eb = EventBatch.from_numpy([GENERATORS[dist](N, W, H, seed=7000 + i) for i in range(B)], H, W)
fn = {"optimized_f64": lambda: eb.optimized(), "optimized_f32": lambda: eb.optimized(dtype=torch.float32),
      "event_stack_f32": lambda: eb.event_stack(), "time_surface_f64": lambda: eb.time_surface(), "voxel5_f64": lambda: eb.voxel(5)}[name]
for _ in range(iters):
    eb.rebin(); fn()
torch.cuda.synchronize()
