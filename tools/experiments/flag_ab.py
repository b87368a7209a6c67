#!/usr/bin/env python3
"""Interleaved A/B of one plan flag inside ONE process (boxes and clocks drift: separate runs differ by 10 %):
   python tools/experiments/flag_ab.py EVREP_X_MDES_NO_COOP optimized_f32 c2 c2@circle c3@circle ..."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_sweep  # noqa: E402
from event_representation_study_amd.engine import EventBatch  # noqa: E402
from event_representation_study_amd.synthetic import GENERATORS  # noqa: E402

flag, builder, tags = sys.argv[1], sys.argv[2], sys.argv[3:]
for tag in tags:
    base, dist = bench_sweep.split_tag(tag)
    W, H, N, B = bench_sweep.CONFIGS[base]
    wins = [GENERATORS[dist](N, W, H, seed=7000 + i) for i in range(B)]
    ebs = []
    for on in (False, True):
        if on:
            os.environ[flag] = "1"
        else:
            os.environ.pop(flag, None)
        ebs.append(EventBatch.from_numpy(wins, H, W))
    os.environ.pop(flag, None)
    tn = torch.rand(ebs[0].total, dtype=torch.float64, device="cuda:0")
    fns = {"optimized_f32": lambda eb, o: eb.optimized(dtype=torch.float32, out=o), "optimized_f64": lambda eb, o: eb.optimized(out=o),
           "voxel5_f64": lambda eb, o: eb.voxel(5, out=o), "tore_full_frame_f32": lambda eb, o: eb.tore(6, frame_mode=2, out=o),
           "time_surface_f64": lambda eb, o: eb.time_surface(out=o), "event_stack_f32": lambda eb, o: eb.event_stack(out=o),
           "nimagenet_acc_all_f32": lambda eb, o: eb.polstats(tn, [1, 2, 1, 2, 1, 2], [0, 0, 1, 1, 2, 2], out=o),
           "bin": lambda eb, o: eb.rebin()}   # (the binning pass itself)
    shape = {"optimized_f32": (12, torch.float32), "optimized_f64": (12, torch.float64), "voxel5_f64": (5, torch.float64),
             "tore_full_frame_f32": (12, torch.float32), "time_surface_f64": (12, torch.float64), "event_stack_f32": (12, torch.float32),
             "nimagenet_acc_all_f32": (6, torch.float32), "bin": (1, torch.float32)}[builder]
    out = torch.empty((B, H, W, shape[0]), dtype=shape[1], device="cuda:0")
    for eb in ebs:
        eb.bin()
    res = [[], []]
    for rnd in range(7):
        for k, eb in enumerate(ebs):
            res[k].append(bench_sweep.timed(lambda: fns[builder](eb, out), 30) * 1e3)
    print("%-12s %-22s %s off: %.1f us   on: %.1f us   (medians of 7 interleaved rounds; min %.1f / %.1f)" % (
        tag, builder, flag, np.median(res[0]), np.median(res[1]), min(res[0]), min(res[1])), flush=True)
